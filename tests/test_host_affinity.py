"""NUMA placement of the host-fed batch workers (include/aptgpu.h 2b; SURVEY.md 8(e): one host thread per GPU,
fed over PCIe): the lookup "which CPUs sit next to this GPU" on mocked sysfs trees.  No GPU needed."""
import os

import pytest

import noaa_apt_amd as apt


def _tree(root, devices, nodes):
    for bdf, (numa, local) in devices.items():
        d = os.path.join(root, "bus", "pci", "devices", bdf)
        os.makedirs(d)
        if numa is not None:
            open(os.path.join(d, "numa_node"), "w").write(f"{numa}\n")
        if local is not None:
            open(os.path.join(d, "local_cpulist"), "w").write(f"{local}\n")
    for n, cpus in nodes.items():
        d = os.path.join(root, "devices", "system", "node", f"node{n}")
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(f"{cpus}\n")


def test_two_socket_node_with_eight_gpus(tmp_path):
    """The shape of an 8-GPU MI355X host: four GPUs per socket, SMT siblings listed as a second range."""
    root = str(tmp_path)
    devs = {f"0000:{b:02x}:00.0": (0 if i < 4 else 1, None) for i, b in enumerate((0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5))}
    _tree(root, devs, {0: "0-63,128-191", 1: "64-127,192-255"})
    for i, bdf in enumerate(devs):
        node, cpus = apt.host_affinity_from_sysfs(root, bdf)
        assert node == (0 if i < 4 else 1)
        assert cpus == ("0-63,128-191" if i < 4 else "64-127,192-255")
    # hipDeviceGetPCIBusId may spell the address in upper case: sysfs does not
    assert apt.host_affinity_from_sysfs(root, "0000:E5:00.0") == (1, "64-127,192-255")


def test_unknown_placement_means_no_pinning(tmp_path):
    root = str(tmp_path)
    _tree(root, {"0000:05:00.0": (-1, "0-15"), "0000:06:00.0": (None, "0-15"), "0000:07:00.0": (3, None),
                 "0000:08:00.0": (2, "8-11")}, {2: ""})
    assert apt.host_affinity_from_sysfs(root, "0000:05:00.0") == (-1, "")   # numa_node = -1: single-socket / no SRAT
    assert apt.host_affinity_from_sysfs(root, "0000:06:00.0") == (-1, "")   # no numa_node file
    assert apt.host_affinity_from_sysfs(root, "0000:07:00.0") == (-1, "")   # a node without a CPU list
    assert apt.host_affinity_from_sysfs(root, "0000:09:00.0") == (-1, "")   # no such device
    assert apt.host_affinity_from_sysfs(root, "0000:08:00.0") == (2, "8-11")  # node list empty: the device's local_cpulist


@pytest.mark.parametrize("bad", ["0-", "a-3", "5-2", "1,,x"])
def test_malformed_cpu_lists_are_refused(tmp_path, bad):
    root = str(tmp_path)
    _tree(root, {"0000:05:00.0": (0, None)}, {0: bad})
    assert apt.host_affinity_from_sysfs(root, "0000:05:00.0") == (-1, "")
