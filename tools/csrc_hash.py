#!/usr/bin/env python3
"""sha256 (first 16 hex digits) of the kernel sources under noaa_apt_amd/csrc (*.hip, *.hpp, *.cpp, Makefile) — the
stamp that ties a committed profile (profiles/r0N_hbm_traffic_*.json, r0N_sq_counters_*.json) to the code it was
collected on; bench.py only quotes a profile whose stamp matches the sources it runs.  No git needed (the GPU box
has no .git).

`csrc_sha16()` covers everything.  `group_sha16(group)` covers only the files a group of kernels is COMPILED from (its
translation units, every header they include, the Makefile with the compiler flags): a counter profile of the front end
is not invalidated by an edit to the peak picker and vice versa.  Profiles collected from round 5 on carry both
(`csrc_sha16`, `csrc_groups_sha16`); bench.py accepts a profile when either the whole-tree stamp or the stamp of the
group it measured matches.  `python tools/csrc_hash.py` prints the whole-tree hash, `--groups` all of them as JSON."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# headers every kernel translation unit includes, and the build flags
_COMMON = ("apt_kernels.hpp", "apt_sync_corr.hpp", "apt_envelope.hpp", "Makefile")
_GROUPS = {
    # k_fused<...>: the specialised front ends (SPLIT / TABLE / PHASE), one instantiation per file
    "front_end": lambda n: n.startswith("apt_kernels_fused") and "fused_any" not in n,
    # k_fused_any: the run-time front end
    "front_end_any": lambda n: n.startswith("apt_kernels_fused_any"),
    # k_sync_words / k_sync_slots / k_sync_orbit_global, the gathers and the unfused stages
    "chain": lambda n: n in ("apt_kernels_sync.hip", "apt_kernels_generic.hip"),
}


def _makefile_flags(path):
    """What of the Makefile decides how a kernel is compiled: the compiler, the architecture and the flags — not the list of
    translation units (round 6: adding an instantiation file used to invalidate every group's profiles)."""
    keep = []
    with open(path, "rb") as f:
        for line in f.read().splitlines():
            if line.startswith((b"HIPCC", b"ARCH", b"CXXFLAGS")) or b"$(HIPCC)" in line:
                keep.append(line)
    return b"\n".join(keep)


def _hash(root, keep):
    d = os.path.join(root, "noaa_apt_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if keep(name):
            h.update(name.encode())
            if name == "Makefile":
                h.update(_makefile_flags(os.path.join(d, name)))
                continue
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def csrc_sha16(root=ROOT):
    return _hash(root, lambda n: n.endswith((".hip", ".hpp", ".cpp")) or n == "Makefile")


def group_sha16(group, root=ROOT):
    sel = _GROUPS[group]
    return _hash(root, lambda n: n in _COMMON or ((n.endswith((".hip", ".hpp")) and sel(n))))


def groups_sha16(root=ROOT):
    return {g: group_sha16(g, root) for g in sorted(_GROUPS)}


def stamp(d, root=ROOT):
    """Add both stamps to a profile dictionary (in place) and return it."""
    d["csrc_sha16"] = csrc_sha16(root)
    d["csrc_groups_sha16"] = groups_sha16(root)
    return d


def matches(d, group, root=ROOT):
    """Was profile `d` collected on the sources here — the whole tree, or at least the group it measured?"""
    if not isinstance(d, dict):
        return False
    if d.get("csrc_sha16") and d.get("csrc_sha16") == csrc_sha16(root):
        return True
    g = (d.get("csrc_groups_sha16") or {}).get(group)
    return bool(g) and g == group_sha16(group, root)


if __name__ == "__main__":
    if "--groups" in sys.argv:
        print(json.dumps({"csrc_sha16": csrc_sha16(), "csrc_groups_sha16": groups_sha16()}))
    else:
        print(csrc_sha16())
