"""CPU-side checks of the drop-in boundary: libaptgpu.so loads, exports every symbol
include/aptgpu.h declares, and its host-side math (no GPU involved) is bit-identical to the
oracle's.  No compute entry point is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

import noaa_apt_amd as apt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(apt.lib_path()):
        apt.build()
    return apt.lib()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(ROOT, "include", "aptgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(aptgpu_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    raw = ctypes.CDLL(apt.lib_path())
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_version_and_device_count(L):
    assert apt.version().startswith("aptgpu")
    hdr = open(os.path.join(ROOT, "include", "aptgpu.h")).read()
    assert apt.abi_version() == int(re.search(r"#define APTGPU_ABI_VERSION (\d+)", hdr).group(1))
    assert apt.device_count() >= 0


def test_struct_layouts_match_header(L):
    # sizes the C side static_asserts / relies on
    assert ctypes.sizeof(apt.Result) == 32
    assert ctypes.sizeof(apt.KernelTime) == 64
    # aptgpu_batch_stats: struct_size + reserved in front of the 0.1.0 fields (include/aptgpu.h)
    assert ctypes.sizeof(apt.BatchStats) == 88 and apt.BatchStats.struct_size.offset == 0 and apt.BatchStats.seconds.offset == 8


def test_batch_stats_without_struct_size_is_refused(L):
    """aptgpu_batch_stats.struct_size is the caller's sizeof; 0 — a zero-initialised struct, or what a
    0.1.0 caller's leading `double seconds` happens to hold — must not be read as "this header's size" (the library would
    write 88 bytes into whatever the caller has).  Refused before anything runs: no GPU needed to see it."""
    import ctypes as C
    cs = apt.Settings()._c()
    cctx = apt.Context()._c()
    st = apt.BatchStats()  # struct_size == 0
    err = C.create_string_buffer(512)
    rc = L.aptgpu_decode_batch(C.byref(cctx), C.byref(cs), 48000, 1, 0, None, None, None, 0, 0, None, None, None, None,
                               C.byref(st), err, 512)
    assert rc == apt.InvalidError.code and b"struct_size" in err.value
    assert st.seconds == 0.0 and st.workers == 0  # nothing written


def test_no_environment_variable_redirects_the_loader(L, monkeypatch):
    """the shipped module loads the in-tree product library, whatever APTGPU_LIB says; the probe build is
    selected in code (use_library), before the first load, and announced on stderr."""
    monkeypatch.setenv("APTGPU_LIB", "/nonexistent/libevil.so")
    import importlib
    assert apt.lib_path().endswith(os.path.join("noaa_apt_amd", "libaptgpu.so"))
    with pytest.raises(RuntimeError):
        apt.use_library("/tmp/other.so")  # the library is loaded already (fixture L)


@pytest.mark.parametrize("kind,cut,atten,dw", [
    ("Lowpass", 1 / 4, 20., 1 / 10), ("Lowpass", 1 / 3, 35., 1 / 30), ("Lowpass", 2 / 5, 60., 1 / 20),
    ("LowpassDcRemoval", 1 / 4, 20., 1 / 10), ("LowpassDcRemoval", 1 / 3, 35., 1 / 30),
    ("LowpassDcRemoval", 2 / 5, 60., 1 / 20)])
def test_filter_design_bitexact_vs_oracle(L, oracle, kind, cut, atten, dw):
    f = getattr(apt, kind)(apt.Freq.pi_rad(cut), atten, apt.Freq.pi_rad(dw))
    ok = oracle.LOWPASS if kind == "Lowpass" else oracle.LOWPASS_DC_REMOVAL
    want = oracle.filter_design(ok, np.float32(cut), atten, np.float32(dw))
    got = f.design()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("rate", [48000, 96000, 11025, 44100, 22050, 8000])
@pytest.mark.parametrize("profile", ["standard", "fast", "slow"])
def test_decode_filters_bitexact_vs_oracle(L, oracle, rate, profile):
    """The two designs decode() makes (decode.rs:65-76, 95-100) for every profile/rate."""
    s = apt.Settings.profile(profile)
    in_rate, work = apt.Rate.hz(rate), apt.Rate.hz(s.work_rate)
    g = np.gcd(rate, s.work_rate)
    l = s.work_rate // g
    f = apt.LowpassDcRemoval(apt.Freq.hz(s.resample_cutout, in_rate), s.resample_atten,
                             apt.Freq.hz(s.resample_delta_freq, in_rate))
    cut, dw = oracle.freq_hz(s.resample_cutout, rate), oracle.freq_hz(s.resample_delta_freq, rate)
    if l > 1:
        f.resample(in_rate, apt.Rate.hz(rate * l))
        _, cut, _, dw = oracle.filter_resample(oracle.LOWPASS_DC_REMOVAL, cut, s.resample_atten, dw,
                                               rate, rate * l)
    want = oracle.filter_design(oracle.LOWPASS_DC_REMOVAL, cut, s.resample_atten, dw)
    got = f.design()
    assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    c2 = apt.Freq.pi_rad(np.float32(4160) / np.float32(work.get_hz()))
    got2 = apt.Lowpass(c2, s.demodulation_atten, c2 / 5.0).design()
    want2 = oracle.filter_design(oracle.LOWPASS, c2.get_pi_rad(), s.demodulation_atten,
                                 (c2 / 5.0).get_pi_rad())
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))


def test_no_filter_and_resample(L):
    assert apt.NoFilter().design().tolist() == [1.0]
    f = apt.Lowpass(apt.Freq.hz(123., apt.Rate.hz(1000)), 40., apt.Freq.hz(12., apt.Rate.hz(1000)))
    f.resample(apt.Rate.hz(1000), apt.Rate.hz(3000))
    # filters.rs:384-398 asserts exact equality with the filter designed at 3000 Hz
    assert np.float32(f.cutout.get_pi_rad()) == np.float32(apt.Freq.hz(123., apt.Rate.hz(3000)).get_pi_rad())
    assert np.float32(f.delta_w.get_pi_rad()) == np.float32(apt.Freq.hz(12., apt.Rate.hz(3000)).get_pi_rad())


def test_sync_frame_matches_reference_vectors(L):
    # decode.rs:271-319
    g5 = apt.generate_sync_frame(apt.Rate.hz(4160 * 5))
    assert g5.tolist() == [-1] * 20 + ([1] * 10 + [-1] * 10) * 7 + [-1] * 30
    g2 = apt.generate_sync_frame(apt.Rate.hz(4160 * 2))
    assert g2.tolist() == [-1] * 8 + ([1] * 4 + [-1] * 4) * 7 + [-1] * 12
    with pytest.raises(apt.InternalError) as e:
        apt.generate_sync_frame(apt.Rate.hz(11025))
    assert str(e.value) == "work_rate is not multiple of FINAL_RATE"


def test_product_does_not_touch_the_oracle():
    """The shipped library must not link, load or reference anything under oracle/."""
    blob = open(apt.lib_path(), "rb").read()
    assert b"aptoracle" not in blob and b"apt_oracle" not in blob
    for fn in os.listdir(os.path.join(ROOT, "noaa_apt_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "noaa_apt_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src


def test_c_example_compiles_against_the_header(tmp_path):
    """examples/aptgpu_decode.c is plain C99: the header must be C-clean (-pedantic) and the
    library must link without Python or PyTorch."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "aptgpu_decode"
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror",
                           "-I", os.path.join(root, "include"), "-o", str(exe),
                           os.path.join(root, "examples", "aptgpu_decode.c"),
                           "-L", os.path.dirname(apt.lib_path()), "-laptgpu",
                           "-Wl,-rpath," + os.path.dirname(apt.lib_path()), "-Wl,-rpath,/opt/rocm/lib"])
    # usage message without arguments (no GPU needed)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr
