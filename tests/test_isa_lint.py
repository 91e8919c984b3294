"""The specialised front end keeps scalar loads in flight behind the compiler's back (inline assembly into pinned SGPR
tuples, waited for one chunk later): tools/isa_lint.py checks the generated code for the one thing that would break it —
an instruction that touches a tuple between its load and its wait.  Here: the check itself on hand-made listings, and
on the freshly compiled listing of two of the kernels (hipcc cross-compiles without a GPU; `make -C noaa_apt_amd/csrc lint`
does all eight)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

GOOD = """
k:
	;;#ASMSTART
	s_load_dwordx16 s[36:51], s[4:5], 0x68
	s_load_dwordx2 s[84:85], s[4:5], 0xc8
	;;#ASMEND
	ds_read_b64 v[8:9], v3 offset:8
	v_pk_mul_f32 v[6:7], v[2:3], s[52:53] op_sel_hi:[0,1]
	s_mov_b32 s1, s0
	s_waitcnt lgkmcnt(0)
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_pk_mul_f32 v[6:7], v[2:3], s[36:37] op_sel_hi:[0,1]
	s_endpgm
"""
BAD_READ = GOOD.replace("s_mov_b32 s1, s0", "s_mov_b64 s[6:7], s[40:41]")     # copies a register whose load is in flight
BAD_WRITE = GOOD.replace("s_mov_b32 s1, s0", "s_mov_b32 s85, 0")              # writes one
BAD_VALU = GOOD.replace("s[52:53] op_sel_hi:[0,1]\n\ts_mov", "s[50:51] op_sel_hi:[0,1]\n\ts_mov")


def _lint_text(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return isa_lint.lint(str(p))


def test_lint_accepts_the_intended_shape_and_flags_every_kind_of_touch(tmp_path):
    assert _lint_text(tmp_path, GOOD) == []
    for bad in (BAD_READ, BAD_WRITE, BAD_VALU):
        hits = _lint_text(tmp_path, bad)
        assert len(hits) == 1 and hits[0][1] in (4, 5), hits


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiled_kernels_keep_their_hands_off_taps_in_flight(tmp_path):
    csrc = os.path.join(ROOT, "noaa_apt_amd", "csrc")
    flags = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "
             "-fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize --cuda-device-only -S").split()
    tus = ["fused_48k_f32", "fused_96k_fast_f32"]
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-o", str(tmp_path / f"{k}.s"), f"apt_kernels_{k}.hip"],
                              cwd=csrc, stderr=subprocess.PIPE) for k in tus]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-2000:]
    for k in tus:
        path = str(tmp_path / f"{k}.s")
        n_loads = sum(1 for line in open(path) if isa_lint.LOAD.match(line.strip()))
        assert n_loads > 150, (k, n_loads)   # the chunk loads are there (and were recognised)
        assert isa_lint.lint(path) == [], k
