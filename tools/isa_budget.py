#!/usr/bin/env python3
"""tools/isa_budget.py <file.s> [--kernel N] : per-segment instruction counts of a gfx950 assembly listing.

A segment ends at a label, a branch or an s_barrier.  Classes: V = VALU (packed counted apart as P, transcendental
as T), S = SALU, M = SMEM (s_load/s_buffer_load), L = LDS (ds_*), G = global/flat/buffer memory, W = s_waitcnt /
s_nop, B = branch/barrier/other.  With --path a,b,c... (label names or line numbers) the counts of the segments
between consecutive marks are summed instead: `--sum 100-200,300-400` sums the instructions of line ranges.
"""
import re
import sys
from collections import Counter

TRANS = ("v_sqrt", "v_rcp", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")


def classify(op):
    if op.startswith("v_pk_"):
        return "P"
    if op.startswith(TRANS):
        return "T"
    if op.startswith("v_"):
        return "V"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "M"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "W"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc", "s_setprio", "s_sethalt", "s_trap")):
        return "B"
    if op.startswith("s_"):
        return "S"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "G"
    return "?"


# Measured issue cost, ns per wave-instruction per SIMD at 3 waves per SIMD (tools/ubench/rates2.hip, rates3.hip;
# gpurun_out/r3a): VGPR-only add/sub/mul/fma/and/mov ~1.2; packed f32, any SGPR operand, min/max/med3, compares,
# v_cndmask with an SGPR mask, 3-operand integer ops ~1.8; transcendentals ~3.5; v_cndmask_b32_e32 (implicit VCC) ~3.6
# behind a compare (9.5 on a stale VCC).
SLOW = ("v_pk_", "v_max", "v_min", "v_med3", "v_cmp", "v_lshl_add", "v_lshl_or", "v_add3", "v_mad_", "v_mul_lo", "v_mul_hi",
        "v_bfe", "v_bfi", "v_perm", "v_readlane", "v_writelane", "v_readfirstlane", "v_mbcnt", "v_and_or", "v_or3")


def cost(text):
    op = text.split()[0]
    if op.startswith(TRANS):
        return 3.5
    if op.startswith("v_cndmask_b32_e32"):
        return 3.6
    if op.startswith(SLOW) or op.startswith("v_cndmask"):
        return 1.8
    args = text[len(op):]
    if re.search(r"(^|[ ,\[])s\[?\d", args) or "vcc" in args or "exec" in args:
        return 1.8
    return 1.2


def parse(path):
    """-> list of (lineno, kind, text) with kind 'label' | class letter"""
    out = []
    inside = False
    for i, line in enumerate(open(path), 1):
        s = line.strip()
        if s.startswith("; APTMARK "):
            out.append((i, "mark", s[len("; APTMARK "):].strip()))
            continue
        if not s or s.startswith((";", "//")):
            continue
        if s.startswith(".") and not s.startswith(".LBB"):
            if s.startswith(".amdhsa_kernel") or s.startswith(".section\t.rodata"):
                inside = False
            continue
        m = re.match(r"^([A-Za-z_.$0-9]+):", s)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                inside = True
            out.append((i, "label", name))
            continue
        if not inside:
            continue
        op = s.split()[0]
        out.append((i, classify(op), s))
    return out


def fmt(c):
    v = c["V"] + c["P"] + c["T"]
    return (f"VALU {v:5d} (pk {c['P']:4d} tr {c['T']:3d})  SALU {c['S']:4d}  SMEM {c['M']:4d}  LDS {c['L']:4d}  "
            f"VMEM {c['G']:3d}  wait/nop {c['W']:4d}  br {c['B']:3d}")


def main():
    path = sys.argv[1]
    items = parse(path)
    if "--marks" in sys.argv:
        # sums between "; APTMARK BEGIN x" and "; APTMARK END x" (apt_kernels_fused_impl.hpp, -DAPT_FUSED_MARKS=1):
        # the code an interior tile executes, stage by stage
        open_marks, sums, order = {}, {}, []
        for ln, k, t in items:
            if k == "mark":
                what, name = t.split(None, 1)
                if what == "BEGIN":
                    open_marks[name] = Counter()
                    if name not in sums:
                        sums[name] = Counter()
                        order.append(name)
                elif name in open_marks:
                    sums[name].update(open_marks.pop(name))
                continue
            if k == "label":
                continue
            for c in open_marks.values():
                c[k] += 1
        total = Counter()
        for name in order:
            print(f"{name:>24s}  {fmt(sums[name])}")
            total.update(sums[name])
        print(f"{'total':>24s}  {fmt(total)}")
        return
    items = [it for it in items if it[1] != "mark"]
    if "--sum" in sys.argv:
        spec = sys.argv[sys.argv.index("--sum") + 1]
        total = Counter()
        for part in spec.split(","):
            name = None
            if "=" in part:
                name, part = part.split("=")
            a, b = (int(t) for t in part.split("-"))
            c = Counter(k for (ln, k, _) in items if a <= ln <= b and k != "label")
            ns = sum(cost(t) for (ln, k, t) in items if a <= ln <= b and k in "VPT")
            total.update(c)
            total_ns = total_ns + ns if "total_ns" in dir() else ns
            print(f"{(name or part):>24s}  {fmt(c)}  ~{ns:6.0f} ns")
        print(f"{'total':>24s}  {fmt(total)}  ~{total_ns:6.0f} ns VALU issue per wave (cost model)")
        if "--ops" in sys.argv:
            ops = Counter()
            for part in spec.split(","):
                if "=" in part:
                    part = part.split("=")[1]
                a, b = (int(t) for t in part.split("-"))
                ops.update(t.split()[0] for (ln, k, t) in items if a <= ln <= b and k in "VPT")
            for op, n in ops.most_common(40):
                print(f"    {op:28s} {n}")
        return
    seg = Counter()
    start = None
    for ln, k, t in items:
        if start is None:
            start = ln
        if k == "label":
            if sum(seg.values()):
                print(f"{start:6d}-{ln - 1:6d}  {fmt(seg)}")
            print(f"{ln:6d}  {t}:")
            seg = Counter()
            start = ln + 1
            continue
        seg[k] += 1
        if k == "B":
            print(f"{start:6d}-{ln:6d}  {fmt(seg)}   {t}")
            seg = Counter()
            start = None
    if sum(seg.values()):
        print(f"{start:6d}-{ln:6d}  {fmt(seg)}")


if __name__ == "__main__":
    main()
