#!/usr/bin/env python3
"""tools/isa_lint.py <file.s>... : the specialised front end's stage 1 fetches its taps with scalar loads written as
inline assembly into PINNED SGPR tuples, and waits for them one chunk later with an explicit s_waitcnt lgkmcnt(0)
(apt_kernels_fused_impl.hpp).  The compiler's own dependency tracking does not see those loads, so nothing but the
shape of the generated code keeps it from touching a tuple whose load is still in flight — a register copy inserted by
the allocator between load and wait would read garbage (seen once, with three-sample chunks and 80 pinned SGPRs).
This check walks every kernel's listing: from each pinned load to the next `s_waitcnt lgkmcnt(0)`, no instruction may
name a register of the tuple being loaded, as source or destination.  Exit status 1 and a report if one does.
"""
import re
import sys

LOAD = re.compile(r"^\s*s_load_dwordx(\d+)\s+s\[(\d+):(\d+)\]")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def regs_of(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def lint(path):
    bad = []
    lines = open(path).read().splitlines()
    in_asm = False
    pending = set()      # registers with a load in flight (issued inside an inline-asm block)
    issued_at = 0
    for i, line in enumerate(lines, 1):
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        m = LOAD.match(s)
        if in_asm and m:
            pending.update(range(int(m.group(2)), int(m.group(3)) + 1))
            issued_at = i
            continue
        if s.startswith("s_waitcnt") and "lgkmcnt(0)" in s:
            pending.clear()
            continue
        if s.startswith(("s_endpgm", "s_barrier")):
            pending.clear()
            continue
        if pending:
            op, _, rest = s.partition(" ")
            hit = regs_of(rest) & pending
            if hit:
                bad.append((i, issued_at, s, sorted(hit)[:4]))
    return bad


def main():
    rc = 0
    for path in sys.argv[1:]:
        bad = lint(path)
        n_loads = sum(1 for l in open(path) if LOAD.match(l.strip()))
        if bad:
            rc = 1
            print(f"{path}: {len(bad)} instruction(s) touch a tap tuple whose load is in flight:")
            for i, at, s, hit in bad[:12]:
                print(f"  line {i} (load issued at line {at}): {s}    [s{hit[0]}...]")
        else:
            print(f"{path}: ok ({n_loads} scalar loads checked)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
