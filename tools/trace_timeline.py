#!/usr/bin/env python3
"""A readable timeline out of a rocprofv3 --kernel-trace CSV of the pipelined loop (profiles/r*_pipeline_trace*.txt).

    python tools/trace_timeline.py kernel_trace.csv [--steps 6] [--skip-front-ends 30]

Prints start / end / duration (us, relative to a front end's start) per kernel for a few consecutive steps from the
middle of the run, then means over the pipelined part: front-end duration, the gap from one front end's end to the
next one's start, the period, and the duration of each chain kernel beside the front end.
"""
import argparse
import csv
import re
import statistics as st


def short(name):
    m = re.search(r"(k_\w+|__amd_\w+)", name)
    return m.group(1) if m else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--skip-front-ends", type=int, default=30, help="front ends to skip from the start (warm-up, other plans)")
    ap.add_argument("--front", default="k_fused")
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])))
    rows.sort()
    # the pipelined part: front ends of the largest grid
    big = max(g for (_, _, _, n, g) in rows if n == args.front)
    fes = [r for r in rows if r[3] == args.front and r[4] == big]
    # consecutive front ends closer than 2 periods apart
    run = []
    best = []
    for f in fes:
        if run and f[0] - run[-1][1] > 200_000:
            if len(run) > len(best):
                best = run
            run = []
        run.append(f)
    if len(run) > len(best):
        best = run
    fes = best
    k0 = min(args.skip_front_ends, max(0, len(fes) - args.steps - 2))
    t0 = fes[k0][0]
    t1 = fes[min(len(fes) - 1, k0 + args.steps)][0]
    queues = {}
    print(f"# {len(fes)} front ends back to back in the trace; steps {k0} .. {k0 + args.steps} of them below")
    for (a, b, q, n, g) in rows:
        if a < t0 - 20_000 or a > t1:
            continue
        qn = queues.setdefault(q, len(queues) + 1)
        print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f}  queue {qn}  {n}")
    inner = fes[4:-2] if len(fes) > 10 else fes
    dur = [(b - a) / 1e3 for (a, b, *_) in inner]
    gap = [(inner[i + 1][0] - inner[i][1]) / 1e3 for i in range(len(inner) - 1)]
    per = [(inner[i + 1][0] - inner[i][0]) / 1e3 for i in range(len(inner) - 1)]
    print(f"# front end, {len(inner)} launches of the pipelined part: duration mean {st.mean(dur):.1f} us, gap to the next front end "
          f"mean {st.mean(gap):.1f} us, period mean {st.mean(per):.1f} us (under rocprofv3)")
    lo, hi = inner[0][0], inner[-1][1]
    names = sorted({n for (_, _, _, n, _) in rows if n != args.front})
    for n in names:
        d = [(b - a) / 1e3 for (a, b, _, nn, _) in rows if nn == n and lo <= a <= hi]
        if d:
            print(f"# {n}: {len(d)} launches, mean {st.mean(d):.1f} us beside the front end")


if __name__ == "__main__":
    main()
