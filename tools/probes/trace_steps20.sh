#!/bin/bash
# The driver's 20-step shape under rocprofv3 --kernel-trace: where the timed region's time goes (front ends back to back,
# the chain of the last call alone behind the last front end).  -> gpurun_out/s8/steps20_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-single-launch --no-power > $O/bench.json 2>/dev/null
python - $O <<'PY'
import csv, glob, json, re, sys
O = sys.argv[1]
f = sorted(glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24],
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])))
rows.sort()
big = max(g for (_, _, n, g) in rows if n == "k_fused")
fes = [r for r in rows if r[2] == "k_fused" and r[3] == big]
runs, run = [], []
for fe in fes:
    if run and fe[0] - run[-1][1] > 150_000:
        runs.append(run); run = []
    run.append(fe)
runs.append(run)
idx = [i for i, r in enumerate(runs) if len(r) == 20][-1]   # warm-up (5) and the timed steps (20) are separated by the synchronize
timed = runs[idx]
nxt = runs[idx + 1][0][0] if idx + 1 < len(runs) else 1 << 62   # (the call behind the timed region: compared with the oracle)
t0, last_fe_end = timed[0][0], timed[-1][1]
chain = [r for r in rows if r[2] != "k_fused" and timed[0][0] <= r[0] < nxt and r[2].startswith("k_")]
tail = [r for r in chain if r[1] >= timed[-1][0]]
end = max(r[1] for r in chain)
line = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
out = []
out.append(f"bench line of this run (under rocprofv3): ms_per_step {line['ms_per_step']}")
out.append(f"timed region on the device: first front end's start -> last gather's end {(end - t0) / 1e3:.1f} us = {(end - t0) / 20e3:.2f} us per step")
out.append(f"  20 front ends: sum of durations {sum(b - a for (a, b, *_) in timed) / 1e3:.1f} us, first start -> last end {(last_fe_end - t0) / 1e3:.1f} us "
           f"({(last_fe_end - t0) / 20e3:.2f} us per step)")
out.append(f"  behind the last front end (the last call's chain, alone on the GPU): {(end - last_fe_end) / 1e3:.1f} us")
for (a, b, n, g) in tail:
    out.append(f"    {(a - last_fe_end) / 1e3:9.1f} .. {(b - last_fe_end) / 1e3:9.1f} us  {n} ({(b - a) / 1e3:.1f} us)")
import statistics as st
gaps = [(timed[i + 1][0] - timed[i][1]) / 1e3 for i in range(19)]
out.append(f"  front ends in the pipeline: {st.mean((b - a) / 1e3 for (a, b, *_) in timed[3:]):.0f} us each (one at a time: see the bench line's roofline.kernel_avg_ms), "
           f"hand-over gap to the next one {st.mean(gaps):.1f} us")
for nm in ("k_sync_words", "k_sync_slots", "k_sync_orbit", "k_gather"):
    d = [(b - a) / 1e3 for (a, b, n, g) in chain if n.startswith(nm)]
    out.append(f"  {nm}: {len(d)} launches, median {st.median(d):.0f} us beside the front ends, the last one (alone) {d[-1]:.0f} us")
open(O + "/steps20_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
