#!/bin/bash
# the 20-step shape with the orbit kernel's 128 KB all-nodes form (default) and its 24 KB closure form, which fits the LDS one
# retiring front-end workgroup leaves: does the orbit then start before the next front end ends, and does the tail shrink?
mkdir -p gpurun_out/s10
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-single-launch --no-power > gpurun_out/s10/def_$i.json 2>/dev/null
  APTGPU_ORBIT_THREADS=256 APTGPU_ORBIT_LDS=0 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-single-launch --no-power > gpurun_out/s10/closure_$i.json 2>/dev/null
done
APTGPU_ORBIT_THREADS=256 APTGPU_ORBIT_LDS=0 python bench.py --steps 300 --no-extras --no-cpu-baseline --no-single-launch --no-power > gpurun_out/s10/closure_300.json 2>/dev/null
python bench.py --steps 300 --no-extras --no-cpu-baseline --no-single-launch --no-power > gpurun_out/s10/def_300.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/s10/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_step"], d["pipeline"].get("kernels_alone_ms", {}).get("sync_orbit"))
PY
