#!/bin/bash
mkdir -p gpurun_out/s3
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-single-launch > gpurun_out/s3/p_$i.json 2>/dev/null
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-single-launch --no-power > gpurun_out/s3/np_$i.json 2>/dev/null
done
python bench.py --steps 600 --no-extras --no-cpu-baseline --no-single-launch > gpurun_out/s3/p600.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s3/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("power",{}).get("timed_region",{}).get("gfxclk_mhz_at_end"))
    except Exception as e: print(f, "ERR", e)
PY
