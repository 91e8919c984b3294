#!/bin/bash
# A/B of the balanced PHASE slots (TableGeom::sq, rot): front end alone, ms per launch of 16, row checksums
mkdir -p gpurun_out/s4; rm -f gpurun_out/s4/*
run() { # rate profile
  python tools/sweep.py --rate $1 --profile $2 --inputs 4 --steps 30 \
    --configs "strict:16:1:APTGPU_PHASE_BALANCED=0,strict:16:1:APTGPU_PHASE_ROT=0,strict:16:1:APTGPU_PHASE_ROT=1,strict:16:1:APTGPU_PHASE_ROT=2,strict:16:1:APTGPU_PHASE_BALANCED=0,strict:16:1:APTGPU_PHASE_ROT=1,strict:16:1:APTGPU_PHASE_ROT=2" 2>&1 | grep -v "^{\"inputs" > gpurun_out/s4/ab_$2_$1.txt
}
for a in "$@"; do run ${a%%:*} ${a##*:}; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s4/ab_*.txt")):
    for ln in open(f):
        try: d=json.loads(ln)
        except Exception: continue
        print(f.split("/")[-1], d.get("config"), d.get("alone_ms_per_call",{}).get("fused_front_end"), d.get("ms_per_recording"), d.get("rows_checksum"))
PY
