#!/bin/bash
# A/B of the two-pass PHASE stage 1 (APT_PHASE_HALVES) on one box: the tree's library against libaptgpu_nohalves.so
# (tools/probes/halves_variant.sh), front end alone and the pipelined loop, row checksums.  args: rate:profile[:mode] ...
mkdir -p gpurun_out/s7; rm -f gpurun_out/s7/*
for a in "$@"; do
  IFS=: read rate prof mode <<< "$a"; mode=${mode:-strict}
  for rep in 1 2; do
    APTGPU_PROBE_LIB=$PWD/noaa_apt_amd/libaptgpu_nohalves.so python tools/sweep.py --rate $rate --profile $prof --inputs 4 --steps 30 --configs "$mode:16:1,$mode:16:3" 2>&1 | grep "^{\"config" | sed -e "s/^/old /" >> gpurun_out/s7/ab_${prof}_${rate}_$mode.txt
    python tools/sweep.py --rate $rate --profile $prof --inputs 4 --steps 30 --configs "$mode:16:1,$mode:16:3" 2>&1 | grep "^{\"config" | sed -e "s/^/new /" >> gpurun_out/s7/ab_${prof}_${rate}_$mode.txt
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s7/ab_*.txt")):
    for ln in open(f):
        tag, js = ln.split(" ", 1)
        d = json.loads(js)
        print(f.split("/")[-1], tag, d["config"], "front end alone", d.get("alone_ms_per_call", {}).get("fused_front_end"), "ms/rec pipelined", d["ms_per_recording"], d["rows_checksum"], "fused", d["fused"])
PY
