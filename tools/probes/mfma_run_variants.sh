#!/bin/bash
# tools/probes/mfma_run_variants.sh name ... (GPU box): the front end alone (APTGPU_DEBUG_SKIP=7), 16 recordings per
# launch, three calls in flight, for every variant library of tools/probes/mfma_variants.sh and for the VALU fast kernel
show='
import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  loop ms/call", round(d["ms_per_recording"]*16,4), "alone", d["alone_ms_per_call"], "rows", d["rows"], "chk", d["rows_checksum"])'
for n in "$@"; do
  echo "== variant $n"
  APTGPU_PROBE_LIB=noaa_apt_amd/libaptgpu_mfma_$n.so APTGPU_DEBUG_SKIP=7 timeout 200 python tools/sweep.py --configs fast:16:3 --steps 60 --inputs 16 2>&1 | grep config | python -c "$show"
done
echo "== VALU fast kernel"
APTGPU_FAST_MFMA=0 APTGPU_PROBE_LIB=noaa_apt_amd/libaptgpu_probe.so APTGPU_DEBUG_SKIP=7 timeout 200 python tools/sweep.py --configs fast:16:3 --steps 60 --inputs 16 2>&1 | grep config | python -c "$show"
