#!/bin/bash
# tools/collect_sq_kernel.sh <label:kernel-substring>[,<label:kernel-substring>...] <hbm: 0|1> <sweep.py arguments...>
# SQ issue / stall / LDS counters (three rocprofv3 --pmc passes, kernel-trace only) and — hbm = 1 — HBM bytes (FETCH_SIZE,
# WRITE_SIZE: one pass each) of the named kernels of ONE tools/sweep.py run shape, one launch in flight:
#   gpurun_out/prof/sq_<label>.json, gpurun_out/prof/hbm_<label>.json  (stamped with the hash of the kernel sources)
SPEC=$1; HBM=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
TAG=$(echo "$SPEC" | tr ',:' '__')
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
      "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
      "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT")
if [ "$HBM" = 1 ]; then SETS+=("FETCH_SIZE" "WRITE_SIZE"); fi
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/tmp_$TAG/pass$i -- \
    python $R/tools/sweep.py --steps 6 --warmup 2 "$@" > $O/tmp_$TAG.pass$i.log 2>&1
done
cd $R
NOTE="rocprofv3 --pmc, one pass per counter set, tools/sweep.py --steps 6 $* (one launch in flight)"
for LP in $(echo "$SPEC" | tr ',' ' '); do
  LABEL=${LP%%:*}; PAT=${LP##*:}
  python tools/summarize_sq.py "$PAT" $(ls $O/tmp_$TAG/pass[123]/*/*counter_collection.csv) --note "$NOTE" > $O/sq_$LABEL.json
  FILES="$O/sq_$LABEL.json"
  if [ "$HBM" = 1 ]; then
    python tools/summarize_pmc.py $(ls $O/tmp_$TAG/pass4/*/*counter_collection.csv | head -1) $(ls $O/tmp_$TAG/pass5/*/*counter_collection.csv | head -1) "$NOTE; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); uncalibrated (the r04_hbm_traffic_* files carry the pool's calibration factor)" > $O/hbm_$LABEL.json
    FILES="$FILES $O/hbm_$LABEL.json"
  fi
  python - $FILES <<'PY'
import json, sys
sys.path.insert(0, "tools")
from csrc_hash import stamp
for f in sys.argv[1:]:
    try:
        d = stamp(json.load(open(f))); json.dump(d, open(f, "w"), indent=1)
    except Exception as e:
        print("stamp failed", f, e)
PY
done
rm -rf $O/tmp_$TAG $O/tmp_$TAG.pass*.log
