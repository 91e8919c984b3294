#!/bin/bash
# tools/collect_sq_kernel.sh <label> <kernel-substring> <sweep.py arguments...>
# SQ issue / stall / LDS counters (three rocprofv3 --pmc passes, kernel-trace only) and HBM bytes (FETCH_SIZE, WRITE_SIZE:
# one pass each) of one kernel of the decode chain, from tools/sweep.py with one launch in flight:
#   gpurun_out/prof/sq_<label>.json, gpurun_out/prof/hbm_<label>.json  (stamped with the hash of the kernel sources)
LABEL=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/tmp_$LABEL/pass$i -- \
    python $R/tools/sweep.py --steps 6 --warmup 2 "$@" > $O/tmp_$LABEL.pass$i.log 2>&1
done
cd $R
NOTE="rocprofv3 --pmc, one pass per counter set, tools/sweep.py --steps 6 $* (one launch in flight)"
python tools/summarize_sq.py "$PAT" $(ls $O/tmp_$LABEL/pass[123]/*/*counter_collection.csv) --note "$NOTE" > $O/sq_$LABEL.json
python tools/summarize_pmc.py $(ls $O/tmp_$LABEL/pass4/*/*counter_collection.csv | head -1) $(ls $O/tmp_$LABEL/pass5/*/*counter_collection.csv | head -1) "$NOTE; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)" > $O/hbm_$LABEL.json
python - <<PY
import json, sys
sys.path.insert(0, "tools")
from csrc_hash import csrc_sha16
for f in ("$O/sq_$LABEL.json", "$O/hbm_$LABEL.json"):
    try:
        d = json.load(open(f)); d["csrc_sha16"] = csrc_sha16(); json.dump(d, open(f, "w"), indent=1)
    except Exception as e:
        print("stamp failed", f, e)
PY
rm -rf $O/tmp_$LABEL $O/tmp_$LABEL.pass*.log
