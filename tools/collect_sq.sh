#!/bin/bash
# tools/collect_sq.sh <mode> <outdir>: SQ issue / stall counters of the front end (one call in flight at a time)
# rocprofv3 from /tmp, counters in their own passes with --kernel-trace only.
MODE=${1:-fast}; OUT=${2:-gpurun_out/sq_$MODE}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
python -c "import torch" 2>/dev/null
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/pass$i -- \
    python $R/tools/sweep.py --steps 6 --warmup 2 --inputs 4 --configs $MODE:1:1 > $R/$OUT/pass$i.log 2>&1
done
cd $R
python tools/summarize_sq.py k_fused $(ls $OUT/pass*/*/*counter_collection.csv) --note "rocprofv3 --pmc, three passes, tools/sweep.py --configs $MODE:1:1 (one recording per launch, one launch in flight)" > $OUT/summary.json
cat $OUT/summary.json
