#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4h; mkdir -p $O
python -c "import torch" 2>/dev/null
for Q in 4 8; do
  GPU_MAX_HW_QUEUES=$Q timeout 300 python tools/sweep.py --configs "strict:1:3,strict:1:4,strict:1:6,strict:1:8,strict:1:12,strict:16:3,strict:16:4,strict:16:6,strict:4:3,strict:4:6" --steps 300 --warmup 20 --inputs 16 > $O/sweep_q$Q.txt 2> $O/sweep_q$Q.err
done
ls $O
