#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4s; mkdir -p $O
python3 -c "import torch" 2>/dev/null
for i in 1 2 3; do
for ss in 0 40 120; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --settle-steps $ss > $O/b_${ss}_$i.json 2> $O/e
python3 - $O/b_${ss}_$i.json $ss <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("settle",sys.argv[2], d['ms_per_step'], d['power']['timed_region'], d['power']['settled']['ms_per_step'])
PY
done; done
