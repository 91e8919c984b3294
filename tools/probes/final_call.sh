#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/prof gpurun_out/fin
timeout 150 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/fin/tests.txt; cat gpurun_out/fin/tests.txt
grep -q " passed" gpurun_out/fin/tests.txt && ! grep -q "failed" gpurun_out/fin/tests.txt || exit 7
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/fin/bench20.json 2>/dev/null
python3 -c "
import json;d=json.loads(open('gpurun_out/fin/bench20.json').read().strip().splitlines()[-1]);print('bench20',d['ms_per_step'],d['config']['picker'],d['roofline']['traffic'])"
bash tools/collect_sq_kernel.sh words:k_sync_words,slots:k_sync_slots,orbit:k_sync_orbit,gather:k_gather_rows 0 --inputs 4 --configs strict:16:1 > gpurun_out/prof_sq_kernels.log 2>&1
ls gpurun_out/prof/ | head
