#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/orbit; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "picker" 2>&1 | tail -15 > $O/t_picker.txt; cat $O/t_picker.txt
APTGPU_ORBIT_ALG=1 timeout 400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/t_all_env.txt; tail -12 $O/t_all_env.txt
python tools/sweep.py --inputs 16 --steps 300 --warmup 20 --configs "strict:16:3,strict:16:3:APTGPU_ORBIT_ALG=1,strict:16:1,strict:16:1:APTGPU_ORBIT_ALG=1,strict:1:3,strict:1:3:APTGPU_ORBIT_ALG=1" > $O/sweep.txt 2>&1
python3 - <<'PY'
import json
for l in open("gpurun_out/orbit/sweep.txt"):
    if l.startswith("{") and "config" in l:
        d=json.loads(l); print(d["config"], d["ms_per_recording"], d["rows_checksum"], d["alone_ms_per_call"])
PY
for e in 0 1; do APTGPU_ORBIT_ALG=$e python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench20_$e.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/bench20_$e.json').read().strip().splitlines()[-1]);print('alg',$e,d['ms_per_step'],d['config']['picker'])"; done
