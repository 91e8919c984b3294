#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4p; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch > $O/bench_strict_600.json 2> $O/e1
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch --no-power > $O/bench_strict_600_nopower.json 2> $O/e2
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch --mode fast > $O/bench_fast_600.json 2> $O/e3
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch --rate 44100 > $O/bench_44100.json 2> $O/e4
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch --rate 96000 --seconds 3600 --batch 1 > $O/bench_config3.json 2> $O/e5
for f in $O/*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['ms_per_step'], json.dumps(d.get('power'))[:900])
PY
done
