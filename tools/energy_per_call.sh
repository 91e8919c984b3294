#!/bin/bash
# tools/energy_per_call.sh — bench lines with `power`, then joules per call (tools/sweep.py --power) of the whole pipeline, one call in
# flight, the front ends alone and twice (probe library), 44.1 kHz.  One gpurun call; profiles/r04_energy_per_call.txt, r04_bench_power_*.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/energy; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch > $O/bench_strict_600.json 2> $O/e1
python3 bench.py --no-extras --no-cpu-baseline --no-single-launch --mode fast > $O/bench_fast_600.json 2> $O/e3
# energy per call by what runs: the whole pipeline, one call at a time, the front ends alone, the chain alone
python3 tools/sweep.py --power --inputs 16 --steps 1200 --warmup 50 --configs "strict:16:3,strict:16:1,fast:16:3" > $O/sweep_power.txt 2> $O/e4
APTGPU_PROBE_LIB=$R/noaa_apt_amd/libaptgpu_probe.so APTGPU_DEBUG_SKIP=7 python3 tools/sweep.py --power --inputs 16 --steps 1200 --warmup 50 --configs "strict:16:3,fast:16:3" > $O/sweep_power_front_only.txt 2> $O/e5
APTGPU_PROBE_LIB=$R/noaa_apt_amd/libaptgpu_probe.so APTGPU_DEBUG_REPEAT_FRONT=2 python3 tools/sweep.py --power --inputs 16 --steps 800 --warmup 50 --configs "strict:16:3" > $O/sweep_power_front_twice.txt 2> $O/e6
python3 tools/sweep.py --power --inputs 16 --steps 400 --warmup 50 --rate 44100 --configs "strict:16:3" > $O/sweep_power_44100.txt 2> $O/e7
for f in $O/bench*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['ms_per_step'], json.dumps(d.get('power'))[:1200])
PY
done
cut -c1-120 $O/sweep_power*.txt; grep -h -o '"power".*' $O/sweep_power*.txt
