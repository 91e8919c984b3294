#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4e; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
./tools/ubench/lds_phase.bin > $O/lds_phase.txt 2>&1
APTGPU_DEBUG_GEOM=1 timeout 100 python tools/sweep.py --rate 44100 --configs strict:16:3 --steps 20 --inputs 2 > $O/geom.txt 2>&1
timeout 300 python bench.py --no-extras --profile fast --steps 100 > $O/bench_profile_fast.json 2> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile slow --steps 100 > $O/bench_profile_slow.json 2>> $O/bench_profile.err
for RT in 8000 11025 16000 32000; do
  V="strict:16:3,strict:16:3:APTGPU_PHASE_FIRST=1,fast:16:3"
  timeout 200 python tools/sweep.py --rate $RT --configs "$V" --steps 100 --warmup 10 --inputs 16 > $O/sweep_$RT.txt 2> $O/sweep_$RT.err
done
ls -la $O
