#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4g; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
timeout 300 python bench.py --no-extras --profile slow --rate 44100 --steps 60 > $O/bench_profile_slow_44100.json 2> $O/bench.err
APTGPU_MODE_GENERIC_NOTE=1 timeout 300 python bench.py --no-extras --profile slow --rate 44100 --steps 20 --mode generic > $O/bench_profile_slow_44100_generic.json 2>> $O/bench.err
ls -la $O
