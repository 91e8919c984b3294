#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4f; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
timeout 300 python bench.py --no-extras > $O/bench_strict.json 2> $O/bench_strict.err
timeout 300 python bench.py --no-extras --batch 1 > $O/bench_batch1.json 2>> $O/bench_strict.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
ls -la $O
