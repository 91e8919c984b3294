#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py tests/test_gpu_fast.py -x -q -m gpu -k "profile_kernels or specialisations or wav or fast_mode" > gpurun_out/r05/run4_tests.txt 2>&1
tail -3 gpurun_out/r05/run4_tests.txt
timeout 600 python tools/sweep.py --rate 96000 --profile fast --inputs 4 --steps 12 --configs "strict:16:1,fast:16:1" > gpurun_out/r05/run5_sweep_fast_96000.txt 2>&1
grep config gpurun_out/r05/run5_sweep_fast_96000.txt
timeout 600 python tools/sweep.py --rate 96000 --profile fast --pcm16 --inputs 4 --steps 12 --configs "strict:16:1" > gpurun_out/r05/run5_sweep_fast_96000_pcm.txt 2>&1
grep config gpurun_out/r05/run5_sweep_fast_96000_pcm.txt
