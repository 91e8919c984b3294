#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "table_switches or ragged_batched_pcm16" > gpurun_out/r05/run4_tests.txt 2>&1
tail -3 gpurun_out/r05/run4_tests.txt
