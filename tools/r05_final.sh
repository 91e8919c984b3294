#!/bin/bash
# round 5: the GPU suite, then everything under profiles/ for the round (tools/collect_all.sh), in one call
mkdir -p gpurun_out/r05 gpurun_out/prof
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
rm -f gpurun_out/prof/stage_seconds.txt
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r05/gpu_tests_final.txt 2>&1
tail -3 gpurun_out/r05/gpu_tests_final.txt
bash tools/collect_all.sh r05 "$@"
