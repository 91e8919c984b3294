#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05/full_tests.txt 2>&1
tail -8 gpurun_out/r05/full_tests.txt
