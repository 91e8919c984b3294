#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
timeout 600 python tools/sweep.py --rate 11025 --inputs 4 --steps 30 --configs "strict:16:1,fast:16:1,strict:16:3" > gpurun_out/r05/run3_sweep_11025.txt 2>&1
cat gpurun_out/r05/run3_sweep_11025.txt
