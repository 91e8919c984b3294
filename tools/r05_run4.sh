#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_image_structure.py tests/test_gpu_batch.py -x -q -m gpu -k "profile_kernels or specialisations or phase or table or geometry or two_processes" > gpurun_out/r05/run4_tests.txt 2>&1
tail -3 gpurun_out/r05/run4_tests.txt
for spec in "standard 44100" "standard 22050" "standard 11025" "fast 48000"; do
  set -- $spec
  timeout 600 python tools/sweep.py --rate $2 --profile $1 --inputs 4 --steps 12 --configs "strict:16:1,strict:16:1,fast:16:1" > gpurun_out/r05/run4_sweep_$1_$2.txt 2>&1
  grep config gpurun_out/r05/run4_sweep_$1_$2.txt | sed "s/^/$spec: /"
done
