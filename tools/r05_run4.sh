#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
for spec in "standard 44100" "standard 22050" "standard 11025" "fast 48000" "slow 44100" "slow 11025"; do
  set -- $spec
  timeout 600 python tools/sweep.py --rate $2 --profile $1 --inputs 4 --steps 12 --configs "strict:16:1,strict:16:1:APTGPU_PHASE_TT=0,strict:16:1,strict:16:1:APTGPU_PHASE_TT=0" > gpurun_out/r05/run4_sweep_$1_$2.txt 2>&1
  grep config gpurun_out/r05/run4_sweep_$1_$2.txt | sed "s/^/$spec: /"
done
