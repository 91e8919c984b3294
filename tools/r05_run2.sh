#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
bash tools/collect_sq_kernel.sh phase44:k_fused 0 --rate 44100 --inputs 2 --configs strict:1:1 
bash tools/collect_sq_kernel.sh phase2_22:k_fused 0 --rate 22050 --inputs 2 --configs strict:1:1
bash tools/collect_sq_kernel.sh phase4_11:k_fused 0 --rate 11025 --inputs 2 --configs "strict:1:1:APTGPU_PHASE_FIRST=1"
ls gpurun_out/prof/
