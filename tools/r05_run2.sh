#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
bash tools/collect_sq_kernel.sh phase44_b:k_fused 0 --rate 44100 --inputs 16 --configs strict:16:1 
bash tools/collect_sq_kernel.sh phase2_22_b:k_fused 0 --rate 22050 --inputs 16 --configs strict:16:1
bash tools/collect_sq_kernel.sh phase4_11_b:k_fused 0 --rate 11025 --inputs 16 --configs strict:16:1
bash tools/collect_sq_kernel.sh strict48_b:k_fused 0 --rate 48000 --inputs 16 --configs strict:16:1
