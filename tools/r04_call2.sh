#!/bin/bash
# round 4, second GPU call: six front-end workgroups per CU against five (LDS pad), with the chain variants; parity of
# the changed front ends; host-fed path five times per leg.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4b; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
V="strict:16:3:APTGPU_FUSED_LDS_PAD=2048"
V="$V,strict:16:3"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=4"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=8"
V="$V,strict:16:2"
V="$V,strict:16:4"
V="$V,strict:16:3:APTGPU_FUSED_LDS_PAD=2048"
V="$V,strict:16:3"
V="$V,fast:16:3:APTGPU_FUSED_LDS_PAD=2048"
V="$V,fast:16:3"
V="$V,fast:16:3:APTGPU_GATHER_ITERS=4"
V="$V,fast:16:3:APTGPU_FUSED_LDS_PAD=2048"
V="$V,fast:16:3"
V="$V,strict:1:6:APTGPU_FUSED_LDS_PAD=2048"
V="$V,strict:1:6"
timeout 500 python tools/sweep.py --configs "$V" --steps 200 --warmup 20 --inputs 16 > $O/sweep_ab.txt 2> $O/sweep_ab.err
V="strict:16:3:APTGPU_FUSED_LDS_PAD=2048,strict:16:3,fast:16:3"
timeout 300 python tools/sweep.py --pcm16 --configs "$V" --steps 200 --warmup 20 --inputs 16 > $O/sweep_pcm16.txt 2> $O/sweep_pcm16.err
timeout 300 python tools/sweep.py --rate 11025 --configs "$V" --steps 200 --warmup 20 --inputs 16 > $O/sweep_11025.txt 2> $O/sweep_11025.err
timeout 300 python tools/ubench/hostfed.py 32 900 > $O/hostfed.txt 2> $O/hostfed.err
ls -la $O
