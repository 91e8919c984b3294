#!/bin/bash
# tools/pipeline_costs.sh — what the pipelined step (strict, 16 recordings per call, three calls in flight) is made of:
# (1) what ONE MORE launch of each kernel costs it (probe library: every kernel of the chain is idempotent, so the rows
# stay valid), the front ends alone; (2) the shapes that were tried on it, as A/B in one process.  One JSON line per
# configuration (tools/sweep.py); `ms per call` = ms_per_recording x 16.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V="strict:16:3"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_WORDS=2"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_ORBIT=2"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_GATHER=2"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_FRONT=2"
V="$V,strict:16:3:APTGPU_DEBUG_SKIP=7"
V="$V,strict:16:3:APTGPU_DEBUG_SKIP=7;APTGPU_FRONT_SERIAL=0"
V="$V,strict:16:3:APTGPU_FRONT_SERIAL=0"
V="$V,strict:16:3:APTGPU_WORDS_DPP=0"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=0"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=4"
V="$V,strict:16:3:APTGPU_ORBIT_THREADS=256"
V="$V,strict:16:3:APTGPU_ORBIT_THREADS=256;APTGPU_ORBIT_LDS=0"
V="$V,strict:16:3:APTGPU_FUSED_LDS_PAD=2048"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=64"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=64;APTGPU_FRONT_EXCL=1"
V="$V,strict:16:2"
V="$V,strict:16:4"
V="$V,strict:16:3"
APTGPU_PROBE_LIB=$R/noaa_apt_amd/libaptgpu_probe.so timeout 600 python tools/sweep.py --configs "$V" --steps 200 --warmup 20 --inputs 16
