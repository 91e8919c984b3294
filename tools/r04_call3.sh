#!/bin/bash
# round 4, third GPU call: what one more launch of each kernel costs a pipelined step (probe library), orbit / gather
# variants, the PHASE lane permutation, kernel trace of the pipelined run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4c; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
export APTGPU_LIB=$R/noaa_apt_amd/libaptgpu_probe.so
V="strict:16:3"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_WORDS=2"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_WORDS=3"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_ORBIT=2"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_ORBIT=3"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_GATHER=2"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_GATHER=3"
V="$V,strict:16:3:APTGPU_DEBUG_REPEAT_FRONT=2"
V="$V,strict:16:3:APTGPU_DEBUG_SKIP=7"
V="$V,strict:16:3:APTGPU_DEBUG_SKIP=7;APTGPU_FRONT_SERIAL=0"
V="$V,strict:16:3:APTGPU_FRONT_SERIAL=0"
V="$V,strict:16:3"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=1"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2;APTGPU_ORBIT_THREADS=256;APTGPU_ORBIT_LDS=0"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2;APTGPU_ORBIT_LDS=0"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2;APTGPU_FUSED_LDS_PAD=2048"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2;APTGPU_FUSED_LDS_PAD=2048;APTGPU_ORBIT_THREADS=256;APTGPU_ORBIT_LDS=0"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=2;APTGPU_DEBUG_REPEAT_GATHER=2"
V="$V,strict:16:3"
timeout 600 python tools/sweep.py --configs "$V" --steps 200 --warmup 20 --inputs 16 > $O/sweep_repeat.txt 2> $O/sweep_repeat.err
unset APTGPU_LIB
V="strict:16:3:APTGPU_PHASE_PERM=0,strict:16:3,fast:16:3:APTGPU_PHASE_PERM=0,fast:16:3,strict:16:3:APTGPU_PHASE_PERM=0,strict:16:3"
timeout 300 python tools/sweep.py --rate 44100 --configs "$V" --steps 100 --warmup 10 --inputs 16 > $O/sweep_44100.txt 2> $O/sweep_44100.err
timeout 300 python tools/sweep.py --rate 22050 --configs "$V" --steps 100 --warmup 10 --inputs 16 > $O/sweep_22050.txt 2> $O/sweep_22050.err
V="strict:16:3:APTGPU_PHASE_FIRST=1;APTGPU_PHASE_PERM=0,strict:16:3:APTGPU_PHASE_FIRST=1,strict:16:3"
timeout 300 python tools/sweep.py --rate 32000 --configs "$V" --steps 100 --warmup 10 --inputs 16 > $O/sweep_32000.txt 2> $O/sweep_32000.err
# kernel trace of the pipelined run (strict, 16 per call, three calls in flight)
cd /tmp && export TMPDIR=/tmp
APTGPU_GATHER_ITERS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/sweep.py --configs strict:16:3 --steps 30 --inputs 16 > $O/trace.log 2>&1
f=$(ls $O/trace/*/*kernel_trace.csv | head -1); cp $f $O/kernel_trace.csv; rm -rf $O/trace
cd $R; ls -la $O
