#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4z; mkdir -p $O
python -c "import torch" 2>/dev/null
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape_full.json 2> $O/bench_driver_shape_full.err ) 2> $O/bench_driver_shape_full.time
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-extras --no-cpu-baseline --no-single-launch --steps 60 --warmup 10 > $O/trace_bench.json 2> $O/trace.err
f=$(ls $O/trace/*/*kernel_trace.csv | head -1); cp $f $O/kernel_trace.csv; rm -rf $O/trace
ls -la $O
