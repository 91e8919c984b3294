D=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D/trace -- python $GRAFT_REPO_ROOT/tools/sweep.py --configs strict:16:3 --steps 30 --inputs 16 > $D/trace.log 2>&1
f=$(ls $D/trace/*/*kernel_trace.csv | head -1); cp $f $D/kernel_trace.csv; rm -rf $D/trace
