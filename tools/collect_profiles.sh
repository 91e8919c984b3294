set -x
cd /tmp && export TMPDIR=/tmp
python -c "import torch"
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/p_stats_bench.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- python $R/bench.py --no-cpu-baseline --steps 12 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- python $R/bench.py --no-cpu-baseline --steps 12 --warmup 2 > /dev/null 2>&1
cd $R
timeout 300 python bench.py > gpurun_out/bench_default.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 5 --no-extras > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err
tail -c 600 gpurun_out/bench_torchrun.json
ls gpurun_out/p_fetch/*/ gpurun_out/p_write/*/ | head
