#!/bin/bash
# tools/collect_profiles.sh — everything under profiles/ for this round, on the GPU box (one MI355X).
# rocprofv3 is run from /tmp with TMPDIR=/tmp; counters in their own passes with --kernel-trace only.
# Results land in gpurun_out/prof/ and are copied to profiles/ by hand (see profiles/README.md).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for MODE in strict fast; do
  # (1) the bench line itself (default shape: 8 recordings per call), and the driver's shape (20 steps)
  (cd $R && python bench.py --mode $MODE > $O/bench_$MODE.json 2> $O/bench_$MODE.err)
  (cd $R && python bench.py --mode $MODE --steps 20 --warmup 5 --no-extras > $O/bench_${MODE}_steps20.json 2>> $O/bench_$MODE.err)
  # (2) kernel statistics of the same command: pipelined, and with ONE call in flight (durations then are GPU time per launch)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$MODE -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --no-single-launch --steps 100 > $O/stats_${MODE}_bench.json 2>/dev/null
  APTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${MODE}_streams1 -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --no-single-launch --steps 100 > $O/stats_${MODE}_streams1_bench.json 2>/dev/null
  # (3) HBM traffic per launch (FETCH_SIZE and WRITE_SIZE cannot share a pass)
  APTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$MODE -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
  APTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$MODE -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
  (cd $R && python tools/summarize_pmc.py $(ls $O/fetch_$MODE/*/*counter_collection.csv | head -1) $(ls $O/write_$MODE/*/*counter_collection.csv | head -1) \
     "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only, APTGPU_STREAMS=1) on bench.py --mode $MODE --steps 6: per LAUNCH = per call of 8 recordings of config 2; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)" > $O/hbm_traffic_$MODE.json)
done
# (4) the other configs
(cd $R && python bench.py --no-extras --seconds 900 --batch 32 --inputs 32 --steps 12 --warmup 2 > $O/bench_config4_share.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --mode fast --seconds 900 --batch 32 --inputs 32 --steps 12 --warmup 2 > $O/bench_config4_share_fast.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --rate 96000 --seconds 3600 --batch 1 --inputs 2 --steps 20 --warmup 3 > $O/bench_config3.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --mode fp16taps > $O/bench_fp16taps.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --batch 1 > $O/bench_strict_batch1.json 2>> $O/bench_strict.err)
(cd $R && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $O/bench_torchrun_n1.json 2> $O/bench_torchrun.err)
ls $O
