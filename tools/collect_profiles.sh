#!/bin/bash
# tools/collect_profiles.sh — everything under profiles/ for this round, on the GPU box (one MI355X).
# rocprofv3 is run from /tmp with TMPDIR=/tmp; counters in their own passes with --kernel-trace only.
# Two phases (bench.py quotes the counter summaries committed under profiles/, so they come first):
#   collect_profiles.sh counters   rocprofv3 kernel stats, HBM traffic, SQ counters -> gpurun_out/prof/
#   (copy hbm_traffic_*.json / sq_*/summary.json to profiles/r04_*.json — tools/install_profiles.sh — then)
#   collect_profiles.sh bench      the bench lines of every config -> gpurun_out/prof/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
PHASE=${1:-counters}
if [ "$PHASE" = counters ]; then
for MODE in strict fast; do
  # (2) kernel statistics of the bench command: pipelined, and with ONE call in flight (durations then are GPU time per launch)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$MODE -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --no-single-launch --steps 100 > $O/stats_${MODE}_bench.json 2>/dev/null
  APTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${MODE}_streams1 -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --no-single-launch --steps 100 > $O/stats_${MODE}_streams1_bench.json 2>/dev/null
  # (3) HBM traffic per launch (FETCH_SIZE and WRITE_SIZE cannot share a pass)
  APTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$MODE -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --no-single-launch --steps 6 --warmup 2 > /dev/null 2>&1
  APTGPU_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$MODE -- python $R/bench.py --mode $MODE --no-cpu-baseline --no-extras --no-single-launch --steps 6 --warmup 2 > /dev/null 2>&1
  (cd $R && python tools/summarize_pmc.py $(ls $O/fetch_$MODE/*/*counter_collection.csv | head -1) $(ls $O/write_$MODE/*/*counter_collection.csv | head -1) \
     "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only, APTGPU_STREAMS=1) on bench.py --mode $MODE --steps 6: per LAUNCH = per call of 16 recordings of config 2; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)" > $O/hbm_traffic_${MODE}_raw.json && python tools/calibrate_pmc.py $O/hbm_traffic_${MODE}_raw.json 16 1198 28800000 > $O/hbm_traffic_$MODE.json)
  # (4) SQ issue / stall counters of the front end, one recording per launch
  (cd $R && bash tools/collect_sq.sh $MODE gpurun_out/prof/sq_$MODE > /dev/null 2>&1)
  # stamp the counter summaries with the hash of the kernel sources they were collected on (bench.py only quotes a
  # profile whose stamp matches the sources it runs)
  (cd $R && python - <<PY
import json, sys
sys.path.insert(0, "tools")
from csrc_hash import stamp
for f in ("$O/hbm_traffic_$MODE.json", "$O/sq_$MODE/summary.json"):
    try:
        d = stamp(json.load(open(f))); json.dump(d, open(f, "w"), indent=1)
    except Exception as e:
        print("stamp failed", f, e)
PY
  )
  # keep the merge small: the per-dispatch csv files are large
  rm -rf $O/fetch_$MODE $O/write_$MODE $O/sq_$MODE/pass*/
  for d in $O/stats_$MODE $O/stats_${MODE}_streams1; do
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $d.kernel_stats.csv
    rm -rf $d
  done
done
else
for MODE in ${MODES:-strict fast}; do
  # (1) the bench line itself (default shape: 16 recordings per call), and the driver's shape (20 steps)
  (cd $R && python bench.py --mode $MODE > $O/bench_$MODE.json 2> $O/bench_$MODE.err)
  (cd $R && python bench.py --mode $MODE --steps 20 --warmup 5 --no-extras > $O/bench_${MODE}_steps20.json 2>> $O/bench_$MODE.err)
done
# (5) the other configs
(cd $R && python bench.py --no-extras --seconds 900 --batch 32 --inputs 32 --steps 12 --warmup 2 > $O/bench_config4_share.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --mode fast --seconds 900 --batch 32 --inputs 32 --steps 12 --warmup 2 > $O/bench_config4_share_fast.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --rate 96000 --seconds 3600 --batch 1 --inputs 2 --steps 20 --warmup 3 > $O/bench_config3.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --mode fp16taps > $O/bench_fp16taps.json 2>> $O/bench_strict.err)
# APTGPU_MODE_FAST with the resampler on the matrix cores (kModeMfma) at the stock tap count: the A/B of round 6
(cd $R && APTGPU_FAST_MFMA=1 python bench.py --no-extras --no-cpu-baseline --mode fast > $O/bench_fast_mfma.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --batch 1 > $O/bench_strict_batch1.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --config4 > $O/bench_config4.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --rate 44100 > $O/bench_44100.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --rate 11025 > $O/bench_11025.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --rate 22050 > $O/bench_22050.json 2>> $O/bench_strict.err)
# the reference's other two stock profiles (default_settings.toml:120-140)
(cd $R && python bench.py --no-extras --profile fast > $O/bench_profile_fast.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --profile slow > $O/bench_profile_slow.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --profile fast --mode fast > $O/bench_profile_fast_fastmode.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --profile slow --mode fast > $O/bench_profile_slow_fastmode.json 2>> $O/bench_strict.err)
# the stock profiles at the rates recordings come in (round 5: every combination on a k_fused kernel but fast x 96 kHz)
for PR in "fast 11025" "fast 44100" "fast 96000" "slow 11025" "slow 44100" "slow 96000"; do
  set -- $PR
  (cd $R && python bench.py --no-extras --no-cpu-baseline --profile $1 --rate $2 --steps 100 > $O/bench_profile_$1_$2.json 2>> $O/bench_strict.err)
done
(cd $R && python bench.py --no-extras --no-cpu-baseline --rate 44100 --mode fast > $O/bench_44100_fast.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --no-cpu-baseline --rate 22050 --mode fast > $O/bench_22050_fast.json 2>> $O/bench_strict.err)
(cd $R && python bench.py --no-extras --no-cpu-baseline --rate 11025 --mode fast > $O/bench_11025_fast.json 2>> $O/bench_strict.err)
(cd $R && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $O/bench_torchrun_n1.json 2> $O/bench_torchrun.err)
fi
ls $O
