#!/bin/bash
# tools/install_profiles.sh [round]: copy the summaries tools/collect_profiles.sh left under gpurun_out/prof/ into
# profiles/ under this round's names (what bench.py and the docs cite).
R=${1:-r05}
P=gpurun_out/prof
cd "$(dirname "$0")/.." || exit 1
for MODE in strict fast; do
  [ -f $P/hbm_traffic_$MODE.json ] && cp $P/hbm_traffic_$MODE.json profiles/${R}_hbm_traffic_$MODE.json
  [ -f $P/sq_$MODE/summary.json ] && cp $P/sq_$MODE/summary.json profiles/${R}_sq_counters_$MODE.json
  [ -f $P/stats_$MODE.kernel_stats.csv ] && cp $P/stats_$MODE.kernel_stats.csv profiles/${R}_kernel_stats_$MODE.csv
  [ -f $P/stats_${MODE}_streams1.kernel_stats.csv ] && cp $P/stats_${MODE}_streams1.kernel_stats.csv profiles/${R}_kernel_stats_${MODE}_streams1.csv
  [ -f $P/stats_${MODE}_bench.json ] && cp $P/stats_${MODE}_bench.json profiles/${R}_bench_${MODE}_under_rocprof.json
  [ -f $P/stats_${MODE}_streams1_bench.json ] && cp $P/stats_${MODE}_streams1_bench.json profiles/${R}_bench_${MODE}_streams1_under_rocprof.json
done
for f in $P/sq_*.json; do
  [ -f "$f" ] && cp "$f" profiles/${R}_sq_counters_$(basename "$f" | sed -e 's/^sq_//')
done
for f in $P/hbm_*.json; do
  case "$f" in *hbm_traffic_*) continue;; esac
  [ -f "$f" ] && cp "$f" profiles/${R}_hbm_$(basename "$f" | sed -e 's/^hbm_//')
done
for f in $P/bench_*.json; do
  [ -f "$f" ] && cp "$f" profiles/${R}_$(basename "$f")
done
for f in power_regimes rates_power pipeline_costs ubench_lds_bw ubench_lds_pat stage_seconds; do
  [ -f $P/$f.txt ] && cp $P/$f.txt profiles/${R}_$f.txt
done
ls profiles | grep "^${R}_" | wc -l
