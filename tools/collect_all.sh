#!/bin/bash
# tools/collect_all.sh [round] [stage...] — everything under profiles/ for a round in ONE call on the GPU box: the counter
# passes, their summaries installed into profiles/ (so that bench.py finds them, stamped with the hashes of the sources
# it runs), then the bench lines of every configuration.  Afterwards, locally: tools/install_profiles.sh copies
# gpurun_out/prof/* into profiles/ under the round's names.
# Stages (default: all of them, ~14 GPU-minutes; each logs its seconds to gpurun_out/prof/stage_seconds.txt):
#   counters  front-end kernel stats, HBM traffic, SQ counters (strict + fast)        ~4 min   -> must precede `bench`
#   bench     the bench line of every configuration                                   ~5 min
#   chain     SQ counters of k_sync_words / slots / orbit / gather                    ~1.5 min
#   others    SQ counters of the PHASE (1 / 2 / 4 branches) / TABLE / 96 kHz / profile front ends  ~4 min
#   power     socket power, clocks, joules per call (amd-smi)                         ~1.5 min
#   pipeline  marginal cost of each kernel, shape variants, the LDS microbenchmark     ~1.5 min
# An edit confined to one group of kernels (tools/csrc_hash.py --groups) needs only that group's stages again.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
ROUND=${1:-r05}; shift
STAGES=${*:-counters bench chain others power pipeline}
mkdir -p gpurun_out/prof
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
timed() { local t0=$SECONDS; "$@"; echo "$1 $2 $3: $((SECONDS - t0)) s" >> gpurun_out/prof/stage_seconds.txt; }
stage_counters() {
  bash tools/collect_profiles.sh counters > gpurun_out/prof_counters.log 2>&1
  bash tools/install_profiles.sh $ROUND > /dev/null 2>&1
}
stage_bench() { bash tools/collect_profiles.sh bench > gpurun_out/prof_bench.log 2>&1; }
stage_chain() {
  bash tools/collect_sq_kernel.sh words:k_sync_words,slots:k_sync_slots,orbit:k_sync_orbit,gather:k_gather_rows 1 --inputs 16 --configs strict:16:1 > gpurun_out/prof_sq_kernels.log 2>&1
}
stage_others() {
  # the phase-resident stage 1 with one / two / four branches per thread (16 recordings per launch: the bench's shape),
  # the table-driven form it replaced as the default, 96 kHz, the other stock profiles
  bash tools/collect_sq_kernel.sh phase:k_fused 1 --rate 44100 --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh phase2:k_fused 1 --rate 22050 --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh phase4:k_fused 1 --rate 11025 --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh table:k_fused 0 --rate 11025 --inputs 16 --configs "strict:16:1:APTGPU_PHASE_FIRST=0" >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh 96k:k_fused 1 --rate 96000 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh profile_fast:k_fused 0 --profile fast --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh profile_slow:k_fused 0 --profile slow --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh slow_44100:k_fused 0 --profile slow --rate 44100 --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
  bash tools/collect_sq_kernel.sh strict16:k_fused 0 --inputs 16 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
}
stage_power() {
  timeout 200 python tools/power_regimes.py > gpurun_out/prof/power_regimes.txt 2> gpurun_out/prof/power_regimes.err
  timeout 100 python tools/rates_power.py > gpurun_out/prof/rates_power.txt 2> gpurun_out/prof/rates_power.err
  bash tools/energy_per_call.sh > gpurun_out/prof/energy_per_call.log 2>&1
  bash tools/energy_marginal.sh > gpurun_out/prof/energy_marginal.log 2>&1
}
stage_pipeline() {
  bash tools/pipeline_costs.sh > gpurun_out/prof/pipeline_costs.txt 2> gpurun_out/prof/pipeline_costs.err
  ./tools/ubench/lds_bw.bin > gpurun_out/prof/ubench_lds_bw.txt 2>&1
  (./tools/ubench/lds_pat.bin tools/ubench/lds_pat_model.txt; ./tools/ubench/lds_pat.bin tools/ubench/lds_pat_44100.txt) > gpurun_out/prof/ubench_lds_pat.txt 2>&1
}
for S in counters bench chain others power pipeline; do
  if has $S; then timed stage_$S; fi
done
cat gpurun_out/prof/stage_seconds.txt
