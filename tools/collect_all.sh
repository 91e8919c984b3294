#!/bin/bash
# tools/collect_all.sh — everything under profiles/ for this round in ONE call on the GPU box: the counter passes, their
# summaries installed into profiles/ (so that bench.py finds them, stamped with the hash of the sources it runs), then
# the bench lines of every configuration.  Afterwards, locally: tools/install_profiles.sh copies gpurun_out/prof/* into
# profiles/ under the round's names.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
bash tools/collect_profiles.sh counters > gpurun_out/prof_counters.log 2>&1
bash tools/install_profiles.sh ${1:-r04} > /dev/null 2>&1
bash tools/collect_profiles.sh bench > gpurun_out/prof_bench.log 2>&1
# counters of the other kernels: the chain behind the front end, the PHASE / TABLE / 96 kHz front ends, the fast / slow profiles
bash tools/collect_sq_kernel.sh words:k_sync_words,slots:k_sync_slots,orbit:k_sync_orbit,gather:k_gather_rows 1 --inputs 16 --configs strict:16:1 > gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh phase:k_fused 1 --rate 44100 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh table:k_fused 1 --rate 11025 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh 96k:k_fused 1 --rate 96000 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh profile_fast:k_fused 0 --profile fast --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh profile_slow:k_fused 0 --profile slow --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh phase512:k_fused 0 --rate 22050 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh any_fast_11025:k_fused_any 0 --profile fast --rate 11025 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
# the shader clock by regime, the pipeline's marginal costs and shape variants, the LDS microbenchmark
for RG in idle isolated pipeline; do timeout 200 python tools/clock_regimes.py --regime $RG >> gpurun_out/prof/sclk_regimes.txt 2>> gpurun_out/prof/sclk.err; done
APTGPU_LIB=$R/noaa_apt_amd/libaptgpu_probe.so APTGPU_DEBUG_SKIP=7 timeout 200 python tools/clock_regimes.py --regime back_to_back >> gpurun_out/prof/sclk_regimes.txt 2>> gpurun_out/prof/sclk.err
bash tools/pipeline_costs.sh > gpurun_out/prof/pipeline_costs.txt 2> gpurun_out/prof/pipeline_costs.err
./tools/ubench/lds_bw.bin > gpurun_out/prof/ubench_lds_bw.txt 2>&1
