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
bash tools/collect_sq_kernel.sh words k_sync_words --inputs 4 --configs strict:16:1 > gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh slots k_sync_slots --inputs 4 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh orbit k_sync_orbit --inputs 4 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh gather k_gather_rows --inputs 4 --configs strict:16:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh phase k_fused --rate 44100 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh phase512 k_fused --rate 22050 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh table k_fused --rate 11025 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh 96k k_fused --rate 96000 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh profile_fast k_fused --profile fast --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh profile_slow k_fused --profile slow --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
bash tools/collect_sq_kernel.sh any_fast_11025 k_fused_any --profile fast --rate 11025 --inputs 2 --configs strict:1:1 >> gpurun_out/prof_sq_kernels.log 2>&1
