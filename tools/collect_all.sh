#!/bin/bash
# tools/collect_all.sh — everything under profiles/ for this round in ONE call on the GPU box: the counter passes, their
# summaries installed into profiles/ (so that bench.py finds them, stamped with the hash of the sources it runs), then
# the bench lines of every configuration.  Afterwards, locally: tools/install_profiles.sh copies gpurun_out/prof/* into
# profiles/ under the round's names.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
bash tools/collect_profiles.sh counters > gpurun_out/prof_counters.log 2>&1
bash tools/install_profiles.sh ${1:-r03} > /dev/null 2>&1
bash tools/collect_profiles.sh bench > gpurun_out/prof_bench.log 2>&1
