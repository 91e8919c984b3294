#!/bin/bash
# round 6: the GPU suite and the smoke test, the rate table, the tuned-filter lines, then everything under profiles/ for the
# round (tools/collect_all.sh), in one call on the GPU box
mkdir -p gpurun_out/r06 gpurun_out/prof
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
rm -f gpurun_out/prof/stage_seconds.txt
(timeout 900 python -m pytest tests -q -m gpu; python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r06/gpu_tests_final.txt 2>&1
tail -4 gpurun_out/r06/gpu_tests_final.txt
timeout 600 python tools/rate_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/rates.txt
bash tools/collect_tuned.sh r06 > gpurun_out/r06/bench_tuned.txt 2>&1
timeout 100 python tools/orbit_stamps.py 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/orbit_stamps.txt
bash tools/collect_all.sh r06 "$@"
