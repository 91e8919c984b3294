#!/bin/bash
# round 5, GPU call: the PHASE stage 1 (thread assignment lists by half-wave, interior tile loads, software-pipelined taps, NQ branches per thread)
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py tests/test_reference_fixture.py tests/test_gpu_bounds.py -x -q -m gpu -k "phase or table or profile_kernels or specialisations or nonfinite_samples or wav or bounds or fixture" > gpurun_out/r05/run1_tests.txt 2>&1
tail -5 gpurun_out/r05/run1_tests.txt
timeout 600 python tools/sweep.py --rate 44100 --inputs 4 --steps 30 --configs "strict:16:1,fast:16:1,strict:16:1:APTGPU_PHASE_IDENTITY=1" > gpurun_out/r05/run1_sweep_44100.txt 2>&1
cat gpurun_out/r05/run1_sweep_44100.txt
timeout 600 python tools/sweep.py --rate 22050 --inputs 4 --steps 30 --configs "strict:16:1,strict:16:1:APTGPU_PHASE_IDENTITY=1,strict:16:1:APTGPU_PHASE_WIDE=1,fast:16:1" > gpurun_out/r05/run1_sweep_22050.txt 2>&1
cat gpurun_out/r05/run1_sweep_22050.txt
timeout 600 python tools/sweep.py --rate 11025 --inputs 4 --steps 30 --configs "strict:16:1,strict:16:1:APTGPU_PHASE_FIRST=0,fast:16:1" > gpurun_out/r05/run1_sweep_11025.txt 2>&1
cat gpurun_out/r05/run1_sweep_11025.txt
timeout 600 python tools/sweep.py --rate 48000 --profile fast --inputs 4 --steps 30 --configs "strict:16:1,strict:16:1:APTGPU_PHASE_IDENTITY=1,fast:16:1" > gpurun_out/r05/run1_sweep_fastp.txt 2>&1
cat gpurun_out/r05/run1_sweep_fastp.txt
