#!/bin/bash
# tools/r04_call1.sh — round 4, first GPU call: the GPU test suite on the changed kernels, A/B sweeps of the pipeline's
# shape, the shader clock by regime, bench lines of the fast / slow profiles, SQ counters of the PHASE stage 1.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4a; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
# A/B: one process, inputs generated once
V="strict:16:3:APTGPU_WORDS_DPP=0"
V="$V,strict:16:3"
V="$V,strict:16:3:APTGPU_ORBIT_THREADS=256"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=4"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=16"
V="$V,strict:16:3:APTGPU_GATHER_ITERS=16;APTGPU_ORBIT_THREADS=256"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_GATHER_ITERS=16;APTGPU_ORBIT_THREADS=256"
V="$V,strict:16:3:APTGPU_WORDS_DPP=0"
V="$V,strict:16:3"
V="$V,fast:16:3:APTGPU_WORDS_DPP=0"
V="$V,fast:16:3"
V="$V,fast:16:3:APTGPU_FRONT_STREAM=1;APTGPU_GATHER_ITERS=16;APTGPU_ORBIT_THREADS=256"
timeout 400 python tools/sweep.py --configs "$V" --steps 200 --warmup 20 --inputs 16 > $O/sweep_ab.txt 2> $O/sweep_ab.err
V="strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=32"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=64"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=32;APTGPU_FRONT_EXCL=1"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=64;APTGPU_FRONT_EXCL=1"
V="$V,strict:16:3:APTGPU_FRONT_STREAM=1;APTGPU_CHAIN_CUS=96;APTGPU_FRONT_EXCL=1"
V="$V,strict:16:3"
timeout 300 python tools/sweep.py --configs "$V" --steps 200 --warmup 20 --inputs 16 > $O/sweep_cumask.txt 2> $O/sweep_cumask.err
# shader clock by regime
for RG in idle isolated pipeline; do timeout 200 python tools/clock_regimes.py --regime $RG >> $O/sclk.txt 2>> $O/sclk.err; done
APTGPU_LIB=$R/noaa_apt_amd/libaptgpu_probe.so APTGPU_DEBUG_SKIP=7 timeout 200 python tools/clock_regimes.py --regime back_to_back >> $O/sclk.txt 2>> $O/sclk.err
timeout 200 python tools/clock_regimes.py --regime pipeline --mode fast >> $O/sclk.txt 2>> $O/sclk.err
# the fast / slow profiles on the run-time front end (one launch per call now)
timeout 300 python bench.py --no-extras --profile fast --steps 100 > $O/bench_profile_fast.json 2> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile slow --steps 100 > $O/bench_profile_slow.json 2>> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile fast --rate 11025 --steps 100 > $O/bench_profile_fast_11025.json 2>> $O/bench_profile.err
# SQ counters of the PHASE stage 1 (44 100 Hz), one recording per launch
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/sq_phase/pass$i -- \
    python $R/tools/sweep.py --rate 44100 --steps 6 --warmup 2 --inputs 2 --configs strict:1:1 > $O/sq_phase_pass$i.log 2>&1
done
cd $R
python tools/summarize_sq.py k_fused $(ls $O/sq_phase/pass*/*/*counter_collection.csv) --note "rocprofv3 --pmc, three passes, tools/sweep.py --rate 44100 --configs strict:1:1 (PHASE stage 1, one 10-minute recording per launch, one launch in flight)" > $O/sq_counters_phase.json 2>> $O/sq_phase.err
rm -rf $O/sq_phase
ls -la $O
