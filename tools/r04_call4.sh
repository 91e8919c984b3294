#!/bin/bash
# round 4, fourth GPU call: the fast / slow profile kernels (parity, bench), PHASE counters with and without the lane
# permutation, PHASE against TABLE at the rates both serve.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4d; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
timeout 300 python bench.py --no-extras --profile fast --steps 100 > $O/bench_profile_fast.json 2> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile slow --steps 100 > $O/bench_profile_slow.json 2>> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile fast --mode fast --steps 100 > $O/bench_profile_fast_fastmode.json 2>> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile slow --mode fast --steps 100 > $O/bench_profile_slow_fastmode.json 2>> $O/bench_profile.err
timeout 300 python bench.py --no-extras --profile fast --rate 96000 --steps 100 > $O/bench_profile_fast_96k.json 2>> $O/bench_profile.err
for RT in 8000 12000 16000 32000; do
  V="strict:16:3,strict:16:3:APTGPU_PHASE_FIRST=1"
  timeout 200 python tools/sweep.py --rate $RT --configs "$V" --steps 100 --warmup 10 --inputs 16 > $O/sweep_$RT.txt 2> $O/sweep_$RT.err
done
cd /tmp && export TMPDIR=/tmp
for PERM in 0 1; do
  APTGPU_PHASE_PERM=$PERM timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/sq_phase$PERM -- \
    python $R/tools/sweep.py --rate 44100 --steps 6 --warmup 2 --inputs 2 --configs strict:1:1 > $O/sq_phase_perm$PERM.log 2>&1
  (cd $R && python tools/summarize_sq.py k_fused $(ls $O/sq_phase$PERM/*/*counter_collection.csv) --note "PHASE stage 1 at 44 100 Hz, APTGPU_PHASE_PERM=$PERM" > $O/sq_phase_perm$PERM.json)
  rm -rf $O/sq_phase$PERM
done
cd $R; ls -la $O
