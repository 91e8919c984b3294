D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -x -q -m gpu > $D/pytest_batch.txt 2>&1
python bench.py --config4 --steps 8 > $D/bench_config4.json 2> $D/bench_config4.err
python bench.py --steps 100 --warmup 10 > $D/bench_full.json 2> $D/bench_full.err
