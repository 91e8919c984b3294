D=gpurun_out/$1; mkdir -p $D
python -m pytest tests -x -q -m gpu > $D/pytest_full.txt 2>&1
python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $D/bench_strict.json 2> $D/bench_strict.err
python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --mode fast > $D/bench_fast.json 2> $D/bench_fast.err
python bench.py --rate 96000 --seconds 3600 --batch 1 --inputs 2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $D/bench_c3.json 2> $D/bench_c3.err
