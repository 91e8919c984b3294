mkdir -p gpurun_out/r3a
./tools/ubench/rates.bin > gpurun_out/r3a/rates.txt 2>&1
python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r3a/bench_strict.json 2> gpurun_out/r3a/bench_strict.err
python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --mode fast > gpurun_out/r3a/bench_fast.json 2> gpurun_out/r3a/bench_fast.err
