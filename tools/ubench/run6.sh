D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_gpu_torchrun.py -x -q -m gpu > $D/pytest_torchrun.txt 2>&1
for R in 44100 22050 11025; do
python bench.py --rate $R --steps 60 --warmup 10 --no-extras --no-cpu-baseline > $D/bench_$R.json 2> $D/bench_$R.err
done
python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --batch 1 > $D/bench_batch1.json 2> $D/bench_batch1.err
