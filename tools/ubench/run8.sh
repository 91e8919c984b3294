D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_batch.py tests/test_reference_fixture.py -x -q -m gpu > $D/pytest_quick.txt 2>&1
python tools/sweep.py --configs strict:16:3,fast:16:3 --steps 100 --inputs 16 2>/dev/null | grep ms_per > $D/sweep.txt
