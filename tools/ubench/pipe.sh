D=gpurun_out/$1; mkdir -p $D
python tools/sweep.py --configs strict:16:3,strict:16:1,strict:16:2 --steps 100 --inputs 16 2>/dev/null | grep ms_per > $D/pipe.txt
python tools/sweep.py --configs strict:16:3,strict:16:1 --steps 100 --inputs 16 --no-sync 2>/dev/null | grep ms_per | sed 's/^/nosync /' >> $D/pipe.txt
