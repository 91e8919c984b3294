D=gpurun_out/$1; mkdir -p $D
for S in 0 1 2 4 3 7; do
  APTGPU_DEBUG_SKIP=$S python tools/sweep.py --configs strict:16:3 --steps 100 --inputs 16 2>/dev/null | grep ms_per | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('skip', $S, 'ms/call', round(d['ms_per_recording']*16, 4), d['alone_ms_per_call'])"
done > $D/skip.txt 2>&1
