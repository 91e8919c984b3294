D=gpurun_out/$1; mkdir -p $D
for FS in 1 0; do for S in 0 7; do for ST in 3 2; do
  APTGPU_FRONT_SERIAL=$FS APTGPU_DEBUG_SKIP=$S python tools/sweep.py --configs strict:16:$ST --steps 100 --inputs 16 2>/dev/null | grep ms_per | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('front_serial', $FS, 'skip', $S, 'streams', $ST, 'ms/call', round(d['ms_per_recording']*16, 4))"
done; done; done > $D/skip2.txt 2>&1
