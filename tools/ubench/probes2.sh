D=gpurun_out/$1; mkdir -p $D
for P in 16 0; do
  APTGPU_PROBE_STOP=$P python tools/sweep.py --configs strict:16:1 --steps 30 --inputs 16 2>/dev/null | grep alone_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('strict probe', $P, d['alone_ms_per_call'].get('fused_front_end'), d['ms_per_recording'])"
done > $D/probes2.txt 2>&1
for P in 8 0; do
  APTGPU_PROBE_STOP=$P python tools/sweep.py --configs fast:16:1 --steps 30 --inputs 16 2>/dev/null | grep alone_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('fast probe', $P, d['alone_ms_per_call'].get('fused_front_end'), d['ms_per_recording'])"
done >> $D/probes2.txt 2>&1
