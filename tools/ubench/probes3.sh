D=gpurun_out/$1; mkdir -p $D
for P in 17 0; do
  APTGPU_PROBE_STOP=$P python tools/sweep.py --configs strict:16:1,strict:16:3,strict:1:1 --steps 40 --inputs 16 2>/dev/null | grep alone_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('strict probe', $P, d['config'], d['alone_ms_per_call'].get('fused_front_end'), 'ms/rec', d['ms_per_recording'], d['status'][:2], d['rows'])"
done > $D/probes3.txt 2>&1
for P in 9 0; do
  APTGPU_PROBE_STOP=$P python tools/sweep.py --configs fast:16:1,fast:16:3,fast:1:1 --steps 40 --inputs 16 2>/dev/null | grep alone_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('fast probe', $P, d['config'], d['alone_ms_per_call'].get('fused_front_end'), 'ms/rec', d['ms_per_recording'], d['status'][:2], d['rows'])"
done >> $D/probes3.txt 2>&1
APTGPU_PROBE_STOP=17 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode_bitexact" > $D/pytest_persist.txt 2>&1
