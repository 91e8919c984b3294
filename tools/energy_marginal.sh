#!/bin/bash
# tools/energy_marginal.sh — joules ONE more launch of each chain kernel adds to a pipelined call (probe library), PCM16 input, config 3,
# the fast / slow profiles.  One gpurun call; profiles/r04_energy_marginal.txt.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/energy; mkdir -p $O
P=$R/noaa_apt_amd/libaptgpu_probe.so
python3 -c "import torch" 2>/dev/null
echo "## base (probe library, nothing repeated)" > $O/energy_marginal.txt
APTGPU_PROBE_LIB=$P python3 tools/sweep.py --power --inputs 16 --steps 800 --warmup 50 --configs strict:16:3 >> $O/energy_marginal.txt 2>$O/e0
for K in WORDS ORBIT GATHER; do
echo "## APTGPU_DEBUG_REPEAT_$K=2" >> $O/energy_marginal.txt
env APTGPU_PROBE_LIB=$P APTGPU_DEBUG_REPEAT_$K=2 python3 tools/sweep.py --power --inputs 16 --steps 800 --warmup 50 --configs strict:16:3 >> $O/energy_marginal.txt 2>>$O/e0
done
echo "## PCM16 payloads as input (product library)" >> $O/energy_marginal.txt
python3 tools/sweep.py --power --pcm16 --inputs 16 --steps 800 --warmup 50 --configs strict:16:3 >> $O/energy_marginal.txt 2>>$O/e0
echo "## config 3: one hour at 96 kHz, one recording per call" >> $O/energy_marginal.txt
python3 tools/sweep.py --power --rate 96000 --seconds 3600 --inputs 2 --steps 800 --warmup 50 --configs strict:1:3 >> $O/energy_marginal.txt 2>>$O/e0
echo "## fast / slow profile at 48 kHz" >> $O/energy_marginal.txt
python3 tools/sweep.py --power --profile fast --inputs 16 --steps 400 --warmup 30 --configs strict:16:3 >> $O/energy_marginal.txt 2>>$O/e0
python3 tools/sweep.py --power --profile slow --inputs 16 --steps 300 --warmup 30 --configs strict:16:3 >> $O/energy_marginal.txt 2>>$O/e0
python3 - $O/energy_marginal.txt <<'PY'
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('#'): print(l); continue
    d=json.loads(l)
    if 'config' in d:
        p=d.get('power') or {}
        print(d['config'], 'ms/call', round(d['ms_per_recording']*int(d['config'].split(':')[1]),4), 'W', p.get('socket_w_mean_by_energy_counter'), 'clk', (p.get('gfxclk_mhz') or {}).get('mean_over_xcds'), 'thr', p.get('power_limit_throttled_frac'), 'J/call', p.get('joules_per_call'))
PY
