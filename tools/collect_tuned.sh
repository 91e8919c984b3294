#!/bin/bash
# tools/collect_tuned.sh <round-tag>: the settings cliff.  bench.py at 48 kHz with a tuned
# resample_atten / resample_delta_freq — tap counts the exact-count kernels are not compiled for — on the padded strict
# kernel (kModeStrictPad, the default for such plans) and, APTGPU_FUSED_PAD=0 / APTGPU_FAST_MFMA=0, on what served them
# until round 6 (k_fused_any); stock settings beside them.  -> gpurun_out/prof/<tag>_bench_tuned_*.json + a table.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
run() {  # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 300 python bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline "$@" > $O/${TAG}_bench_tuned_$name.json 2> $O/${TAG}_bench_tuned_$name.err
  python - $O/${TAG}_bench_tuned_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    w = d["config"]["workload"]
    taps = w[w.index("resample"):w.index("low-pass") + 8].replace("-> AM envelope -> ", "/ ") if "resample" in w else ""
    print(f"{sys.argv[2]:28s} {taps:46s} ms/step {d['ms_per_step']:.4f}  front end alone {d['roofline']['kernel_avg_ms']:.4f} ms  frac {d['roofline']['frac']:.4f}  parity: {str(d.get('parity'))[:60]}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run stock APTGPU_QUIET=1 --
for kv in resample_atten=29 resample_atten=31 resample_delta_freq=900 resample_delta_freq=1100; do
  run ${kv}_pad APTGPU_QUIET=1 -- --set $kv
  run ${kv}_any APTGPU_QUIET=1 APTGPU_FUSED_PAD=0 -- --set $kv
done
# a tuned demodulation_atten moves the LOW-PASS length (37 taps at 25 dB): kModeStrictPad2 against k_fused_any
for kv in demodulation_atten=24 demodulation_atten=26 demodulation_atten=29; do
  run ${kv}_pad2 APTGPU_QUIET=1 -- --set $kv
  run ${kv}_any APTGPU_QUIET=1 APTGPU_FUSED_PAD=0 -- --set $kv
done
run atten31_demod26_pad2 APTGPU_QUIET=1 -- --set resample_atten=31 --set demodulation_atten=26
# ... and at the rates a sound card records at (the PHASE kernels' kModeStrictPad2 instantiations)
for r in 44100 11025; do
  run ${r}_stock APTGPU_QUIET=1 -- --rate $r
  run ${r}_demodulation_atten=26_pad2 APTGPU_QUIET=1 -- --rate $r --set demodulation_atten=26
  run ${r}_demodulation_atten=26_any APTGPU_QUIET=1 APTGPU_FUSED_PAD=0 -- --rate $r --set demodulation_atten=26 --steps 30
done
run fast_stock APTGPU_QUIET=1 -- --mode fast
run fast_atten31_mfma APTGPU_QUIET=1 -- --mode fast --set resample_atten=31
run fast_atten31_any APTGPU_QUIET=1 APTGPU_FAST_MFMA=0 APTGPU_FUSED_PAD=0 -- --mode fast --set resample_atten=31
rm -f $O/${TAG}_bench_tuned_*.err
