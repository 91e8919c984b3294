#!/usr/bin/env python3
"""Host-fed path on one GPU: BASELINE config 4's per-GPU share (32 recordings of 15 min at 48 kHz, tiled from a few
distinct ones) through aptgpu_decode_batch from pageable host memory; and the one-shot aptgpu_decode()."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
from noaa_apt_amd.testing.wavfile import make_wav

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 900.0
base = [synth_apt(48000, secs, seed=1000 + j) for j in range(4)]
recs = [base[j % 4].copy() for j in range(n_rec)]          # distinct host buffers (pageable)
wavs = [make_wav(base[j % 4].astype(np.int16), 48000) for j in range(4)]
wavs = [bytes(wavs[j % 4]) for j in range(n_rec)]
st = apt.Settings()
rate = apt.Rate.hz(48000)
ctx = apt.Context(device=0)
ref, _ = apt.decode(ctx, st, base[0], rate, True), None
for key, inputs in (("f32", recs), ("pcm16_wav", wavs)):
    for workers in (1, 2):
        for per_call in (4, 8, 16):
            apt.decode_batch(ctx, st, inputs[:2 * per_call], rate, True, devices=(0,) * workers, recordings_per_call=per_call)
            t0 = time.perf_counter()
            got, res, hst = apt.decode_batch(ctx, st, inputs, rate, True, devices=(0,) * workers,
                                             recordings_per_call=per_call, return_stats=True)
            t1 = time.perf_counter()
            ok = all(not isinstance(g, Exception) for g in got)
            same = bool(ok and np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32) if isinstance(ref, tuple) else ref.view(np.uint32)))
            moved = hst.h2d_bytes + hst.d2h_bytes
            print(json.dumps({"input": key, "workers": workers, "per_call": per_call, "seconds": round(hst.seconds, 4),
                              "wall_python": round(t1 - t0, 4), "Gsamples_per_s": round(hst.samples / hst.seconds / 1e9, 2),
                              "pcie_GBps": round(moved / hst.seconds / 1e9, 2), "frac_of_63": round(moved / hst.seconds / 63e9, 3),
                              "ok": ok, "rows_identical": same}), flush=True)
# one-shot decode of a ten-minute recording
x = synth_apt(48000, 600, seed=2)
for _ in range(3):
    apt.decode(ctx, st, x, rate, True)
t0 = time.perf_counter()
for _ in range(10):
    apt.decode(ctx, st, x, rate, True)
t1 = time.perf_counter()
print(json.dumps({"one_shot_decode_ms": round(1e2 * (t1 - t0), 3), "cache": apt.cache_info()}))
