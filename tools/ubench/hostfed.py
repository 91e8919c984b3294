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
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for key, inputs in (("f32", recs), ("pcm16_wav", wavs)):
    for workers in (1, 2):
        for per_call in (8, 16):
            # warm-up over the whole list: every worker has leased (created) its session, host pages are touched
            apt.decode_batch(ctx, st, inputs, rate, True, devices=(0,) * workers, recordings_per_call=per_call)
            runs = []
            for _ in range(REPS):
                got, res, hst = apt.decode_batch(ctx, st, inputs, rate, True, devices=(0,) * workers,
                                                 recordings_per_call=per_call, return_stats=True)
                ok = all(not isinstance(g, Exception) for g in got)
                same = bool(ok and np.array_equal(got[0].view(np.uint32), ref.view(np.uint32)))
                runs.append((hst.seconds, hst.gate_wait_seconds, hst.setup_seconds, hst.sessions_created, hst.workers_pinned, ok and same))
                moved = hst.h2d_bytes + hst.d2h_bytes
                del got
            secs = sorted(r[0] for r in runs)
            print(json.dumps({"input": key, "workers": workers, "per_call": per_call,
                              "seconds_min_median_max": [round(secs[0], 4), round(secs[len(secs) // 2], 4), round(secs[-1], 4)],
                              "frac_of_63_median": round(moved / secs[len(secs) // 2] / 63e9, 3),
                              "frac_of_63_best": round(moved / secs[0] / 63e9, 3),
                              "gate_wait_s": [round(r[1], 4) for r in runs], "setup_s": [round(r[2], 4) for r in runs],
                              "sessions_created": [r[3] for r in runs], "workers_pinned": [r[4] for r in runs],
                              "all_ok_and_identical": all(r[5] for r in runs)}), flush=True)
print(json.dumps({"host_affinity": apt.host_affinity(0)}))
# one-shot decode of a ten-minute recording
x = synth_apt(48000, 600, seed=2)
for _ in range(3):
    apt.decode(ctx, st, x, rate, True)
t0 = time.perf_counter()
for _ in range(10):
    apt.decode(ctx, st, x, rate, True)
t1 = time.perf_counter()
print(json.dumps({"one_shot_decode_ms": round(1e2 * (t1 - t0), 3), "cache": apt.cache_info()}))
