#!/usr/bin/env python3
"""Effective shader clock while the decode pipeline runs, by regime.

A sampler kernel (tools/ubench/clocks.hip -> libclockprobe.so: eight single-wave workgroups, one per XCD if the
dispatcher spreads them) stays resident beside the work and records (s_memtime, s_memrealtime) pairs every ~10 us:
what was taken for shader-clock ticks against a constant 100 MHz.  IT IS NOT: s_memtime (and the period of an s_sleep
loop) tick at a constant rate on gfx950 whatever the gfx clock is — this tool reads 2.40 GHz in every regime while the
SMU runs the XCDs at 2.0 GHz under the pipeline (tools/power_regimes.py, noaa_apt_amd/testing/smu.py: ask the SMU).
Kept for the record and for the per-regime front-end durations it logs.  One process per regime (the probe library latches APTGPU_DEBUG_SKIP on first use):

    python tools/clock_regimes.py --regime isolated      # one call at a time, host synchronisation after each
    APTGPU_PROBE_LIB=noaa_apt_amd/libaptgpu_probe.so APTGPU_DEBUG_SKIP=7 python tools/clock_regimes.py --regime back_to_back
    python tools/clock_regimes.py --regime pipeline      # the bench's timed region: three calls in flight
    python tools/clock_regimes.py --regime idle          # the sampler alone

Prints one JSON line: clock quantiles (MHz) per sampler workgroup (with its XCC id) over the window, the front end's
event-timed duration in that regime, ms per call.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regime", default="pipeline", choices=["idle", "isolated", "back_to_back", "pipeline"])
    ap.add_argument("--mode", default="strict")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--ms", type=float, default=60.0, help="how long the regime runs under the sampler")
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--seconds", type=float, default=600.0)
    args = ap.parse_args()

    import torch
    import noaa_apt_amd as apt
    if os.environ.get("APTGPU_PROBE_LIB"):  # (the probe build: APTGPU_DEBUG_* switches; timing only)
        apt.use_library(os.environ["APTGPU_PROBE_LIB"])
    from noaa_apt_amd.testing.synth import synth_apt

    probe = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libclockprobe.so"))
    probe.clockprobe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    B = args.batch
    xs = [synth_apt(args.rate, args.seconds, seed=2 + 1000 * j) for j in range(B)]
    n = xs[0].size
    d_xs = [torch.from_numpy(v).to(dev) for v in xs]
    modes = {"strict": apt.MODE_STRICT, "fast": apt.MODE_FAST}
    plan = apt.Plan(apt.Settings(), apt.Rate.hz(args.rate), True, max_samples=n, max_batch=B, mode=modes[args.mode])
    cap = int(plan.info.max_rows)
    S = 3
    outs = [[torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in range(B)] for _ in range(S)]
    k = [0]

    def step():
        j = k[0]
        k[0] += 1
        sig = [d_xs[(j + b) % B].data_ptr() for b in range(B)]
        plan.decode_device(sig, [n] * B, [t.data_ptr() for t in outs[j % S]], [cap] * B)

    # warm up (clocks, caches, lazily created buffers), and find out how long a call takes in this regime
    for _ in range(40):
        step()
        if args.regime == "isolated":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
        if args.regime == "isolated":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ms_call = 1e3 * (time.perf_counter() - t0) / 20
    calls = max(8, int(args.ms / ms_call))

    n_wgs, sleeps = 8, 3
    lead_ms = 3.0
    n_samples = int((args.ms + 2 * lead_ms + 8.0) * 1e3 / 10.5)  # ~10.5 us per sample at 2.4 GHz, more at lower clocks
    d_out = torch.zeros(n_wgs * n_samples * 2, dtype=torch.int64, device=dev)
    d_meta = torch.zeros(n_wgs, dtype=torch.int64, device=dev)
    ss = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    rc = probe.clockprobe_launch(C.c_void_p(ss.cuda_stream), C.c_void_p(d_out.data_ptr()), n_wgs, n_samples, sleeps,
                                 C.c_void_p(d_meta.data_ptr()))
    assert rc == 0, rc
    time.sleep(lead_ms * 1e-3)
    plan.enable_timing(2 if args.regime == "isolated" else 1)
    w0 = time.perf_counter()
    if args.regime != "idle":
        for _ in range(calls):
            step()
            if args.regime == "isolated":
                torch.cuda.synchronize()
        plan.synchronize()
    else:
        time.sleep(args.ms * 1e-3)
    w1 = time.perf_counter()
    ktimes = plan.collect_timing() if args.regime != "idle" else {}
    plan.enable_timing(0)
    torch.cuda.synchronize()
    samples = d_out.cpu().numpy().reshape(n_wgs, n_samples, 2).astype(np.int64)
    meta = d_meta.cpu().numpy()
    busy_ms = 1e3 * (w1 - w0)
    per_wg = []
    for g in range(n_wgs):
        t, r = samples[g, :, 0], samples[g, :, 1]
        rel_ms = (r - r[0]) / 1e5  # 100 MHz ticks -> ms
        lo, hi = lead_ms + 0.2 * busy_ms, lead_ms + 0.8 * busy_ms
        sel = np.nonzero((rel_ms >= lo) & (rel_ms <= hi))[0]
        if sel.size < 8:
            per_wg.append({"xcc": int(meta[g] >> 32), "samples": int(sel.size)})
            continue
        a, b = sel[0], sel[-1]
        mean_mhz = float(t[b] - t[a]) / float(r[b] - r[a]) * 100.0
        # per-interval clocks over windows of 16 samples (~170 us)
        step_ = 16
        tt, rr = t[a:b:step_], r[a:b:step_]
        inst = np.diff(tt) / np.maximum(np.diff(rr), 1) * 100.0
        idle = slice(2, max(3, int(0.6 * lead_ms * 1e3 / 10.5)))
        idle_mhz = float(t[idle][-1] - t[idle][0]) / float(max(1, r[idle][-1] - r[idle][0])) * 100.0
        per_wg.append({"xcc": int(meta[g] >> 32) & 7, "samples": int(sel.size), "mean_mhz": round(mean_mhz, 1),
                       "p05_mhz": round(float(np.percentile(inst, 5)), 1), "p50_mhz": round(float(np.percentile(inst, 50)), 1),
                       "p95_mhz": round(float(np.percentile(inst, 95)), 1), "before_work_mhz": round(idle_mhz, 1),
                       "sample_period_us": round(float(np.median(np.diff(rel_ms))) * 1e3, 2)})
    good = [w["mean_mhz"] for w in per_wg if "mean_mhz" in w]
    print(json.dumps({
        "regime": args.regime, "mode": args.mode, "recordings_per_call": B, "calls": calls, "lib": os.path.basename(apt.lib_path()),
        "debug_skip": os.environ.get("APTGPU_DEBUG_SKIP", "0"),
        "ms_per_call_in_regime": round(busy_ms / max(1, calls), 4) if args.regime != "idle" else None,
        "kernels_ms": {kk: round(v[0], 5) for kk, v in sorted(ktimes.items())},
        "shader_clock_mhz": {"mean_over_sampler_waves": round(float(np.mean(good)), 1) if good else None,
                             "min": min(good) if good else None, "max": max(good) if good else None},
        "sampler_waves": per_wg,
    }), flush=True)
    plan.close()


if __name__ == "__main__":
    main()
