#!/usr/bin/env python3
"""tools/rate_timing.py [rate[:profile] ...] — which kernel path serves an input rate and what a decode costs on the
device: per-kernel times (one call in flight) of a 60 s recording, and the front end's time per WORK sample against
48 kHz at the same profile.  Without arguments: every rate of tests/test_gpu_rates.py (the sweep whose parity that test
checks) x the three stock profiles -> the table of profiles/r06_rates.txt.  GPU box."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import noaa_apt_amd as apt  # noqa: E402
from noaa_apt_amd.testing.synth import synth_apt  # noqa: E402

PATHS = {0: "generic", 1: "k_fused SPLIT", 2: "k_fused_any", 3: "k_fused TABLE", 4: "k_fused PHASE"}
SECONDS = 60


def one(rate, profile, dev):
    s = apt.Settings.profile(profile)
    l = s.work_rate // math.gcd(rate, s.work_rate)
    if l > 1 and rate * l > (1 << 32) - 1:
        return None
    x = synth_apt(rate, SECONDS, seed=1)
    plan = apt.Plan(s, apt.Rate.hz(rate), True, max_samples=x.size)
    d_in = torch.from_numpy(x).to(dev)
    cap = int(plan.info.max_rows)
    d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
    torch.cuda.synchronize()
    plan.enable_timing(2)
    for _ in range(5):
        plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
        torch.cuda.synchronize()
    t = {k: v[0] for k, v in plan.collect_timing().items()}
    res = plan.results(1)[0]
    info = plan.info
    out = dict(rate=rate, profile=profile, l=int(info.l), m=int(info.m), taps=int(info.n_resample_taps), path=PATHS[int(info.fused)],
               work=int(res.work_len), front=sum(v for k, v in t.items() if k not in ("sync_nodes", "sync_orbit", "gather_rows")),
               total=sum(t.values()))
    plan.close()
    return out


def main():
    dev = torch.device("cuda", 0)
    if len(sys.argv) > 1:
        cases = []
        for spec in sys.argv[1:]:
            r, _, p = spec.partition(":")
            cases.append((int(r), p or "standard"))
    else:
        from test_gpu_rates import FIXED, RANDOM, PROFILES, _edge_rates
        cases = [(r, p) for r in FIXED + RANDOM for p in PROFILES]
        cases += [(_edge_rates(apt.Settings.profile(p).work_rate)[0], p) for p in PROFILES]
    base = {}
    print(f"(a {SECONDS} s recording per case, one call in flight; 'x 48 kHz': front-end time — every kernel in front of the peak picker — per work sample against 48 000 Hz at the same profile)")
    for p in sorted({p for _, p in cases}):
        base[p] = one(48000, p, dev)
        b = base[p]
        print(f"{48000:7d} {p:9s} l={b['l']:6d} m={b['m']:6d} taps={b['taps']:9d} {b['path']:14s} front end {b['front']:8.4f} ms  all kernels {b['total']:8.4f} ms  "
              f"{1e6 * b['front'] / b['work']:7.3f} ns per work sample")
    for rate, p in cases:
        r = one(rate, p, dev)
        if r is None:
            print(f"{rate:7d} {p:9s} RateOverflow (in * l does not fit u32: dsp.rs:82-91)")
            continue
        rel = (r["front"] / r["work"]) / (base[p]["front"] / base[p]["work"])
        print(f"{rate:7d} {p:9s} l={r['l']:6d} m={r['m']:6d} taps={r['taps']:9d} {r['path']:14s} front end {r['front']:8.4f} ms  all kernels {r['total']:8.4f} ms  "
              f"{1e6 * r['front'] / r['work']:7.3f} ns per work sample = {rel:6.2f} x 48 kHz", flush=True)


if __name__ == "__main__":
    main()
