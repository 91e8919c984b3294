#!/usr/bin/env python3
"""tools/rate_timing.py rate[:profile] ... — per-kernel times of one device-resident decode at odd input rates
(which kernel path serves it, where the time goes).  GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import noaa_apt_amd as apt  # noqa: E402
from noaa_apt_amd.testing.synth import synth_apt  # noqa: E402

dev = torch.device("cuda", 0)
for spec in sys.argv[1:]:
    rate, _, profile = spec.partition(":")
    rate = int(rate)
    profile = profile or "standard"
    x = synth_apt(rate, 10, seed=1)
    plan = apt.Plan(apt.Settings.profile(profile), apt.Rate.hz(rate), True, max_samples=x.size)
    d_in = torch.from_numpy(x).to(dev)
    cap = int(plan.info.max_rows)
    d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
    torch.cuda.synchronize()
    plan.enable_timing(2)
    for _ in range(4):
        plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
        torch.cuda.synchronize()
    t = plan.collect_timing()
    info = plan.info
    print(json.dumps({"rate": rate, "profile": profile, "l": int(info.l), "m": int(info.m), "fused": int(info.fused),
                      "taps": int(info.n_resample_taps), "ms": {k: round(v[0], 4) for k, v in sorted(t.items())}}), flush=True)
    plan.close()
