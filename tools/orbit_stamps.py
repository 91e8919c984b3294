#!/usr/bin/env python3
"""tools/orbit_stamps.py [batch]: where k_sync_orbit_global's time goes (cycle stamps its first thread leaves in the slot's
flags; one decode of config 2's recording).  GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import noaa_apt_amd as apt  # noqa: E402
from noaa_apt_amd.testing.synth import synth_apt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
rate = int(sys.argv[3]) if len(sys.argv) > 3 else 48000
dev = torch.device("cuda", 0)
x = synth_apt(rate, secs, seed=2)
plan = apt.Plan(apt.Settings(), apt.Rate.hz(rate), True, max_samples=x.size, max_batch=B)
d_in = torch.from_numpy(x).to(dev)
cap = int(plan.info.max_rows)
outs = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in range(B)]
for _ in range(3):
    plan.decode_device([d_in.data_ptr()] * B, [x.size] * B, [o.data_ptr() for o in outs], [cap] * B)
torch.cuda.synchronize()
plan.enable_timing(2)
for _ in range(4):
    plan.decode_device([d_in.data_ptr()] * B, [x.size] * B, [o.data_ptr() for o in outs], [cap] * B)
    torch.cuda.synchronize()
t = plan.collect_timing()
f = plan.read_internal("picker_flags", np.uint32, 32)
print("kernels alone, ms:", {k: round(v[0], 4) for k, v in sorted(t.items())})
print("orbit form", int(f[6]), "nodes", int(f[4]), "with a predecessor", int(f[13]), "stamps (successors / orbit / peaks):", [int(v) for v in f[8:11]])
names = ["arguments", "counts + entries fetched, scanned", "list in LDS", "predecessors numbered", "orbit known"]
prev = 0
for k, nm in enumerate(names):
    v = int(f[16 + k])
    print(f"  {nm:36s} at {v:7d} cycles (+{v - prev})")
    prev = v
