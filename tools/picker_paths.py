#!/usr/bin/env python3
"""Which path the peak picker takes on each synthetic recording (one GPU): tools/picker_paths.py [--inputs 8]."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=600.0)
    args = ap.parse_args()
    import torch
    import noaa_apt_amd as apt
    from noaa_apt_amd.testing.synth import synth_apt
    dev = torch.device("cuda", 0)
    for j in range(args.inputs):
        x = synth_apt(48000, args.seconds, seed=2 + 1000 * j)
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=x.size)
        d = torch.from_numpy(x).to(dev)
        cap = int(plan.info.max_rows)
        out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            plan.decode_device([d.data_ptr()], [x.size], [out.data_ptr()], [cap])
        res = plan.results(1)[0]
        f = plan.read_internal("picker_flags", np.uint32, 32)
        plan.enable_timing(2)
        for _ in range(4):
            plan.decode_device([d.data_ptr()], [x.size], [out.data_ptr()], [cap])
            torch.cuda.synchronize()
        t = plan.collect_timing()
        print(f"seed {2 + 1000 * j}: rows {res.n_rows} path {int(f[1])} direct {int(f[6])} nodes {int(f[3])} visited {int(f[4])} levels {int(f[12])} "
              f"stamps {[int(v) for v in f[8:11]]} orbit_ms {t.get('sync_orbit', (0, 0))[0]:.4f} nodes_ms {t.get('sync_nodes', (0, 0))[0]:.4f}")
        plan.close()


if __name__ == "__main__":
    main()
