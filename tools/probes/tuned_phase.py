import sys, numpy as np
sys.path.insert(0, ".")
import torch, noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
dev = torch.device("cuda", 0)
for rate in (44100, 22050, 11025):
    for kw in ({}, dict(resample_atten=31.0), dict(resample_delta_freq=900.0), dict(resample_atten=35.0), dict(resample_delta_freq=800.0)):
        s = apt.Settings(**kw)
        x = synth_apt(rate, 60, seed=1)
        plan = apt.Plan(s, apt.Rate.hz(rate), True, max_samples=x.size)
        d_in = torch.from_numpy(x).to(dev); cap = int(plan.info.max_rows)
        d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
        for _ in range(3): plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
        torch.cuda.synchronize(); plan.enable_timing(2)
        for _ in range(4):
            plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap]); torch.cuda.synchronize()
        t = plan.collect_timing()
        print(rate, kw, "fused", int(plan.info.fused), "taps", int(plan.info.n_resample_taps), "front end ms", round(t["fused_front_end"][0], 4), flush=True)
        plan.close()
