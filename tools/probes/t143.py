import sys, time, numpy as np
sys.path.insert(0, ".")
import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
ctx = apt.Context(device=0)
for rate in (143054, 113779):
    x = synth_apt(rate, 10, seed=1)
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); got, st = apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(rate), True, return_stats=True); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(rate, ts, st.fused, st.orbit_path, apt.cache_info())
    msgs = []
    c2 = apt.Context(device=0, ui_callback=lambda p, t: msgs.append((round((time.perf_counter()-t0)*1e3,2), p, t)))
    t0 = time.perf_counter(); apt.decode(c2, apt.Settings(), x, apt.Rate.hz(rate), True); print(msgs, round((time.perf_counter()-t0)*1e3,2))
