import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
from noaa_apt_amd.testing.wavfile import make_wav
x = synth_apt(48000, 600, seed=2)
wav = make_wav(x.astype(np.int16), 48000)
ctx = apt.Context(device=0)
for name, fn in (("decode(f32 host buffer)", lambda: apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(48000), True)),
                 ("decode_wav(PCM16 file image)", lambda: apt.decode_wav(ctx, apt.Settings(), wav, True))):
    ts = []
    for i in range(6):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
    print(name, "first %.1f ms, then" % (ts[0] * 1e3), " ".join("%.1f" % (t * 1e3) for t in ts[1:]), "ms; rows", r.size // 2080)
