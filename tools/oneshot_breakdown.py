#!/usr/bin/env python3
"""Where the one-shot aptgpu_decode() of a ten-minute recording spends its time (bench.py's `one_shot_decode`): the
status callbacks fire on the calling thread — 0.1 before the upload is enqueued, 0.5 after the kernels are enqueued
(the pageable upload has blocked the host for most of its duration by then), 0.9 after the result record was read
(= every kernel finished), return after the rows are on the host — so their timestamps split a call into upload /
kernels' tail / download without a profiler.  Variants: pageable and pinned input, the same recording as a PCM16 WAV."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import noaa_apt_amd as apt  # noqa: E402
from noaa_apt_amd import api  # noqa: E402
from noaa_apt_amd.testing.synth import synth_apt  # noqa: E402
from noaa_apt_amd.testing.wavfile import make_wav  # noqa: E402


def run(name, call, stamps, reps=11):
    rows = []
    for k in range(3 + reps):
        stamps.clear()
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        if k >= 3:
            s = dict(stamps)
            rows.append((t1 - t0, s.get(0.1, t0) - t0, s.get(0.5, t0) - s.get(0.1, t0), s.get(0.9, t0) - s.get(0.5, t0),
                         t1 - s.get(0.9, t1)))
    a = np.median(np.array(rows), axis=0) * 1e3
    print(json.dumps({"case": name, "ms": round(a[0], 3), "before_upload": round(a[1], 3),
                      "upload_and_enqueue": round(a[2], 3), "kernels_tail": round(a[3], 3),
                      "rows_to_host": round(a[4], 3)}), flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    x = synth_apt(48000, seconds, 2)
    stamps = []
    ctx = apt.Context(ui_callback=lambda p, t: stamps.append((round(p, 2), time.perf_counter())), device=0)
    s = apt.Settings()
    rate = apt.Rate.hz(48000)
    run("wrapper, pageable f32", lambda: apt.decode(ctx, s, x, rate, True), stamps)
    # the C call alone
    cctx, cs = ctx._c(), s._c()
    xp = x.ctypes.data_as(api._f32p)
    out, n, st = api._f32p(), C.c_size_t(), api.Stats()
    err = C.create_string_buffer(1024)
    L = apt.lib()

    def c_call(ptr=xp):
        rc = L.aptgpu_decode(C.byref(cctx), C.byref(cs), ptr, x.size, 48000, 1, C.byref(out), C.byref(n), C.byref(st), err, 1024)
        assert rc == 0, err.value
        L.aptgpu_free(C.cast(out, C.c_void_p))
    run("C call, pageable f32", c_call, stamps)
    pin = apt.host_alloc_f32(x.size)
    pin[:] = x
    pp = pin.ctypes.data_as(api._f32p)
    run("C call, pinned f32", lambda: c_call(pp), stamps)
    apt.host_free(pin)
    wav = make_wav(x.astype(np.int16), 48000)
    run("wrapper, PCM16 WAV image (pageable)", lambda: apt.decode_wav(ctx, s, wav, True), stamps)


if __name__ == "__main__":
    main()
