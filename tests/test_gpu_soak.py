"""Differential soak of decode() against the oracle on the GPU: seeded random rates x settings x lengths x signals.

Every case draws an input rate (the stock rates often, else anything in [6 000, 260 000] Hz), a settings profile or a
tuned Settings (work_rate, resample_atten / delta_freq / cutout, demodulation_atten), a length (now and then too short
for ten rows, a few samples off whole seconds), sync or no-sync, and a signal: synthetic APT, APT with NaN / +-Inf
samples, a silent stretch, a constant, amplitudes of 1e-30 / 1e30, pure noise.  The product's rows must be bit-identical
to the oracle's, or both must fail with the same message.  The suite runs APT_SOAK_CASES cases per seed (default 40:
about 20 s); `APT_SOAK_CASES=700 python -m pytest tests/test_gpu_soak.py -m gpu -s` is the long run whose summary lines
are kept in profiles/r06_soak.txt.  A failure message names the seed and the case index that reproduce it.
"""
import json
import os
import time

import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt

pytestmark = pytest.mark.gpu


def _seeds(default):
    """APT_SOAK_SEEDS=101,102,... replaces a test's own seeds (the long runs of profiles/r06_soak.txt)."""
    e = os.environ.get("APT_SOAK_SEEDS", "")
    return [int(v) for v in e.split(",") if v.strip()] or default

STOCK_RATES = [48000, 44100, 22050, 11025, 96000, 8000, 16000, 32000, 24000, 60000, 192000, 250000, 12000, 20800]


def draw_case(rng):
    c = {}
    c["rate"] = int(rng.choice(STOCK_RATES)) if rng.random() < 0.7 else int(rng.integers(6000, 260000))
    u = rng.random()
    if u < 0.45:
        c["profile"] = "standard"
    elif u < 0.6:
        c["profile"] = "fast"
    elif u < 0.7:
        c["profile"] = "slow"
    else:
        c["profile"] = None
        c["settings"] = dict(
            work_rate=int(rng.choice([12480, 16640, 20800, 8320, 4160, 24960])),
            resample_atten=float(np.float32(rng.uniform(18, 48))),
            resample_delta_freq=float(np.float32(rng.uniform(400, 3500))),
            resample_cutout=float(np.float32(rng.uniform(3000, 6500))),
            demodulation_atten=float(np.float32(rng.uniform(12, 45))))
    c["sync"] = bool(rng.random() < 0.75)
    v = rng.random()
    c["seconds"] = float(rng.uniform(1.0, 5.5)) if v < 0.06 else float(rng.uniform(5.5, 9.0)) if v < 0.2 else float(rng.uniform(9.0, 45.0))
    c["trim"] = int(rng.integers(0, 7))  # samples cut from the end: lengths that are not whole seconds
    c["kind"] = str(rng.choice(["apt", "apt", "apt", "apt", "nan", "inf", "silence", "const", "tiny", "huge", "noise"]))
    c["seed"] = int(rng.integers(1, 1 << 30))
    return c


def make_signal(c, synth_apt):
    rate, kind = c["rate"], c["kind"]
    rng = np.random.default_rng(c["seed"])
    if kind == "noise":
        x = (rng.standard_normal(int(rate * c["seconds"])) * 3000).astype(np.float32)
    elif kind == "const":
        x = np.full(int(rate * c["seconds"]), 1234.0, np.float32)
    else:
        x = synth_apt(rate, c["seconds"], seed=c["seed"] % 100000)
    if c["trim"]:
        x = x[:x.size - c["trim"]]
    n = x.size
    if kind == "nan":
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, max(1, n - 50)))
            x[a:a + int(rng.integers(1, 40))] = np.nan
    elif kind == "inf":
        for _ in range(int(rng.integers(1, 4))):
            x[int(rng.integers(0, n))] = np.inf if rng.random() < 0.5 else -np.inf
    elif kind == "silence":
        a = int(rng.integers(0, max(1, n // 2)))
        x[a:a + int(rng.integers(n // 20 + 1, n // 3 + 2))] = 0.0
    elif kind == "tiny":
        x = (x * np.float32(1e-30)).astype(np.float32)
    elif kind == "huge":
        x = (x * np.float32(1e30)).astype(np.float32)
    return x




@pytest.mark.parametrize("seed", _seeds([1, 2, 3]))
def test_soak_decode_against_oracle(oracle, seed):
    cases = int(os.environ.get("APT_SOAK_CASES", "40"))
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    bad, err_match, paths, kinds = [], 0, {}, {}
    for i in range(cases):
        c = draw_case(rng)
        x = make_signal(c, synth_apt)
        s = apt.Settings.profile(c["profile"]) if c["profile"] else apt.Settings(**c["settings"])
        os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq", "resample_cutout",
                                          "demodulation_atten")}
        want = want_err = got = got_err = st = None
        try:
            want = oracle.decode(x, c["rate"], c["sync"], settings=os_)
        except Exception as e:  # noqa: BLE001
            want_err = str(e)
        try:
            got, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(c["rate"]), c["sync"], return_stats=True)
        except Exception as e:  # noqa: BLE001
            got_err = str(e)
        kinds[c["kind"]] = kinds.get(c["kind"], 0) + 1
        if st is not None:
            paths[int(st.fused)] = paths.get(int(st.fused), 0) + 1
        if want_err is not None or got_err is not None:
            ok = want_err == got_err
            err_match += 1 if ok else 0
            detail = f"oracle error {want_err!r}, product error {got_err!r}"
        else:
            ok = want.size == got.size and np.array_equal(want.view(np.uint32), got.view(np.uint32))
            detail = f"sizes {want.size} / {got.size}"
            if not ok and want.size == got.size:
                detail += f", first difference at {int(np.flatnonzero(want.view(np.uint32) != got.view(np.uint32))[0])}"
        if not ok:
            bad.append(f"seed {seed} case {i}: {json.dumps(c)} n={x.size} fused={None if st is None else int(st.fused)}: {detail}")
        if (i + 1) % 50 == 0:
            apt.cache_clear()  # (the sessions and plans of fifty settings: keep the cache from growing)
    apt.cache_clear()
    print("\nsoak " + json.dumps({"seed": seed, "cases": cases, "mismatches": len(bad), "errors_that_matched": err_match,
                                   "kernel_paths_stats_fused": dict(sorted(paths.items())),
                                   "signal_kinds": dict(sorted(kinds.items())),
                                   "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed", _seeds([11, 12]))
def test_soak_decode_batch_against_oracle(oracle, seed):
    """The same draw for aptgpu_decode_batch: one rate / Settings / sync per batch, two to nine recordings of mixed
    lengths and kinds (some too short: their error is the oracle's, the others still decode), one to three worker
    entries on the device, one to eight recordings per call."""
    batches = max(1, int(os.environ.get("APT_SOAK_CASES", "40")) // 8)
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    bad, n_rec, n_err = [], 0, 0
    for b in range(batches):
        head = draw_case(rng)
        s = apt.Settings.profile(head["profile"]) if head["profile"] else apt.Settings(**head["settings"])
        os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq", "resample_cutout",
                                          "demodulation_atten")}
        recs, cases = [], []
        for _ in range(int(rng.integers(2, 10))):
            c = draw_case(rng)
            c.update(rate=head["rate"], profile=head["profile"], sync=head["sync"])
            c["seconds"] = min(c["seconds"], 30.0)
            cases.append(c)
            recs.append(make_signal(c, synth_apt))
        workers = tuple([0] * int(rng.integers(1, 4)))
        per_call = int(rng.integers(1, 9))
        try:
            got = apt.decode_batch(apt.Context(device=0), s, recs, apt.Rate.hz(head["rate"]), head["sync"],
                                   devices=workers, recordings_per_call=per_call)
        except Exception as e:  # noqa: BLE001 - a batch-level failure: the reference's decode() fails the same way for each
            got = [e] * len(recs)
        for i, (c, x) in enumerate(zip(cases, recs)):
            n_rec += 1
            want = want_err = None
            try:
                want = oracle.decode(x, head["rate"], head["sync"], settings=os_)
            except Exception as e:  # noqa: BLE001
                want_err = str(e)
            g = got[i]
            if want_err is not None or isinstance(g, Exception):
                ok = isinstance(g, Exception) and str(g) == want_err
                n_err += 1 if ok else 0
                detail = f"oracle error {want_err!r}, product {g!r}"
            else:
                ok = want.size == g.size and np.array_equal(want.view(np.uint32), g.view(np.uint32))
                detail = f"sizes {want.size} / {g.size}"
            if not ok:
                bad.append(f"seed {seed} batch {b} recording {i} (workers {len(workers)}, per call {per_call}): "
                           f"{json.dumps(c)} n={x.size}: {detail}")
        apt.cache_clear()
    print("\nsoak-batch " + json.dumps({"seed": seed, "batches": batches, "recordings": n_rec, "mismatches": len(bad),
                                        "errors_that_matched": n_err, "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed", _seeds([21]))
def test_soak_fast_mode_within_its_tolerance(oracle, seed):
    """APTGPU_MODE_FAST over the same draw of rates and Settings (finite signals: synthetic APT, the kinds whose sync
    positions are well defined): SURVEY.md 8(d)'s tolerance — same row count, sync positions identical on >= 99.9 % of the
    rows and never off by more than one sample, |d px| <= 1e-4 max |px| on rows with identical position — whatever kernel
    the plan lands on (the fast instantiations, the matrix-core kernel for tuned tap counts, strict kernels elsewhere)."""
    from test_gpu_fast import check_tolerance, decode_on_plan
    cases = max(4, int(os.environ.get("APT_SOAK_CASES", "40")) // 2)
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    bad, worst, done, paths = [], 0.0, 0, {}
    for i in range(cases):
        c = draw_case(rng)
        c["kind"], c["sync"] = "apt", True
        c["seconds"] = max(c["seconds"], 9.0)
        x = make_signal(c, synth_apt)
        s = apt.Settings.profile(c["profile"]) if c["profile"] else apt.Settings(**c["settings"])
        os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq", "resample_cutout",
                                          "demodulation_atten")}
        try:
            want, st = oracle.decode(x, c["rate"], True, settings=os_, want_steps=True)
        except Exception:  # noqa: BLE001 - too short / rate overflow / work_rate: the strict soak compares the errors
            continue
        try:
            rows, pos, res, fused = decode_on_plan(x, c["rate"], apt.MODE_FAST, settings=s)
            assert res.status == 0, (res.status, res.reason)
            _, err = check_tolerance(rows, pos, want, st["sync_pos"], f"seed {seed} case {i}")
            worst = max(worst, err)
            paths[fused] = paths.get(fused, 0) + 1
            done += 1
        except AssertionError as e:
            bad.append(f"seed {seed} case {i}: {json.dumps(c)} n={x.size}: {e}")
        if (i + 1) % 50 == 0:
            apt.cache_clear()
    apt.cache_clear()
    print("\nsoak-fast " + json.dumps({"seed": seed, "cases": done, "out_of_tolerance": len(bad), "worst_px_error_of_full_scale": worst,
                                       "kernel_paths_stats_fused": dict(sorted(paths.items())),
                                       "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed", _seeds([31]))
def test_soak_image_stage_and_wav_ingest_against_oracle(seed):
    """The rows either side of the path, same idea: random WAV file images (8 / 16 / 24 / 32-bit PCM,
    float, one to three channels, junk chunks that move the payload to odd addresses) through load(); random row images
    (a noisy frame, with NaN / Inf pixels, constant, tiny, negated) through process() with every contrast mode and
    rotation, and through read_telemetry().  Bit-identical to the oracle or the same error."""
    from oracle import image_binding as oi, wav_binding as ow
    from noaa_apt_amd.testing.synth import make_image
    from noaa_apt_amd.testing.wavfile import make_wav
    cases = max(6, int(os.environ.get("APT_SOAK_CASES", "40")) // 2)
    rng = np.random.default_rng(seed)
    bad = []
    f32 = np.float32
    same = lambda a, b: np.asarray(a, f32).tobytes() == np.asarray(b, f32).tobytes()  # noqa: E731
    for i in range(cases):
        # ---- N1: load()
        bits = int(rng.choice([8, 16, 24, 32]))
        is_float = bool(rng.random() < 0.25)
        channels = int(rng.integers(1, 4))
        frames = int(rng.choice([1, 2, 7, 255, 4099, int(rng.integers(10, 300000))]))
        if is_float:
            vals = (rng.standard_normal(frames * channels) * 10.0 ** rng.integers(-3, 6)).astype(f32)
            if vals.size > 3:
                vals[int(rng.integers(0, vals.size))] = np.nan
                vals[int(rng.integers(0, vals.size))] = -np.inf
            kw = dict(is_float=True, channels=channels)
        else:
            lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
            vals = rng.integers(lo, hi + 1, size=frames * channels, dtype=np.int64)
            kw = dict(bits=bits, channels=channels)
        if rng.random() < 0.5:
            kw["extra_chunks"] = [(b"junk", bytes(int(rng.integers(1, 9))))]
        rate_hz = int(rng.choice([11025, 44100, 48000, 96000, 8000]))
        data = make_wav(vals, rate_hz, **kw)
        what = f"seed {seed} case {i} load {kw} frames={frames}"
        try:
            want, _ = ow.load_wav(data)
            want_err = None
        except Exception as e:  # noqa: BLE001
            want_err = str(e)
        try:
            got, rate = apt.load(data)
            got_err = None
        except Exception as e:  # noqa: BLE001
            got_err = str(e)
        if want_err is not None or got_err is not None:
            if want_err != got_err:
                bad.append(f"{what}: oracle error {want_err!r}, product error {got_err!r}")
        elif not (same(got, want) and rate.get_hz() == rate_hz):
            bad.append(f"{what}: samples differ")
        # ---- N2 / N3: process(), read_telemetry()
        rows = int(rng.choice([1, 3, 199, 200, 201, int(rng.integers(202, 1400))]))
        kind = str(rng.choice(["frame", "frame", "nan", "inf", "const", "tiny", "negated"]))
        img = make_image(rows, seed=int(rng.integers(1, 1 << 20)))
        sig = (img * f32(37.5) + rng.standard_normal(img.shape).astype(f32) * f32(rng.uniform(1.0, 300.0))).astype(f32).ravel()
        if kind == "nan":
            sig[rng.integers(0, sig.size, 5)] = np.nan
        elif kind == "inf":
            sig[rng.integers(0, sig.size, 3)] = np.inf
            sig[rng.integers(0, sig.size, 2)] = -np.inf
        elif kind == "const":
            sig[:] = f32(rng.uniform(-5, 5))
        elif kind == "tiny":
            sig *= f32(1e-35)
        elif kind == "negated":
            sig = -sig
        cname = str(rng.choice(["telemetry", "percent", "minmax"]))
        p = float(np.float32(rng.choice([0.98, 0.9, 0.5, 1.0, 0.0, float(rng.uniform(0, 1))])))
        ca = {"telemetry": apt.Contrast.TELEMETRY, "percent": apt.Contrast.Percent(p), "minmax": apt.Contrast.MINMAX}[cname]
        ck = {"telemetry": oi.CONTRAST_TELEMETRY, "percent": oi.CONTRAST_PERCENT, "minmax": oi.CONTRAST_MINMAX}[cname]
        what = f"seed {seed} case {i} process rows={rows} kind={kind} contrast={cname} p={p}"
        try:
            want_img, lo_, hi_ = oi.process_gray(sig, ck, p)
            want_err = None
        except Exception as e:  # noqa: BLE001
            want_err = str(e)
        try:
            got_img, info = apt.process(apt.Context(device=0), sig, ca, rotate=0, return_info=True)
            got_err = None
        except Exception as e:  # noqa: BLE001
            got_err = str(e)
        if want_err is not None or got_err is not None:
            if want_err != got_err:
                bad.append(f"{what}: oracle error {want_err!r}, product error {got_err!r}")
        elif not (np.array_equal(got_img.ravel(), np.asarray(want_img).ravel()) and same(info.low, lo_) and same(info.high, hi_)):
            bad.append(f"{what}: image or limits differ")
        what = f"seed {seed} case {i} telemetry rows={rows} kind={kind}"
        try:
            wt = oi.read_telemetry(sig)
            want_err = None
        except Exception as e:  # noqa: BLE001
            want_err = str(e)
        try:
            gt = apt.read_telemetry(apt.Context(device=0), sig)
            got_err = None
        except Exception as e:  # noqa: BLE001
            got_err = str(e)
        if want_err is not None or got_err is not None:
            if want_err != got_err:
                bad.append(f"{what}: oracle error {want_err!r}, product error {got_err!r}")
        elif not (gt.row == wt.row and same(gt.quality, wt.quality) and gt.values_a.tobytes() == wt.values_a.tobytes()
                  and gt.values_b.tobytes() == wt.values_b.tobytes()):
            bad.append(f"{what}: telemetry differs")
    print("\nsoak-sides " + json.dumps({"seed": seed, "cases": cases, "mismatches": len(bad)}), flush=True)
    assert not bad, "\n".join(bad[:20])
