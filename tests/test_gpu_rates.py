"""Parity over arbitrary input sample rates (docs/usage.md:74: "Works with WAV files of any sample rate").

decode() at rates a sound card never writes — SDR front ends (60 000 / 192 000 / 250 000 Hz), rates coprime to the
work rate (44 101 Hz: l = 12 480, ~0.84 M resampler taps), rates just under the RateOverflow edge of dsp.rs:82-91
(in * l still fits u32) and just over it, and rates drawn from a seeded generator — at the three stock profiles,
sync and no-sync, bit for bit against the oracle.  Every case checks that a documented kernel path served it
(stats.fused); `APTGPU_RATES_REPORT=<file>` appends "rate -> l / m / taps -> kernel path -> one-shot ms" (host time
included).  The table on device time — profiles/r06_rates.txt — is tools/rate_timing.py over the same rates.

Tolerance: NONE (strict mode; uint32 views compared).
"""
import math
import os
import time

import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt

pytestmark = pytest.mark.gpu

f32 = np.float32
PROFILES = ("standard", "fast", "slow")
# stats.fused: 0 unfused generic kernels, 1 specialised k_fused (SPLIT), 2 k_fused_any, 3 k_fused with the
# table-driven stage 1, 4 k_fused with the phase-resident stage 1
PATHS = {0: "generic", 1: "k_fused SPLIT", 2: "k_fused_any", 3: "k_fused TABLE", 4: "k_fused PHASE"}


def _edge_rates(work):
    """The largest input rate coprime to `work` whose in * l (= in * work) fits u32 and the smallest one that does
    not (dsp.rs:82-91: `input_rate.checked_mul(l)`)."""
    lim = (1 << 32) - 1
    hi = lim // work
    while math.gcd(hi, work) != 1:
        hi -= 1
    lo = lim // work + 1
    while math.gcd(lo, work) != 1:
        lo += 1
    assert hi * work <= lim < lo * work
    return hi, lo


FIXED = [60000, 192000, 250000, 20833, 37500, 44101, 47999, 8001]
RANDOM = [int(r) for r in np.random.default_rng(20260930).integers(6000, 300001, size=8)]


def _settings(profile):
    s = apt.Settings.profile(profile)
    return s, {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                           "resample_cutout", "demodulation_atten")}


def _seconds(rate):
    # >= 10 rows of work samples with margin, bounded in input samples (the oracle finishes in seconds)
    return 12 if rate <= 100000 else 10


def _report(line):
    path = os.environ.get("APTGPU_RATES_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


@pytest.fixture(scope="module")
def ctx():
    assert apt.device_count() >= 1, "no HIP device: the GPU tests must run on the GPU box"
    return apt.Context(device=0)


def _one(ctx, oracle, rate, profile):
    s, os_ = _settings(profile)
    x = synth_apt(rate, _seconds(rate), seed=rate % 1009 + len(profile))
    l = s.work_rate // math.gcd(rate, s.work_rate)
    if l > 1 and rate * l > (1 << 32) - 1:
        # `input_rate.checked_mul(l)` fails (dsp.rs:82-91): the same error, the same text, from both
        with pytest.raises(oracle.OracleError) as eo:
            oracle.decode(x, rate, True, settings=os_)
        with pytest.raises(apt.RateOverflowError) as eg:
            apt.decode(ctx, s, x, apt.Rate.hz(rate), True)
        assert eo.value.code == 2 and str(eg.value) == str(eo.value)
        _report(f"{rate:7d} {profile:9s} l={l:6d} RateOverflow (in * l = {rate * l} > u32)")
        return
    for sync in (True, False):
        want, st = oracle.decode(x, rate, sync, settings=os_, want_steps=True)
        apt.decode(ctx, s, x, apt.Rate.hz(rate), sync)  # (plan creation, tap design and upload: not timed)
        t0 = time.perf_counter()
        got, stats = apt.decode(ctx, s, x, apt.Rate.hz(rate), sync, return_stats=True)
        ms = (time.perf_counter() - t0) * 1e3
        assert stats.fused in PATHS, stats.fused
        assert stats.n_resample_taps == st["resample_filter"].size, (rate, profile)
        assert stats.work_len == st["resampled"].size, (rate, profile)
        got = np.asarray(got, f32)
        want = np.asarray(want, f32)
        assert got.shape == want.shape, (rate, profile, sync, got.shape, want.shape)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rate, profile, sync)
        if sync:
            assert stats.n_sync == st["sync_pos"].size
            _report(f"{rate:7d} {profile:9s} l={stats.l:6d} m={stats.m:6d} taps={stats.n_resample_taps:8d} "
                    f"path={PATHS[stats.fused]:14s} n={x.size:8d} one-shot {ms:9.2f} ms  "
                    f"{ms * 1e3 / stats.work_len:8.4f} us per work sample")


@pytest.mark.parametrize("profile", PROFILES)
@pytest.mark.parametrize("rate", FIXED)
def test_decode_fixed_odd_rates(ctx, oracle, rate, profile):
    _one(ctx, oracle, rate, profile)


@pytest.mark.parametrize("profile", PROFILES)
@pytest.mark.parametrize("rate", RANDOM)
def test_decode_random_rates(ctx, oracle, rate, profile):
    _one(ctx, oracle, rate, profile)


@pytest.mark.parametrize("profile", PROFILES)
def test_decode_at_the_rate_overflow_edge(ctx, oracle, profile):
    """in * l = in * work_rate just fits u32: decodes, bit-exact; one coprime rate further on: RateOverflow, from the
    product and from the oracle alike (dsp.rs:82-91)."""
    s, os_ = _settings(profile)
    fits, overflows = _edge_rates(s.work_rate)
    _one(ctx, oracle, fits, profile)
    x = synth_apt(overflows, 4, seed=3)
    with pytest.raises(apt.RateOverflowError):
        apt.decode(ctx, s, x, apt.Rate.hz(overflows), True)
    with pytest.raises(oracle.OracleError) as e:
        oracle.decode(x, overflows, True, settings=os_)
    assert e.value.code == 2  # APT_ORACLE_ERR_RATE_OVERFLOW
