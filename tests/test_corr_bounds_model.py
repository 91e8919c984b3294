"""CPU check of the inequality the strict front ends rely on (csrc/apt_kernels_fused_impl.hpp, stage 4):

    | pulse-sum evaluation  -  the reference's sequential chain |  <=  138 u * sum|F|      (u = 2^-24, pw = 3)

for the sync correlation of find_sync() (decode.rs:225-233).  The chain is the oracle's (the C restatement of
the reference); the pulse-sum evaluation is re-stated here in numpy float32 with exactly the association the
kernels use (csrc/apt_sync_corr.hpp: B2[i] = F[i] + F[i+1]; B[i] = (B2[i] + B2[i+2]) + B2[i+4];
corr[i] = -B[i] - B[i+6] + B[i+12] - ... left to right).  The GPU tests (tests/test_gpu_bounds.py) check the
kernels' bounds against the exact maxima; this one checks the inequality itself, on inputs built to stress it —
and that it is not vacuous (the two evaluations really differ, by far less than the bound).
"""
import numpy as np
import pytest

f32 = np.float32
U = 2.0 ** -24
WORK_RATE = 12480
PW = 3
G = 38 * PW


def pulse_sum_corr(f):
    """The fast / bounding evaluation, every operation rounded to f32 in the kernels' order."""
    f = np.asarray(f, f32)
    n = f.size - G
    b2 = (f[:-1] + f[1:]).astype(f32)                       # B2[i] = F[i] + F[i+1]
    m = b2.size - 4
    b = ((b2[:m] + b2[2:m + 2]).astype(f32) + b2[4:m + 4]).astype(f32)   # pulse sums over 2*pw = 6 samples
    plus = [k >= 2 and k <= 14 and k % 2 == 0 for k in range(19)]
    c = (-b[0:n]).astype(f32)
    for k in range(1, 19):
        t = b[6 * k:6 * k + n]
        c = (c + t).astype(f32) if plus[k] else (c - t).astype(f32)
    return c


def cases():
    rng = np.random.default_rng(7)
    n = 20000
    yield "envelope-like", (1500 + 800 * np.sin(np.arange(n) / 37.0) + 200 * rng.standard_normal(n)).astype(f32)
    yield "white noise", (3000 * rng.standard_normal(n)).astype(f32)
    yield "uniform positive", rng.uniform(0, 1, n).astype(f32)
    yield "constant (pure cancellation)", np.full(n, 1234.567, f32)
    yield "alternating sign", (np.where(np.arange(n) % 2 == 0, 1, -1) * rng.uniform(1, 2, n)).astype(f32)
    yield "wide dynamic range", (rng.standard_normal(n) * 10.0 ** rng.uniform(-20, 20, n)).astype(f32)
    yield "huge", (rng.uniform(0.5, 1, n) * 1e34).astype(f32)
    yield "denormal", (rng.uniform(0, 1, n) * 1e-41).astype(f32)
    yield "spikes", np.where(rng.random(n) < 0.01, 1e9, 1.0).astype(f32) * rng.uniform(0.9, 1.1, n).astype(f32)
    # the template's own shape at every scale: the largest partial sums the chain can see
    t = np.tile(np.repeat([0.0, 1.0], 2 * PW), n // (4 * PW) + 1)[:n]
    yield "sync pulses", (t * 20000 + rng.uniform(0, 50, n)).astype(f32)
    yield "integers (exact sums)", rng.integers(0, 4096, n).astype(f32)


@pytest.mark.parametrize("name,f", list(cases()), ids=[c[0] for c in cases()])
def test_pulse_sum_evaluation_is_within_the_bound_of_the_chain(oracle, name, f):
    pos, chain = oracle.find_sync(f, WORK_RATE, return_correlation=True)
    approx = pulse_sum_corr(f)
    assert chain.size == approx.size == f.size - G
    af = np.abs(f.astype(np.float64))
    cs = np.concatenate([[0.0], np.cumsum(af)])
    win = cs[G:G + chain.size] - cs[:chain.size]              # sum |F[i .. i+113]| per position
    diff = np.abs(approx.astype(np.float64) - chain.astype(np.float64))
    bound = 138 * U * win
    assert np.all(diff <= bound), (name, float(np.max(diff / np.maximum(bound, 1e-300))))
    # and the analytic figure behind it: gamma(113) + gamma(21) < 134.01 u
    assert np.all(diff <= 134.01 * U * win * (1 + 1e-9) + 0.0), name
    if name in ("integers (exact sums)", "denormal"):
        # every partial sum is an integer < 2^24, or a subnormal (whose additions are exact — why the bound has no
        # underflow term): no rounding at all
        assert np.all(diff == 0)
    elif name not in ("constant (pure cancellation)",):
        assert np.any(diff > 0), name                          # the two really are different evaluations


def test_the_bound_is_what_the_kernels_use():
    """fused_gm_slack(pw = 3) of csrc/apt_kernels_fused.hip, re-stated: (38 pw - 1 + pw + 18) * 1.03 * u * 1.0001."""
    slack = f32(f32(38 * PW - 1 + PW + 18) * f32(1.03) * f32(2.0 ** -24) * f32(1.0001))
    assert 134.01 * U * 1.02 < float(slack) < 138.1 * U
