"""The strict front ends hand the picker BOUNDS of the per-group maxima of the sync correlation
(pulse sums + a rounding-error bound) instead of evaluating the reference's 114-term chain at every
position; k_sync_nodes evaluates the exact chain where it matters.  Two things are checked here:

* the bounds are bounds: lo <= (exact maximum of the group, from the oracle's correlation) <= hi for
  every group of every input, and they are tight (a few 1e-6 of the window's sum of |F|);
* the result never depends on them: with the bounds widened a thousand- and a million-fold
  (APTGPU_GM_SLACK_SCALE) every comparison the bounds can no longer decide is settled by exact
  evaluation — rows bit-identical, and the counter of such settlements shows the code ran.

Tolerance: none (uint32 views).
"""
import functools

import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt, synth_noise
from test_gpu_parity import assert_bitexact, assert_same_values

pytestmark = pytest.mark.gpu

f32 = np.float32
GS = 52


NAMES = ["apt48k", "apt48k-ppm", "noise48k", "heavy-noise", "weak", "apt96k", "apt11025", "zeros", "dc", "loud",
         "tiny", "nonfinite"]


@functools.lru_cache(maxsize=None)
def _case(name):
    if name == "apt48k":
        return synth_apt(48000, 20, 5), 48000
    if name == "apt48k-ppm":
        return synth_apt(48000, 14, 12, ppm=40.0), 48000
    if name == "noise48k":
        return synth_noise(48000, 20.0, 5, sigma=4000.0), 48000
    if name == "heavy-noise":
        return synth_apt(48000, 16, 8, noise_sigma=6000.0), 48000
    if name == "weak":
        return synth_apt(48000, 14, 9, amplitude=300.0, noise_sigma=400.0), 48000
    if name == "apt96k":
        return synth_apt(96000, 14, 3), 96000
    if name == "apt11025":  # table-driven stage 1, same work-rate stages
        return synth_apt(11025, 24, 1), 11025
    if name == "zeros":
        return np.zeros(48000 * 12, f32), 48000
    if name == "dc":
        return np.full(48000 * 12, 1234.0, f32), 48000
    if name == "loud":  # radicands past 2^100: general envelope, huge F
        return synth_apt(48000, 12, 31) * f32(3e14), 48000
    if name == "tiny":  # radicands in the denormal range
        return synth_apt(48000, 12, 32) * f32(1e-25), 48000
    if name == "nonfinite":
        x = synth_apt(48000, 14, 61)
        rng = np.random.default_rng(5)
        for s0 in rng.integers(48000, x.size - 48000, 7):
            x[s0:s0 + int(rng.integers(1, 300))] = np.nan if rng.random() < 0.6 else np.inf
        return x, 48000
    raise KeyError(name)


def _decode_plan(x, rate, torch):
    dev = torch.device("cuda:0")
    plan = apt.Plan(apt.Settings(), apt.Rate.hz(rate), True, max_samples=x.size)
    d_in = torch.from_numpy(x).to(dev)
    cap = int(plan.info.max_rows)
    d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
    res = plan.results(1)[0]
    rows = d_out[:res.n_out].cpu().numpy()
    return plan, res, rows


@pytest.mark.parametrize("name", NAMES)
def test_bounds_hold_and_are_tight(oracle, name):
    torch = pytest.importorskip("torch")
    x, rate = _case(name)
    want, st = oracle.decode(x, rate, True, want_steps=True)
    plan, res, rows = _decode_plan(x, rate, torch)
    assert plan.info.fused in (1, 3, 4)
    (assert_same_values if name == "nonfinite" else assert_bitexact)(rows, want, name)
    corr = st["correlation"].astype(f32).copy()
    filt = st["filtered"]
    n_corr = corr.size
    ng = (n_corr + GS - 1) // GS
    gm = plan.read_internal("group_max", f32, 2 * ng).reshape(ng, 2)
    hi, lo = gm[:, 0], gm[:, 1]
    plan.close()
    # exact maxima per group: NaNs left out, position 0 clamped to >= 0 (decode.rs:208)
    if not corr[0] > 0:
        corr[0] = 0.0
    pad = np.full(ng * GS, -np.inf, f32)
    pad[:n_corr] = np.where(np.isnan(corr), -np.inf, corr)
    exact = pad.reshape(ng, GS).max(axis=1)
    flagged = np.isinf(hi) & np.isinf(lo) & (hi > 0) & (lo < 0)
    ok = ~flagged
    assert np.all(lo[ok] <= exact[ok]) and np.all(exact[ok] <= hi[ok]), (
        name, int(np.flatnonzero(ok & ~((lo <= exact) & (exact <= hi)))[0]))
    # a group is flagged when its F window is not finite (the kernel sums |F| over 13 threads' worth of
    # positions — 169 >= the 165 of the window — so a non-finite F just behind the window flags too)
    af = np.abs(filt.astype(np.float64))
    cs = np.concatenate([[0.0], np.cumsum(np.where(np.isfinite(af), af, 0.0))])
    bad = np.concatenate([[0], np.cumsum(~np.isfinite(af))])
    g0 = np.arange(ng) * GS
    must = (bad[np.minimum(g0 + GS + 114 - 1, filt.size)] - bad[g0]) > 0
    may = (bad[np.minimum(g0 + 169, filt.size)] - bad[g0]) > 0
    assert np.all(flagged[must]), name
    assert not np.any(flagged & ~may), name
    # tight: the interval is a few 1e-6 of the window's sum of |F| wide (138 u * A on each side; the
    # kernel sums |F| over 13 threads' worth of positions, 169 >= 165)
    a_win = cs[np.minimum(g0 + 169, filt.size)] - cs[g0]
    width = (hi[ok].astype(np.float64) - lo[ok].astype(np.float64))
    assert np.all(width <= 2 * 140 * 2.0 ** -24 * a_win[ok] * 1.001 + 1e-45), name
    if name in ("apt48k", "noise48k", "apt96k", "apt11025"):
        assert np.all(width[a_win[ok] > 0] > 0), name


@pytest.mark.parametrize("scale", ["1e3", "1e6"])
@pytest.mark.parametrize("name", NAMES)
def test_result_does_not_depend_on_the_bounds(oracle, monkeypatch, name, scale):
    torch = pytest.importorskip("torch")
    x, rate = _case(name)
    want, st = oracle.decode(x, rate, True, want_steps=True)
    monkeypatch.setenv("APTGPU_GM_SLACK_SCALE", scale)
    plan, res, rows = _decode_plan(x, rate, torch)
    flags = plan.read_internal("picker_flags", np.uint32, 32)
    pos = plan.sync_positions(0)
    plan.close()
    assert res.n_sync == st["sync_pos"].size
    assert pos.tolist() == st["sync_pos"].tolist()
    (assert_same_values if name == "nonfinite" else assert_bitexact)(rows, want, f"{name} slack x{scale}")
    if name not in ("zeros",):  # all-zero F: the bounds have zero width whatever the scale
        assert int(flags[11]) > 0, (name, scale, "no open comparison was settled exactly: test has no teeth")


def test_true_bounds_rarely_need_settling(oracle):
    """With the real slack the exact settlement is the exception on APT data: the half-width of the
    bounds is 138 u x sum|F| ~ 1e-3 of a typical correlation value, so about one candidate group in
    fifty has a position whose comparison the bounds leave open (and then one or two of the 95 groups
    in between are evaluated, not all of them).  The counter also counts the candidates k_sync_nodes
    evaluates twice (its look-behind halo), hence the loose limit."""
    torch = pytest.importorskip("torch")
    for name in ("apt48k", "heavy-noise", "apt96k"):
        x, rate = _case(name)
        plan, res, rows = _decode_plan(x, rate, torch)
        flags = plan.read_internal("picker_flags", np.uint32, 32)
        plan.close()
        assert_bitexact(rows, oracle.decode(x, rate, True), name)
        assert int(flags[11]) <= res.n_sync // 2 + 2, (name, int(flags[11]), res.n_sync)
