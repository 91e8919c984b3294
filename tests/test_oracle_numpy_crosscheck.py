"""Cross-check oracle/apt_oracle.c against the independent numpy-f32 model
(tests/np_model.py).  Bit-exact wherever only +,-,*,/,sqrt are involved."""
import numpy as np
import pytest

from noaa_apt_amd.testing.synth import synth_apt, synth_noise
from tests import np_model as M

f32 = np.float32


def _eq(a, b):
    a = np.asarray(a, f32)
    b = np.asarray(b, f32)
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("atten,dw", [(30., 0.0032051282), (25., 1 / 15), (40., 0.01), (60., 0.05), (20., 0.1)])
def test_kaiser_bitexact(oracle, atten, dw):
    _eq(oracle.kaiser(atten, dw), M.kaiser(atten, dw))


def test_designs_bitexact(oracle):
    cut = oracle.freq_hz(4800., 48000)
    dw = oracle.freq_hz(1000., 48000)
    _, cut13, _, dw13 = oracle.filter_resample(oracle.LOWPASS_DC_REMOVAL, cut, 30., dw, 48000, 48000 * 13)
    _eq(oracle.filter_design(oracle.LOWPASS_DC_REMOVAL, cut13, 30., dw13),
        M.design("dcremoval", cut13, 30., dw13))
    c2 = f32(4160) / f32(12480)
    _eq(oracle.filter_design(oracle.LOWPASS, c2, 25., c2 / f32(5)),
        M.design("lowpass", c2, 25., f32(c2 / f32(5))))


@pytest.mark.parametrize("rate,l,m", [(48000, 13, 50), (96000, 13, 100), (11025, 832, 735), (44100, 208, 735)])
def test_polyphase_reformulation_bitexact(oracle, rate, l, m):
    x = synth_noise(rate, 0.35, 3)
    cut = oracle.freq_hz(4800., rate)
    dw = oracle.freq_hz(1000., rate)
    _, cut_l, _, dw_l = oracle.filter_resample(oracle.LOWPASS_DC_REMOVAL, cut, 30., dw, rate, rate * l)
    coeff = oracle.filter_design(oracle.LOWPASS_DC_REMOVAL, cut_l, 30., dw_l)
    _eq(oracle.fast_resampling(x, l, m, coeff), M.resample_poly(x, l, m, coeff))


def test_polyphase_edge_cases(oracle):
    rng = np.random.default_rng(0)
    for (n, l, m, t) in [(1000, 3, 2, 101), (100, 3, 2, 1001), (50, 7, 5, 33), (17, 2, 3, 5),
                         (64, 5, 1, 21), (1, 13, 50, 959), (200, 13, 50, 959), (40, 3, 7, 1)]:
        x = rng.standard_normal(n).astype(f32)
        c = rng.standard_normal(t).astype(f32)
        _eq(oracle.fast_resampling(x, l, m, c), M.resample_poly(x, l, m, c))


def test_stages_bitexact(oracle):
    x = synth_apt(48000, 11, 11)
    rows, st = oracle.decode(x, 48000, True, want_steps=True)
    _eq(st["demodulated"], M.demodulate(st["resampled"], 12480))
    _eq(st["filtered"], M.fir_causal(st["demodulated"], st["filter_filter"]))
    g = M.sync_template(12480)
    assert np.array_equal(g, oracle.generate_sync_frame(12480))
    _eq(st["correlation"], M.correlate(st["filtered"], g))
    peaks = M.find_sync_orbit(st["correlation"], 6240, 4992)
    assert peaks == st["sync_pos"].astype(np.int64).tolist()
    _eq(rows, M.gather_rows(st["filtered"], peaks, 6240))


def _fsm_cases():
    rng = np.random.default_rng(42)
    n = 2080 * 23 + 977
    cases = {
        "zeros": np.zeros(n, f32),
        "noise": rng.standard_normal(n).astype(f32),
        "noise_pos": (rng.standard_normal(n) + 5).astype(f32),
        "noise_neg": (rng.standard_normal(n) - 5).astype(f32),
        "ramp_up": np.arange(n, dtype=f32),
        "ramp_down": -np.arange(n, dtype=f32),
        "plateaus": np.repeat(rng.integers(0, 4, n // 64 + 1), 64)[:n].astype(f32),
        "quantised": rng.integers(-3, 4, n).astype(f32),
        "sparse_spikes": np.where(rng.random(n) < 0.0007, rng.random(n) * 100, 0).astype(f32),
        "slow_sine": np.sin(np.arange(n) / 700.0).astype(f32),
        "rising_sine": (np.sin(np.arange(n) / 37.0) + np.arange(n) / 900.0).astype(f32),
    }
    return cases


@pytest.mark.parametrize("name", list(_fsm_cases().keys()))
@pytest.mark.parametrize("work_rate", [4160, 8320])
def test_peak_fsm_reformulation(oracle, name, work_rate):
    """terminals+orbit == the reference's sequential FSM, on adversarial inputs."""
    f = _fsm_cases()[name]
    spr = 2080 * work_rate // 4160
    md = spr * 8 // 10
    pos, corr = oracle.find_sync(f, work_rate, return_correlation=True)
    assert M.find_sync_orbit(corr, spr, md) == pos.astype(np.int64).tolist()


def test_peak_fsm_random_sweep(oracle):
    rng = np.random.default_rng(7)
    for trial in range(40):
        n = int(rng.integers(2080 * 3, 2080 * 30))
        kind = trial % 4
        if kind == 0:
            f = rng.standard_normal(n)
        elif kind == 1:
            f = np.cumsum(rng.standard_normal(n)) * 0.05
        elif kind == 2:
            f = rng.integers(-2, 3, n)
        else:
            f = np.sin(np.arange(n) * rng.uniform(0.001, 0.1)) * rng.uniform(0.1, 10) + rng.standard_normal(n) * 0.2
        f = f.astype(f32)
        pos, corr = oracle.find_sync(f, 4160, return_correlation=True)
        assert M.find_sync_orbit(corr, 2080, 1664) == pos.astype(np.int64).tolist(), trial
