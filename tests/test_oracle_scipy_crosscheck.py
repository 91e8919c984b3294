"""A THIRD-PARTY check of the oracle's structure: scipy.signal's implementations of the published algorithms
(polyphase up-FIR-down, direct-form FIR, cross-correlation, frequency response) in float64, with the reference's
documented quirks applied by hand, against oracle/apt_oracle.c.

tests/np_model.py was written by the same hand as the oracle; this file uses nothing of ours on the scipy side, so a
shared structural misreading of the Rust (tap order, centring, which samples are skipped, the template's direction)
would show here.  It does NOT pin decode()'s bits — float64 library code cannot — and it does not replace the
comparison with the reference binary (oracle/ref_harness/, `make -C oracle _ref`): tolerance 1e-5 of the output's
scale, the accumulated rounding of <= ~2000 sequential f32 operations being ~1e-6.

Reference: dsp.rs:186-289 (fast_resampling), dsp.rs:386-410 (filter), dsp.rs:350-383 (demodulate),
decode.rs:171-233 (sync template and correlation), filters.rs:57-183 (designs).
"""
import numpy as np
import pytest

scipy_signal = pytest.importorskip("scipy.signal")

from noaa_apt_amd.testing.synth import synth_apt, synth_noise  # noqa: E402

f32 = np.float32
RTOL = 1e-5


def _close(got, want64, what):
    got = np.asarray(got, np.float64)
    want64 = np.asarray(want64, np.float64)
    assert got.shape == want64.shape, (what, got.shape, want64.shape)
    if got.size == 0:
        return
    scale = float(np.max(np.abs(want64))) or 1.0
    err = float(np.max(np.abs(got - want64))) / scale
    assert err <= RTOL, (what, err)


def _upfirdn_reference_centring(x, l, m, h):
    """fast_resampling (dsp.rs:186-289) through scipy.signal.upfirdn.

    The reference walks t = off, off + m, ... < N*l over the zero-stuffed signal xu (xu[n*l] = x[n]) and sums
    coeff[n + off - t] * xu[n] for n in [t - off, t + off], off = (T - 1) / 2: tap index ASCENDING with n — a
    correlation with h, centred on t, i.e. a convolution with h reversed.  upfirdn(g, x, up=l)[j] = sum_n g[j - n] xu[n]
    with g = h[::-1] gives sum_n h[T - 1 - j + n] xu[n]; the two agree for j = t + off.  Samples at or past N*l are
    skipped by the reference (dsp.rs:257) and are zeros of the full convolution here.  Output k is therefore centred
    on up-sampled index off + k*m (the first off/m outputs a conventional resampler would produce do not exist), and
    there are ceil((N*l - off) / m) of them."""
    T = len(h)
    off = (T - 1) // 2
    y = scipy_signal.upfirdn(np.asarray(h, np.float64)[::-1], np.asarray(x, np.float64), up=l, down=1)
    n_up = len(x) * l
    ts = np.arange(off, n_up, m)
    # (the full convolution has N*l - l + T entries; t + off can point past it only where every term is a skipped one)
    idx = ts + off
    out = np.zeros(len(ts))
    ok = idx < len(y)
    out[ok] = y[idx[ok]]
    return out


@pytest.mark.parametrize("rate,l,m", [(48000, 13, 50), (96000, 13, 100), (11025, 832, 735), (44100, 208, 735),
                                      (22050, 416, 735)])
def test_fast_resampling_vs_upfirdn_designed_taps(oracle, rate, l, m):
    x = synth_noise(rate, 0.25, 5)
    cut = oracle.freq_hz(4800., rate)
    dw = oracle.freq_hz(1000., rate)
    _, cut_l, _, dw_l = oracle.filter_resample(oracle.LOWPASS_DC_REMOVAL, cut, 30., dw, rate, rate * l)
    h = oracle.filter_design(oracle.LOWPASS_DC_REMOVAL, cut_l, 30., dw_l)
    got = oracle.fast_resampling(x, l, m, h)
    want = _upfirdn_reference_centring(x, l, m, h)
    assert got.size == -(-(x.size * l - (h.size - 1) // 2) // m)  # ceil((N*l - off) / m), SURVEY.md 8(a) A4
    _close(got, want, f"fast_resampling {rate}")


def test_fast_resampling_vs_upfirdn_asymmetric_taps(oracle):
    """Random (asymmetric) taps: a convolution / correlation mix-up cannot hide behind the designs' symmetry; taps
    longer than the signal, l > m, l = 1-adjacent factors."""
    rng = np.random.default_rng(11)
    for (n, l, m, t) in [(1000, 3, 2, 101), (100, 3, 2, 1001), (50, 7, 5, 33), (17, 2, 3, 5), (64, 5, 1, 21),
                         (1, 13, 50, 959), (200, 13, 50, 959), (40, 3, 7, 1), (300, 13, 100, 1915)]:
        x = rng.standard_normal(n).astype(f32)
        h = rng.standard_normal(t).astype(f32)
        _close(oracle.fast_resampling(x, l, m, h), _upfirdn_reference_centring(x, l, m, h), (n, l, m, t))


def test_filter_vs_lfilter(oracle):
    """filter() (dsp.rs:386-410) is the causal direct form with the `i > j` guard: x[0] never contributes and out[0]
    is 0 — scipy.signal.lfilter on the signal with its first sample zeroed."""
    rng = np.random.default_rng(12)
    c2 = f32(4160) / f32(12480)
    designed = oracle.filter_design(oracle.LOWPASS, c2, 25., c2 / f32(5))
    for h in (designed, rng.standard_normal(61).astype(f32), np.array([1.0], f32)):
        for n in (1, 5, 36, 37, 38, 4000):
            x = (rng.standard_normal(n) * 1000).astype(f32)
            xz = x.astype(np.float64).copy()
            xz[0] = 0.0
            want = scipy_signal.lfilter(np.asarray(h, np.float64), [1.0], xz)
            got = oracle.fir(x, h)
            assert got[0] == 0.0
            _close(got, want, ("fir", len(h), n))


def test_sync_correlation_vs_scipy_correlate(oracle):
    """find_sync's correlation (decode.rs:225-233): corr[i] = sum_j guide[j] * F[i + j] for i in 0 .. W - G — the
    first W - G entries of the 'valid' cross-correlation with the +-1 template of generate_sync_frame."""
    for work in (12480, 16640, 20800):
        g = oracle.generate_sync_frame(work).astype(np.float64)
        assert g.size == 38 * (work // 4160) and set(np.unique(g)) == {-1.0, 1.0}
        x = synth_apt(work, 3.0, 21)  # any signal with structure; find_sync takes it as the filtered work-rate signal
        F = np.abs(x).astype(f32)
        _, corr = oracle.find_sync(F, work, return_correlation=True)
        want = scipy_signal.correlate(F.astype(np.float64), g, mode="valid")[:F.size - g.size]
        assert corr.size == F.size - g.size
        _close(corr, want, ("correlation", work))


def test_demodulate_is_the_two_sample_envelope_at_twice_the_angle(oracle):
    """demodulate (dsp.rs:350-383) is the two-sample envelope estimator sqrt(p^2 + c^2 - 2 p c cos(phi)) / sin(phi),
    which is exact for a carrier that advances phi per sample — and the reference sets phi = 2 * get_rad()
    (dsp.rs:360; get_rad() already is 2 pi f / rate), i.e. TWICE the 2400 Hz carrier's advance.  Checked against the
    analytic envelope: exact (to f32 rounding and the slow modulation) for a carrier at 2 * 2400 Hz, and NOT the
    envelope of a 2400 Hz carrier — the quirk is the reference's, the oracle and the kernels reproduce it."""
    work, carrier = 12480, 2400.0
    t = np.arange(20000) / work
    a = 1000.0 * (1.0 + 0.5 * np.sin(2 * np.pi * 3.0 * t))
    x2 = (a * np.cos(2 * np.pi * (2 * carrier) * t + 0.3)).astype(f32)
    got = oracle.demodulate(x2, oracle.freq_hz(carrier, work))
    assert got[0] == 0.0
    # (the estimator assumes a constant envelope over its two samples: a 3 Hz modulation moves it by ~1e-3 of full scale)
    err = np.max(np.abs(got[1:].astype(np.float64) - a[1:])) / 1500.0
    assert err < 5e-3, err
    x1 = (a * np.cos(2 * np.pi * carrier * t + 0.3)).astype(f32)
    got1 = oracle.demodulate(x1, oracle.freq_hz(carrier, work))
    assert np.max(np.abs(got1[1:].astype(np.float64) - a[1:])) / 1500.0 > 0.2
    # and the formula itself in float64, on noise
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(5000) * 3000).astype(f32)
    phi = 2.0 * (2.0 * carrier / work) * np.pi
    p, c = x[:-1].astype(np.float64), x[1:].astype(np.float64)
    want = np.concatenate([[0.0], np.sqrt(np.maximum(p * p + c * c - p * c * 2 * np.cos(phi), 0.0)) / np.sin(phi)])
    _close(oracle.demodulate(x, oracle.freq_hz(carrier, work)), want, "demodulate formula")


@pytest.mark.parametrize("kind,cut_hz,atten,dw_hz,rate", [("lowpass", 4160.0 / 2, 25.0, 4160.0 / 10, 12480),
                                                           ("dcremoval", 4800.0, 30.0, 1000.0, 48000),
                                                           ("dcremoval", 4800.0, 40.0, 500.0, 48000)])
def test_designs_meet_their_spec_by_freqz(oracle, kind, cut_hz, atten, dw_hz, rate):
    """The designed taps through scipy.signal.freqz: unity pass band within the Kaiser ripple, stop band below
    -atten dB (the bounds filters.rs:243-366 checks with its own FFT), zero gain at DC for LowpassDcRemoval."""
    cut, dw = oracle.freq_hz(cut_hz, rate), oracle.freq_hz(dw_hz, rate)
    k = oracle.LOWPASS if kind == "lowpass" else oracle.LOWPASS_DC_REMOVAL
    h = oracle.filter_design(k, cut, atten, dw).astype(np.float64)
    assert h.size % 2 == 1 and np.allclose(h, h[::-1], rtol=0, atol=1e-9)  # linear phase
    w, H = scipy_signal.freqz(h, worN=8192)
    mag = np.abs(H)
    ripple = 10 ** (-atten / 20)
    c, d = cut * np.pi, dw * np.pi
    lo_edge = d if kind == "dcremoval" else 0.0       # (DC removal: a transition band of dw above 0 too)
    pas = (w >= lo_edge + d / 2) & (w <= c - d / 2)
    stop = w >= c + d / 2
    assert pas.any() and stop.any()
    assert np.max(np.abs(mag[pas] - 1.0)) <= 1.5 * ripple, float(np.max(np.abs(mag[pas] - 1.0)))
    assert np.max(mag[stop]) <= 1.5 * ripple, float(np.max(mag[stop]))
    if kind == "dcremoval":
        assert mag[0] <= 2.0 * ripple  # (filters.rs:336-339: "2*ripple otherwise it fails ... it's only 3dB")
