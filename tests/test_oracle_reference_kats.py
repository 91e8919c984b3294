"""Pin the CPU oracle against every known-answer test the reference holds for the
decode() path.  Each test names the reference test it ports.
"""
import math

import numpy as np
import pytest

FINAL_RATE = 4160


def _ulps(a, b):
    a = np.float32(a).view(np.int32).astype(np.int64)
    b = np.float32(b).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-(2 ** 31)) - a, a)
    b = np.where(b < 0, np.int64(-(2 ** 31)) - b, b)
    return int(abs(int(a) - int(b)))


# ---- src/decode.rs:271-319  test_sample_sync_frame (exact) --------------------------

def _expected_sync(pw):
    v = [-1] * (2 * pw)
    for _ in range(7):
        v += [-1] * (2 * pw) + [1] * (2 * pw)
    v += [-1] * (8 * pw)
    return v


def test_sample_sync_frame(oracle):
    # the two vectors spelled out in decode.rs:274-318 (5x and 2x FINAL_RATE)
    g5 = oracle.generate_sync_frame(FINAL_RATE * 5)
    assert g5.tolist() == (
        [-1] * 20 + ([1] * 10 + [-1] * 10) * 7 + [-1] * 30)
    assert len(g5) == 190
    g2 = oracle.generate_sync_frame(FINAL_RATE * 2)
    assert g2.tolist() == ([-1] * 8 + ([1] * 4 + [-1] * 4) * 7 + [-1] * 12)
    assert len(g2) == 76
    for pw in (1, 3, 4):
        assert oracle.generate_sync_frame(FINAL_RATE * pw).tolist() == _expected_sync(pw)


def test_sync_frame_requires_multiple_of_final_rate(oracle):
    # decode.rs:172-176
    with pytest.raises(oracle.OracleError) as e:
        oracle.generate_sync_frame(11025)
    assert e.value.code == oracle.ERR_INTERNAL
    assert str(e.value) == "work_rate is not multiple of FINAL_RATE"


# ---- src/misc.rs:493-513  test_bessel_i0 (rel 1e-3 vs GNU Octave) -------------------

BESSEL = [(0., 1.00000000000000), (0.5, 1.06348337074132), (1., 1.26606587775201),
          (1.5, 1.64672318977289), (2., 2.27958530233607), (2.5, 3.28983914405012),
          (3., 4.88079258586502), (3.5, 7.37820343222548), (4., 11.3019219521363),
          (4.5, 17.4811718556093), (5., 27.2398718236044), (5.5, 42.6946451518478),
          (6., 67.2344069764780), (6.5, 106.292858243996), (7., 168.593908510290)]


@pytest.mark.parametrize("x,expected", BESSEL)
def test_bessel_i0(oracle, x, expected):
    assert oracle.bessel_i0(x) == pytest.approx(expected, rel=1e-3)


# ---- src/filters.rs:243-366  test_lowpass / test_lowpass_dc_removal -----------------

FILTER_PARAMS = [(1. / 4., 20., 1. / 10.), (1. / 3., 35., 1. / 30.), (2. / 5., 60., 1. / 20.)]


def _abs_fft(c):
    return np.abs(np.fft.fft(c.astype(np.float64)))


@pytest.mark.parametrize("cutout,atten,delta_w", FILTER_PARAMS)
def test_lowpass(oracle, cutout, atten, delta_w):
    ripple = 10.0 ** (-atten / 20.0)
    coeff = oracle.filter_design(oracle.LOWPASS, cutout, atten, delta_w)
    assert coeff.size % 2 == 1
    fft = _abs_fft(coeff)
    for i, v in enumerate(fft):
        w = 2.0 * i / fft.size
        if w < cutout - delta_w / 2.0:
            assert 1.0 - ripple < v < 1.0 + ripple
        elif cutout + delta_w / 2.0 < w < 1.0:
            assert v < ripple


@pytest.mark.parametrize("cutout,atten,delta_w", FILTER_PARAMS)
def test_lowpass_dc_removal(oracle, cutout, atten, delta_w):
    ripple = 10.0 ** (-atten / 20.0)
    coeff = oracle.filter_design(oracle.LOWPASS_DC_REMOVAL, cutout, atten, delta_w)
    fft = _abs_fft(coeff)
    for i, v in enumerate(fft):
        w = 2.0 * i / fft.size
        if i == 0:
            assert v < 2.0 * ripple
        if delta_w < w < cutout - delta_w / 2.0:
            assert 1.0 - ripple < v < 1.0 + ripple
        elif cutout + delta_w / 2.0 < w < 1.0:
            assert v < ripple


# ---- src/filters.rs:368-423  test_no_filter / *_resample ----------------------------

def test_no_filter(oracle):
    assert oracle.filter_design(oracle.NOFILTER).tolist() == [1.0]


@pytest.mark.parametrize("kind", ["LOWPASS", "LOWPASS_DC_REMOVAL"])
def test_filter_resample(oracle, kind):
    k = getattr(oracle, kind)
    cut = oracle.freq_hz(123., 1000)
    dw = oracle.freq_hz(12., 1000)
    _, cut_r, atten, dw_r = oracle.filter_resample(k, cut, 40., dw, 1000, 3000)
    # filters.rs:384-398: `assert!(filter == expected)` is exact f32 equality
    assert np.float32(cut_r) == np.float32(oracle.freq_hz(123., 3000))
    assert np.float32(dw_r) == np.float32(oracle.freq_hz(12., 3000))
    assert atten == 40.


def test_no_filter_resample(oracle):
    assert oracle.filter_resample(oracle.NOFILTER, 0.25, 1., 0.5, 1000, 3000) == (
        oracle.NOFILTER, 0.25, 1., 0.5)


# ---- src/frequency.rs:325-416  test_frequency_conversion (10 ULP) -------------------

PI = math.pi
EQUIV = [(0.435374149659864, 1.367768230134332, 2400., 11025),
         (-0.435374149659864, -1.367768230134332, -2400., 11025),
         (0.1, 0.3141592653589793, 100., 2000), (-0.1, -0.3141592653589793, -100., 2000),
         (0., 0., 0., 11025), (1., PI, 5512.5, 11025), (-1., -PI, -5512.5, 11025),
         (2., 2. * PI, 11025., 11025), (-2., -2. * PI, -11025., 11025),
         (300., 300. * PI, 150., 1), (-300., -300. * PI, -150., 1)]


@pytest.mark.parametrize("pi_rad,rad,hz,rate", EQUIV)
def test_frequency_conversion(oracle, pi_rad, rad, hz, rate):
    for f in (np.float32(pi_rad), np.float32(oracle.freq_rad(rad)),
              np.float32(oracle.freq_hz(hz, rate))):
        assert _ulps(f, pi_rad) <= 10
        assert _ulps(oracle.freq_get_rad(f), rad) <= 10
        assert _ulps(oracle.freq_get_hz(f, rate), hz) <= 10


# ---- src/dsp.rs:420-468  test_rate_overflow / test_fast_resampling(_short) ----------

def test_rate_overflow(oracle):
    with pytest.raises(oracle.OracleError) as e:
        oracle.resample_with_filter(np.zeros(1000, np.float32), 99371, 93911, oracle.NOFILTER)
    assert e.value.code == oracle.ERR_RATE_OVERFLOW


def test_fast_resampling_ok(oracle):
    out = oracle.fast_resampling(np.zeros(1000, np.float32), 3, 2, np.zeros(100, np.float32))
    # t = 49, 51, ... < 3000
    assert out.size == len(range(49, 3000, 2))
    assert not out.any()


def test_fast_resampling_short(oracle):
    out = oracle.fast_resampling(np.zeros(100, np.float32), 3, 2, np.zeros(1000, np.float32))
    assert out.size == 0  # offset 499 >= interpolated_len 300: the while loop never runs


def test_resample_to_zero_hz(oracle):
    # dsp.rs:69-71
    with pytest.raises(oracle.OracleError) as e:
        oracle.resample_with_filter(np.zeros(10, np.float32), 48000, 0, oracle.NOFILTER)
    assert e.value.code == oracle.ERR_INTERNAL and str(e.value) == "Can't resample to 0Hz"


# ---- decode(): error paths of src/decode.rs:79-83,112-118 ---------------------------

def test_decode_too_short(oracle):
    with pytest.raises(oracle.OracleError) as e:
        oracle.decode(np.zeros(48000, np.float32), 48000)
    assert str(e.value) == "Got less than 10 rows of samples, audio file is too short"


def test_decode_constants_standard_48k(oracle):
    """Derived constants tabulated in SURVEY.md §8 for the standard profile."""
    from noaa_apt_amd.testing.synth import synth_apt
    x = synth_apt(48000, 12, 5)
    rows, st = oracle.decode(x, 48000, True, want_steps=True)
    assert st["resample_filter"].size == 959
    assert st["filter_filter"].size == 37
    n = x.size
    off = (959 - 1) // 2
    assert st["resampled"].size == -(-(n * 13 - off) // 50)  # ceil((N*l - off)/m)
    assert st["correlation"].size == st["filtered"].size - 114
    assert rows.size % 2080 == 0
    assert rows[0] == 0.0  # pixel (0,0) is 0: `i > j` guard in filter([1.]) (dsp.rs:399)
