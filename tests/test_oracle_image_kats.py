"""The image-side oracle (contrast limits, u8 mapping, telemetry) pinned by the reference's own
tests, re-stated here value for value, plus an independent numpy-f32 re-derivation.

Reference tests mirrored:
    noaa_apt.rs:266-281   test_map                      (exact u8 vector)
    misc.rs:515-543       test_percent                  (1 % bounds)
    telemetry.rs:255-311  test_telemetry_from_bands     (10 ULP)
    telemetry.rs:313-348  test_telemetry_get_channel    (exact names)
"""
import numpy as np
import pytest

from oracle import image_binding as oi
from oracle.binding import OracleError

f32 = np.float32


def _ulps(a, b):
    a, b = f32(a), f32(b)
    return abs(int(a.view(np.int32)) - int(b.view(np.int32)))


def test_map_reference_vector():
    expected = [0, 0, 0, 0, 1, 2, 50, 120, 200, 255, 255, 255]
    values = np.array([-10., -5., -1., 0., 1., 2.4, 50., 120., 199.6, 255., 256., 300.], f32)
    shifted = values * f32(123.123) - f32(234.234)
    low = f32(0.) * f32(123.123) - f32(234.234)
    high = f32(255.) * f32(123.123) - f32(234.234)
    assert oi.map_signal_u8(shifted, low, high).tolist() == expected


def test_percent_reference_bounds():
    sig = np.arange(10000, dtype=f32)
    for value in (1., 0.95, 0.90, 0.80, 0.50):
        lo, hi = oi.percent(sig, value)
        rem = (1. - value) / 2.
        assert rem - 0.005 < lo / 10000. < rem + 0.005
        assert 1. - (rem + 0.005) < hi / 10000. < 1. - (rem - 0.005)


def test_percent_rejects_out_of_range():
    for p in (-0.1, 1.1):
        with pytest.raises(OracleError, match="Percent given should be between 0 and 1"):
            oi.percent(np.arange(10, dtype=f32), p)


def test_min_max_empty_errors():
    with pytest.raises(OracleError, match="maximum of a zero length"):
        oi.get_max(np.zeros(0, f32))
    with pytest.raises(OracleError, match="minimum of a zero length"):
        oi.get_min(np.zeros(0, f32))


def _bands_fixture():
    wedge = np.array([1., 1.2, 0.8, 1.1, 0.9, 0.7, 1.3, 1.], f32)
    factors = [-5234.] + list(range(1, 17)) + list(range(1, 10)) + [-5234.]
    means_a = np.concatenate([wedge * f32(k) for k in factors]).astype(f32)
    return means_a, (means_a + f32(1.)).astype(f32)


def test_telemetry_from_bands_reference():
    means_a, means_b = _bands_fixture()
    t = oi.telemetry_from_bands(means_a, means_b, 8)
    for wedge in range(1, 17):
        assert _ulps(t.get_wedge_value(wedge, "A"), wedge) <= 10
        assert _ulps(t.get_wedge_value(wedge, "B"), wedge + 1.) <= 10
        assert _ulps(t.get_wedge_value(wedge, None), wedge + 0.5) <= 10


def test_telemetry_get_channel_reference():
    means = [1., 2., 3., 4., 5., 6., 7., 8., 9., 3., 3., 3., 3., 3., 3.]
    cases = [("1", 1., "2", 2.), ("3a", 3., "3b", 6.), ("4", 4., "5", 5.),
             ("Unknown", 7., "Unknown", 8.), ("Unknown", 9., "Unknown", 1000.),
             ("1", 1.4, "2", 1.6), ("3a", 2.6, "3a", 3.4), ("1", -1000., "5", 5.4)]
    for name_a, va, name_b, vb in cases:
        t = oi.Telemetry(means + [va], means + [vb])
        assert t.get_channel_name("A") == name_a
        assert t.get_channel_name("B") == name_b


# ------------------------------------------------------------------ numpy-f32 re-derivation
def _np_percent(x, p):
    x = np.asarray(x, f32)
    rem = (f32(1.) - f32(p)) / f32(2.)
    mn, mx = x.min(), x.max()
    rng = f32(mx - mn)
    with np.errstate(all="ignore"):
        b = np.trunc((x - mn) / rng * f32(1000.))
    b = np.where(np.isnan(b), 0, b)
    b = np.clip(b, 0, 999).astype(np.int64)
    counts = np.bincount(b, minlength=1000).astype(np.uint32)
    acc = np.cumsum(counts.astype(np.uint64)).astype(np.uint32)
    frac = acc.astype(f32) / f32(x.size)
    low_b = high_b = None
    for i in range(1000):
        if low_b is None and frac[i] > rem:
            low_b = i
        elif high_b is None and frac[i] > f32(1.) - rem:
            high_b = i
    if high_b is None:
        high_b = 999
    return (f32(low_b) / f32(1000.) * rng + mn, f32(high_b) / f32(1000.) * rng + mn, counts)


def _np_map(x, low, high):
    x = np.asarray(x, f32)
    rng = f32(high) - f32(low)
    with np.errstate(all="ignore"):
        v = (x - f32(low)) / rng * f32(255.)
    v = np.where(np.isnan(v), f32(0.), v)
    v = np.minimum(np.maximum(v, f32(0.)), f32(255.))
    # round half away from zero (v >= 0 here)
    r = np.floor(v)
    r = np.where(v - r >= f32(0.5), r + 1, r)
    return r.astype(np.uint8)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_percent_and_map_vs_numpy(seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(200_000) * 3000 + 9000).astype(f32)
    if seed == 2:
        x = np.abs(x) ** f32(1.7)
    for p in (0.98, 0.9, 1.0, 0.0):
        lo, hi, counts = oi.percent(x, p, want_buckets=True)
        nlo, nhi, ncounts = _np_percent(x, p)
        assert np.array_equal(counts, ncounts)
        assert lo.tobytes() == f32(nlo).tobytes() and hi.tobytes() == f32(nhi).tobytes()
        assert np.array_equal(oi.map_signal_u8(x, lo, hi), _np_map(x, lo, hi))
    assert oi.get_min(x).tobytes() == x.min().tobytes()
    assert oi.get_max(x).tobytes() == x.max().tobytes()


def test_map_degenerate_ranges():
    x = np.array([1., 2., 3., np.nan, np.inf, -np.inf], f32)
    # range 0: (x-low)/0 -> +-inf or NaN; NaN -> 0
    assert oi.map_signal_u8(x, 2., 2.).tolist() == [0, 0, 255, 0, 255, 0]
    # inverted range
    assert oi.map_signal_u8(x[:3], 3., 1.).tolist() == [255, 128, 0]


def test_percent_constant_signal():
    x = np.full(5000, 7.5, f32)
    lo, hi = oi.percent(x, 0.98)
    assert lo == f32(7.5) and hi == f32(7.5)


def _np_read_telemetry(sig):
    sig = np.asarray(sig, f32)
    rows = sig.size // 2080
    img = sig[:rows * 2080].reshape(rows, 2080)
    a, b = img[:, 994:994 + 44], img[:, 2034:2034 + 44]

    def seq_sum(m):  # left-to-right f32 sum along axis 1
        s = np.zeros(m.shape[0], f32)
        for j in range(m.shape[1]):
            s = s + m[:, j]
        return s
    ma, mb = seq_sum(a) / f32(44.), seq_sum(b) / f32(44.)
    da, db = a - ma[:, None], b - mb[:, None]
    var = (seq_sum(da * da) + seq_sum(db * db)) / f32(88.)
    wedges = [31., 63., 95., 127., 159., 191., 224., 255., 0.] + [0.] * 7 + \
             [31., 63., 95., 127., 159., 191., 224., 255., 0.]
    sample = np.repeat(np.array(wedges, f32), 8)
    nc = rows - 200
    corr = np.zeros(nc, f32)
    sd = np.zeros(nc, f32)
    sq = np.sqrt(var)
    for j in range(200):
        corr = corr + sample[j] * ma[j:j + nc]
        corr = corr + sample[j] * mb[j:j + nc]
        sd = sd + sq[j:j + nc]
    with np.errstate(all="ignore"):
        q = corr / sd
    best, best_q = 0, f32(0.)
    for i in range(nc):
        if q[i] > best_q:
            best, best_q = i, q[i]
    return ma, mb, var, corr, q, best


def test_read_telemetry_vs_numpy():
    from noaa_apt_amd.testing.synth import make_image
    rng = np.random.default_rng(5)
    img = make_image(420, seed=9)
    sig = (img * f32(37.5) + rng.standard_normal(img.shape).astype(f32) * f32(120.)).astype(f32).ravel()
    t = oi.read_telemetry(sig)
    ma, mb, var, corr, q, best = _np_read_telemetry(sig)
    for name, want in (("telemetry_a", ma), ("telemetry_b", mb), ("telemetry_variance", var),
                       ("telemetry_correlation", corr), ("telemetry_quality", q)):
        assert np.array_equal(t.steps[name].view(np.uint32), want.view(np.uint32)), name
    assert t.row == best
    # the synthetic wedge cycle is 128 rows long and starts at row 0: wedge 1 begins at a
    # multiple of 128
    assert t.row % 128 == 0
    want = oi.telemetry_from_bands(ma, mb, best)
    assert np.array_equal(t.values_a, want.values_a) and np.array_equal(t.values_b, want.values_b)
    # wedge 8 (brightest) above wedge 9 (black): contrast limits are ordered
    assert t.get_wedge_value(8) > t.get_wedge_value(9)


def test_read_telemetry_too_short():
    with pytest.raises(OracleError, match="Recording too short for telemetry decoding"):
        oi.read_telemetry(np.zeros(199 * 2080, f32))


@pytest.mark.parametrize("contrast", [oi.CONTRAST_TELEMETRY, oi.CONTRAST_PERCENT, oi.CONTRAST_MINMAX])
def test_process_gray_composition(contrast):
    from noaa_apt_amd.testing.synth import make_image
    sig = (make_image(300, seed=4) * f32(11.)).astype(f32).ravel()
    img, lo, hi = oi.process_gray(sig, contrast, 0.98)
    if contrast == oi.CONTRAST_TELEMETRY:
        t = oi.read_telemetry(sig)
        want = (t.get_wedge_value(9), t.get_wedge_value(8))
    elif contrast == oi.CONTRAST_PERCENT:
        want = oi.percent(sig, 0.98)
    else:
        want = (oi.get_min(sig), oi.get_max(sig))
    assert (lo, hi) == want
    assert np.array_equal(img, oi.map_signal_u8(sig, lo, hi))
