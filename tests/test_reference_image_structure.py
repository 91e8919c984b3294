"""The one artefact of the reference's OUTPUT that exists — docs/examples/argentina.png, an image the reference decoded
from a real NOAA pass (its input WAV is not in the repository) — as a check of decode()'s row GEOMETRY: a recording
synthesised from a band of that image's own rows (tests/golden/reference_image/, cut by make_reference_image_rows.py)
must decode to rows whose sync A, sync B and telemetry columns sit where the reference's image has them
(decode.rs:16-35: 39 px sync, 47 px space, 909 px image, 45 px telemetry, twice).  This pins row length, row start
and column offsets to something the reference produced; it does NOT pin pixel values (oracle)."""
import os

import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt

HERE = os.path.dirname(os.path.abspath(__file__))
BAND = np.load(os.path.join(HERE, "golden", "reference_image", "argentina_rows.npy"))  # 96 x 2080 u8


def _check_geometry(rows_flat, what):
    assert rows_flat.size % 2080 == 0 and rows_flat.size >= 60 * 2080, (what, rows_flat.size)
    got = rows_flat.reshape(-1, 2080).astype(np.float64)
    src = BAND.astype(np.float64)
    n_src = src.shape[0]

    def ncc(a, b):
        a = a - a.mean()
        b = b - b.mean()
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))

    # 1. every decoded row (but the first and the last two: the picker's edge rows) is one row of the source band — the
    # same one modulo the band's length for consecutive rows — and lines up with it at ONE column lag, the same for
    # every row, of at most 2 px.  (Not 0: the reference's image carries its sync A pulses at columns 2.5-4.5 where the
    # template of decode.rs:188-198 has them at 4-6, and a decode puts what it is sent half a pixel late — the causal
    # low-pass is not delay-compensated, dsp.rs:386-410 — so this SECOND-generation decode sits a constant 2 px right.)
    # (neighbouring rows of a real image resemble each other: the row offset is the most frequent best match, and every
    # row must then match ITS row of the band)
    offs = [(max(range(n_src), key=lambda s: ncc(got[r], src[s])) - r) % n_src for r in range(2, got.shape[0] - 2)]
    k = max(set(offs), key=offs.count)
    lag_of_row = []
    for r in range(2, got.shape[0] - 2):
        mine = src[(r + k) % n_src]
        lags = {lag: ncc(got[r, 16:-16], np.roll(mine, lag)[16:-16]) for lag in range(-6, 7)}
        lag_of_row.append(max(lags, key=lags.get))
        assert lags[lag_of_row[-1]] > 0.93, (what, r, lags)
    assert len(set(lag_of_row)) == 1 and abs(lag_of_row[0]) <= 2, (what, lag_of_row)
    lag = lag_of_row[0]
    # 2. the column layout of decode.rs:16-35, read off the reference's own image: after that shift and the affine map
    # that takes the decoded amplitudes to pixel values, the column means of the four marker regions agree with the image's
    want = np.roll(np.stack([src[(r + k) % n_src] for r in range(2, got.shape[0] - 2)]).mean(axis=0), lag)
    have = got[2:-2].mean(axis=0)
    a, b = np.polyfit(have, want, 1)
    err = np.abs(a * have + b - want)
    for name, lo, hi in (("telemetry A", 995, 1040), ("telemetry B", 2035, 2076)):   # flat regions: absolute agreement
        assert err[lo:hi].max() < 12.0, (what, name, float(err[lo:hi].max()))
    for name, lo, hi in (("sync A + space", 0, 86), ("sync B + space", 1040, 1126)):  # pulse trains: shape agreement
        assert ncc(have[lo:hi], want[lo:hi]) > 0.9, (what, name, ncc(have[lo:hi], want[lo:hi]))
    # the seven sync A pulses: maxima every 4 px, minima in between (the template of decode.rs:188-198); sync B: seven
    # pulses every 5 px — at the columns the reference's image has them, plus the lag
    span = have.max() - have.min()
    assert all(have[3 + lag + 4 * i] > have[5 + lag + 4 * i] + 0.4 * span for i in range(7)), what
    assert all(have[1043 + lag + 5 * i] > have[1046 + lag + 5 * i] + 0.4 * span for i in range(7)), what


@pytest.mark.parametrize("rate", [11025, 48000])
def test_oracle_rows_have_the_reference_images_geometry(oracle, rate):
    x = synth_apt(rate, 40.0, seed=11, image=BAND, noise_sigma=100.0, start_px=1000.0)
    _check_geometry(oracle.decode(x, rate, True), f"oracle {rate}")


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [11025, 48000])
def test_product_rows_have_the_reference_images_geometry(oracle, rate):
    x = synth_apt(rate, 40.0, seed=11, image=BAND, noise_sigma=100.0, start_px=1000.0)
    rows = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(rate), True)
    _check_geometry(rows, f"product {rate}")
    assert np.array_equal(rows.view(np.uint32), oracle.decode(x, rate, True).view(np.uint32))
