"""APTGPU_MODE_FAST with the resampler on the matrix cores (kModeMfma, csrc/apt_kernels_fused_launch.hpp): stage 1 of the
48 / 96 kHz standard-profile front ends as a banded Toeplitz product through v_mfma_f32_16x16x32_bf16 on bf16 pieces of
the f32 taps (three: exact) and samples (two: exact for 16-bit data), f32 accumulation — against the oracle with fast
mode's stated tolerance (tests/test_gpu_fast.py):

  * row count identical; sync positions identical on >= 99.9 % of the rows, never off by more than one work sample;
  * on rows with identical position max |px - ref| <= 1e-4 max |ref|

— and tighter than that where it is a property of the arithmetic: the pixels agree with the oracle to PX_TIGHT of full
scale whatever the scale of the input (bf16 carries f32's exponent: nothing is scaled), with mono PCM16 input, with a
user-tuned tap count (the table is zero-padded: no "exact tap count" dispatch in this mode — the plans this kernel is
selected for by default), and a tile that holds a NaN or an infinity takes the scalar path (the outputs of the VALU
fast kernel).  `APTGPU_FAST_MFMA=1` (read at plan creation) selects it for the stock tap counts too, `=0` never: the
A/B switch these tests use.
"""
import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
from test_gpu_fast import check_tolerance, decode_on_plan, PX_TOL

pytestmark = pytest.mark.gpu

f32 = np.float32
PX_TIGHT = 5e-6  # of max |ref px| (measured: ~5e-7; the VALU fast kernels measure 4e-7)


def _valu(monkeypatch):
    monkeypatch.setenv("APTGPU_FAST_MFMA", "0")


@pytest.fixture(autouse=True)
def _mfma(monkeypatch):
    monkeypatch.setenv("APTGPU_FAST_MFMA", "1")


@pytest.mark.parametrize("rate,seconds,kw", [
    (48000, 14, dict(seed=2)),
    (48000, 40, dict(seed=12, ppm=40.0)),
    (96000, 12, dict(seed=3)),
    (48000, 20, dict(seed=8, noise_sigma=6000.0)),
    (48000, 20, dict(seed=9, amplitude=2000.0)),
])
def test_mfma_tolerance_and_it_is_the_matrix_path(oracle, monkeypatch, rate, seconds, kw):
    x = synth_apt(rate, seconds, **kw)
    want, st = oracle.decode(x, rate, True, want_steps=True)
    rows, pos, res, fused = decode_on_plan(x, rate, apt.MODE_FAST)
    assert fused == 1 and res.status == 0
    frac, err = check_tolerance(rows, pos, want, st["sync_pos"], f"mfma {rate} {kw}")
    assert 0 < err <= PX_TIGHT, err
    # the VALU fast kernel on the same input: another arithmetic (other bits), the same tolerance
    _valu(monkeypatch)
    rows_v, pos_v, _, fused_v = decode_on_plan(x, rate, apt.MODE_FAST)
    assert fused_v == 1
    assert rows_v.shape == rows.shape and not np.array_equal(rows_v.view(np.uint32), rows.view(np.uint32))
    check_tolerance(rows_v, pos_v, want, st["sync_pos"], "valu fast")


@pytest.mark.parametrize("scale_exp", [-15, -40, 14, 40])
def test_mfma_any_input_scale(oracle, scale_exp):
    """A float WAV's +-1 range, 2^29-valued samples (32-bit integer WAVs arrive unscaled, wav.rs:37), and values far out
    in the f32 range decode as well as 16-bit ones: the pieces are bf16, nothing is scaled.  (The oracle runs on the scaled
    input itself; beyond 2^+-45 or so its own envelope squares leave the f32 range.)"""
    x = synth_apt(48000, 12, seed=21) * f32(2.0 ** scale_exp)
    want, st = oracle.decode(x, 48000, True, want_steps=True)
    rows, pos, res, fused = decode_on_plan(x, 48000, apt.MODE_FAST)
    assert fused == 1 and res.status == 0
    frac, err = check_tolerance(rows, pos, want, st["sync_pos"], f"scale 2^{scale_exp}")
    assert err <= PX_TIGHT, err


def test_mfma_mixed_scales_inside_one_recording(oracle):
    """Loud and quiet passages (40 dB apart) in one recording."""
    x = synth_apt(48000, 16, seed=22)
    g = np.where((np.arange(x.size) // 9000) % 2 == 0, f32(1.0), f32(0.01)).astype(f32)
    x = (x * g).astype(f32)
    want, st = oracle.decode(x, 48000, True, want_steps=True)
    rows, pos, res, _ = decode_on_plan(x, 48000, apt.MODE_FAST)
    frac, err = check_tolerance(rows, pos, want, st["sync_pos"], "mixed scales")
    assert err <= PX_TIGHT, err


def test_mfma_pcm16_input(oracle):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    for rate in (48000, 96000):
        x = synth_apt(rate, 12, 7)
        want, st = oracle.decode(x, rate, True, want_steps=True)
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(rate), True, max_samples=x.size, mode=apt.MODE_FAST)
        d_pcm = torch.from_numpy(x.astype(np.int16)).to(dev)
        cap = int(plan.info.max_rows)
        d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        spec = apt.WavSpec(1, 16, 2, 0, rate, 1, 0, 2 * x.size, x.size, x.size)
        plan.decode_device_wav([d_pcm.data_ptr()], [spec], [d_out.data_ptr()], [cap])
        res = plan.results(1)[0]
        pos = plan.sync_positions(0)
        frac, err = check_tolerance(d_out[:res.n_out].cpu().numpy(), pos, want, st["sync_pos"], f"pcm16 {rate}")
        assert err <= PX_TIGHT, err
        plan.close()


@pytest.mark.parametrize("rate,kw", [(48000, dict(resample_atten=29.0)), (48000, dict(resample_atten=31.0)),
                                     (48000, dict(resample_delta_freq=1100.0)), (48000, dict(resample_delta_freq=950.0)),
                                     (96000, dict(resample_atten=32.0)), (96000, dict(resample_delta_freq=920.0))])
def test_mfma_takes_tuned_tap_counts(oracle, monkeypatch, rate, kw):
    """default_settings.toml:108-140 is a user-editable file: a tuned attenuation / transition width changes the tap COUNT
    (959 / 1915 at the stock values).  This mode's table is zero-padded to the kernel's K, so such a plan stays on the
    specialised kernel (stats.fused == 1) while its taps per branch fit."""
    monkeypatch.delenv("APTGPU_FAST_MFMA")  # (the default: this kernel where the exact-count ones do not apply)
    s = apt.Settings(**kw)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    x = synth_apt(rate, 12, seed=33)
    want, st = oracle.decode(x, rate, True, settings=os_, want_steps=True)
    assert st["resample_filter"].size not in (959, 1915)
    rows, pos, res, fused = decode_on_plan(x, rate, apt.MODE_FAST, settings=s)
    assert fused == 1 and res.status == 0, (fused, st["resample_filter"].size)
    frac, err = check_tolerance(rows, pos, want, st["sync_pos"], f"tuned {kw}")
    assert err <= PX_TIGHT, err


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_mfma_non_finite_sample_takes_the_scalar_path(monkeypatch, bad):
    """A NaN / infinity in the input: its tile is evaluated sample by sample (a matrix product would spread it over whole
    groups of 16 windows) — the outputs of the VALU fast kernel, NaN for NaN."""
    x = synth_apt(48000, 12, seed=41)
    for i in (123457, 300001, 300002, x.size - 5):
        x[i] = bad
    rows, pos, res, _ = decode_on_plan(x, 48000, apt.MODE_FAST)
    _valu(monkeypatch)
    rows_v, pos_v, res_v, _ = decode_on_plan(x, 48000, apt.MODE_FAST)
    assert res.status == res_v.status and rows.shape == rows_v.shape
    assert pos.tolist() == pos_v.tolist()
    assert np.array_equal(np.isnan(rows), np.isnan(rows_v))
    ok = ~np.isnan(rows_v)
    fin = np.isfinite(rows_v)
    assert np.array_equal(np.isfinite(rows), fin)
    scale = np.max(np.abs(rows_v[fin]))
    assert np.max(np.abs(rows[fin] - rows_v[fin])) <= PX_TIGHT * scale
    assert np.array_equal(rows[ok & ~fin], rows_v[ok & ~fin])  # the infinities, sign for sign


def test_mfma_full_size_config2(oracle):
    x = synth_apt(48000, 600, seed=2)
    want, st = oracle.decode(x, 48000, True, want_steps=True)
    rows, pos, res, fused = decode_on_plan(x, 48000, apt.MODE_FAST)
    assert fused == 1 and res.n_rows == want.size // 2080
    frac, err = check_tolerance(rows, pos, want, st["sync_pos"], "config 2")
    print(f"config 2 on the matrix cores: positions identical {frac:.5f}, max px err {err:.3e} of full scale")
    assert err <= PX_TIGHT
