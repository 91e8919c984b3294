"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerance: NONE.  Strict mode rounds every product and sum separately in the reference's
order, so every comparison below is bit-for-bit (uint32 views), positions exact.
"""
import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt, synth_noise

pytestmark = pytest.mark.gpu

f32 = np.float32


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


def assert_bitexact(got, want, what=""):
    got = np.asarray(got, f32)
    want = np.asarray(want, f32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(_bits(got), _bits(want)):
        bad = np.flatnonzero(_bits(got) != _bits(want))
        raise AssertionError(f"{what}: {bad.size} of {got.size} differ, first at {bad[0]}: "
                             f"{got[bad[0]]!r} vs {want[bad[0]]!r}")


def assert_same_values(got, want, what=""):
    """Bit-exact on every non-NaN element, NaN exactly where the oracle has NaN (the sign and
    payload of a generated NaN are not part of IEEE arithmetic: x86 and gfx950 differ there)."""
    got = np.asarray(got, f32)
    want = np.asarray(want, f32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), (what, "NaN positions differ", int(np.flatnonzero(gn != wn)[0]))
    ok = ~wn
    assert np.array_equal(_bits(got)[ok], _bits(want)[ok]), what


@pytest.fixture(scope="module")
def ctx():
    assert apt.device_count() >= 1, "no HIP device: the GPU tests must run on the GPU box"
    return apt.Context(device=0)


# ------------------------------------------------------------------ stages
@pytest.mark.parametrize("rate", [48000, 96000, 11025, 44100, 22050])
def test_resample_with_filter_polyphase(ctx, oracle, rate):
    x = synth_noise(rate, 1.5, 3)
    work = 12480
    cut, dw = oracle.freq_hz(4800., rate), oracle.freq_hz(1000., rate)
    want = oracle.resample_with_filter(x, rate, work, oracle.LOWPASS_DC_REMOVAL, cut, 30., dw)
    got = apt.resample_with_filter(ctx, x, apt.Rate.hz(rate), apt.Rate.hz(work),
                                   apt.LowpassDcRemoval(apt.Freq.pi_rad(cut), 30., apt.Freq.pi_rad(dw)))
    assert_bitexact(got, want, f"resample {rate}->12480")


@pytest.mark.parametrize("in_rate,out_rate", [(24960, 12480), (12480, 4160), (12480, 12480), (37440, 12480)])
def test_resample_with_filter_l_equals_1(ctx, oracle, in_rate, out_rate):
    """l == 1 branch: filter() then decimate() (dsp.rs:106-116)."""
    x = synth_noise(in_rate, 0.7, 4)
    cut, dw = oracle.freq_hz(4800., in_rate), oracle.freq_hz(1000., in_rate)
    for kind, args in ((oracle.NOFILTER, (0., 0., 0.)), (oracle.LOWPASS, (cut, 30., dw))):
        want = oracle.resample_with_filter(x, in_rate, out_rate, kind, *args)
        filt = apt.NoFilter() if kind == oracle.NOFILTER else apt.Lowpass(apt.Freq.pi_rad(cut), 30., apt.Freq.pi_rad(dw))
        got = apt.resample_with_filter(ctx, x, apt.Rate.hz(in_rate), apt.Rate.hz(out_rate), filt)
        assert_bitexact(got, want, f"l==1 {in_rate}->{out_rate} kind {kind}")


def test_resample_tool_path(ctx, oracle):
    """dsp::resample (WAV->WAV tool) incl. up-sampling; rate pairs of test/test.sh:48-52."""
    x = synth_noise(11025, 0.8, 9)
    for out_rate in (48000, 6000, 3675, 80000):
        want = oracle.resample(x, 11025, out_rate, 40., 0.1)
        got = apt.resample(ctx, x, apt.Rate.hz(11025), apt.Rate.hz(out_rate), 40., apt.Freq.pi_rad(0.1))
        assert_bitexact(got, want, f"resample 11025->{out_rate}")


def test_resample_edge_cases(ctx, oracle):
    rng = np.random.default_rng(1)
    # taps longer than the signal (dsp.rs:454-468), tiny signals, 1-sample signal
    for n in (1, 2, 17, 100, 1000):
        x = rng.standard_normal(n).astype(f32) * 1000
        want = oracle.resample_with_filter(x, 48000, 12480, oracle.LOWPASS_DC_REMOVAL,
                                           oracle.freq_hz(4800., 48000), 30., oracle.freq_hz(1000., 48000))
        got = apt.resample_with_filter(ctx, x, apt.Rate.hz(48000), apt.Rate.hz(12480),
                                       apt.LowpassDcRemoval(apt.Freq.hz(4800., apt.Rate.hz(48000)), 30.,
                                                            apt.Freq.hz(1000., apt.Rate.hz(48000))))
        assert_bitexact(got, want, f"short n={n}")


def test_rate_overflow(ctx):
    # dsp.rs:420-434
    with pytest.raises(apt.RateOverflowError):
        apt.resample_with_filter(ctx, np.zeros(1000, f32), apt.Rate.hz(99371), apt.Rate.hz(93911), apt.NoFilter())
    with pytest.raises(apt.InternalError) as e:
        apt.resample_with_filter(ctx, np.zeros(10, f32), apt.Rate.hz(48000), apt.Rate.hz(0), apt.NoFilter())
    assert str(e.value) == "Can't resample to 0Hz"


def test_demodulate(ctx, oracle):
    x = synth_apt(12480, 3, 2, amplitude=1500., noise_sigma=30.)
    for work in (12480, 16640, 20800):
        pr = oracle.freq_hz(2400., work)
        assert_bitexact(apt.demodulate(ctx, x, apt.Freq.pi_rad(pr)), oracle.demodulate(x, pr), f"demod {work}")
    # values that make the radicand slightly negative / zero / denormal
    y = np.array([0, 0, 1e-30, -1e-30, 1, 1, -1, 3e38, 3e38, 1e-45, 0], f32)
    pr = oracle.freq_hz(2400., 12480)
    assert_bitexact(apt.demodulate(ctx, y, apt.Freq.pi_rad(pr)), oracle.demodulate(y, pr), "demod specials")


def test_filter(ctx, oracle):
    x = synth_noise(12480, 2, 6)
    c2 = f32(4160) / f32(12480)
    want = oracle.fir(x, oracle.filter_design(oracle.LOWPASS, c2, 25., c2 / f32(5)))
    got = apt.filter(ctx, x, apt.Lowpass(apt.Freq.pi_rad(c2), 25., apt.Freq.pi_rad(c2) / 5.0))
    assert_bitexact(got, want, "lowpass")
    assert got[0] == 0.0  # the `i > j` guard (dsp.rs:399)
    assert_bitexact(apt.filter(ctx, x[:5], apt.NoFilter()), oracle.fir(x[:5], np.ones(1, f32)), "nofilter")


def _fsm_cases():
    rng = np.random.default_rng(42)
    n = 2080 * 23 + 977
    return {
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


@pytest.mark.parametrize("name", list(_fsm_cases().keys()))
@pytest.mark.parametrize("work_rate", [4160, 8320, 12480])
def test_find_sync_adversarial(ctx, oracle, name, work_rate):
    f = _fsm_cases()[name]
    want_pos, want_corr = oracle.find_sync(f, work_rate, return_correlation=True)
    got_pos, got_corr = apt.find_sync(ctx, f, apt.Rate.hz(work_rate), return_correlation=True)
    assert_bitexact(got_corr, want_corr, "correlation")
    assert got_pos.tolist() == want_pos.tolist()


def _nonfinite_cases():
    """Float WAVs (wav.rs:41-50) can carry NaN / +-Inf samples; they reach the picker as NaN / Inf
    correlations.  `corr > last` is false for a NaN on either side (decode.rs:250): a NaN position is
    never replaced once it is the peak, and never replaces one."""
    rng = np.random.default_rng(11)
    n = 60000
    base = (rng.standard_normal(n) * 50).astype(f32)
    out = {}
    v = base.copy()
    for s0 in (100, 5000, 5003, 17000, 30500, 59990):
        v[s0:s0 + 30] = np.nan
    out["nan_bursts"] = v
    v = base.copy()
    v[rng.integers(0, n, 40)] = np.nan
    out["nan_sparse"] = v
    v = base.copy()
    idx = rng.integers(0, n, 60)
    v[idx[:20]] = np.inf
    v[idx[20:40]] = -np.inf
    v[idx[40:]] = np.nan
    out["inf_mix"] = v
    out["pos_inf_only"] = np.where(rng.random(n) < 0.001, np.inf, base).astype(f32)
    out["neg_zero"] = np.full(n, -0.0, f32)
    out["mixed_zero"] = np.where(rng.random(n) < 0.5, -0.0, 0.0).astype(f32)
    out["all_nan"] = np.full(n, np.nan, f32)
    v = base.copy()
    v[:200] = np.nan       # the picker's first window, position 0 included
    v[-400:] = np.nan      # the last positions
    out["nan_edges"] = v
    return out


@pytest.mark.parametrize("name", list(_nonfinite_cases().keys()))
@pytest.mark.parametrize("work_rate", [4160, 12480])
def test_find_sync_nonfinite(ctx, oracle, name, work_rate, monkeypatch):
    f = _nonfinite_cases()[name]
    want_pos, want_corr = oracle.find_sync(f, work_rate, return_correlation=True)
    got_pos, got_corr = apt.find_sync(ctx, f, apt.Rate.hz(work_rate), return_correlation=True)
    assert_same_values(got_corr, want_corr, "correlation")
    assert got_pos.tolist() == want_pos.tolist()
    # ... and through the reference-shaped picker and the sequential fallback
    gen = apt.Context(device=0, mode=apt.MODE_GENERIC)
    assert apt.find_sync(gen, f, apt.Rate.hz(work_rate)).tolist() == want_pos.tolist()
    monkeypatch.setenv("APTGPU_FORCE_WALK", "1")
    assert apt.find_sync(apt.Context(device=0), f, apt.Rate.hz(work_rate)).tolist() == want_pos.tolist()


@pytest.mark.parametrize("rate,fused_any", [(48000, False), (48000, True), (96000, False), (11025, True)])
@pytest.mark.parametrize("kind", ["nan", "inf"])
def test_decode_nonfinite_samples(oracle, monkeypatch, rate, fused_any, kind):
    """A float Signal with NaN / Inf bursts through the whole decode(): the fused front ends leave
    NaNs out of the group maxima and flag the group, k_sync_nodes re-evaluates the correlation
    of the flagged groups, and the rows come out as the reference's — NaNs where it has NaNs."""
    if fused_any:
        monkeypatch.setenv("APTGPU_FUSED_ANY", "1")
    x = synth_apt(rate, 16, seed=61)
    rng = np.random.default_rng(5)
    bad = np.nan if kind == "nan" else np.inf
    for s0 in rng.integers(rate, x.size - rate, 9):
        x[s0:s0 + int(rng.integers(1, 400))] = bad if rng.random() < 0.7 else -bad
    x[3] = bad  # inside the very first tile / first group
    want, st = oracle.decode(x, rate, True, want_steps=True)
    got, stats = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(rate), True, return_stats=True)
    assert stats.n_sync == st["sync_pos"].size
    assert_same_values(got, want, f"decode with {kind} bursts at {rate} Hz")
    assert np.isnan(want).any()


@pytest.mark.parametrize("name", ["zeros", "noise", "ramp_up", "plateaus", "rising_sine"])
def test_find_sync_alternate_pickers(oracle, name, monkeypatch):
    """The same answers from the reference-shaped picker (MODE_GENERIC: full sliding-window
    terminals + sequential orbit) and from the doubling picker's fallback walk."""
    f = _fsm_cases()[name]
    want = oracle.find_sync(f, 4160).tolist()
    gen = apt.Context(device=0, mode=apt.MODE_GENERIC)
    assert apt.find_sync(gen, f, apt.Rate.hz(4160)).tolist() == want
    monkeypatch.setenv("APTGPU_FORCE_WALK", "1")
    assert apt.find_sync(apt.Context(device=0), f, apt.Rate.hz(4160)).tolist() == want


def test_find_sync_short_and_edges(ctx, oracle):
    rng = np.random.default_rng(3)
    for n in (38, 39, 100, 1664, 1665, 2080, 2081, 4160, 4161, 5000):
        f = rng.standard_normal(n).astype(f32)
        want = oracle.find_sync(f, 4160)
        got = apt.find_sync(ctx, f, apt.Rate.hz(4160))
        assert got.tolist() == want.tolist(), n
    with pytest.raises(apt.InternalError):
        apt.find_sync(ctx, np.zeros(5000, f32), apt.Rate.hz(11025))


# ------------------------------------------------------------------ decode()
DECODE_CASES = [
    # (rate, seconds, seed, profile, kwargs)
    (48000, 14, 2, "standard", {}),
    (48000, 14, 12, "standard", dict(ppm=40.0)),
    (96000, 12, 3, "standard", {}),
    (11025, 20, 1, "standard", {}),
    (44100, 12, 4, "standard", {}),
    (48000, 12, 5, "fast", {}),
    (48000, 12, 6, "slow", {}),
    (24960, 12, 7, "standard", {}),   # l == 1 first stage
    (48000, 12, 8, "standard", dict(noise_sigma=6000.0)),  # heavy noise
]


@pytest.mark.parametrize("rate,seconds,seed,profile,kw", DECODE_CASES)
@pytest.mark.parametrize("sync", [True, False])
def test_decode_bitexact(ctx, oracle, rate, seconds, seed, profile, kw, sync):
    x = synth_apt(rate, seconds, seed, **kw)
    s = apt.Settings.profile(profile)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    want, st = oracle.decode(x, rate, sync, settings=os_, want_steps=True)
    got, stats = apt.decode(ctx, s, x, apt.Rate.hz(rate), sync, return_stats=True)
    assert stats.n_resample_taps == st["resample_filter"].size
    assert stats.n_lowpass_taps == st["filter_filter"].size
    assert stats.work_len == st["resampled"].size
    if sync:
        assert stats.n_sync == st["sync_pos"].size
    assert_bitexact(got, want, f"decode {rate} {profile} sync={sync}")
    assert got.size % 2080 == 0


@pytest.mark.parametrize("mode", ["generic", "walk"])
def test_decode_alternate_paths(oracle, mode, monkeypatch):
    """decode() through the unfused generic kernels, and with the picker's fallback walk."""
    x = synth_apt(48000, 14, 2)
    want = oracle.decode(x, 48000, True)
    if mode == "walk":
        monkeypatch.setenv("APTGPU_FORCE_WALK", "1")
        c = apt.Context(device=0)
    else:
        c = apt.Context(device=0, mode=apt.MODE_GENERIC)
    got, st = apt.decode(c, apt.Settings(), x, apt.Rate.hz(48000), True, return_stats=True)
    assert_bitexact(got, want, f"decode via {mode}")
    assert st.fused == (0 if mode == "generic" else 1)


def test_fused_specialisations_are_used(ctx):
    """48 kHz and 96 kHz (standard profile) run the compile-time specialised front end (1); 11 025 Hz
    and the other rates whose tap table and input tile fit run the table-driven stage 1 in front of
    the specialised work-rate stages (3); 44 100 Hz the phase-resident stage 1 in front of the same stages
    (4); the rest the run-time fused kernel (2); l == 1 the unfused kernels (0)."""
    apt.cache_clear()  # (a session another test built under an APTGPU_* switch would answer for its own kernel path)
    for rate, profile, want in ((48000, "standard", 1), (96000, "standard", 1), (44100, "standard", 4),
                                (11025, "standard", 4), (8000, "standard", 4), (22050, "standard", 4),
                                (48000, "fast", 4), (48000, "slow", 1), (96000, "fast", 1), (16000, "fast", 4),
                                (11025, "fast", 4), (96000, "slow", 1), (24960, "standard", 0)):
        _, st = apt.decode(ctx, apt.Settings.profile(profile), synth_apt(rate, 11, 3), apt.Rate.hz(rate),
                           True, return_stats=True)
        assert st.fused == want, (rate, profile)


TABLE_CASES = [  # (rate, seconds): rates served by k_fused's table-driven stage 1 (fused == 3)
    (11025, 40), (8000, 40), (12000, 20), (15600, 20), (20800, 15), (16000, 20), (11025, 11),
]


@pytest.mark.parametrize("rate,seconds", TABLE_CASES)
@pytest.mark.parametrize("sync", [True, False])
def test_table_stage1_bitexact(ctx, oracle, monkeypatch, rate, seconds, sync):
    monkeypatch.setenv("APTGPU_PHASE_FIRST", "0")  # (the phase-resident stage 1 is the default where both exist)
    apt.cache_clear()
    x = synth_apt(rate, seconds, seed=rate % 97 + seconds)
    want = oracle.decode(x, rate, sync)
    got, st = apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(rate), sync, return_stats=True)
    assert st.fused == 3, (rate, st.l, st.m, st.n_resample_taps)
    assert_bitexact(got, want, f"table stage 1 {rate} sync={sync}")


def test_table_stage1_long_ragged_batched_and_pcm16(oracle, monkeypatch):
    """Many tiles, lengths that end mid-tile / mid-group, several recordings per call, PCM16 payloads at
    odd byte offsets, and the fast mode's tolerance — all at 11 025 Hz."""
    monkeypatch.setenv("APTGPU_PHASE_FIRST", "0")
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    recs = [synth_apt(11025, 900, seed=5), synth_apt(11025, 20, seed=6)[:11025 * 20 - 3],
            synth_apt(11025, 33, seed=7)[:11025 * 33 - 1], synth_apt(11025, 64, seed=8)]
    wants = [oracle.decode(r, 11025, True, want_steps=True) for r in recs]
    nmax = max(r.size for r in recs)
    for mode in (apt.MODE_STRICT, apt.MODE_FAST):
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(11025), True, max_samples=nmax, max_batch=len(recs), mode=mode)
        assert plan.info.fused == 3
        d_in = [torch.from_numpy(r).to(dev) for r in recs]
        cap = int(plan.info.max_rows)
        d_out = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
        torch.cuda.synchronize()
        for _ in range(2):
            plan.decode_device([t.data_ptr() for t in d_in], [r.size for r in recs], [t.data_ptr() for t in d_out],
                               [cap] * len(recs))
        res = plan.results(len(recs))
        for i, (want, st) in enumerate(wants):
            got = d_out[i][:res[i].n_out].cpu().numpy()
            if mode == apt.MODE_STRICT:
                assert_bitexact(got, want, f"table batch {i}")
                assert plan.sync_positions(i).tolist() == st["sync_pos"].tolist()
            else:
                from test_gpu_fast import check_tolerance
                check_tolerance(got, plan.sync_positions(i), want, st["sync_pos"], f"table fast {i}")
        if mode == apt.MODE_STRICT:
            # the same recordings as PCM16 payloads, the second one at an odd 2-byte offset
            pcm = [torch.from_numpy(np.concatenate([np.zeros(1 if i == 1 else 0, np.int16), r.astype(np.int16)])).to(dev)
                   for i, r in enumerate(recs)]
            ptrs = [t.data_ptr() + (2 if i == 1 else 0) for i, t in enumerate(pcm)]
            specs = [apt.WavSpec(1, 16, 2, 0, 11025, 1, 0, 2 * r.size, r.size, r.size) for r in recs]
            plan.decode_device_wav(ptrs, specs, [t.data_ptr() for t in d_out], [cap] * len(recs))
            res = plan.results(len(recs))
            for i, (want, _) in enumerate(wants):
                assert_bitexact(d_out[i][:res[i].n_out].cpu().numpy(), want, f"table pcm16 {i}")
        plan.close()


PHASE_CASES = [  # (rate, seconds): rates k_fused's phase-resident stage 1 (fused == 4) can serve
    (44100, 40), (44100, 11), (32000, 20), (20800, 15), (16000, 20), (24000, 15), (40000, 12), (15600, 20),
    (22050, 40), (22050, 11),  # l = 416: 512-thread workgroups
    (11025, 40), (11025, 11),  # l = 832: 1024-thread workgroups
]


@pytest.mark.parametrize("rate,seconds", PHASE_CASES)
@pytest.mark.parametrize("sync", [True, False])
def test_phase_stage1_bitexact(oracle, monkeypatch, rate, seconds, sync):
    monkeypatch.setenv("APTGPU_PHASE_FIRST", "1")
    apt.cache_clear()
    x = synth_apt(rate, seconds, seed=rate % 89 + seconds)
    want = oracle.decode(x, rate, sync)
    got, st = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(rate), sync, return_stats=True)
    assert st.fused == 4, (rate, st.l, st.m, st.n_resample_taps)
    assert_bitexact(got, want, f"phase stage 1 {rate} sync={sync}")


@pytest.mark.parametrize("rate,seconds", [(22050, 40), (22050, 11), (11025, 40), (11025, 11)])
@pytest.mark.parametrize("identity", [False, True])
def test_phase_stage1_wide_workgroups_bitexact(oracle, monkeypatch, rate, seconds, identity):
    """The 512- / 1024-thread forms of the phase-resident stage 1 (one branch per thread; APTGPU_PHASE_WIDE=1 — the
    default for these rates is 256 threads with two / four branches each), with the host's thread assignment lists and
    with slot = thread."""
    monkeypatch.setenv("APTGPU_PHASE_FIRST", "1")
    monkeypatch.setenv("APTGPU_PHASE_WIDE", "1")
    if identity:
        monkeypatch.setenv("APTGPU_PHASE_IDENTITY", "1")
    apt.cache_clear()
    x = synth_apt(rate, seconds, seed=rate % 89 + seconds)
    want = oracle.decode(x, rate, True)
    got, st = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(rate), True, return_stats=True)
    apt.cache_clear()
    assert st.fused == 4, (rate, st.l, st.m, st.n_resample_taps)
    assert_bitexact(got, want, f"phase stage 1 (wide) {rate}")


@pytest.mark.parametrize("rate,seconds,profile", [(44100, 14, "standard"), (11025, 30, "standard"), (22050, 20, "slow")])
@pytest.mark.parametrize("switch", ["APTGPU_PHASE_TT", "APTGPU_PHASE_IDENTITY"])
def test_phase_stage1_table_switches_bitexact(oracle, monkeypatch, rate, seconds, profile, switch):
    """The A/B switches of the phase-resident stage 1's host tables: taps from the phase-major rows only
    (APTGPU_PHASE_TT=0: no thread-order copy) and slot = thread (APTGPU_PHASE_IDENTITY=1: no assignment lists) —
    other memory layouts and lane assignments, the same arithmetic."""
    monkeypatch.setenv(switch, "0" if switch.endswith("TT") else "1")
    apt.cache_clear()
    x = synth_apt(rate, seconds, seed=rate % 83 + seconds)
    s = apt.Settings.profile(profile)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    want = oracle.decode(x, rate, True, settings=os_)
    got, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), True, return_stats=True)
    apt.cache_clear()
    assert st.fused == 4
    assert_bitexact(got, want, f"phase stage 1 with {switch} at {rate} Hz, {profile}")


@pytest.mark.parametrize("rate,seconds,profile", [
    (11025, 30, "standard"), (22050, 20, "standard"), (8000, 30, "standard"),   # l = 832 (nq 4), 416 (2), 39 (1: untouched)
    (13000, 25, "standard"), (9600, 25, "standard"),                               # l = 24 / 13 ... whatever the plan picks
    (44100, 14, "fast"), (22050, 20, "fast"), (11025, 30, "fast"),                 # l = 832 (4), 1664 (8), 3328 (16)
    (11025, 30, "slow"), (22050, 20, "slow"),                                      # streamed taps, nq 4 / 2
])
@pytest.mark.parametrize("balanced", ["0", "1"])
def test_phase_stage1_slot_stride_bitexact(oracle, monkeypatch, rate, seconds, profile, balanced):
    """TableGeom::sq both ways for every number of branches per thread: l / nq slots of nq branches each
    (APTGPU_PHASE_BALANCED=0) and one slot per thread, 256 apart, the slots >= l missing (=1: the last branch exists for
    some threads only — skipped by a wave none of whose lanes has it, computed on window 0 / branch 0 and dropped by a lane
    that sits beside one that has).  The same outputs from other threads: bit-identical to the oracle either way."""
    monkeypatch.setenv("APTGPU_PHASE_BALANCED", balanced)
    apt.cache_clear()
    x = synth_apt(rate, seconds, seed=rate % 79 + seconds)
    x[x.size // 3] = np.nan  # (a non-finite sample must stay where the reference has it: no lane's garbage is ever stored)
    s = apt.Settings.profile(profile)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    want = oracle.decode(x, rate, True, settings=os_)
    got, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), True, return_stats=True)
    apt.cache_clear()
    assert st.fused in (1, 2, 3, 4), (rate, profile, st.fused)
    assert_bitexact(got, want, f"slot stride (balanced={balanced}) at {rate} Hz, {profile}")


@pytest.mark.parametrize("rate,seconds,profile", [
    (48035, 14, "fast"), (22035, 20, "fast"),      # l = 256 (m = 739 / 339): sixteen tile phases — no per-phase tables (exact == 0)
    (47970, 14, "fast"),                            # l = 128
    (44100, 14, "standard"), (8000, 30, "standard"), (12000, 25, "standard"), (32000, 15, "standard"),  # l = 208, 39, 26, 39
])
@pytest.mark.parametrize("sync", [True, False])
def test_phase_one_branch_tile_in_two_halves_bitexact(oracle, rate, seconds, profile, sync):
    """The PHASE stage 1 with one branch per thread takes its paired input tile through LDS in two halves (windows 0-7, then
    8-15: APT_PHASE_HALVES), the standard profile's long branches in segments fetched once per half — with the per-phase
    tables and without them (l = 256 at the fast profile: sixteen tile phases, the kernel divides), short and long
    branches, a NaN and an infinity in the input, the recording's first and last tiles."""
    x = synth_apt(rate, seconds, seed=rate % 71 + seconds)
    x[x.size // 2] = np.nan
    x[7] = np.inf
    s = apt.Settings.profile(profile)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    try:
        want = oracle.decode(x, rate, sync, settings=os_)
    except oracle.OracleError as e:
        with pytest.raises(apt.AptError) as ge:
            apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), sync)
        assert str(ge.value) == str(e)
        return
    got, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), sync, return_stats=True)
    assert st.fused == 4, (rate, profile, st.l, st.m, st.fused)
    assert_same_values(got, want, f"one-branch PHASE, two halves: {rate} Hz, {profile}, sync={sync}")


def test_phase_stage1_long_ragged_batched_and_pcm16(oracle):
    """Many tiles, lengths that end mid-tile / mid-group, several recordings per call, PCM16 payloads at
    odd 2-byte offsets, non-finite samples, and the fast mode's tolerance — all at 44 100 Hz."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    bad = synth_apt(44100, 25, seed=9)
    bad[70000:70100] = np.nan
    bad[500000] = np.inf
    bad[3] = -np.inf
    recs = [synth_apt(44100, 300, seed=5), synth_apt(44100, 20, seed=6)[:44100 * 20 - 3],
            synth_apt(44100, 33, seed=7)[:44100 * 33 - 1], synth_apt(44100, 64, seed=8), bad]
    wants = [oracle.decode(r, 44100, True, want_steps=True) for r in recs]
    nmax = max(r.size for r in recs)
    for mode in (apt.MODE_STRICT, apt.MODE_FAST):
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(44100), True, max_samples=nmax, max_batch=len(recs), mode=mode)
        assert plan.info.fused == 4
        d_in = [torch.from_numpy(r).to(dev) for r in recs]
        cap = int(plan.info.max_rows)
        d_out = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
        torch.cuda.synchronize()
        for _ in range(2):
            plan.decode_device([t.data_ptr() for t in d_in], [r.size for r in recs], [t.data_ptr() for t in d_out],
                               [cap] * len(recs))
        res = plan.results(len(recs))
        for i, (want, st) in enumerate(wants):
            got = d_out[i][:res[i].n_out].cpu().numpy()
            if mode == apt.MODE_STRICT:
                assert_same_values(got, want, f"phase batch {i}")
                assert plan.sync_positions(i).tolist() == st["sync_pos"].tolist()
            elif i < 4:
                from test_gpu_fast import check_tolerance
                check_tolerance(got, plan.sync_positions(i), want, st["sync_pos"], f"phase fast {i}")
        if mode == apt.MODE_STRICT:
            # the same recordings as PCM16 payloads, the second one at an odd 2-byte offset
            pcm = [torch.from_numpy(np.concatenate([np.zeros(1 if i == 1 else 0, np.int16), r.astype(np.int16)])).to(dev)
                   for i, r in enumerate(recs[:4])]
            ptrs = [t.data_ptr() + (2 if i == 1 else 0) for i, t in enumerate(pcm)]
            specs = [apt.WavSpec(1, 16, 2, 0, 44100, 1, 0, 2 * r.size, r.size, r.size) for r in recs[:4]]
            plan.decode_device_wav(ptrs, specs, [t.data_ptr() for t in d_out[:4]], [cap] * 4)
            res = plan.results(4)
            for i, (want, _) in enumerate(wants[:4]):
                assert_bitexact(d_out[i][:res[i].n_out].cpu().numpy(), want, f"phase pcm16 {i}")
        plan.close()


PROFILE_CASES = [  # (rate, seconds, profile, fused): the fast and slow profiles on the specialised kernels (round 4)
    (48000, 14, "slow", 1),    # SPLIT stage 1 for 13 / 30 with 2783 taps; 61-tap low-pass, pixel width 5
    (48000, 14, "fast", 4),    # phase-resident stage 1 (l = 26) + the fast profile's work-rate stages (43 taps, pw 4)
    (96000, 12, "fast", 1),    # l = 13, m = 75 (odd): SPLIT stage 1 with 4-byte window reads (round 5; until then the run-time kernel)
    (16000, 30, "fast", 4), (32000, 16, "fast", 4), (8000, 50, "fast", 4), (12000, 40, "fast", 4), (24000, 20, "fast", 4),
    # round 5: four / eight branches per thread in front of the fast profile's stages (l = 832 / 1664)
    (44100, 14, "fast", 4), (22050, 20, "fast", 4),
    (11025, 30, "fast", 4), (11025, 61, "fast", 4),   # l = 3328: sixteen branches per thread, one output each per tile
    # round 5: the slow profile at the sound-card rates — 197 taps per branch streamed from the table (l = 208 / 416 / 832)
    (44100, 14, "slow", 4), (22050, 20, "slow", 4), (11025, 30, "slow", 4), (44100, 41, "slow", 4),
    (96000, 12, "slow", 1),    # round 5: SPLIT stage 1 for 13 / 60 with 5565 taps
]


@pytest.mark.parametrize("rate,seconds,profile,fused", PROFILE_CASES)
@pytest.mark.parametrize("sync", [True, False])
def test_profile_kernels_bitexact(oracle, rate, seconds, profile, fused, sync):
    """default_settings.toml:120-140: the reference's other two stock profiles on the same kernels as the standard one —
    the work-rate stages are templates over the low-pass length and the pixel width, stage 4 in its general form."""
    x = synth_apt(rate, seconds, seed=rate % 97 + seconds)
    s = apt.Settings.profile(profile)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    want = oracle.decode(x, rate, sync, settings=os_)
    got, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), sync, return_stats=True)
    assert st.fused == fused, (st.fused, st.l, st.m, st.n_resample_taps, st.n_lowpass_taps)
    assert_bitexact(got, want, f"profile kernel {rate} {profile} sync={sync}")


def test_profile_kernels_ragged_batched_pcm16_and_nonfinite(oracle):
    """The same kernels over a call of ragged recordings, as PCM16 payloads, with NaN / Inf samples, and with
    recordings that end around the tile boundaries."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    for profile, rate in (("slow", 48000), ("fast", 48000), ("slow", 44100), ("fast", 44100), ("slow", 11025), ("fast", 22050), ("fast", 11025),
                          ("fast", 96000), ("slow", 96000)):
        s = apt.Settings.profile(profile)
        os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                           "resample_cutout", "demodulation_atten")}
        base = synth_apt(rate, 16, seed=5 + len(profile))
        ns = [base.size, base.size - 1, base.size - 997, int(rate * 11.3), int(rate * 12.7) + 3, int(rate * 14.01)]
        recs = [np.ascontiguousarray(base[:n]) for n in ns]
        bad = recs[2].copy()
        bad[rate * 3: rate * 3 + 40] = np.nan
        bad[rate * 7] = np.inf
        recs.append(bad)
        wants = [oracle.decode(r, rate, True, settings=os_) for r in recs]
        plan = apt.Plan(s, apt.Rate.hz(rate), True, max_samples=base.size, max_batch=len(recs))
        cap = int(plan.info.max_rows)
        d_in = [torch.from_numpy(r).to(dev) for r in recs]
        d_out = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
        torch.cuda.synchronize()
        for _ in range(2):
            plan.decode_device([t.data_ptr() for t in d_in], [r.size for r in recs], [t.data_ptr() for t in d_out], [cap] * len(recs))
        res = plan.results(len(recs))
        for i, want in enumerate(wants):
            assert res[i].status == 0
            assert_bitexact(d_out[i][:res[i].n_out].cpu().numpy(), want, f"{profile} batched {i}")
        # PCM16 payloads of the first four
        pcm = [r.astype(np.int16) for r in recs[:4]]
        wants16 = [oracle.decode(p.astype(np.float32), rate, True, settings=os_) for p in pcm]
        d_pcm = [torch.from_numpy(p).to(dev) for p in pcm]
        specs = [apt.WavSpec(1, 16, 2, 0, rate, 1, 0, 2 * p.size, p.size, p.size) for p in pcm]
        torch.cuda.synchronize()
        plan.decode_device_wav([t.data_ptr() for t in d_pcm], specs, [t.data_ptr() for t in d_out[:4]], [cap] * 4)
        res = plan.results(4)
        for i, want in enumerate(wants16):
            assert_bitexact(d_out[i][:res[i].n_out].cpu().numpy(), want, f"{profile} pcm16 {i}")
        plan.close()


ANY_CASES = [  # (rate, seconds, profile): the run-time fused kernel on every kind of geometry
    (11025, 40, "standard"), (44100, 14, "standard"), (22050, 14, "standard"), (8000, 40, "standard"),
    (32000, 14, "standard"), (12000, 20, "standard"), (48000, 14, "fast"), (48000, 14, "slow"),
    (11025, 30, "fast"), (44100, 14, "slow"), (96000, 12, "fast"), (20800, 15, "standard"), (15600, 20, "standard"),
    (48000, 14, "standard"), (96000, 12, "standard"),   # forced over the specialisations (env below)
]


@pytest.mark.parametrize("rate,seconds,profile", ANY_CASES)
@pytest.mark.parametrize("sync", [True, False])
def test_runtime_fused_kernel_bitexact(oracle, monkeypatch, rate, seconds, profile, sync):
    monkeypatch.setenv("APTGPU_FUSED_ANY", "1")
    apt.cache_clear()  # (the switch is read at plan creation: a cached session of an earlier test would ignore it)
    x = synth_apt(rate, seconds, seed=rate % 89 + seconds)
    s = apt.Settings.profile(profile)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    want = oracle.decode(x, rate, sync, settings=os_)
    got, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), sync, return_stats=True)
    # (44 100 Hz at the slow profile: its polyphase table — 29 k taps — does not fit LDS and is read from HBM / L2)
    assert st.fused == 2, (st.l, st.m, st.n_resample_taps)
    assert_bitexact(got, want, f"run-time fused {rate} {profile} sync={sync}")


def test_runtime_fused_kernel_long_and_ragged(oracle, monkeypatch):
    """Many tiles, lengths that end mid-tile / mid-group, and a 15-minute 11 025 Hz pass."""
    monkeypatch.setenv("APTGPU_FUSED_ANY", "1")
    for rate, n in ((11025, 11025 * 900), (11025, 11025 * 20 + 1), (44100, 44100 * 12 + 735),
                    (48000, 48000 * 14 - 1), (48000, 48000 * 13 + 50)):
        x = synth_apt(rate, n // rate + 1, seed=n % 101)[:n]
        want = oracle.decode(x, rate, True)
        got, st = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(rate), True,
                             return_stats=True)
        assert st.fused == 2
        assert_bitexact(got, want, f"run-time fused {rate} n={n}")


def test_decode_long_recordings(ctx, oracle):
    """Long recordings (15 min @ 48 kHz, 5 min @ 96 kHz): the picker's tables live in HBM and have
    no size limit."""
    for rate, seconds, seed in ((48000, 900, 41), (96000, 300, 42)):
        x = synth_apt(rate, seconds, seed)
        want = oracle.decode(x, rate, True)
        got, st = apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(rate), True, return_stats=True)
        assert st.fused == 1
        assert_bitexact(got, want, f"long {rate} Hz {seconds} s")


@pytest.mark.parametrize("rate,seed", [(48000, 2), (96000, 3), (11025, 1)])
def test_decode_fp16_taps_mode(oracle, rate, seed):
    """BASELINE config 5: fp16 taps + fp16 samples, f32 accumulate.  Tolerance (stated):
    identical sync positions and row count, |px - ref| <= 2e-3 * max|ref| (measured ~5e-4)."""
    x = synth_apt(rate, 14, seed)
    want, st = oracle.decode(x, rate, True, want_steps=True)
    c = apt.Context(device=0, mode=apt.MODE_FP16_TAPS)
    got, stats = apt.decode(c, apt.Settings(), x, apt.Rate.hz(rate), True, return_stats=True)
    # 48 kHz standard: fp16 stage 1 inside the specialised fused kernel; other rates: generic kernel
    assert stats.fused == (1 if rate == 48000 else 0) and stats.n_sync == st["sync_pos"].size
    assert got.shape == want.shape
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err <= 2e-3, err
    assert err > 0  # it really is the reduced-precision path


def test_decode_noise_fixture_like(ctx, oracle):
    """Stand-in for test/noise_48000hz.wav (really 11025 Hz, 30 s of noise)."""
    x = synth_noise(11025, 30.0, 77)
    assert_bitexact(apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(11025), True),
                    oracle.decode(x, 11025, True), "noise decode")


def test_decode_zeros_and_dc(ctx, oracle):
    for x in (np.zeros(48000 * 8, f32), np.full(48000 * 8, 1234.0, f32)):
        assert_bitexact(apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(48000), True),
                        oracle.decode(x, 48000, True), "zeros/dc")


def test_decode_errors(ctx):
    with pytest.raises(apt.InternalError) as e:
        apt.decode(ctx, apt.Settings(), np.zeros(48000, f32), apt.Rate.hz(48000), True)
    assert str(e.value) == "Got less than 10 rows of samples, audio file is too short"
    with pytest.raises(apt.InternalError) as e:  # work_rate not a multiple of 4160 -> find_sync error
        apt.decode(ctx, apt.Settings(work_rate=11025), synth_apt(48000, 12, 1), apt.Rate.hz(48000), True)
    assert str(e.value) == "work_rate is not multiple of FINAL_RATE"
    with pytest.raises(apt.RateOverflowError):
        apt.decode(ctx, apt.Settings(work_rate=93911), np.zeros(99371 * 12, f32), apt.Rate.hz(99371), True)


def test_decode_nosync_odd_work_rate(ctx, oracle):
    """sync == false with a work_rate that is not a multiple of 4160: the final stage goes
    through fast_resampling with NoFilter taps (l > 1)."""
    x = synth_apt(48000, 12, 21)
    s = apt.Settings(work_rate=11025)
    os_ = dict(work_rate=11025, resample_atten=30., resample_delta_freq=1000., resample_cutout=4800.,
               demodulation_atten=25.)
    assert_bitexact(apt.decode(ctx, s, x, apt.Rate.hz(48000), False),
                    oracle.decode(x, 48000, False, settings=os_), "nosync odd work rate")


def test_status_callbacks_in_reference_order(ctx):
    seen = []
    c = apt.Context(ui_callback=lambda p, t: seen.append((round(p, 2), t)), device=0)
    apt.decode(c, apt.Settings(), synth_apt(48000, 12, 1), apt.Rate.hz(48000), True)
    assert seen == [(0.1, "Resampling to 12480"), (0.4, "Demodulating"), (0.42, "Filtering"),
                    (0.5, "Syncing"), (0.9, "Resampling to 4160")]
    seen.clear()
    apt.decode(c, apt.Settings(), synth_apt(48000, 12, 1), apt.Rate.hz(48000), False)
    assert seen[3] == (0.5, "Skipping Syncing")


def test_steps_export(ctx, oracle):
    """Context::step: every intermediate the reference would export, bit-identical."""
    x = synth_apt(48000, 12, 31)
    got = []
    c = apt.Context(step_callback=lambda i, v, d, r: got.append((i, v, d, r)), device=0)
    s = apt.Settings(export_wav=True)
    rows = apt.decode(c, s, x, apt.Rate.hz(48000), True)
    want_rows, st = oracle.decode(x, 48000, True, want_steps=True)
    assert_bitexact(rows, want_rows)
    ids = [g[0] for g in got]
    assert ids == ["input", "resample_filter", "resample_filtered", "resample_decimated",
                   "demodulation_result", "filter_filter", "filter_result", "sync_correlation",
                   "sync_result", "resample_filter", "filter_filter", "filter_result",
                   "resample_filtered", "resample_decimated"]
    by = {}
    for i, v, d, r in got:
        by.setdefault(i, []).append((v, d, r))
    assert_bitexact(by["input"][0][1], x)
    assert_bitexact(by["resample_filter"][0][1], st["resample_filter"])
    assert by["resample_filtered"][0][1].size == 0 and by["resample_filtered"][0][2] == 48000 * 13
    assert_bitexact(by["resample_decimated"][0][1], st["resampled"])
    assert_bitexact(by["demodulation_result"][0][1], st["demodulated"])
    assert_bitexact(by["filter_filter"][0][1], st["filter_filter"])
    assert_bitexact(by["filter_result"][0][1], st["filtered"])
    assert_bitexact(by["sync_correlation"][0][1], st["correlation"])
    assert_bitexact(by["sync_result"][0][1], st["aligned"])
    assert by["resample_filter"][1][1].tolist() == [1.0]
    assert_bitexact(by["resample_decimated"][1][1], want_rows)


def test_steps_export_nosync(ctx, oracle):
    """The no-sync branch exports the same step sequence as the reference: the dummy
    sync_correlation (decode.rs:139), the cropped signal, then the steps of the final
    resample_with_filter(NoFilter) (dsp.rs:96-122)."""
    x = synth_apt(48000, 12, 33)
    got = []
    c = apt.Context(step_callback=lambda i, v, d, r: got.append((i, v, d, r)), device=0)
    rows = apt.decode(c, apt.Settings(export_wav=True), x, apt.Rate.hz(48000), False)
    want_rows, st = oracle.decode(x, 48000, False, want_steps=True)
    assert_bitexact(rows, want_rows)
    ids = [g[0] for g in got]
    assert ids == ["input", "resample_filter", "resample_filtered", "resample_decimated",
                   "demodulation_result", "filter_filter", "filter_result", "sync_correlation",
                   "sync_result", "resample_filter", "filter_filter", "filter_result",
                   "resample_filtered", "resample_decimated"]
    by = {}
    for i, v, d, r in got:
        by.setdefault(i, []).append((v, d, r))
    assert by["sync_correlation"][0][1].size == 0 and by["sync_correlation"][0][2] == 12480
    aligned = st["filtered"][:st["filtered"].size // 6240 * 6240]
    assert_bitexact(by["sync_result"][0][1], aligned)
    assert by["resample_filter"][1][1].tolist() == [1.0] and by["filter_filter"][1][1].tolist() == [1.0]
    # filter([1.]): 0.0 + x*1.0 and sample 0 dropped by the `i > j` guard (dsp.rs:399)
    fr = aligned.copy()
    fr[0] = 0.0
    assert_bitexact(by["filter_result"][1][1], fr)
    assert_bitexact(by["resample_filtered"][1][1], fr)
    assert by["resample_filtered"][1][2] == 12480
    assert_bitexact(by["resample_decimated"][1][1], want_rows)
    # a work rate that is not a multiple of 4160: the final stage is fast_resampling (l > 1)
    got.clear()
    apt.decode(c, apt.Settings(export_wav=True, work_rate=11025), x, apt.Rate.hz(48000), False)
    ids = [g[0] for g in got]
    assert ids[-4:] == ["sync_result", "resample_filter", "resample_filtered", "resample_decimated"]
    assert got[-2][2].size == 0 and got[-2][3] == 11025 * 832  # l2 = 4160 / gcd(11025, 4160) = 832


def test_rows_cap_clamps_the_result_record(oracle):
    """A rows buffer smaller than the recording: the record reports what was written (reason 4)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    x = synth_apt(48000, 14, 2)
    want = oracle.decode(x, 48000, True)
    plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=x.size)
    d_in = torch.from_numpy(x).to(dev)
    cap = 7
    d_out = torch.full(((cap + 1) * 2080,), -7.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
    res = plan.results(1)[0]
    assert res.status == 0 and res.reason == 4 and res.n_rows == cap and res.n_out == cap * 2080
    out = d_out.cpu().numpy()
    assert_bitexact(out[:cap * 2080], want[:cap * 2080], "clamped rows")
    assert np.all(out[cap * 2080:] == -7.0)  # nothing written past the caller's capacity
    plan.close()
    with pytest.raises(apt.AptError):
        apt.Plan(apt.Settings(work_rate=1), apt.Rate.hz(48000), False, max_samples=48000)  # spr == 0


@pytest.mark.parametrize("lds", [True, False])
def test_picker_paths_direct_and_doubling(oracle, monkeypatch, lds):
    """The closure form of the picker (APTGPU_ORBIT_ALG=0; what runs when a recording's tables do not fit the default
    form's LDS): it reads the orbit off directly on a confluent (continuous APT) recording and extracts it by pointer
    doubling otherwise — with its jump tables in LDS, or (forced here; else only for recordings whose visited nodes do
    not fit) through global memory; all bit-exact."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("APTGPU_ORBIT_ALG", "0")
    if not lds:
        monkeypatch.setenv("APTGPU_ORBIT_LDS", "0")
    dev = torch.device("cuda:0")
    cases = [("apt", synth_apt(48000, 20, 5)), ("noise", synth_noise(48000, 20.0, 5, sigma=4000.0)),
             ("noise-long", synth_noise(48000, 120.0, 6, sigma=3000.0)),
             ("gaps", np.concatenate([synth_apt(48000, 8, 7), synth_noise(48000, 3.0, 8, sigma=500.0),
                                      synth_apt(48000, 9, 9), np.zeros(48000, f32), synth_apt(48000, 7, 10)]))]
    seen = {}
    for name, x in cases:
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=x.size)
        d_in = torch.from_numpy(x).to(dev)
        cap = int(plan.info.max_rows)
        d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
        res = plan.results(1)[0]
        flags = plan.read_internal("picker_flags", np.uint32, 32)
        want, st = oracle.decode(x, 48000, True, want_steps=True)
        assert_bitexact(d_out[:res.n_out].cpu().numpy(), want, name)
        assert plan.sync_positions(0).tolist() == st["sync_pos"].tolist(), name
        seen[name] = (int(flags[1]), int(flags[6]))
        plan.close()
    assert seen["apt"] == (2, 1)       # global kernel, direct orbit
    assert all(v[0] in (1, 2) for v in seen.values())  # (1: the walk, when a chunk's node list overflowed — the zeros)
    doubling = [v[1] for v in seen.values() if v[0] == 2 and v[1] != 1]
    assert doubling, "no case took the doubling path: the test has no teeth"
    assert all(d == (2 if lds else 0) for d in doubling), seen


def test_picker_all_nodes_doubling(oracle, monkeypatch):
    """The default picker: the successors of ALL possible starts and the root's orbit by doubling — no breadth-first
    closure — on confluent and non-confluent recordings; bit-exact, and the path is the one that ran."""
    torch = pytest.importorskip("torch")
    monkeypatch.delenv("APTGPU_ORBIT_ALG", raising=False)
    dev = torch.device("cuda:0")
    cases = [("apt", synth_apt(48000, 20, 5)), ("noise", synth_noise(48000, 20.0, 5, sigma=4000.0)),
             ("noise-long", synth_noise(48000, 120.0, 6, sigma=3000.0)), ("apt-long", synth_apt(48000, 600, 2)),
             ("nan", np.where(np.arange(48000 * 20) % 9001 == 17, np.float32("nan"), synth_apt(48000, 20, 11)).astype(f32)),
             ("gaps", np.concatenate([synth_apt(48000, 8, 7), synth_noise(48000, 3.0, 8, sigma=500.0),
                                      synth_apt(48000, 9, 9), np.zeros(48000, f32), synth_apt(48000, 7, 10)]))]
    seen = {}
    for name, x in cases:
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=x.size)
        d_in = torch.from_numpy(x).to(dev)
        cap = int(plan.info.max_rows)
        d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):  # twice: the plan's scratch is reused
            plan.decode_device([d_in.data_ptr()], [x.size], [d_out.data_ptr()], [cap])
        res = plan.results(1)[0]
        flags = plan.read_internal("picker_flags", np.uint32, 32)
        want, st = oracle.decode(x, 48000, True, want_steps=True)
        assert_bitexact(d_out[:res.n_out].cpu().numpy(), want, name)
        assert plan.sync_positions(0).tolist() == st["sync_pos"].tolist(), name
        seen[name] = (int(flags[1]), int(flags[6]))
        plan.close()
    assert seen["apt"] == (2, 3) and seen["apt-long"] == (2, 3) and seen["noise-long"] == (2, 3), seen
    assert all(v == (2, 3) or v[0] == 1 for v in seen.values()), seen  # (1: the walk, when a chunk's node list overflowed)


# ------------------------------------------------------------------ plans / batch
def test_plan_device_resident_batch(ctx, oracle):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    recs = [synth_apt(48000, 11 + i, 100 + i, ppm=10.0 * i) for i in range(3)]
    nmax = max(r.size for r in recs)
    plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=nmax, max_batch=3,
                    stream=torch.cuda.current_stream().cuda_stream)
    d_in = [torch.from_numpy(r).to(dev) for r in recs]
    cap = int(plan.info.max_rows)
    d_out = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
    for _ in range(2):  # run twice: the plan is reusable
        plan.decode_device([t.data_ptr() for t in d_in], [r.size for r in recs],
                           [t.data_ptr() for t in d_out], [cap] * 3)
    res = plan.results(3)
    for i, r in enumerate(recs):
        want, st = oracle.decode(r, 48000, True, want_steps=True)
        assert res[i].status == 0 and res[i].n_out == want.size
        assert_bitexact(d_out[i][:res[i].n_out].cpu().numpy(), want, f"batch {i}")
        assert plan.sync_positions(i).tolist() == st["sync_pos"].tolist()
    plan.close()


# ------------------------------------------------------------------ exact fast envelope
def _plan_inv_sinphi(profile, rate=48000):
    torch = pytest.importorskip("torch")
    x = torch.zeros(rate * 12, dtype=torch.float32, device="cuda:0")
    plan = apt.Plan(apt.Settings.profile(profile), apt.Rate.hz(rate), True, max_samples=x.numel(), max_batch=1)
    cap = int(plan.info.max_rows)
    rows = torch.empty(cap * 2080, dtype=torch.float32, device="cuda:0")
    plan.decode_device([x.data_ptr()], [x.numel()], [rows.data_ptr()], [cap])
    plan.results(1)
    v = float(plan.read_internal("inv_sinphi", np.float32, 1)[0])
    plan.close()
    return v


@pytest.mark.parametrize("profile", ["standard", "fast", "slow"])
def test_fast_exact_divide_is_verified_and_used(profile):
    """The reciprocal-plus-correction divide is only switched on for a sin(phi) that passed the
    exhaustive 2^24-significand check at plan creation; the three stock profiles do."""
    s = apt.Settings.profile(profile)
    phi = np.float32(2.0) * (np.float32(2.0) * np.float32(2400.0) / np.float32(s.work_rate) * np.float32(np.pi))
    want = np.float32(1.0) / np.sin(phi, dtype=np.float32)
    got = _plan_inv_sinphi(profile)
    assert got != 0.0
    assert abs(got - float(want)) <= 2e-7 * abs(float(want))


@pytest.mark.parametrize("scale,what", [(1e-25, "denormal-range radicands"), (3e14, "radicands past 2^100"),
                                        (1.0, "half digital silence")])
def test_envelope_out_of_range_values_take_the_general_path(ctx, oracle, scale, what):
    """Radicands outside [2^-96, 2^100] (and exact zeros) leave the fast exact envelope and go
    through the general correctly rounded code, wave by wave: still bit-identical."""
    x = synth_apt(48000, 14, seed=33)
    if what == "half digital silence":
        x = x.copy()
        x[x.size // 3: 2 * x.size // 3] = 0.0
    else:
        x = (x * np.float32(scale)).astype(np.float32)
    for sync in (True, False):
        try:
            want = oracle.decode(x, 48000, sync)
        except Exception as e:  # "less than 5 sync frames" on both sides then
            with pytest.raises(apt.InternalError, match=str(e)[:30]):
                apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(48000), sync)
            continue
        got = apt.decode(ctx, apt.Settings(), x, apt.Rate.hz(48000), sync)
        assert_bitexact(got, want, f"{what} sync={sync}")


def test_general_envelope_forced(oracle, monkeypatch):
    """APTGPU_GENERAL_ENVELOPE=1 keeps the compiler's general sqrt / divide everywhere."""
    monkeypatch.setenv("APTGPU_GENERAL_ENVELOPE", "1")
    for rate in (48000, 11025):
        x = synth_apt(rate, 14, seed=3)
        got = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(rate), True)
        assert_bitexact(got, oracle.decode(x, rate, True), f"general envelope {rate}")
    assert _plan_inv_sinphi("standard") == 0.0


# ------------------------------------------------------------------ threads / graph capture
def test_decode_is_reentrant_across_threads(oracle):
    """aptgpu_decode from several host threads at once (the GUI calls it from a worker thread,
    gui/work.rs:174): every call owns its plan, stream and buffers."""
    import threading
    recs = [synth_apt(48000, 12 + i, seed=60 + i) for i in range(4)]
    wants = [oracle.decode(r, 48000, True) for r in recs]
    out, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                out[i] = apt.decode(apt.Context(device=0), apt.Settings(), recs[i], apt.Rate.hz(48000), True)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert_bitexact(out[i], wants[i], f"thread {i}")


def test_plan_decode_captured_in_a_hip_graph(oracle):
    """decode_device + join issue no host synchronisation, so the whole fork/join over the
    plan's internal streams can be captured from ctx.stream into a graph and replayed."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    recs = [synth_apt(48000, 12 + i, seed=80 + i) for i in range(2)]
    n = max(r.size for r in recs)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=n, max_batch=2,
                        stream=stream.cuda_stream)
        cap = int(plan.info.max_rows)
        d_in = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in recs]
        d_rows = [torch.zeros(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
        args = ([t.data_ptr() for t in d_in], [r.size for r in recs], [t.data_ptr() for t in d_rows], [cap] * 2)
        plan.decode_device(*args)  # warm-up outside the capture (lazy allocations, attributes)
        plan.join()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            plan.decode_device(*args)
            plan.join()
        for rep in range(2):
            for t, r in zip(d_in, recs):
                t[:r.size].copy_(torch.from_numpy(r if rep == 0 else r[::-1].copy()))
            for t in d_rows:
                t.zero_()
            graph.replay()
            stream.synchronize()
            for i, r in enumerate(recs):
                src = r if rep == 0 else r[::-1].copy()
                try:
                    want = oracle.decode(src, 48000, True)
                except Exception:
                    continue
                assert_bitexact(d_rows[i][:want.size].cpu().numpy(), want, f"graph replay {rep} rec {i}")
    plan.close()


# ------------------------------------------------------------------ SPLIT stage 1 (48 / 96 kHz front ends)
@pytest.mark.parametrize("rate", [48000, 96000])
def test_split_stage1_recording_ends_around_the_sub_tile_boundaries(oracle, rate):
    """The specialised front ends run stage 1 over two sub-tiles of 128 windows (apt_kernels_fused_launch.hpp); a
    tile owns 244 windows = 244 m input samples.  Recordings that end just inside a tile, one window either side of
    the sub-tile boundary, deep in the second sub-tile, on the tile's last sample and on its boundary — as one ragged
    batch (f32 Signals and PCM16 WAV images), so that the last tile's loads, its zero fill and the edge copies of the
    stages behind see every case.  Bit-exact, sync and no-sync."""
    from noaa_apt_amd.testing.wavfile import make_wav
    m = rate * 13 // 12480            # input samples per window (l = 13 outputs)
    assert m in (50, 100)
    tile = 244 * m
    tiles = (12 * rate) // tile       # ~12 s: more than the 10 rows decode() needs
    full = synth_apt(rate, 14, seed=77 + rate % 7)
    ends = [3 * m - 7, 125 * m, 128 * m - 1, 129 * m + 1, 131 * m + m // 2, 200 * m + 17, tile - 1, tile]
    recs = [full[:tiles * tile + e] for e in ends]
    for sync in (True, False):
        wants = [oracle.decode(r, rate, sync) for r in recs]
        got = apt.decode_batch(apt.Context(device=0), apt.Settings(), recs, apt.Rate.hz(rate), sync, recordings_per_call=len(recs))
        for e, g, w in zip(ends, got, wants):
            assert not isinstance(g, Exception), (e, g)
            assert_bitexact(g, w, f"{rate} Hz, recording ends {e} samples into a tile, sync={sync}")
    wavs = [make_wav(r.astype(np.int16), rate) for r in recs]
    wants = [oracle.decode(r.astype(np.int16).astype(f32), rate, True) for r in recs]
    got = apt.decode_batch(apt.Context(device=0), apt.Settings(), wavs, apt.Rate.hz(rate), True, recordings_per_call=3)
    for e, g, w in zip(ends, got, wants):
        assert not isinstance(g, Exception), (e, g)
        assert_bitexact(g, w, f"{rate} Hz PCM16, recording ends {e} samples into a tile")


# ------------------------------------------------------------------ user-tuned tap counts (kModeStrictPad)
TUNED = [(48000, dict(resample_atten=29.0)), (48000, dict(resample_atten=31.0)), (48000, dict(resample_delta_freq=900.0)),
         (48000, dict(resample_delta_freq=1100.0)), (96000, dict(resample_atten=31.0)), (96000, dict(resample_delta_freq=900.0)),
         (96000, dict(resample_delta_freq=1100.0)), (48000, dict(resample_cutout=5200.0))]


TUNED_PROFILES = [(48000, "slow", dict(resample_atten=41.0)), (48000, "slow", dict(resample_delta_freq=520.0)),
                  (96000, "slow", dict(resample_atten=39.0)), (96000, "fast", dict(resample_atten=31.0)),
                  (96000, "fast", dict(resample_delta_freq=2900.0))]


@pytest.mark.parametrize("rate,profile,kw", TUNED_PROFILES)
def test_tuned_settings_of_the_other_profiles(ctx, oracle, rate, profile, kw):
    """The slow profile at 48 / 96 kHz and the fast profile at 96 kHz (l = 13: the SPLIT kernels) with a tuned filter."""
    s = apt.Settings.profile(profile)
    for k, v in kw.items():
        setattr(s, k, v)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    x = synth_apt(rate, 12, seed=54)
    for sync in (True, False):
        want, st = oracle.decode(x, rate, sync, settings=os_, want_steps=True)
        got, stats = apt.decode(ctx, s, x, apt.Rate.hz(rate), sync, return_stats=True)
        assert stats.n_resample_taps == st["resample_filter"].size and stats.n_resample_taps not in (2783, 5565, 639)
        assert stats.fused == 1, (rate, profile, kw, stats.n_resample_taps)
        assert_bitexact(got, want, f"tuned {rate} {profile} {kw} sync={sync}")


@pytest.mark.parametrize("rate,kw", TUNED)
@pytest.mark.parametrize("sync", [True, False])
def test_tuned_settings_stay_on_the_specialised_kernel(ctx, oracle, rate, kw, sync):
    """default_settings.toml:108-140 is a user-editable file.  A tuned resample_atten / resample_delta_freq changes the tap
    COUNT (959 / 1915 at the stock values): such a plan runs the strict SPLIT kernel compiled for a tap-count bound, with a
    zero-padded table (kModeStrictPad) — stats.fused == 1, bit-exact (until round 6: k_fused_any)."""
    s = apt.Settings(**kw)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    x = synth_apt(rate, 12, seed=51)
    want, st = oracle.decode(x, rate, sync, settings=os_, want_steps=True)
    got, stats = apt.decode(ctx, s, x, apt.Rate.hz(rate), sync, return_stats=True)
    assert stats.n_resample_taps == st["resample_filter"].size
    assert stats.fused == 1, (rate, kw, stats.n_resample_taps)
    assert_bitexact(got, want, f"tuned {rate} {kw} sync={sync}")


def test_padded_kernel_on_the_stock_tap_counts_and_non_finite_samples(oracle, monkeypatch):
    """APTGPU_FUSED_PAD=1: the padded kernels serve the stock tap counts too (74 of their 83 / 148 of 165 taps per branch
    real, the rest zeros) — bit-exact, also where a NaN or an infinity sits under a zero tap (0 x inf = NaN where the
    reference, which skips that tap, has none: such a tile is evaluated again sample by sample), and with PCM16 input."""
    monkeypatch.setenv("APTGPU_FUSED_PAD", "1")
    apt.cache_clear()
    c = apt.Context(device=0)
    for rate in (48000, 96000):
        x = synth_apt(rate, 12, seed=52)
        got, stats = apt.decode(c, apt.Settings(), x, apt.Rate.hz(rate), True, return_stats=True)
        assert stats.fused == 1
        assert_bitexact(got, oracle.decode(x, rate, True), f"padded, stock taps, {rate}")
        for bad in (np.nan, np.inf, -np.inf):
            y = x.copy()
            for i in (5, 77777, 200001, 200002, y.size - 3):
                y[i] = bad
            for sync in (True, False):
                try:
                    want = oracle.decode(y, rate, sync)
                except oracle.OracleError as e:
                    with pytest.raises(apt.AptError) as ge:
                        apt.decode(c, apt.Settings(), y, apt.Rate.hz(rate), sync)
                    assert str(ge.value) == str(e)
                    continue
                assert_same_values(apt.decode(c, apt.Settings(), y, apt.Rate.hz(rate), sync), want, f"padded {rate} {bad} sync={sync}")
    # PCM16 payload
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    x = synth_apt(48000, 12, seed=53)
    want = oracle.decode(x, 48000, True)
    plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=x.size)
    d_pcm = torch.from_numpy(x.astype(np.int16)).to(dev)
    cap = int(plan.info.max_rows)
    d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    spec = apt.WavSpec(1, 16, 2, 0, 48000, 1, 0, 2 * x.size, x.size, x.size)
    plan.decode_device_wav([d_pcm.data_ptr()], [spec], [d_out.data_ptr()], [cap])
    res = plan.results(1)[0]
    assert_bitexact(d_out[:res.n_out].cpu().numpy(), want, "padded, PCM16")
    plan.close()
    apt.cache_clear()


# ------------------------------------------------------------------ a tuned demodulation_atten (kModeStrictPad2)
DEMOD_TUNED = [(48000, dict(demodulation_atten=20.0)), (48000, dict(demodulation_atten=24.0)),
               (48000, dict(demodulation_atten=26.0)), (48000, dict(demodulation_atten=27.5)),
               (48000, dict(demodulation_atten=29.0)), (96000, dict(demodulation_atten=24.0)),
               (96000, dict(demodulation_atten=28.0)), (48000, dict(demodulation_atten=26.0, resample_atten=31.0)),
               (96000, dict(demodulation_atten=23.0, resample_delta_freq=1100.0))]


@pytest.mark.parametrize("rate,kw", DEMOD_TUNED)
@pytest.mark.parametrize("sync", [True, False])
def test_tuned_demodulation_atten_stays_on_the_specialised_kernel(ctx, oracle, rate, kw, sync):
    """demodulation_atten (default_settings.toml:116) moves the Kaiser length of the LOW-PASS (25 dB: 37 taps): such a plan
    runs the padded strict kernel whose low-pass length is a bound too (kModeStrictPad2: up to 45 taps, zero-padded h2 /
    h2p) — stats.fused == 1, bit-exact, alone or together with a tuned resampler (until this round: k_fused_any)."""
    s = apt.Settings(**kw)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    x = synth_apt(rate, 12, seed=57)
    want, st = oracle.decode(x, rate, sync, settings=os_, want_steps=True)
    got, stats = apt.decode(ctx, s, x, apt.Rate.hz(rate), sync, return_stats=True)
    assert stats.n_lowpass_taps == st["filter_filter"].size and stats.n_lowpass_taps != 37 and stats.n_lowpass_taps <= 45
    assert stats.fused == 1, (rate, kw, stats.n_resample_taps, stats.n_lowpass_taps)
    assert_bitexact(got, want, f"tuned {rate} {kw} sync={sync}")


def test_tuned_demodulation_atten_non_finite_samples_pcm16_and_the_bound(oracle):
    """kModeStrictPad2 where a zero tap of the padded low-pass meets a non-finite envelope value (0 x inf = NaN where the
    reference, whose filter ends before it, has none: such a tile is filtered again with the run-time tap count); from the
    first samples of a recording (the `i > j` guard of dsp.rs:396-404 with fewer taps than the kernel's bound); PCM16
    input; several recordings per call; and a low-pass longer than the bound (47 taps: k_fused_any, same results)."""
    c = apt.Context(device=0)
    s = apt.Settings(demodulation_atten=24.0)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    for rate in (48000, 96000):
        x = synth_apt(rate, 12, seed=58)
        for bad in (np.nan, np.inf, -np.inf):
            y = x.copy()
            for i in (0, 5, 77777, 200001, 200002, y.size - 3):
                y[i] = bad
            y[300000:300400] = bad
            for sync in (True, False):
                try:
                    want = oracle.decode(y, rate, sync, settings=os_)
                except oracle.OracleError as e:
                    with pytest.raises(apt.AptError) as ge:
                        apt.decode(c, s, y, apt.Rate.hz(rate), sync)
                    assert str(ge.value) == str(e)
                    continue
                got, stats = apt.decode(c, s, y, apt.Rate.hz(rate), sync, return_stats=True)
                assert stats.fused == 1 and stats.n_lowpass_taps == 35
                assert_same_values(got, want, f"padded low-pass {rate} {bad} sync={sync}")
        # amplitudes whose F overflows: the same infinities, through the run-time loop
        z = (x * np.float32(3e33)).astype(np.float32)
        assert_same_values(apt.decode(c, s, z, apt.Rate.hz(rate), True), oracle.decode(z, rate, True, settings=os_),
                           f"padded low-pass {rate}, overflow")
    # PCM16 payloads, three recordings of different lengths in one call
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    recs = [synth_apt(48000, 12, seed=59), synth_apt(48000, 31, seed=60)[:48000 * 31 - 7], synth_apt(48000, 9, seed=61)]
    plan = apt.Plan(s, apt.Rate.hz(48000), True, max_samples=max(r.size for r in recs), max_batch=3)
    assert plan.info.fused == 1 and plan.info.n_lowpass_taps == 35
    d_pcm = [torch.from_numpy(r.astype(np.int16)).to(dev) for r in recs]
    cap = int(plan.info.max_rows)
    d_out = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
    torch.cuda.synchronize()
    specs = [apt.WavSpec(1, 16, 2, 0, 48000, 1, 0, 2 * r.size, r.size, r.size) for r in recs]
    plan.decode_device_wav([d.data_ptr() for d in d_pcm], specs, [d.data_ptr() for d in d_out], [cap] * 3)
    for i, (r, res) in enumerate(zip(recs, plan.results(3))):
        assert_bitexact(d_out[i][:res.n_out].cpu().numpy(), oracle.decode(r, 48000, True, settings=os_), f"padded low-pass, PCM16 {i}")
    plan.close()
    # beyond the bound
    s2 = apt.Settings(demodulation_atten=30.0)
    os2 = dict(os_, demodulation_atten=30.0)
    x = synth_apt(48000, 12, seed=62)
    got, stats = apt.decode(c, s2, x, apt.Rate.hz(48000), True, return_stats=True)
    assert stats.n_lowpass_taps > 45 and stats.fused == 2
    assert_bitexact(got, oracle.decode(x, 48000, True, settings=os2), "low-pass beyond the padded kernel's bound")
    apt.cache_clear()


DEMOD_TUNED_PHASE = [(44100, 14, dict(demodulation_atten=24.0)), (44100, 14, dict(demodulation_atten=28.0)),
                     (22050, 20, dict(demodulation_atten=26.0)), (11025, 30, dict(demodulation_atten=24.0)),
                     (11025, 30, dict(demodulation_atten=29.0, resample_atten=31.0)), (16000, 20, dict(demodulation_atten=20.0)),
                     (32000, 15, dict(demodulation_atten=27.5))]


@pytest.mark.parametrize("rate,seconds,kw", DEMOD_TUNED_PHASE)
@pytest.mark.parametrize("sync", [True, False])
def test_tuned_demodulation_atten_at_the_sound_card_rates(oracle, rate, seconds, kw, sync):
    """kModeStrictPad2 on the PHASE kernels (one, two, four branches per thread): a tuned demodulation_atten at 44 100 /
    22 050 / 11 025 Hz (and 16 / 32 kHz) keeps stats.fused == 4 — the low-pass tables zero-padded to 45 taps, a tile whose F
    is not all finite filtered again with the run-time count — bit-exact, with a NaN and an infinity in the input, in
    strict mode and (the same strict kernels) in APTGPU_MODE_FAST; APTGPU_FUSED_PAD=0: k_fused_any as until round 6."""
    s = apt.Settings(**kw)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    x = synth_apt(rate, seconds, seed=rate % 67 + seconds)
    x[x.size // 3] = np.nan
    x[11] = -np.inf
    want = oracle.decode(x, rate, sync, settings=os_)
    for mode in (apt.MODE_STRICT, apt.MODE_FAST):
        got, stats = apt.decode(apt.Context(device=0, mode=mode), s, x, apt.Rate.hz(rate), sync, return_stats=True)
        assert stats.fused == 4 and stats.n_lowpass_taps != 37, (rate, kw, stats.fused, stats.n_lowpass_taps)
        assert_same_values(got, want, f"tuned low-pass on the PHASE kernels: {rate} {kw} sync={sync} mode={mode}")


def test_tuned_demodulation_atten_phase_switch_off_and_pcm16(oracle, monkeypatch):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    s = apt.Settings(demodulation_atten=26.0)
    os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                       "resample_cutout", "demodulation_atten")}
    # PCM16 payloads, two recordings per call, 11 025 Hz
    recs = [synth_apt(11025, 40, seed=71), synth_apt(11025, 25, seed=72)[:11025 * 25 - 5]]
    plan = apt.Plan(s, apt.Rate.hz(11025), True, max_samples=max(r.size for r in recs), max_batch=2)
    assert plan.info.fused == 4 and plan.info.n_lowpass_taps == 39
    d_pcm = [torch.from_numpy(r.astype(np.int16)).to(dev) for r in recs]
    cap = int(plan.info.max_rows)
    d_out = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in recs]
    torch.cuda.synchronize()
    specs = [apt.WavSpec(1, 16, 2, 0, 11025, 1, 0, 2 * r.size, r.size, r.size) for r in recs]
    plan.decode_device_wav([d.data_ptr() for d in d_pcm], specs, [d.data_ptr() for d in d_out], [cap] * 2)
    for i, (r, res) in enumerate(zip(recs, plan.results(2))):
        assert_bitexact(d_out[i][:res.n_out].cpu().numpy(), oracle.decode(r, 11025, True, settings=os_), f"PHASE pad2, PCM16 {i}")
    plan.close()
    # the switch
    monkeypatch.setenv("APTGPU_FUSED_PAD", "0")
    apt.cache_clear()
    x = synth_apt(44100, 14, seed=73)
    got, stats = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(44100), True, return_stats=True)
    assert stats.fused == 2
    assert_bitexact(got, oracle.decode(x, 44100, True, settings=os_), "tuned low-pass, APTGPU_FUSED_PAD=0")
    apt.cache_clear()


def _random_tunings(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append(dict(resample_atten=float(np.round(rng.uniform(20.0, 42.0), 1)),
                        resample_delta_freq=float(np.round(rng.uniform(650.0, 1600.0))),
                        resample_cutout=float(np.round(rng.uniform(4200.0, 5400.0)))))
    return out


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("rate", [48000, 96000])
def test_random_tunings_of_the_standard_profile(oracle, rate, mode):
    """Seeded random (attenuation, transition width, cutout) triples around the standard profile: tap counts on both
    sides of the padded kernels' bounds (1079 / 2145) and of the matrix-core kernel's (1053 / 2119) — whichever kernel the
    plan lands on, strict mode is bit-exact and fast mode inside its tolerance; a count above every bound takes the
    run-time kernel (stats.fused == 2)."""
    from test_gpu_fast import check_tolerance, decode_on_plan
    seen = set()
    for kw in _random_tunings(7, 600 + rate // 1000):
        s = apt.Settings(**kw)
        os_ = {k: getattr(s, k) for k in ("work_rate", "resample_atten", "resample_delta_freq",
                                           "resample_cutout", "demodulation_atten")}
        x = synth_apt(rate, 11, seed=int(kw["resample_delta_freq"]))
        want, st = oracle.decode(x, rate, True, settings=os_, want_steps=True)
        t1 = st["resample_filter"].size
        bound = 1079 if rate == 48000 else 2145
        if mode == "strict":
            got, stats = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(rate), True, return_stats=True)
            assert stats.n_resample_taps == t1
            assert stats.fused == (1 if t1 <= bound else 2), (kw, t1, stats.fused)
            assert_bitexact(got, want, f"random tuning {rate} {kw} ({t1} taps)")
            seen.add(stats.fused)
        else:
            rows, pos, res, fused = decode_on_plan(x, rate, apt.MODE_FAST, settings=s)
            assert fused == (1 if t1 <= bound else 2) and res.status == 0, (kw, t1, fused)
            check_tolerance(rows, pos, want, st["sync_pos"], f"random tuning, fast mode, {rate} {kw} ({t1} taps)")
            seen.add(fused)
    assert seen == {1, 2}, seen  # the draw covers both sides of the bound
