"""Settings.export_resample_filtered on the GPU path (/root/reference/src/config.rs:83 -> context.rs:113): the flag
moves the decimation phase of fast_resampling (dsp.rs:265-273: (t + 1) % m == 0 instead of t = offset + k*m), so it
changes decode()'s rows whether or not anything is exported, and with export_wav it delivers the expanded signal as
the "resample_filtered" step (dsp.rs:269,281-285).  Bit-exact against the oracle's restatement of that branch
(tests/test_oracle_export_filtered.py pins the restatement to a transcription of the loop)."""
import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
from noaa_apt_amd.testing.wavfile import make_wav

pytestmark = pytest.mark.gpu

f32 = np.float32


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


def assert_bitexact(got, want, what=""):
    got, want = np.asarray(got, f32), np.asarray(want, f32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(_bits(got), _bits(want)), what


@pytest.fixture(scope="module")
def ctx():
    assert apt.device_count() >= 1, "no HIP device: the GPU tests must run on the GPU box"
    return apt.Context(device=0)


def _settings(profile, **kw):
    s = apt.Settings.profile(profile)
    s.export_resample_filtered = True
    for k, v in kw.items():
        setattr(s, k, v)
    return s


CASES = [(48000, 7, "standard"), (11025, 7, "standard"), (44100, 7, "standard"), (48000, 7, "fast"), (48000, 7, "slow"),
         (12480, 7, "standard")]  # the last one: l == 1, fast_resampling is not called and the flag changes nothing


@pytest.mark.parametrize("sync", [True, False])
@pytest.mark.parametrize("rate,seconds,profile", CASES)
def test_flag_moves_the_decimation_phase(ctx, oracle, rate, seconds, profile, sync):
    x = synth_apt(rate, seconds, 40 + rate % 97)
    o = {"standard": oracle.STANDARD, "fast": oracle.FAST, "slow": oracle.SLOW}[profile]
    rows, st = apt.decode(ctx, _settings(profile), x, apt.Rate.hz(rate), sync, return_stats=True)
    want = oracle.decode(x, rate, sync, settings=o, export_resample_filtered=True)
    assert_bitexact(rows, want, "rows under export_resample_filtered")
    normal = apt.decode(ctx, apt.Settings.profile(profile), x, apt.Rate.hz(rate), sync)
    if st.l > 1:
        assert st.fused == 0  # the unfused kernels serve this mode
        assert rows.size != normal.size or not np.array_equal(_bits(rows), _bits(normal))
    else:
        assert_bitexact(rows, normal)
    # a second call comes from the session cache (its key holds the flag) and a normal call in between must not
    # have replaced the plan
    again = apt.decode(ctx, _settings(profile), x, apt.Rate.hz(rate), sync)
    assert_bitexact(again, want)


@pytest.mark.parametrize("sync", [True, False])
def test_flag_is_honoured_in_fp16_taps_mode(oracle, sync):
    """APTGPU_MODE_FP16_TAPS at 48 kHz normally runs stage 1 inside the specialised fused kernel, which decimates at
    t = off + k m; with export_resample_filtered the plan must leave it (advisor, round 5): same row count and the same
    rows as the oracle's flagged decode within that mode's tolerance (2e-3 of full scale), not the unflagged phase."""
    x = synth_apt(48000, 9, 71)
    c = apt.Context(device=0, mode=apt.MODE_FP16_TAPS)
    rows, st = apt.decode(c, _settings("standard"), x, apt.Rate.hz(48000), sync, return_stats=True)
    want = oracle.decode(x, 48000, sync, settings=oracle.STANDARD, export_resample_filtered=True)
    assert st.fused == 0  # not the fused fp16 kernel
    assert rows.shape == np.asarray(want).shape
    err = np.max(np.abs(rows - want)) / np.max(np.abs(want))
    assert err <= 2e-3, err
    unflagged = oracle.decode(x, 48000, sync, settings=oracle.STANDARD)
    if np.asarray(unflagged).shape == rows.shape:
        assert np.max(np.abs(rows - unflagged)) / np.max(np.abs(want)) > err  # closer to the flagged decode


@pytest.mark.parametrize("sync", [True, False])
def test_expanded_signal_is_exported(oracle, sync):
    x = synth_apt(48000, 7, 77)
    got = []
    c = apt.Context(step_callback=lambda i, v, d, r: got.append((i, v, d, r)), device=0)
    rows = apt.decode(c, _settings("standard", export_wav=True), x, apt.Rate.hz(48000), sync)
    want_rows, st = oracle.decode(x, 48000, sync, want_steps=True, export_resample_filtered=True)
    assert_bitexact(rows, want_rows)
    by = {}
    for i, v, d, r in got:
        by.setdefault(i, []).append((v, d, r))
    assert [g[0] for g in got][:4] == ["input", "resample_filter", "resample_filtered", "resample_decimated"]
    assert by["resample_filtered"][0][2] == 48000 * 13
    assert_bitexact(by["resample_filtered"][0][1], st["expanded1"], "expanded signal")
    assert_bitexact(by["resample_decimated"][0][1], st["resampled"])
    assert_bitexact(by["demodulation_result"][0][1], st["demodulated"])
    assert_bitexact(by["filter_result"][0][1], st["filtered"])
    if sync:
        assert_bitexact(by["sync_correlation"][0][1], st["correlation"])
        assert_bitexact(by["sync_result"][0][1], st["aligned"])
    # the final stage of a stock profile is filter + decimate (l == 1): its "resample_filtered" is the filtered signal
    assert_bitexact(by["resample_filtered"][1][1], st["expanded2"])
    assert_bitexact(by["resample_decimated"][1][1], want_rows)


def test_expanded_signal_of_the_final_stage(oracle):
    """work_rate 11025 (4160 does not divide it; no-sync only): the NoFilter resample to 4160 Hz is fast_resampling
    too (l2 = 832, m2 = 2205) and follows the flag."""
    x = synth_apt(48000, 6, 78)
    got = []
    c = apt.Context(step_callback=lambda i, v, d, r: got.append((i, v, d, r)), device=0)
    rows = apt.decode(c, _settings("standard", export_wav=True, work_rate=11025), x, apt.Rate.hz(48000), False)
    s = dict(oracle.STANDARD, work_rate=11025)
    want_rows, st = oracle.decode(x, 48000, False, settings=s, want_steps=True, export_resample_filtered=True)
    assert_bitexact(rows, want_rows)
    ex = [g for g in got if g[0] == "resample_filtered"]
    assert len(ex) == 2 and ex[0][3] == 48000 * 147 and ex[1][3] == 11025 * 832
    assert_bitexact(ex[0][2], st["expanded1"], "expanded signal, first resample")
    assert_bitexact(ex[1][2], st["expanded2"], "expanded signal, final resample")
    # and without the step export the rows are the same ones
    rows2 = apt.decode(None, _settings("standard", work_rate=11025), x, apt.Rate.hz(48000), False)
    assert_bitexact(rows2, want_rows)


def test_wav_input_and_batch_follow_the_flag(ctx, oracle):
    xs = [synth_apt(11025, 7, 91), synth_apt(11025, 8, 92)]
    want = [oracle.decode(x, 11025, True, export_resample_filtered=True) for x in xs]
    out = apt.decode_batch(ctx, _settings("standard"), xs, apt.Rate.hz(11025), True)
    for r, w in zip(out, want):
        assert not isinstance(r, Exception), r
        assert_bitexact(r, w)
    wav = make_wav(xs[0].astype(np.int16), 11025)  # the synthetic recordings are int16-valued (wav.rs:37 hands them over unscaled)
    rows, st = apt.decode_wav(ctx, _settings("standard"), wav, True, return_stats=True)
    assert st.fused == 0
    assert_bitexact(rows, want[0])


# ---- the WAV -> WAV tool (resample.rs:17-71; main.rs:125-130 hands settings.export_resample_filtered to
# Context::resample): the steps of Context::resample in the reference's order, and the flag
@pytest.fixture(scope="module")
def ow():
    from oracle import wav_binding
    wav_binding.lib()
    return wav_binding


@pytest.mark.parametrize("flag", [False, True])
@pytest.mark.parametrize("in_rate,out_rate", [(11025, 48000), (11025, 6000), (11025, 3675), (48000, 11025)])
def test_resample_tool_steps_and_flag(ow, in_rate, out_rate, flag):
    x = synth_apt(in_rate, 2, seed=out_rate % 89)
    data = make_wav(x.astype(np.int16), in_rate)
    s = apt.Settings(export_resample_filtered=flag)
    got = []
    c = apt.Context(step_callback=lambda ident, variant, arr, rate: got.append((ident, variant, arr, rate)), device=0)
    out = apt.resample_wav(c, s, data, "out.wav", out_rate)
    want, st = ow.resample_wav(data, out_rate, s.wav_resample_atten, s.wav_resample_delta_freq,
                               export_resample_filtered=flag, return_steps=True)
    assert out == want
    assert [g[0] for g in got] == ["input", "resample_filter", "resample_filtered", "resample_decimated"]
    by = {g[0]: g for g in got}
    assert_bitexact(by["input"][2], st["input"])
    assert by["resample_filter"][1] == 1
    assert_bitexact(by["resample_filter"][2], st["resample_filter"])
    g = int(np.gcd(in_rate, out_rate))
    l = out_rate // g
    assert_bitexact(by["resample_filtered"][2], st["resample_filtered"], "resample_filtered")
    assert by["resample_filtered"][3] == in_rate * l  # (input_rate itself where l == 1, dsp.rs:110-114)
    assert (by["resample_filtered"][2].size > 0) == (flag or l == 1)
    assert_bitexact(by["resample_decimated"][2], st["resample_decimated"])
    assert by["resample_decimated"][3] == out_rate
    # without a step callback: the same file
    assert apt.resample_wav(None, s, data, None, out_rate) == want
    if flag and l > 1:
        assert want != ow.resample_wav(data, out_rate, s.wav_resample_atten, s.wav_resample_delta_freq)


@pytest.mark.parametrize("flag", [False, True])
def test_first_resample_with_l_equal_1_delivers_the_filtered_signal(oracle, flag):
    """decode() of a 24 960 Hz recording: the first resample is the l == 1 branch (filter + decimate, dsp.rs:106-116),
    whose "resample_filtered" step carries the filtered signal at the input rate whatever the flag says (the flag only
    matters inside Context::step, context.rs:157).  Until round 6 decode_host never delivered it (advisor, round 5)."""
    x = synth_apt(24960, 8, 77)
    got = []
    c = apt.Context(step_callback=lambda i, v, d, r: got.append((i, np.array(d, copy=True), r)), device=0)
    rows = apt.decode(c, apt.Settings(export_wav=True, export_resample_filtered=flag), x, apt.Rate.hz(24960), True)
    want_rows = oracle.decode(x, 24960, True, export_resample_filtered=flag)
    # (the oracle hands out the "resample_filtered" signals with the flag set; at l == 1 the flag changes nothing else)
    _, st = oracle.decode(x, 24960, True, want_steps=True, export_resample_filtered=True)
    assert_bitexact(rows, want_rows)
    assert [g[0] for g in got][:4] == ["input", "resample_filter", "resample_filtered", "resample_decimated"]
    first = [g for g in got if g[0] == "resample_filtered"][0]
    assert first[2] == 24960 and first[1].size == x.size
    assert_bitexact(first[1], st["expanded1"], "filtered signal of the l == 1 branch")
