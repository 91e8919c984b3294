"""GPU parity of the WAV ingest: file image -> Signal on the device, and
file image -> pixel rows without the host f32 detour, bit-for-bit against the oracle."""
import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
from noaa_apt_amd.testing.wavfile import make_wav

pytestmark = pytest.mark.gpu

f32 = np.float32


@pytest.fixture(scope="module")
def ow():
    from oracle import wav_binding
    return wav_binding


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


def _same(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return a.shape == b.shape and np.array_equal(_bits(a), _bits(b))


def _int_values(bits, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    v = rng.integers(lo, hi + 1, size=n, dtype=np.int64)
    k = min(4, n)
    v[:k] = [lo, hi, 0, -1][:k]
    return v


LOAD_CASES = [
    ("pcm16_mono", dict(), 16, 1),
    ("pcm16_mono_odd_offset", dict(extra_chunks=[(b"junk", b"abc")]), 16, 1),     # data at an odd address
    ("pcm16_mono_2mod4_offset", dict(extra_chunks=[(b"junk", b"ab")]), 16, 1),    # 2-byte aligned only
    ("pcm16_stereo", dict(channels=2), 16, 2),
    ("pcm16_6ch", dict(channels=6), 16, 6),
    ("pcm8", dict(bits=8), 8, 1),
    ("pcm8_stereo", dict(bits=8, channels=2), 8, 2),
    ("pcm24", dict(bits=24), 24, 1),
    ("pcm24_stereo", dict(bits=24, channels=2, fmt_len=18), 24, 2),
    ("pcm24_in_4", dict(bits=24, container_bytes=4, extensible=True), 24, 1),
    ("pcm32", dict(bits=32), 32, 1),     # |x| > 2^24: `as f32` rounds to nearest even
    ("float32", dict(is_float=True), 32, 1),
    ("float32_stereo", dict(is_float=True, channels=2, fmt_len=18), 32, 2),
]


@pytest.mark.parametrize("name,kw,bits,channels", LOAD_CASES, ids=[c[0] for c in LOAD_CASES])
@pytest.mark.parametrize("frames", [1, 7, 8, 9, 100_003])
def test_load_matches_oracle(ow, name, kw, bits, channels, frames):
    if kw.get("is_float"):
        rng = np.random.default_rng(frames)
        vals = (rng.standard_normal(frames * channels) * 1e4).astype(f32)
        vals[0] = np.float32(-0.0)
        if vals.size > 2:
            vals[1], vals[2] = np.float32(np.nan), np.float32(1e-41)
    else:
        vals = _int_values(bits, frames * channels, frames)
    data = make_wav(vals, 44100, **kw)
    want, ospec = ow.load_wav(data)
    got, rate, spec = apt.load(data, return_spec=True)
    assert rate.get_hz() == 44100 == ospec.sample_rate
    assert _same(got, want)
    assert (spec.channels, spec.bits_per_sample, spec.n_frames) == (channels, bits, frames)


def test_load_empty_and_file_path(ow, tmp_path):
    data = make_wav(np.zeros(0, np.int16), 8000)
    got, rate = apt.load(data)
    assert got.size == 0 and rate.get_hz() == 8000
    vals = _int_values(16, 5000, 3)
    p = tmp_path / "rec.wav"
    p.write_bytes(make_wav(vals, 11025))
    got, rate = apt.load(str(p))
    assert rate.get_hz() == 11025 and _same(got, ow.load_wav(p.read_bytes())[0])
    with pytest.raises(apt.IoError):
        apt.load(str(tmp_path / "missing.wav"))


def test_load_errors():
    good = make_wav(np.arange(100), 8000)
    with pytest.raises(apt.WavOpenError, match="no RIFF tag found"):
        apt.load(b"RIFX" + good[4:])
    with pytest.raises(apt.IoError, match="Failed to read enough bytes"):
        apt.load(good[:100])
    with pytest.raises(apt.WavOpenError, match="not supported"):
        apt.load(make_wav(np.arange(4), 8000, format_tag=2))


def _pcm16(x):
    """The synthetic recordings are int16-valued floats (wav.rs hands decode() unscaled ints)."""
    assert np.array_equal(x, np.round(x)) and np.abs(x).max() <= 32767
    return x.astype(np.int16)


@pytest.mark.parametrize("rate,seconds,kw,fused", [
    (48000, 20, dict(), 1),                                           # mono PCM16 -> fused int16 front end
    (96000, 12, dict(), 1),
    (48000, 20, dict(extra_chunks=[(b"junk", b"abc")]), 1),           # misaligned payload -> staging + fused f32
    (48000, 20, dict(channels=2), 1),                                 # stereo: first channel only
    (11025, 30, dict(), 4),                                           # phase-resident stage 1 (four branches per thread), int16 input
    (11025, 30, dict(extra_chunks=[(b"junk", b"abc")]), 4),          # odd payload address -> staging + the f32 kernel
    (48000, 20, dict(is_float=True), 1),
    (48000, 20, dict(bits=32), 1),
])
def test_decode_wav_matches_oracle(oracle, ow, rate, seconds, kw, fused):
    x = synth_apt(rate, seconds, seed=rate // 1000 + seconds)
    channels = kw.get("channels", 1)
    if kw.get("is_float"):
        vals = x
    else:
        vals = _pcm16(x).astype(np.int64)
        if kw.get("bits") == 32:
            vals = vals * 65536 + 12345  # needs rounding when converted to f32
    if channels > 1:
        inter = np.empty(vals.size * channels, vals.dtype)
        inter[0::channels] = vals
        for c in range(1, channels):
            inter[c::channels] = vals[::-1]  # the other channel must be ignored
        vals = inter
    data = make_wav(vals, rate, **kw)
    sig, _ = ow.load_wav(data)
    want = oracle.decode(sig, rate, True)
    got, st = apt.decode_wav(apt.Context(device=0), apt.Settings(), data, True, return_stats=True)
    assert st.fused == fused
    assert _same(got, want)


def test_decode_wav_steps_export_input(oracle, ow):
    x = synth_apt(48000, 12, seed=5)
    data = make_wav(_pcm16(x), 48000)
    steps, status = {}, []
    c = apt.Context(ui_callback=lambda p, t: status.append(t),
                    step_callback=lambda ident, variant, arr, rate: steps.setdefault(ident, (arr, rate)))
    s = apt.Settings()
    s.export_wav = True
    got = apt.decode_wav(c, s, data, True)
    sig = ow.load_wav(data)[0]  # (x itself may hold -0.0, which PCM16 cannot)
    assert _same(steps["input"][0], sig) and steps["input"][1] == 48000
    assert status[0] == "Resampling to 12480"
    assert _same(got, oracle.decode(sig, 48000, True))


def test_decode_wav_errors():
    with pytest.raises(apt.WavOpenError):
        apt.decode_wav(None, apt.Settings(), b"not a wav file at all", True)
    short = make_wav(np.zeros(48000, np.int16), 48000)  # 1 s: fewer than 10 rows
    with pytest.raises(apt.InternalError, match="Got less than 10 rows"):
        apt.decode_wav(None, apt.Settings(), short, True)


def test_plan_decode_device_wav_batch(oracle, ow):
    """Device-resident batch of WAV payloads: aligned mono PCM16 (fused int16 front end),
    2-byte-aligned mono PCM16 and stereo (staging buffer), all in one call."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    recs = [synth_apt(48000, 14 + 2 * i, 700 + i, ppm=20.0 * i) for i in range(3)]
    files = [make_wav(_pcm16(recs[0]), 48000),
             make_wav(_pcm16(recs[1]), 48000, extra_chunks=[(b"junk", b"ab")]),
             make_wav(np.stack([_pcm16(recs[2]), _pcm16(recs[2])[::-1]], 1).ravel(), 48000, channels=2)]
    specs = [apt.wav_parse(f) for f in files]
    nmax = max(int(s.n_frames) for s in specs)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=nmax, max_batch=3,
                        stream=stream.cuda_stream)
        cap = int(plan.info.max_rows)
        # whole file images in HBM; the payload pointer is base + data_offset
        d_files = [torch.frombuffer(bytearray(f), dtype=torch.uint8).to(dev) for f in files]
        d_data = [t.data_ptr() + int(s.data_offset) for t, s in zip(d_files, specs)]
        assert d_data[0] % 4 == 0 and d_data[1] % 4 == 2
        d_rows = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in files]
        for _ in range(2):
            plan.decode_device_wav(d_data, specs, [t.data_ptr() for t in d_rows], [cap] * 3)
        res = plan.results(3)
    for i, f in enumerate(files):
        want = oracle.decode(ow.load_wav(f)[0], 48000, True)
        assert res[i].status == 0 and res[i].n_out == want.size
        assert _same(d_rows[i][:want.size].cpu().numpy(), want), i
    with pytest.raises(apt.InvalidError, match="sample rate"):
        bad = apt.wav_parse(make_wav(np.zeros(100, np.int16), 44100))
        plan.decode_device_wav([d_data[0]], [bad], [d_rows[0].data_ptr()], [cap])
    plan.close()


# ------------------------------------------------------------------ write_wav / resample tool (N4)
def test_write_wav_matches_oracle(ow):
    rng = np.random.default_rng(21)
    cases = {
        "normal": (rng.standard_normal(100_003) * 900).astype(f32),
        "with_nan": np.concatenate([[1.0], [np.nan] * 3, rng.standard_normal(50)]).astype(f32),
        "negative_peak": np.array([1.0, -3.0, 0.5, -0.99999], f32),
        "all_negative": np.array([-1.0, -3.0, -0.5], f32),     # max < 0: signs flip, as in the reference
        "zeros": np.zeros(17, f32),                             # 0/0 = NaN -> 0
        "one": np.array([42.0], f32),
    }
    for name, x in cases.items():
        assert apt.write_wav(x, apt.Rate.hz(6000)) == ow.write_wav_i16(x, 6000), name
    with pytest.raises(apt.InternalError, match="maximum of a zero length vector"):
        apt.write_wav(np.zeros(0, f32), apt.Rate.hz(6000))


# test/test.sh:48-52 of the reference: up-sampling, down-sampling, pure decimation
@pytest.mark.parametrize("in_rate,out_rate", [(11025, 48000), (11025, 6000), (11025, 3675),
                                              (48000, 80000), (48000, 11025)])
@pytest.mark.parametrize("profile", ["standard", "fast"])
def test_resample_wav_tool(ow, in_rate, out_rate, profile):
    x = synth_apt(in_rate, 6, seed=out_rate % 97)
    data = make_wav(_pcm16(x), in_rate)
    s = apt.Settings.profile(profile)
    status, steps = [], {}
    c = apt.Context(ui_callback=lambda p, t: status.append((round(p, 2), t)),
                    step_callback=lambda ident, variant, arr, rate: steps.setdefault(ident, (arr, rate)))
    got = apt.resample_wav(c, s, data, "out.wav", out_rate)
    want = ow.resample_wav(data, out_rate, s.wav_resample_atten, s.wav_resample_delta_freq)
    assert got == want
    assert status == [(0.0, "Reading WAV file"), (0.2, f"Resampling to {out_rate}"),
                      (0.8, "Writing WAV to 'out.wav'"), (1.0, "Finished")]
    assert _same(steps["input"][0], ow.load_wav(data)[0]) and steps["input"][1] == in_rate


def test_resample_wav_files_and_errors(ow, tmp_path):
    import os
    x = synth_apt(11025, 4, seed=9)
    src, dst = tmp_path / "in.wav", tmp_path / "out.wav"
    src.write_bytes(make_wav(_pcm16(x), 11025))
    os.utime(src, (1_500_000_000, 1_545_511_181))
    s = apt.Settings()
    apt.resample_wav(None, s, str(src), str(dst), 20800)
    assert dst.read_bytes() == ow.resample_wav(src.read_bytes(), 20800, s.wav_resample_atten,
                                               s.wav_resample_delta_freq)
    assert int(os.stat(dst).st_mtime) == 1_545_511_181       # misc::write_timestamp
    with pytest.raises(apt.InternalError, match="Can't resample to 0Hz"):
        apt.resample_wav(None, s, src.read_bytes(), None, 0)
    with pytest.raises(apt.InternalError, match="Got zero samples after resampling"):
        apt.resample_wav(None, s, make_wav(np.zeros(3, np.int16), 48000), None, 4160)
    with pytest.raises(apt.RateOverflowError):
        apt.resample_wav(None, s, make_wav(np.zeros(1000, np.int16), 99371), None, 93911)
    with pytest.raises(apt.IoError):
        apt.resample_wav(None, s, str(tmp_path / "nope.wav"), str(dst), 8000)


def test_c_example_wav_to_pgm(oracle, ow, tmp_path):
    """The plain-C caller (examples/aptgpu_decode.c): WAV file in, PGM out, identical to the
    oracle's decode -> 98 % contrast -> u8 image."""
    import os
    import shutil
    import subprocess
    from oracle import image_binding as oi
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "aptgpu_decode"
    libdir = os.path.dirname(apt.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(root, "include"), "-o", str(exe),
                           os.path.join(root, "examples", "aptgpu_decode.c"), "-L", libdir, "-laptgpu",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    x = synth_apt(11025, 130, seed=12)
    wav = tmp_path / "pass.wav"
    wav.write_bytes(make_wav(_pcm16(x), 11025))
    for contrast, kind in (("percent", oi.CONTRAST_PERCENT), ("telemetry", oi.CONTRAST_TELEMETRY)):
        pgm = tmp_path / f"{contrast}.pgm"
        r = subprocess.run([str(exe), str(wav), str(pgm), contrast], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Resampling to 12480" in r.stderr and "Generating image" in r.stderr
        rows = oracle.decode(ow.load_wav(wav.read_bytes())[0], 11025, True)
        want, lo, hi = oi.process_gray(rows, kind, 0.98)
        data = pgm.read_bytes()
        header = f"P5\n2080 {rows.size // 2080}\n255\n".encode()
        assert data.startswith(header)
        assert data[len(header):] == want.tobytes()
