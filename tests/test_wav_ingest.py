"""WAV ingest on the CPU side: the oracle (wav::load_wav over hound 3.5.1's reader semantics)
against independent readers (Python's `wave`, scipy.io.wavfile, direct numpy decoding), and the
product's host-side header parser against the oracle.  No GPU needed: aptgpu_wav_parse is pure
host code."""
import io
import os
import wave

import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.wavfile import first_channel_f32, make_wav
from oracle import wav_binding as ow
from oracle.binding import OracleError

f32 = np.float32
REF_FIXTURE = "/root/reference/test/noise_48000hz.wav"


def _values(bits, n, rng):
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    v = rng.integers(lo, hi + 1, size=n, dtype=np.int64)
    v[:4] = [lo, hi, 0, -1]
    return v


CASES = [  # (name, kwargs for make_wav, bits, is_float, channels)
    ("pcm16_mono", dict(), 16, False, 1),
    ("pcm16_stereo", dict(channels=2), 16, False, 2),
    ("pcm16_5ch", dict(channels=5), 16, False, 5),
    ("pcm8_mono", dict(bits=8), 8, False, 1),
    ("pcm8_stereo", dict(bits=8, channels=2), 8, False, 2),
    ("pcm24_mono", dict(bits=24), 24, False, 1),
    ("pcm24_stereo_ex18", dict(bits=24, channels=2, fmt_len=18), 24, False, 2),
    ("pcm32_mono", dict(bits=32), 32, False, 1),
    ("pcm24_in_4", dict(bits=24, container_bytes=4, extensible=True), 24, False, 1),
    ("pcm16_fmt18", dict(fmt_len=18), 16, False, 1),
    ("pcm16_fmt40", dict(fmt_len=40), 16, False, 1),
    ("pcm16_extensible", dict(extensible=True, channels=2), 16, False, 2),
    ("float32_mono", dict(is_float=True), 32, True, 1),
    ("float32_stereo_ex18", dict(is_float=True, channels=2, fmt_len=18), 32, True, 2),
    ("float32_extensible", dict(is_float=True, extensible=True), 32, True, 1),
    ("pcm16_list_chunk", dict(extra_chunks=[(b"LIST", b"INFOsoft" * 3)]), 16, False, 1),
    ("pcm16_odd_chunk", dict(extra_chunks=[(b"junk", b"abc")]), 16, False, 1),
    ("pcm16_fact", dict(fact=True), 16, False, 1),
]


def _case(name, kw, bits, is_float, channels, frames=1001, seed=0):
    rng = np.random.default_rng(seed + len(name))
    if is_float:
        vals = (rng.standard_normal(frames * channels) * 0.3).astype(f32)
        vals[:3] = [np.float32(-0.0), np.float32(1e-41), np.float32(3.0e38)]
    else:
        vals = _values(bits, frames * channels, rng)
    return make_wav(vals, 11025, **kw), first_channel_f32(vals, channels, bits, is_float)


@pytest.mark.parametrize("name,kw,bits,is_float,channels", CASES, ids=[c[0] for c in CASES])
def test_oracle_and_parser_on_generated_files(name, kw, bits, is_float, channels):
    data, want = _case(name, kw, bits, is_float, channels)
    sig, spec = ow.load_wav(data)
    assert sig.tobytes() == want.tobytes()
    assert (spec.channels, spec.sample_rate, spec.bits_per_sample) == (channels, 11025, bits)
    assert spec.sample_format == int(is_float)
    # the product's host-side parser agrees with the oracle field by field
    ps = apt.wav_parse(data)
    for field in ("channels", "bits_per_sample", "bytes_per_sample", "sample_format", "sample_rate",
                  "data_offset", "data_len", "n_samples"):
        assert getattr(ps, field) == getattr(spec, field), field
    assert ps.n_frames == want.size


@pytest.mark.parametrize("bits,channels", [(8, 1), (16, 1), (16, 2), (24, 1), (32, 2)])
def test_oracle_vs_python_wave_module(bits, channels):
    """Files written by Python's own `wave` writer, decoded by its reader + numpy."""
    rng = np.random.default_rng(bits + channels)
    frames = 777
    vals = _values(bits, frames * channels, rng)
    from noaa_apt_amd.testing.wavfile import encode_samples
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(bits // 8)
        w.setframerate(48000)
        w.writeframes(encode_samples(vals, bits))
    data = buf.getvalue()
    with wave.open(io.BytesIO(data), "rb") as r:
        raw = r.readframes(r.getnframes())
        assert (r.getnchannels(), r.getsampwidth(), r.getframerate()) == (channels, bits // 8, 48000)
    if bits == 8:
        dec = np.frombuffer(raw, np.uint8).astype(np.int64) - 128
    elif bits == 24:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int64)
        dec = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        dec = np.where(dec >= 1 << 23, dec - (1 << 24), dec)
    else:
        dec = np.frombuffer(raw, {16: "<i2", 32: "<i4"}[bits]).astype(np.int64)
    sig, spec = ow.load_wav(data)
    assert sig.tobytes() == dec[::channels].astype(f32).tobytes()
    assert spec.sample_rate == 48000


def test_oracle_vs_scipy_float_and_int():
    wavfile = pytest.importorskip("scipy.io.wavfile")
    rng = np.random.default_rng(7)
    for dtype, channels in ((np.float32, 1), (np.float32, 2), (np.int16, 2), (np.int32, 1), (np.uint8, 1)):
        shape = (500, channels) if channels > 1 else (500,)
        if dtype == np.float32:
            a = rng.standard_normal(shape).astype(np.float32)
        else:
            info = np.iinfo(dtype)
            a = rng.integers(info.min, info.max + 1, size=shape, dtype=np.int64).astype(dtype)
        buf = io.BytesIO()
        wavfile.write(buf, 22050, a)
        sig, spec = ow.load_wav(buf.getvalue())
        ch0 = a[:, 0] if channels > 1 else a
        want = (ch0.astype(np.int64) - 128).astype(f32) if dtype == np.uint8 else ch0.astype(f32)
        assert sig.tobytes() == want.tobytes(), (dtype, channels)
        assert spec.sample_rate == 22050 and spec.channels == channels


@pytest.mark.skipif(not os.path.exists(REF_FIXTURE), reason="reference tree not present")
def test_reference_fixture():
    """test/noise_48000hz.wav of the reference (despite its name: 11025 Hz, mono, 16 bit)."""
    data = open(REF_FIXTURE, "rb").read()
    with wave.open(io.BytesIO(data), "rb") as r:
        raw = r.readframes(r.getnframes())
        meta = (r.getnchannels(), r.getsampwidth(), r.getframerate(), r.getnframes())
    assert meta == (1, 2, 11025, 330745)
    sig, spec = ow.load_wav(data)
    assert sig.tobytes() == np.frombuffer(raw, "<i2").astype(f32).tobytes()
    ps = apt.wav_parse(data)
    assert (ps.channels, ps.bits_per_sample, ps.sample_rate, ps.n_frames, ps.codec) == (1, 16, 11025, 330745, 1)


def test_empty_data_chunk():
    data = make_wav(np.zeros(0, np.int16), 8000)
    sig, spec = ow.load_wav(data)
    assert sig.size == 0 and apt.wav_parse(data).n_frames == 0


ERRORS = [  # (name, file image, oracle code, message fragment)
    ("no_riff", b"RIFX" + make_wav([1, 2], 8000)[4:], ow.ERR_WAV_OPEN, "no RIFF tag found"),
    ("no_wave", make_wav([1, 2], 8000)[:8] + b"WAVX" + make_wav([1, 2], 8000)[12:], ow.ERR_WAV_OPEN,
     "no WAVE tag found"),
    ("truncated_data", make_wav(np.arange(100), 8000, truncate=44 + 150), ow.ERR_IO, "Failed to read enough bytes"),
    ("truncated_header", make_wav(np.arange(100), 8000, truncate=30), ow.ERR_IO, "Failed to read enough bytes"),
    ("no_data_chunk", make_wav(np.arange(4), 8000)[:36], ow.ERR_IO, "Failed to read enough bytes"),
    ("odd_data_len", make_wav(np.arange(4), 8000, data_len_override=7), ow.ERR_WAV_OPEN,
     "data chunk length is not a multiple of sample size"),
    ("partial_frame", make_wav(np.arange(5), 8000, channels=2), ow.ERR_WAV_OPEN, "invalid data chunk length"),
    ("adpcm", make_wav(np.arange(4), 8000, format_tag=2), ow.ERR_WAV_OPEN, "not supported"),
    ("mulaw", make_wav(np.arange(4), 8000, format_tag=7), ow.ERR_WAV_OPEN, "not supported"),
    ("float64", make_wav(np.zeros(4), 8000, is_float=True, bits=64, container_bytes=8, format_tag=3),
     ow.ERR_WAV_OPEN, "bits per sample is not 32"),
    ("bad_byte_rate", make_wav(np.arange(4), 8000, byte_rate_override=1234), ow.ERR_WAV_OPEN,
     "inconsistent fmt chunk"),
    ("zero_channels", make_wav(np.arange(4), 8000)[:22] + b"\x00\x00" + make_wav(np.arange(4), 8000)[24:],
     ow.ERR_WAV_OPEN, "file contains zero channels"),
    ("pcm_fmt_20", make_wav(np.arange(4), 8000)[:16] + b"\x14\x00\x00\x00" + make_wav(np.arange(4), 8000)[20:36] +
     b"\0\0\0\0" + make_wav(np.arange(4), 8000)[36:], ow.ERR_WAV_OPEN, "unexpected fmt chunk size"),
]


@pytest.mark.parametrize("name,data,code,frag", ERRORS, ids=[e[0] for e in ERRORS])
def test_errors_oracle_and_parser_agree(name, data, code, frag):
    with pytest.raises(OracleError) as eo:
        ow.load_wav(data)
    assert eo.value.code == code and frag in str(eo.value), str(eo.value)
    want_cls = {ow.ERR_WAV_OPEN: apt.WavOpenError, ow.ERR_IO: apt.IoError}[code]
    with pytest.raises(want_cls) as ep:
        apt.wav_parse(data)
    assert str(ep.value) == str(eo.value)


def test_missing_fmt_chunk():
    good = make_wav(np.arange(4), 8000)
    data = good[:12] + good[36:]  # RIFF header + data chunk only
    for fn, exc in ((ow.load_wav, OracleError), (apt.wav_parse, apt.WavOpenError)):
        with pytest.raises(exc, match="missing fmt chunk"):
            fn(data)


# ------------------------------------------------------------------ write_wav (oracle)
def test_write_wav_oracle_vs_numpy_and_wave_reader():
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(5001) * 900).astype(f32)
    x[7] = np.nan
    data = ow.write_wav_i16(x, 6000)
    with wave.open(io.BytesIO(data), "rb") as r:
        assert (r.getnchannels(), r.getsampwidth(), r.getframerate(), r.getnframes()) == (1, 2, 6000, 5001)
        got = np.frombuffer(r.readframes(5001), "<i2")
    mx = np.nanmax(x)  # x[0] is not NaN, so get_max skips the NaN
    with np.errstate(invalid="ignore"):
        v = x / f32(mx) * f32(32767.)
    want = np.where(np.isnan(v), 0, np.clip(np.trunc(v), -32768, 32767)).astype(np.int16)
    assert np.array_equal(got, want)
    assert len(data) == 44 + 2 * 5001 and data[:4] == b"RIFF" and data[36:40] == b"data"
    # the product's parser reads the oracle's writer back
    spec = apt.wav_parse(data)
    assert (spec.channels, spec.bits_per_sample, spec.sample_rate, spec.n_frames) == (1, 16, 6000, 5001)


def test_write_wav_negative_peak_saturates():
    x = np.array([1.0, -3.0, 0.5, -0.99999], f32)  # max = 1: -3 -> -98301 -> saturates
    got = np.frombuffer(ow.write_wav_i16(x, 8000)[44:], "<i2")
    assert got.tolist() == [32767, -32768, 16383, -32766]


def test_write_wav_empty():
    with pytest.raises(OracleError, match="maximum of a zero length vector"):
        ow.write_wav_i16(np.zeros(0, f32), 8000)
