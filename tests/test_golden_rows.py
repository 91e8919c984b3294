"""Golden vectors for the rows either side of decode() (tests/golden/rows/*.json, made by
make_golden_rows.py from the oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path
reproduces them without the oracle."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

from noaa_apt_amd.testing.synth import synth_apt
from noaa_apt_amd.testing.wavfile import make_wav

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = os.path.join(HERE, "golden", "rows")
_spec = importlib.util.spec_from_file_location("make_golden_rows", os.path.join(ROWS, "make_golden_rows.py"))


def sha(a):
    return hashlib.sha256(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(v):
    return [int(b) for b in np.asarray(v, np.float32).ravel().view(np.uint32)]


def _gen():
    m = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(m)  # imports the oracle bindings lazily (not its lib)
    return m


@pytest.fixture(scope="module")
def image_golden():
    return json.load(open(os.path.join(ROWS, "image_apt48k_120s.json")))


def test_oracle_image_golden(oracle, image_golden):
    from oracle import image_binding as oi
    g = image_golden
    rows = oracle.decode(synth_apt(**g["input"]), 48000, True)
    assert sha(rows) == g["rows_sha256"]
    for name, kind in (("telemetry", 0), ("percent", 1), ("minmax", 2)):
        img, lo, hi = oi.process_gray(rows, kind, 0.98)
        c = g["contrast"][name]
        assert (sha(img), bits(lo)[0], bits(hi)[0]) == (c["image_sha256"], c["low_bits"], c["high_bits"]), name
    t = oi.read_telemetry(rows)
    assert (t.row, bits(t.values_a), bits(t.values_b)) == (g["telemetry"]["row"], g["telemetry"]["values_a_bits"],
                                                          g["telemetry"]["values_b_bits"])


def test_oracle_wav_and_tool_golden():
    from oracle import wav_binding as ow
    gen = _gen()
    golden = {e["name"]: e for e in json.load(open(os.path.join(ROWS, "wav_ingest.json")))}
    for name, kw, ibits, channels, frames, seed in gen.wav_cases():
        data = gen.wav_file(kw, ibits, channels, frames, seed)
        e = golden[name]
        assert sha(data) == e["file_sha256"], "WAV generator drifted: regenerate the goldens"
        sig, spec = ow.load_wav(data)
        assert (sha(sig), sig.size, spec.data_offset) == (e["signal_sha256"], e["n_frames"], e["data_offset"]), name
    for e in json.load(open(os.path.join(ROWS, "resample_tool.json"))):
        data = make_wav(synth_apt(e["in_rate"], 3, seed=e["seed"]).astype(np.int16), e["in_rate"])
        assert sha(data) == e["input_sha256"]
        assert sha(ow.resample_wav(data, e["out_rate"], 40.0, 0.1)) == e["output_sha256"]


@pytest.mark.gpu
def test_gpu_image_golden(image_golden):
    import noaa_apt_amd as apt
    g = image_golden
    rows = apt.decode(apt.Context(device=0), apt.Settings(), synth_apt(**g["input"]), apt.Rate.hz(48000), True)
    assert sha(rows) == g["rows_sha256"]
    for name, ca in (("telemetry", apt.Contrast.TELEMETRY), ("percent", apt.Contrast.Percent(0.98)),
                     ("minmax", apt.Contrast.MINMAX)):
        img, info = apt.process(None, rows, ca, return_info=True)
        c = g["contrast"][name]
        assert (sha(img), bits(info.low)[0], bits(info.high)[0]) == (c["image_sha256"], c["low_bits"], c["high_bits"]), name
    steps = {}
    t = apt.read_telemetry(apt.Context(step_callback=lambda i, v, d, r: steps.__setitem__(i, d)), rows)
    gt = g["telemetry"]
    assert (t.row, bits(t.quality)[0], bits(t.values_a), bits(t.values_b)) == \
        (gt["row"], gt["quality_bits"], gt["values_a_bits"], gt["values_b_bits"])
    assert (t.get_channel_name("A"), t.get_channel_name("B")) == (gt["channel_a"], gt["channel_b"])
    assert {k: sha(v) for k, v in steps.items()} == gt["steps_sha256"]


@pytest.mark.gpu
def test_gpu_wav_and_tool_golden():
    import noaa_apt_amd as apt
    gen = _gen()
    golden = {e["name"]: e for e in json.load(open(os.path.join(ROWS, "wav_ingest.json")))}
    for name, kw, ibits, channels, frames, seed in gen.wav_cases():
        data = gen.wav_file(kw, ibits, channels, frames, seed)
        e = golden[name]
        assert sha(data) == e["file_sha256"]
        sig, rate, spec = apt.load(data, return_spec=True)
        assert (sha(sig), sig.size, rate.get_hz(), spec.channels, spec.bits_per_sample, spec.data_offset) == \
            (e["signal_sha256"], e["n_frames"], e["sample_rate"], e["channels"], e["bits_per_sample"], e["data_offset"]), name
        assert bits(sig[:8]) == e["first8_bits"]
    s = apt.Settings()
    for e in json.load(open(os.path.join(ROWS, "resample_tool.json"))):
        data = make_wav(synth_apt(e["in_rate"], 3, seed=e["seed"]).astype(np.int16), e["in_rate"])
        out = apt.resample_wav(None, s, data, None, e["out_rate"])
        assert (sha(out), len(out)) == (e["output_sha256"], e["output_bytes"]), (e["in_rate"], e["out_rate"])
