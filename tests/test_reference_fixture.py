"""The reference's one real WAV fixture (test/noise_48000hz.wav — despite its name 11 025 Hz, mono,
16 bit, 330 745 frames; a copy sits in tests/golden/reference_fixture/) through the commands of the reference's
test/test.sh:46,50-51: decode, resample to 80 000 Hz, resample to 11 025 Hz.

The CPU tests check the oracle against hashes frozen in tests/golden/reference_fixture/reference_fixture.json
(made by make_reference_fixture_golden.py next to it); the GPU tests check the HIP path, through
the C ABI on the file image exactly as it is on disk, against the oracle AND the frozen hashes.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import noaa_apt_amd as apt

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "reference_fixture", "noise_48000hz.wav")
GOLDEN = json.load(open(os.path.join(HERE, "golden", "reference_fixture", "reference_fixture.json")))
f32 = np.float32


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def data():
    d = open(FIXTURE, "rb").read()
    assert sha(d) == GOLDEN["file_sha256"]
    return d


@pytest.fixture(scope="module")
def ow():
    from oracle import wav_binding
    wav_binding.lib()
    return wav_binding


def test_fixture_is_the_references_file(data):
    ref = "/root/reference/test/noise_48000hz.wav"
    if os.path.exists(ref):  # (the reference tree only exists in the build container)
        assert open(ref, "rb").read() == data


@pytest.mark.parametrize("sync", [True, False])
def test_oracle_decode_matches_frozen_hash(oracle, ow, data, sync):
    sig, spec = ow.load_wav(data)
    assert (sig.size, spec.sample_rate) == (GOLDEN["frames"], GOLDEN["sample_rate"]) == (330745, 11025)
    rows, st = oracle.decode(sig, spec.sample_rate, sync, want_steps=True)
    g = GOLDEN[f"decode_sync_{int(sync)}"]
    assert rows.size == g["rows"] * 2080 and sha(rows.astype("<f4").tobytes()) == g["sha256"]
    assert st["sync_pos"].size == g["n_sync"] and sha(st["sync_pos"].astype("<u8").tobytes()) == g["sync_pos_sha256"]


@pytest.mark.parametrize("rate", [80000, 11025])
def test_oracle_resample_matches_frozen_hash(ow, data, rate):
    s = apt.Settings()
    out = ow.resample_wav(data, rate, s.wav_resample_atten, s.wav_resample_delta_freq)
    g = GOLDEN[f"resample_{rate}"]
    assert len(out) == g["bytes"] and sha(out) == g["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("sync", [True, False])
@pytest.mark.parametrize("mode", ["strict", "generic"])
def test_gpu_decode_wav(oracle, ow, data, sync, mode):
    """`noaa-apt noise_48000hz.wav -o decoded_noise.png` up to the pixel rows (test.sh:46)."""
    ctx = apt.Context(device=0, mode=apt.MODE_GENERIC if mode == "generic" else apt.MODE_STRICT)
    rows, st = apt.decode_wav(ctx, apt.Settings(), data, sync, return_stats=True)
    sig, spec = ow.load_wav(data)
    want = oracle.decode(sig, spec.sample_rate, sync)
    assert rows.size == want.size and np.array_equal(rows.view(np.uint32), want.view(np.uint32))
    g = GOLDEN[f"decode_sync_{int(sync)}"]
    assert sha(rows.astype("<f4").tobytes()) == g["sha256"] and st.n_sync == g["n_sync"]
    # the f32 Signal route (load_wav, then decode) gives the same rows
    sig_gpu, rate = apt.load(data, ctx)
    assert rate.get_hz() == 11025 and np.array_equal(sig_gpu.view(np.uint32), sig.view(np.uint32))
    rows2 = apt.decode(ctx, apt.Settings(), sig_gpu, rate, sync)
    assert np.array_equal(rows2.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_gpu_decode_wav_fast_mode_within_tolerance(oracle, ow, data):
    """APTGPU_MODE_FAST on the fixture (11 025 Hz: phase-resident stage 1 + the fast work-rate stages,
    PCM16 payload straight from the file image): the tolerance of SURVEY.md §8(d) — same rows, and (a
    moved sync position would move a whole row) every pixel within 1e-4 of full scale."""
    rows, st = apt.decode_wav(apt.Context(device=0, mode=apt.MODE_FAST), apt.Settings(), data, True,
                              return_stats=True)
    sig, spec = ow.load_wav(data)
    want = oracle.decode(sig, spec.sample_rate, True)
    g = GOLDEN["decode_sync_1"]
    assert st.fused == 4 and st.n_sync == g["n_sync"] and rows.size == want.size == g["rows"] * 2080
    err = float(np.max(np.abs(rows - want))) / float(np.max(np.abs(want)))
    assert 0 < err <= 1e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [80000, 11025])
def test_gpu_resample_wav(ow, data, rate):
    """`noaa-apt noise_48000hz.wav -r 80000 / -r 11025` (test.sh:50-51)."""
    s = apt.Settings()
    got = apt.resample_wav(apt.Context(device=0), s, data, "out.wav", rate)
    want = ow.resample_wav(data, rate, s.wav_resample_atten, s.wav_resample_delta_freq)
    assert got == want
    g = GOLDEN[f"resample_{rate}"]
    assert len(got) == g["bytes"] and sha(got) == g["sha256"]
