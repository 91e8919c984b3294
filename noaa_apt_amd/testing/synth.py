"""Deterministic synthetic APT recordings (SURVEY.md §8(d) "Synthetic APT generator").

Test and bench tooling only.  Builds an APT frame stream following the line
layout of /root/reference/src/decode.rs:16-35 (sync | space | image | telemetry,
two channels of 1040 px), zero-order-holds the 4160 px/s stream to the WAV rate,
AM-modulates the 2400 Hz sub-carrier, adds Gaussian noise and rounds to int16,
then converts to f32 WITHOUT scaling — exactly what wav::load_wav hands to
decode() (/root/reference/src/wav.rs:30-51: `*x as f32`).
"""
import numpy as np

PX_PER_ROW = 2080
PX_PER_CHANNEL = 1040
PX_SYNC_FRAME = 39
PX_SPACE_DATA = 47
PX_CHANNEL_IMAGE_DATA = 909
PX_TELEMETRY_DATA = 45
FINAL_RATE = 4160
CARRIER_HZ = 2400.0


def _sync_a():
    # 2 low, 7 x [2 low, 2 high], 8 low = 38 px (matches generate_sync_frame,
    # decode.rs:188-198), padded with 1 low px to PX_SYNC_FRAME
    px = [0, 0]
    for _ in range(7):
        px += [0, 0, 255, 255]
    px += [0] * 8
    px += [0]
    return np.array(px, dtype=np.float32)


def _sync_b():
    # sync B: 7 x [3 high, 2 low] px preceded by 4 low px = 39
    px = [0, 0, 0, 0]
    for _ in range(7):
        px += [255, 255, 255, 0, 0]
    return np.array(px, dtype=np.float32)


_WEDGES = np.array([31, 63, 95, 127, 159, 191, 223, 255, 0, 80, 120, 60, 200, 140, 100, 180],
                   dtype=np.float32)


def make_image(n_rows, seed):
    """n_rows x 2080 u8-valued (float32) APT frame rows."""
    rng = np.random.default_rng(seed)
    img = np.zeros((n_rows, PX_PER_ROW), dtype=np.float32)
    r = np.arange(n_rows, dtype=np.float32)[:, None]
    for ch in range(2):
        base = ch * PX_PER_CHANNEL
        img[:, base:base + PX_SYNC_FRAME] = (_sync_a() if ch == 0 else _sync_b())[None, :]
        o = base + PX_SYNC_FRAME
        img[:, o:o + PX_SPACE_DATA] = 0.0 if ch == 0 else 255.0
        o += PX_SPACE_DATA
        c = np.arange(PX_CHANNEL_IMAGE_DATA, dtype=np.float32)[None, :]
        # smooth 2-D gradient + seeded band-limited texture
        lo = rng.standard_normal((n_rows // 16 + 2, PX_CHANNEL_IMAGE_DATA // 16 + 2)).astype(np.float32)
        tex = np.kron(lo, np.ones((16, 16), dtype=np.float32))[:n_rows, :PX_CHANNEL_IMAGE_DATA]
        body = 128.0 + 60.0 * np.sin(2 * np.pi * (c / 400.0 + r / 300.0 + 0.37 * ch)) + 35.0 * tex
        img[:, o:o + PX_CHANNEL_IMAGE_DATA] = np.clip(body, 0.0, 255.0)
        o += PX_CHANNEL_IMAGE_DATA
        wedge = _WEDGES[((np.arange(n_rows) // 8) % 16)]
        img[:, o:o + PX_TELEMETRY_DATA] = wedge[:, None]
    return np.floor(img)


def synth_apt(rate_hz, seconds, seed, *, noise_sigma=400.0, amplitude=20000.0, ppm=0.0,
              start_px=None, phase=None, chunk=1 << 22, image=None):
    """Return the signal (f32, int16-valued).  Deterministic in (rate, seconds, seed, kwargs).
    image: rows x 2080 pixel values 0..255 to transmit instead of make_image()'s frame (repeated as often as the
    duration needs) — e.g. rows of an image the reference itself decoded (tests/test_reference_image_structure.py)."""
    n = int(round(rate_hz * seconds))
    rng = np.random.default_rng(seed)
    if start_px is None:
        start_px = float(rng.integers(0, PX_PER_ROW))
    if phase is None:
        phase = float(rng.uniform(0, 2 * np.pi))
    px_per_sample = FINAL_RATE / (rate_hz * (1.0 + ppm * 1e-6))
    n_rows = int(np.ceil((n * px_per_sample + start_px) / PX_PER_ROW)) + 2
    if image is None:
        img = make_image(n_rows, seed + 7919).reshape(-1)
    else:
        src = np.asarray(image, dtype=np.float32)
        assert src.ndim == 2 and src.shape[1] == PX_PER_ROW, src.shape
        img = np.tile(src, ((n_rows + src.shape[0] - 1) // src.shape[0], 1))[:n_rows].reshape(-1)
    out = np.empty(n, dtype=np.float32)
    w = 2.0 * np.pi * CARRIER_HZ / rate_hz
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(s, e, dtype=np.float64)
        px = np.floor(idx * px_per_sample + start_px).astype(np.int64)
        v = img[px]
        env = 0.05 + 0.95 * v / 255.0
        sig = amplitude * env * np.cos(w * idx + phase)
        sig += noise_sigma * rng.standard_normal(e - s)
        out[s:e] = np.clip(np.rint(sig), -32768, 32767).astype(np.float32)
    return out


def synth_noise(rate_hz, seconds, seed, sigma=3000.0):
    """int16-valued white noise as f32 (stand-in for test/noise_48000hz.wav, which is
    11025 Hz / 16-bit / mono / 30 s of noise and is not shipped to the GPU box)."""
    n = int(round(rate_hz * seconds))
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(sigma * rng.standard_normal(n)), -32768, 32767).astype(np.float32)
