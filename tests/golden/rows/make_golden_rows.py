#!/usr/bin/env python3
"""Generates tests/golden/rows/*.json — golden vectors for the rows either side of decode():
WAV ingest, contrast limits / u8 image / telemetry, the WAV->WAV resample tool.

PROVENANCE: as for tests/golden/make_golden.py — the reference cannot be run here, so these come
from the CPU oracle (oracle/apt_oracle_{image,wav}.c), which the reference's own unit tests pin
(tests/test_oracle_image_kats.py, tests/test_wav_ingest.py).  They freeze the oracle and let the
GPU tests run without it.  Re-run:  python tests/golden/rows/make_golden_rows.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))

from noaa_apt_amd.testing.synth import synth_apt  # noqa: E402
from noaa_apt_amd.testing.wavfile import make_wav  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from oracle import image_binding as oi  # noqa: E402
from oracle import wav_binding as ow  # noqa: E402


def sha(a):
    return hashlib.sha256(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(v):
    return [int(b) for b in np.asarray(v, np.float32).ravel().view(np.uint32)]


def wav_cases():
    """(name, make_wav kwargs, integer bits or None for float, channels, frames, seed)"""
    return [
        ("pcm16_mono", dict(), 16, 1, 4001, 1),
        ("pcm16_stereo", dict(channels=2), 16, 2, 3000, 2),
        ("pcm8_mono", dict(bits=8), 8, 1, 2500, 3),
        ("pcm24_stereo", dict(bits=24, channels=2, fmt_len=18), 24, 2, 1501, 4),
        ("pcm32_mono", dict(bits=32), 32, 1, 999, 5),
        ("pcm24_in_4_ext", dict(bits=24, container_bytes=4, extensible=True), 24, 1, 777, 6),
        ("float32_stereo", dict(is_float=True, channels=2), None, 2, 2048, 7),
        ("pcm16_odd_offset", dict(extra_chunks=[(b"junk", b"abc")]), 16, 1, 1234, 8),
    ]


def wav_file(kw, ibits, channels, frames, seed):
    rng = np.random.default_rng(seed)
    if ibits is None:
        vals = (rng.standard_normal(frames * channels) * 0.25).astype(np.float32)
    else:
        lo, hi = -(1 << (ibits - 1)), (1 << (ibits - 1)) - 1
        vals = rng.integers(lo, hi + 1, size=frames * channels, dtype=np.int64)
        vals[:4] = [lo, hi, 0, -1]
    return make_wav(vals, 11025, **kw)


def main():
    # ---- image stage on a decoded pass
    x = synth_apt(48000, 120, seed=9)
    rows = oracle.decode(x, 48000, True)
    g = {"input": dict(rate_hz=48000, seconds=120, seed=9), "rows_sha256": sha(rows), "n_rows": rows.size // 2080,
         "contrast": {}}
    for name, kind in (("telemetry", oi.CONTRAST_TELEMETRY), ("percent", oi.CONTRAST_PERCENT), ("minmax", oi.CONTRAST_MINMAX)):
        img, lo, hi = oi.process_gray(rows, kind, 0.98)
        g["contrast"][name] = {"image_sha256": sha(img), "low_bits": bits(lo)[0], "high_bits": bits(hi)[0]}
    t = oi.read_telemetry(rows)
    g["telemetry"] = {"row": t.row, "quality_bits": bits(t.quality)[0], "values_a_bits": bits(t.values_a),
                      "values_b_bits": bits(t.values_b), "channel_a": t.get_channel_name("A"),
                      "channel_b": t.get_channel_name("B"),
                      "steps_sha256": {k: sha(v) for k, v in sorted(t.steps.items())}}
    json.dump(g, open(os.path.join(HERE, "image_apt48k_120s.json"), "w"), indent=1)
    print("image", g["n_rows"], "rows")

    # ---- WAV ingest
    out = []
    for name, kw, ibits, channels, frames, seed in wav_cases():
        data = wav_file(kw, ibits, channels, frames, seed)
        sig, spec = ow.load_wav(data)
        out.append({"name": name, "file_sha256": sha(data), "signal_sha256": sha(sig), "n_frames": int(sig.size),
                    "channels": spec.channels, "bits_per_sample": spec.bits_per_sample,
                    "sample_rate": spec.sample_rate, "data_offset": int(spec.data_offset),
                    "first8_bits": bits(sig[:8])})
    json.dump(out, open(os.path.join(HERE, "wav_ingest.json"), "w"), indent=1)
    print("wav", len(out), "files")

    # ---- resample tool
    tool = []
    for in_rate, out_rate, seed in ((11025, 48000, 11), (11025, 6000, 12), (11025, 3675, 13), (48000, 11025, 14)):
        xs = synth_apt(in_rate, 3, seed=seed).astype(np.int16)
        data = make_wav(xs, in_rate)
        res = ow.resample_wav(data, out_rate, 40.0, 0.1)
        tool.append({"in_rate": in_rate, "out_rate": out_rate, "seed": seed, "input_sha256": sha(data),
                     "output_sha256": sha(res), "output_bytes": len(res)})
    json.dump(tool, open(os.path.join(HERE, "resample_tool.json"), "w"), indent=1)
    print("resample tool", len(tool), "cases")


if __name__ == "__main__":
    main()
