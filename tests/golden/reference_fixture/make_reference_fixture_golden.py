#!/usr/bin/env python3
"""Freeze the oracle's answers for the reference's one real WAV fixture (test/noise_48000hz.wav,
copied next to this script — despite its name 11 025 Hz, mono, 16 bit, 330 745 frames):
the commands of /root/reference/test/test.sh:46,50-51

    noaa-apt noise_48000hz.wav -o decoded_noise.png        -> decode() rows (before the PNG stage)
    noaa-apt noise_48000hz.wav -r 80000 -o upsampled.wav   -> resample tool output file
    noaa-apt noise_48000hz.wav -r 11025 -o downsampled.wav -> resample tool output file

Writes reference_fixture.json next to this script (sha256 of the raw little-endian bytes + sizes).  The
oracle is the C restatement of the reference (oracle/); the reference itself (Rust) cannot be run
in this image, so these hashes pin "what the oracle said on the day they were frozen", not the
reference binary — see DESIGN.md §3.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))

from oracle import binding as oracle  # noqa: E402
from oracle import wav_binding as ow  # noqa: E402

ATTEN, DELTA = 40.0, 0.1  # standard profile: wav_resample_atten / wav_resample_delta_freq (default_settings.toml:115-116)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    data = open(os.path.join(HERE, "noise_48000hz.wav"), "rb").read()
    sig, spec = ow.load_wav(data)
    out = {"file_sha256": sha(data), "frames": int(sig.size), "sample_rate": int(spec.sample_rate)}
    for sync in (True, False):
        rows, st = oracle.decode(sig, spec.sample_rate, sync, want_steps=True)
        out[f"decode_sync_{int(sync)}"] = {"rows": int(rows.size // 2080), "sha256": sha(rows.astype("<f4").tobytes()),
                                          "n_sync": int(st["sync_pos"].size),
                                          "sync_pos_sha256": sha(st["sync_pos"].astype("<u8").tobytes())}
    for rate in (80000, 11025):
        f = ow.resample_wav(data, rate, ATTEN, DELTA)
        out[f"resample_{rate}"] = {"bytes": len(f), "sha256": sha(f)}
    json.dump(out, open(os.path.join(HERE, "reference_fixture.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
