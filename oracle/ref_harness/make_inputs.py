#!/usr/bin/env python3
"""Inputs for the reference run: oracle/_ref/in/<case>.wav (mono PCM16, canonical 44-byte header)."""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from noaa_apt_amd.testing.synth import synth_apt  # noqa: E402
from noaa_apt_amd.testing.wavfile import make_wav  # noqa: E402

CASES = {  # name -> (rate, seconds, seed)
    "apt48k_14s": (48000, 14, 2),
    "apt11025_20s": (11025, 20, 1),
}


def main():
    out = os.path.join(ROOT, "oracle", "_ref", "in")
    os.makedirs(out, exist_ok=True)
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "reference_fixture", "noise_48000hz.wav"), os.path.join(out, "noise_fixture.wav"))
    for name, (rate, seconds, seed) in CASES.items():
        x = synth_apt(rate, seconds, seed)  # integer-valued f32 within int16 range
        open(os.path.join(out, name + ".wav"), "wb").write(make_wav(x.astype(np.int16), rate))
    print("inputs in", out)


if __name__ == "__main__":
    main()
