#!/usr/bin/env python3
"""Cuts a band of rows out of the reference's only OUTPUT artefact, /root/reference/docs/examples/argentina.png
(2080 x 1619, 8-bit gray: a decode of a real NOAA pass by the reference itself; its input WAV is not in the repository),
into tests/golden/reference_image/argentina_rows.npy.  Run here (the reference is not on the GPU box):

    python tests/golden/reference_image/make_reference_image_rows.py

The rows are used by tests/test_reference_image_structure.py: a recording synthesised FROM them must decode to rows
whose sync A / sync B / telemetry columns sit where the reference's image has them (decode.rs:16-35)."""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/docs/examples/argentina.png"
ROW0, ROWS = 700, 96
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    img = np.asarray(Image.open(SRC))
    assert img.shape == (1619, 2080) and img.dtype == np.uint8, (img.shape, img.dtype)
    band = np.ascontiguousarray(img[ROW0:ROW0 + ROWS])
    np.save(os.path.join(HERE, "argentina_rows.npy"), band)
    print("rows", ROW0, ROW0 + ROWS, "mean", float(band.mean()))
