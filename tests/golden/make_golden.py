#!/usr/bin/env python3
"""Generates tests/golden/*.json — golden vectors for the decode() path.

PROVENANCE: the reference (Rust) cannot be built or run in this environment (no rustc/cargo,
crates not vendored) and holds no golden output of its own for decode(), so
these vectors come from the CPU oracle (oracle/apt_oracle.c, the line-by-line C restatement of
the reference, cross-checked by tests/np_model.py).  They pin the oracle and the HIP path
against regressions and make the GPU parity tests independent of the oracle build on the GPU
box.  Re-run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from noaa_apt_amd.testing.synth import synth_apt, synth_noise  # noqa: E402
from oracle import binding as oracle  # noqa: E402

CASES = [
    # name, generator, args, rate, profile, sync
    ("apt48k_std", "apt", dict(rate_hz=48000, seconds=13, seed=2), 48000, "STANDARD", True),
    ("apt48k_std_nosync", "apt", dict(rate_hz=48000, seconds=13, seed=2), 48000, "STANDARD", False),
    ("apt96k_std", "apt", dict(rate_hz=96000, seconds=12, seed=3), 96000, "STANDARD", True),
    ("apt11025_std", "apt", dict(rate_hz=11025, seconds=16, seed=1), 11025, "STANDARD", True),
    ("apt48k_fast", "apt", dict(rate_hz=48000, seconds=12, seed=5), 48000, "FAST", True),
    ("apt48k_slow", "apt", dict(rate_hz=48000, seconds=12, seed=6), 48000, "SLOW", True),
    ("noise11025_std", "noise", dict(rate_hz=11025, seconds=30.0, seed=77), 11025, "STANDARD", True),
    # the rates served by k_fused's phase-resident stage 1 (256- and 512-thread workgroups)
    ("apt44100_std", "apt", dict(rate_hz=44100, seconds=13, seed=8), 44100, "STANDARD", True),
    ("apt22050_std", "apt", dict(rate_hz=22050, seconds=14, seed=9), 22050, "STANDARD", True),
    # Settings.export_resample_filtered: fast_resampling's other decimation phase (dsp.rs:265-273) and the expanded signal
    ("apt48k_std_export_filtered", "apt", dict(rate_hz=48000, seconds=9, seed=12), 48000, "STANDARD", True, True),
    ("apt11025_std_export_filtered", "apt", dict(rate_hz=11025, seconds=8, seed=13), 11025, "STANDARD", False, True),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_input(kind, args):
    return synth_apt(**args) if kind == "apt" else synth_noise(**args)


def main():
    for name, kind, args, rate, profile, sync, *rest in CASES:
        export = bool(rest and rest[0])
        x = make_input(kind, args)
        rows, st = oracle.decode(x, rate, sync, settings=getattr(oracle, profile), want_steps=True,
                                 export_resample_filtered=export)
        img = rows.reshape(-1, 2080)
        g = {
            "name": name, "generator": kind, "args": args, "rate": rate, "profile": profile,
            "sync": sync, "input_sha256": sha(x), "n_in": int(x.size),
            "n_resample_taps": int(st["resample_filter"].size),
            "n_lowpass_taps": int(st["filter_filter"].size),
            "resample_filter_sha256": sha(st["resample_filter"]),
            "filter_filter_sha256": sha(st["filter_filter"]),
            "work_len": int(st["resampled"].size),
            "resampled_sha256": sha(st["resampled"]),
            "demodulated_sha256": sha(st["demodulated"]),
            "filtered_sha256": sha(st["filtered"]),
            "correlation_sha256": sha(st["correlation"]) if sync else None,
            "sync_pos": [int(v) for v in st["sync_pos"]] if sync else None,
            "n_rows": int(img.shape[0]),
            "rows_sha256": sha(rows),
            "row_sha256": [sha(r) for r in img[:4]],
            # a few literal values as u32 bit patterns so a human can diff them
            "row3_first8_bits": [int(v) for v in img[3, :8].view(np.uint32)] if img.shape[0] > 3 else [],
        }
        if export:  # (keys only where set: the files of the other cases stay byte-identical)
            g["export_resample_filtered"] = True
            g["n_expanded"] = int(st["expanded1"].size)
            g["expanded_sha256"] = sha(st["expanded1"])
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(g, f, indent=1)
        print(name, g["n_rows"], "rows", g["rows_sha256"][:16])


if __name__ == "__main__":
    main()
