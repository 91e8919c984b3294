"""Golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them without needing the
oracle at all (hashes of every stage + sync positions + rows)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from noaa_apt_amd.testing.synth import synth_apt, synth_noise

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(path):
    g = json.load(open(path))
    x = synth_apt(**g["args"]) if g["generator"] == "apt" else synth_noise(**g["args"])
    assert sha(x) == g["input_sha256"], "synthetic generator drifted: regenerate the goldens"
    return g, x


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
def test_oracle_matches_golden(oracle, path):
    g, x = load(path)
    export = bool(g.get("export_resample_filtered"))
    rows, st = oracle.decode(x, g["rate"], g["sync"], settings=getattr(oracle, g["profile"]), want_steps=True,
                             export_resample_filtered=export)
    if export:
        assert st["expanded1"].size == g["n_expanded"] and sha(st["expanded1"]) == g["expanded_sha256"]
    assert st["resample_filter"].size == g["n_resample_taps"]
    assert sha(st["resample_filter"]) == g["resample_filter_sha256"]
    assert sha(st["filter_filter"]) == g["filter_filter_sha256"]
    assert sha(st["resampled"]) == g["resampled_sha256"]
    assert sha(st["demodulated"]) == g["demodulated_sha256"]
    assert sha(st["filtered"]) == g["filtered_sha256"]
    if g["sync"]:
        assert sha(st["correlation"]) == g["correlation_sha256"]
        assert [int(v) for v in st["sync_pos"]] == g["sync_pos"]
    assert sha(rows) == g["rows_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
def test_gpu_matches_golden(path):
    import noaa_apt_amd as apt
    g, x = load(path)
    s = apt.Settings.profile(g["profile"].lower())
    s.export_resample_filtered = bool(g.get("export_resample_filtered"))
    # every exported step, hashed
    got = {}
    c = apt.Context(step_callback=lambda i, v, d, r: got.setdefault(i, []).append(d), device=0)
    rows = apt.decode(c, apt.Settings(**{**s.__dict__, "export_wav": True}), x, apt.Rate.hz(g["rate"]), g["sync"])
    assert sha(got["resample_filter"][0]) == g["resample_filter_sha256"]
    assert sha(got["resample_decimated"][0]) == g["resampled_sha256"]
    if s.export_resample_filtered:
        assert got["resample_filtered"][0].size == g["n_expanded"] and sha(got["resample_filtered"][0]) == g["expanded_sha256"]
    assert sha(got["demodulation_result"][0]) == g["demodulated_sha256"]
    assert sha(got["filter_result"][0]) == g["filtered_sha256"]
    if g["sync"]:
        assert sha(got["sync_correlation"][0]) == g["correlation_sha256"]
    assert sha(rows) == g["rows_sha256"]
    # and the fast path (fused kernels where specialised, parallel picker), no steps
    rows2, st = apt.decode(apt.Context(device=0), s, x, apt.Rate.hz(g["rate"]), g["sync"], return_stats=True)
    assert sha(rows2) == g["rows_sha256"]
    assert st.n_rows == g["n_rows"]
    img = rows2.reshape(-1, 2080)
    assert [sha(r) for r in img[:4]] == g["row_sha256"]
    if g["sync"]:
        pos = apt.find_sync(apt.Context(device=0), got["filter_result"][0], apt.Rate.hz(s.work_rate))
        assert [int(v) for v in pos] == g["sync_pos"]
