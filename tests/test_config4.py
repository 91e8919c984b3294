"""BASELINE config 4's recording list and its sharding (bench.py --config4): host logic only."""
import importlib.util
import os

import numpy as np

from noaa_apt_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)


def test_config4_list_is_deterministic_ragged_and_within_50ppm():
    n_nom = 48000 * 900
    la, _ = bench.config4_recordings(256)
    lb, _ = bench.config4_recordings(256)
    assert la == lb and len(la) == 256
    assert len(set(la)) > 200                                   # ragged: (almost) every recording has its own length
    assert all(abs(n - n_nom) <= n_nom * 50e-6 + 1 for n in la)
    assert 256 * n_nom * 0.9999 < sum(la) < 256 * n_nom * 1.0001  # 11.06 G samples in all


def test_config4_shares_are_balanced_and_cover_every_recording_once():
    lengths, _ = bench.config4_recordings(256)
    shares = shard.assign(lengths, 8)
    assert sorted(i for s in shares for i in s) == list(range(256))
    assert all(len(s) == 32 for s in shares)
    loads = [sum(lengths[i] for i in s) for s in shares]
    assert (max(loads) - min(loads)) / max(loads) < 1e-4          # LPT on near-equal lengths: within 0.01 %


def test_config4_recordings_are_rotations_of_a_few_bases_cut_to_their_length():
    lengths, make = bench.config4_recordings(6, rate=11025, seconds=8.0, distinct=2)
    recs = [make(i) for i in range(6)]
    assert [r.size for r in recs] == lengths and all(r.dtype == np.float32 and r.flags.c_contiguous for r in recs)
    # recordings 0, 2, 4 share a base (different starts), 1, 3, 5 the other: same multiset of samples up to the cut
    assert not np.array_equal(recs[0][:1000], recs[2][:1000])
    assert abs(float(np.sort(recs[0])[1000:-1000].mean()) - float(np.sort(recs[2])[1000:-1000].mean())) < 5.0
