"""Multi-process (gloo, world_size 2, CPU) test of the N>1 path: recordings shard across ranks
with no data-path collective; only a barrier and scalar reductions cross ranks.  Each rank runs the host side of
its shard's decodes through the product's C ABI (tap design, geometry); the kernels themselves need a GPU
(tests/test_gpu_batch.py: two processes on one device)."""
import os
import socket

import numpy as np
import pytest

from noaa_apt_amd import shard


def test_assign_partitions_every_recording_once():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        lengths = rng.integers(1_000_000, 50_000_000, size=37).tolist()
        parts = shard.assign(lengths, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(37))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lengths)  # LPT bound
    # BASELINE config 4: 256 equal recordings over 8 GPUs -> 32 each
    parts = shard.assign([43_200_000] * 256, 8)
    assert [len(p) for p in parts] == [32] * 8
    assert shard.assign([], 4) == [[], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _plan_digest(rate, n_samples):
    """What a rank prepares on the host for one recording of its shard, through the PRODUCT's CPU-callable pieces
    (aptgpu_filter_resample + aptgpu_filter_design, include/aptgpu.h — the designs decode() makes, decode.rs:65-76,
    95-100) and the geometry decode() derives from them (dsp.rs:73-75, 226-277): a hash of the resampling taps, the
    low-pass taps, (l, m, work samples, rows)."""
    import hashlib
    import noaa_apt_amd as apt
    s = apt.Settings()
    in_rate = apt.Rate.hz(rate)
    g = int(np.gcd(rate, s.work_rate))
    l, m = s.work_rate // g, rate // g
    f = apt.LowpassDcRemoval(apt.Freq.hz(s.resample_cutout, in_rate), s.resample_atten,
                             apt.Freq.hz(s.resample_delta_freq, in_rate))
    if l > 1:
        f.resample(in_rate, apt.Rate.hz(rate * l))
    taps = f.design()
    c2 = apt.Freq.pi_rad(np.float32(4160) / np.float32(s.work_rate))
    lp = apt.Lowpass(c2, s.demodulation_atten, c2 / 5.0).design()
    off = (taps.size - 1) // 2
    w = (n_samples * l - off + m - 1) // m if l > 1 else n_samples // m   # fast_resampling's output count (dsp.rs:230-277)
    spr = 2080 * s.work_rate // 4160
    h = hashlib.sha256()
    h.update(taps.tobytes())
    h.update(lp.tobytes())
    h.update(np.array([l, m, w, w // spr, taps.size, lp.size], np.int64).tobytes())
    return int.from_bytes(h.digest()[:7], "little")


def _worker(rank, world, port, recs, out_q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = [n for _, n in recs]
        mine = shard.my_shard(lengths, rank, world)
        # the per-rank host work of a decode, through the product's own C ABI (no GPU here: the kernels cannot run,
        # and the oracle must not stand in for them): tap design + geometry of every recording of the shard
        digests = {i: _plan_digest(recs[i][0], recs[i][1]) for i in mine}
        samples = float(sum(lengths[i] for i in mine))
        dist.barrier()
        elapsed, total = shard.reduce_job(0.5 + 0.25 * rank, samples)
        counts = shard.gather_counts(len(mine))
        d = torch.zeros(len(recs), dtype=torch.int64)
        for i, v in digests.items():
            d[i] = v
        dist.all_reduce(d)  # (bookkeeping of the test, not of the data path: every recording's digest on every rank)
        out_q.put((rank, mine, elapsed, total, counts, d.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_job_over_gloo():
    import torch.multiprocessing as mp
    world = 2
    recs = [(48000, 28_800_000), (48000, 43_200_000), (11025, 11_025 * 600), (44100, 5_000_000), (96000, 43_200_000)]
    lengths = [n for _, n in recs]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, recs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_idx = sorted(i for r in res for i in r[1])
    assert all_idx == list(range(len(recs)))                # each recording prepared exactly once
    want = [_plan_digest(rate, n) for rate, n in recs]      # the same pieces in this process
    for rank, mine, elapsed, total, counts, digests in res:
        assert elapsed == pytest.approx(0.75)               # MAX over ranks
        assert total == float(sum(lengths))                 # whole-job samples
        assert counts == [len(res[0][1]), len(res[1][1])]
        assert digests == want                              # each rank's taps + geometry = the single-process ones
