"""Multi-process (gloo, world_size 2, CPU) test of the N>1 path: recordings shard across ranks
with no data-path collective; only a barrier and scalar reductions cross ranks."""
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


def _worker(rank, world, port, lengths, out_q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.my_shard(lengths, rank, world)
        # stand-in for the per-rank decode: a deterministic per-recording digest (the GPU
        # product path is not callable on a CPU box, and the oracle must not be used here)
        digest = sum((i + 1) * lengths[i] for i in mine)
        samples = float(sum(lengths[i] for i in mine))
        dist.barrier()
        elapsed, total = shard.reduce_job(0.5 + 0.25 * rank, samples)
        counts = shard.gather_counts(len(mine))
        d = torch.tensor([digest], dtype=torch.int64)
        dist.all_reduce(d)
        out_q.put((rank, mine, elapsed, total, counts, int(d.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_job_over_gloo():
    import torch.multiprocessing as mp
    world = 2
    lengths = [28_800_000, 43_200_000, 11_025 * 600, 5_000_000, 43_200_000]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_idx = sorted(i for r in res for i in r[1])
    assert all_idx == list(range(len(lengths)))            # each recording decoded exactly once
    for rank, mine, elapsed, total, counts, digest in res:
        assert elapsed == pytest.approx(0.75)               # MAX over ranks
        assert total == float(sum(lengths))                 # whole-job samples
        assert counts == [len(res[0][1]), len(res[1][1])]
        assert digest == sum((i + 1) * n for i, n in enumerate(lengths))
