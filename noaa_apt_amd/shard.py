"""Multi-GPU driver logic: independent recordings shard across ranks, no data-path collective.

decode() touches only its own arguments (/root/reference/src/decode.rs:43-162), so a batch of
recordings is embarrassingly parallel ACROSS recordings and not shardable WITHIN one (the
peak picker carries state from the first sample to the last).  One process per GPU; each rank
decodes its own recordings on its own device; the only communication is the bookkeeping
below (a barrier and two scalar all-reduces), which works on any torch.distributed backend
(RCCL on GPUs, gloo in the CPU tests).
"""
from typing import List, Sequence, Tuple


def assign(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of recordings (by sample count) to ranks.
    Deterministic; every index appears exactly once; returns one index list per rank."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(lengths[i])
    for lst in out:
        lst.sort()
    return out


def my_shard(lengths: Sequence[int], rank: int, world_size: int) -> List[int]:
    return assign(lengths, world_size)[rank]


def reduce_job(elapsed_s: float, samples: float, device=None) -> Tuple[float, float]:
    """(max elapsed over ranks, total samples over ranks).  No-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), float(samples)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([samples], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), float(s.item())


def gather_counts(value: int, device=None) -> List[int]:
    """Every rank's integer (e.g. rows decoded) on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(value)]
    mine = torch.tensor([int(value)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(o.item()) for o in out]
