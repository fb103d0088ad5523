"""Multi-GPU sharding: independent swarm instances partition contiguously by rank; the only
collective is the gather of results (RCCL all_gather over xGMI, or gloo in CPU rehearsal)."""
from __future__ import annotations

from typing import Tuple


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of a batch of B instances owned by `rank`; remainders go to the low ranks."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local, B: int, group=None):
    """all_gather variable-length shards of [b_local, ...] tensors back into [B, ...] on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(B, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
