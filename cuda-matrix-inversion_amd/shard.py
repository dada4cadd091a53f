"""Batch sharding across the GPUs of one node and result reassembly.

The reference is single-device; this is new capability. Every inversion is independent, so the
batch is block-partitioned by matrix index with no communication during compute; an RCCL all-gather over xGMI
(torch.distributed backend "nccl" on ROCm) is used ONLY to reassemble the result on every rank when the caller
asks for it. One process per GPU.
"""
from __future__ import annotations

from typing import List, Tuple


def partition(batch: int, world: int, multiple: int = 1) -> List[Tuple[int, int]]:
    """Contiguous block partition of [0, batch): rank g gets [start, stop).

    Shard sizes are ceil(batch/world) rounded up to `multiple` (the per-wavefront packing factor of the small-n
    kernels, 64/n matrices per wave) so that no wavefront straddles two ranks; trailing ranks may be short or empty.
    """
    if batch < 0 or world < 1 or multiple < 1:
        raise ValueError("bad partition arguments")
    per = -(-batch // world)
    per = -(-per // multiple) * multiple
    out = []
    for g in range(world):
        lo = min(batch, g * per)
        hi = min(batch, lo + per)
        out.append((lo, hi))
    return out


def packing_multiple(n: int) -> int:
    """Matrices per wavefront in the rowlane family (n <= 16); 1 otherwise."""
    if n <= 8:
        return 8
    if n <= 16:
        return 4
    return 1


def all_gather_shards(local, n: int, batch: int, group=None):
    """Reassemble the full result on every rank from per-rank shards (flat tensors of shard*n*n elements).

    Shards are padded to the common shard size so one fixed-size all_gather_into_tensor moves everything
    (each GPU's contiguous shard goes out once over its xGMI links); the padding is sliced off afterwards.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = partition(batch, world, packing_multiple(n))
    per = max(hi - lo for lo, hi in parts) if parts else 0
    rank = dist.get_rank(group)
    lo, hi = parts[rank]
    if local.numel() != (hi - lo) * n * n:
        raise ValueError(f"rank {rank}: shard has {local.numel()} elements, expected {(hi - lo) * n * n}")
    send = local
    if hi - lo < per:
        send = torch.zeros(per * n * n, dtype=local.dtype, device=local.device)
        send[: local.numel()] = local
    full = torch.empty(world * per * n * n, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, send.contiguous(), group=group)
    if world * per == batch:
        return full
    pieces = [full[g * per * n * n: g * per * n * n + (h - l) * n * n] for g, (l, h) in enumerate(parts)]
    return torch.cat(pieces)
