"""Batch sharding across the GPUs of one node and result reassembly.

The reference is single-device; this is new capability. Every inversion is independent, so the
batch is block-partitioned by matrix index with no communication during compute; an RCCL all-gather over xGMI
(torch.distributed backend "nccl" on ROCm) is used ONLY to reassemble the result on every rank when the caller
asks for it. One process per GPU.
"""
from __future__ import annotations

import ctypes
import os
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


_COMMS = {}  # process group -> ncclComm_t of the C ABI (matinv_comm_init_rank), created on first use


def _c_comm(group, device):
    """The C ABI's own RCCL communicator for `group`: rank 0 draws the unique id (matinv_comm_unique_id), the 128 bytes travel
    through the process group, every rank joins with matinv_comm_init_rank on its device."""
    import torch
    import torch.distributed as dist
    from . import _lib
    key = id(group) if group is not None else 0
    if key in _COMMS:
        return _COMMS[key]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    buf = (ctypes.c_ubyte * 128)()
    if rank == 0:
        _lib.check(_lib.lib().matinv_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
    t = torch.tensor(list(buf), dtype=torch.uint8, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    raw = bytes(t.cpu().tolist())
    comm = ctypes.c_void_p()
    with torch.cuda.device(device):
        _lib.check(_lib.lib().matinv_comm_init_rank(ctypes.byref(comm), world, ctypes.c_char_p(raw), rank))
    _COMMS[key] = comm
    return comm


def all_gather_shards(local, n: int, batch: int, group=None, impl: str | None = None):
    """Reassemble the full result on every rank from per-rank shards (flat tensors of shard*n*n elements).

    Shards are padded to the common shard size so ONE fixed-size all-gather moves everything (each GPU's contiguous shard
    goes out once over its xGMI links); the padding is sliced off afterwards. impl = "c": the C ABI's matinv_allgather_shards
    (ncclAllGather on the library's own communicator) -- the default for device tensors on the "nccl" backend since r04 (the C
    path is the product; MATINV_GATHER=torch switches back); "torch": torch.distributed.all_gather_into_tensor on the group's
    backend (RCCL for "nccl"; gloo for the CPU rehearsal, where it is the only choice).
    """
    import torch
    import torch.distributed as dist
    if impl is None:
        impl = os.environ.get("MATINV_GATHER") or ("c" if (local.is_cuda and dist.get_backend(group) == "nccl") else "torch")
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
    send = send.contiguous()
    if impl == "c" and local.is_cuda:
        from . import _lib
        comm = _c_comm(group, local.device)
        with torch.cuda.device(local.device):
            _lib.check(_lib.lib().matinv_allgather_shards(
                comm, _lib.F64 if local.dtype == torch.float64 else _lib.F32, ctypes.c_void_p(send.data_ptr()),
                ctypes.c_void_p(full.data_ptr()), per * n * n, ctypes.c_void_p(torch.cuda.current_stream(local.device).cuda_stream)))
    else:
        dist.all_gather_into_tensor(full, send, group=group)
    if world * per == batch:
        return full
    pieces = [full[g * per * n * n: g * per * n * n + (h - l) * n * n] for g, (l, h) in enumerate(parts)]
    return torch.cat(pieces)
