"""Multi-GPU path on CPU: world_size-2 gloo processes shard a batch, each 'inverts' its shard (with the oracle --
test infrastructure standing in for the GPU kernel, which needs a device) and all-gathers the result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg, spd_batch


def test_partition_properties():
    shard = pkg("shard")
    for batch in (0, 1, 7, 100, 100_000, 1_000_003):
        for world in (1, 2, 3, 8):
            for mult in (1, 4, 8):
                parts = shard.partition(batch, world, mult)
                assert len(parts) == world
                assert parts[0][0] == 0 and parts[-1][1] == batch
                for (l0, h0), (l1, h1) in zip(parts, parts[1:]):
                    assert h0 == l1 and l0 <= h0
                for lo, hi in parts[:-1]:
                    assert (hi - lo) % mult == 0 or hi == batch
    assert shard.partition(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard.packing_multiple(8) == 8 and shard.packing_multiple(16) == 4 and shard.packing_multiple(64) == 1


def _worker(rank, world, port, n, batch, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    import oracle
    shard = importlib.import_module("cuda-matrix-inversion_amd.shard")
    a = spd_batch(n, batch, seed=77)
    lo, hi = shard.partition(batch, world, shard.packing_multiple(n))[rank]
    local, _ = oracle.inverse_batched(a[lo * n * n: hi * n * n], n)
    full = shard.all_gather_shards(torch.from_numpy(local), n, batch)
    want, _ = oracle.inverse_batched(a, n)
    q.put((rank, bool(np.array_equal(full.numpy(), want))))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,batch", [(8, 37), (16, 64), (32, 5)])
def test_world2_gloo_shard_and_gather(n, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_c_partition_equals_the_python_one():
    """matinv_shard_range (what matinv_inverse_batched_host_multi cuts the batch with) against shard.partition: pure host
    arithmetic, no device needed."""
    import ctypes
    import importlib
    lib = importlib.import_module("cuda-matrix-inversion_amd._lib")
    shard = importlib.import_module("cuda-matrix-inversion_amd.shard")
    L = lib.lib()
    for batch in (0, 1, 7, 1000, 1003, 100_000, 1_000_000):
        for world in (1, 2, 3, 8):
            for n in (4, 8, 12, 16, 17, 64, 200):
                want = shard.partition(batch, world, shard.packing_multiple(n))
                for g in range(world):
                    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
                    assert L.matinv_shard_range(batch, world, n, g, ctypes.byref(lo), ctypes.byref(hi)) == 0
                    assert (lo.value, hi.value) == want[g], (batch, world, n, g)
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.matinv_shard_range(10, 2, 8, 2, ctypes.byref(lo), ctypes.byref(hi)) == lib.ERR_ARG
