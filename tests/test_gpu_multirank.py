"""The N > 1 path WITH the HIP kernels: two ranks share cuda:0 (the GPU box has one card), gloo carries the collective.
(The world_size-2 CPU rehearsal of the same plumbing is tests/test_shard_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _launch(nproc, script, *args, env=None):
    port = 29600 + os.getpid() % 2000
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *args]
    return subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=600, cwd=ROOT)


def test_two_ranks_hip_kernels_and_gather_match_one_launch():
    p = _launch(2, os.path.join(ROOT, "tests", "_two_rank_worker.py"))
    assert p.returncode == 0 and "TWO_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_bench_gpus2_runs_configs3_sharding():
    """bench.py --gpus 2 (gloo rehearsal on one card): total batch block-partitioned by shard.partition, one JSON line."""
    p = _launch(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--total-batch", "20001",
                env={"MATINV_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["total_batch"] == 20001
    assert out["config"]["batch_per_gpu"] == 10001  # rank 0's shard of shard.partition(20001, 2)
    assert out["gather"]["n_ranks_seen"] == 2 and out["gather"]["allgather_ms"] > 0
    assert out["roofline"]["units_per_launch"] == 10001 and 0 < out["roofline"]["frac"] < 1
    assert out["value"] > 0 and out["unit"] == "inversions/s"


def test_nccl_process_group_of_one_runs_the_bench_collectives():
    """The nccl (= RCCL) branch of bench.py on the one GPU there is: a world-size-1 process group (MATINV_BENCH_FORCE_DIST=1),
    so that init_process_group("nccl", device_id=...), the all-reduces and the all-gather of the result -- through
    torch.distributed AND through the C ABI's own communicator (matinv_comm_* / matinv_allgather_shards) -- are at least
    constructed and run before the driver's 8-GPU job meets them."""
    for impl in ("torch", "c"):
        p = _launch(1, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "5000",
                    "--no-others", "--no-cpu-baseline", env={"MATINV_BENCH_FORCE_DIST": "1", "MATINV_GATHER": impl})
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["gather"]["backend"] == "nccl" and out["gather"]["n_ranks_seen"] == 1 and out["gather"]["impl"] == impl
        assert out["gather"]["allgather_ms"] > 0 and out["gather"]["gather_equals_local_result"] is True


def test_c_abi_allgather_single_process():
    """matinv_allgather_local (one process, ncclCommInitAll over the visible devices -- one here) and the rank-wise form with
    a communicator of one: the gathered buffer equals the shard."""
    import ctypes
    import importlib
    import torch
    lib = importlib.import_module("cuda-matrix-inversion_amd._lib")
    L = lib.lib()
    send = torch.rand(4096, dtype=torch.float64, device="cuda")
    recv = torch.zeros_like(send)
    devs = (ctypes.c_int * 1)(0)
    sp = (ctypes.c_void_p * 1)(send.data_ptr())
    rp = (ctypes.c_void_p * 1)(recv.data_ptr())
    torch.cuda.synchronize()
    lib.check(L.matinv_allgather_local(1, devs, lib.F64, sp, rp, send.numel()))
    assert torch.equal(send, recv)
    uid = (ctypes.c_ubyte * 128)()
    lib.check(L.matinv_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p)))
    comm = ctypes.c_void_p()
    lib.check(L.matinv_comm_init_rank(ctypes.byref(comm), 1, ctypes.cast(uid, ctypes.c_void_p), 0))
    recv.zero_()
    lib.check(L.matinv_allgather_shards(comm, lib.F64, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel(),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    lib.check(L.matinv_comm_destroy(comm))


def test_local_allgather_is_ordered_behind_its_producer_stream():
    """matinv_allgather_local ran on the library's own streams with nothing ordering it behind the work that
    produces the shard. matinv_allgather_local_after takes the producer stream (an event recorded there is waited for on the
    gather's stream): a large inversion enqueued on a side stream and gathered AT ONCE, no host synchronisation in between, must
    arrive complete; the plain form synchronises the device on entry and passes the same check."""
    import ctypes
    import importlib
    import torch
    lib = importlib.import_module("cuda-matrix-inversion_amd._lib")
    api = importlib.import_module("cuda-matrix-inversion_amd.api")
    L = lib.lib()
    n, batch = 64, 20_000  # ~0.4 ms of kernel: long enough that an unordered gather would read it half-written
    g = torch.Generator(device="cuda").manual_seed(5)
    r = torch.rand(batch, n, n, generator=g, dtype=torch.float64, device="cuda")
    a = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=torch.float64, device="cuda")).reshape(-1).contiguous()
    want = api.inverse_batched(a, n, api.ALGO_GAUSS_JORDAN, batch=batch).clone()
    torch.cuda.synchronize()
    devs = (ctypes.c_int * 1)(0)
    side = torch.cuda.Stream()
    for ordered in (True, False):
        x = torch.zeros_like(a)
        recv = torch.zeros_like(a)
        torch.cuda.synchronize()
        sp = (ctypes.c_void_p * 1)(x.data_ptr())
        rp = (ctypes.c_void_p * 1)(recv.data_ptr())
        with torch.cuda.stream(side):
            for _ in range(3):  # a few launches deep
                api.inverse_batched(a, n, api.ALGO_GAUSS_JORDAN, out=x, batch=batch)
        if ordered:
            st = (ctypes.c_void_p * 1)(side.cuda_stream)
            lib.check(L.matinv_allgather_local_after(1, devs, lib.F64, sp, rp, x.numel(), st))
        else:
            lib.check(L.matinv_allgather_local(1, devs, lib.F64, sp, rp, x.numel()))
        # the call returns when the gather has completed: recv is final without any synchronisation here
        assert torch.equal(recv, want), "the gather read its shard before the producer had finished"
    torch.cuda.synchronize()
