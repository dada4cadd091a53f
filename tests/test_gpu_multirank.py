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
