#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched inversion hot path on MI355X.

One "step" = one pass of the hot path (one kernel launch) over one device-resident batch of synthetic
matrices. Default workload = BASELINE.json configs[2]: batch 100 000 of 64x64 fp64, Gauss-Jordan, on SPD inputs
generated as R + R^T + n*I (tests/generate_inverse_matrices.m:9-18 of the reference) -- the configuration the
north_star target ("inversions/s of 64x64 fp64 ... fraction of HBM roofline") is quoted on. `--workload` selects
the other single-GPU configs (n16 = configs[1], chol64 = the Cholesky half of configs[2]).

Multi-GPU (`--gpus N`, launched by torch.distributed.run, one rank per GPU): the batch shards by matrix index with
no data-path collective (weak scaling: every rank inverts its own `--batch` matrices); value = matrices all ranks
inverted / max-over-ranks time. The RCCL all-gather that reassembles results is timed separately and reported as
`allgather_ms` (it is not part of the inversion path and not in `value`).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "cuda-matrix-inversion_amd"

F64_MFMA_PEAK_TFLOPS = 78.6  # 1024 SIMDs x 32 flop/clk (v_mfma_f64_16x16x4_f64: 2048 flop per 64-cycle issue) x 2.4 GHz
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

WORKLOADS = {
    # name: (n, algo, description)
    "gj64": (64, "gj", "batch x 64x64 fp64 Gauss-Jordan (partial pivoting), SPD inputs, BASELINE configs[2]"),
    "chol64": (64, "chol", "batch x 64x64 fp64 Cholesky inverse, SPD inputs, BASELINE configs[2]"),
    "gj16": (16, "gj", "batch x 16x16 fp64 Gauss-Jordan (partial pivoting), SPD inputs, BASELINE configs[1]"),
    "gj32": (32, "gj", "batch x 32x32 fp64 Gauss-Jordan"),
    "gj8": (8, "gj", "batch x 8x8 fp64 Gauss-Jordan"),
    "gj128": (128, "gj", "batch x 128x128 fp64 Gauss-Jordan"),
    # GENERAL input: A ~ U(0,1)^(n x n), not symmetric, not dominant (like the reference's tests/square_5_*.mats): every
    # matrix needs row exchanges
    "gj64g": (64, "gj", "batch x 64x64 fp64 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
    "gj32g": (32, "gj", "batch x 32x32 fp64 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
    "gj128g": (128, "gj", "batch x 128x128 fp64 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
}
GENERAL = {"gj64g", "gj32g", "gj128g"}


def make_spd(n, batch, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    r = torch.rand((batch, n, n), generator=g, dtype=torch.float64, device=device)
    a = r + r.transpose(1, 2)
    a.diagonal(dim1=1, dim2=2).add_(float(n))
    return a.reshape(-1).contiguous()


def make_general(n, batch, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.rand((batch * n * n,), generator=g, dtype=torch.float64, device=device)


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(n, algo_name, target_seconds=12.0):
    """Oracle (oracle/liboracle.so, a C port of the reference's CPU algorithms) timed on this box's host cores,
    OpenMP schedule(dynamic,8) over the batch as src/inverse.c:79 of the reference. Bounded sample."""
    import oracle
    cores = usable_cores()
    algo = oracle.ALGO_GJ_PIVOT if algo_name == "gj" else oracle.ALGO_CHOLESKY
    rng = np.random.default_rng(0)

    def sample(k):
        r = rng.random((k, n, n))
        return (r + r.transpose(0, 2, 1) + n * np.eye(n)).reshape(-1)

    probe = sample(max(cores * 16, 256))
    oracle.inverse_batched(probe, n, algo, threads=cores)  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    oracle.inverse_batched(probe, n, algo, threads=cores)
    dt = max(time.perf_counter() - t0, 1e-6)
    rate = (probe.size // (n * n)) / dt
    # a sample of at most ~1 GB, inverted `passes` times back to back so that the timed CPU work is ~target_seconds
    k = int(min(max(rate * target_seconds, 1024), 1e9 / (n * n * 8)))
    passes = max(1, int(round(rate * target_seconds / k)))
    a = sample(k)
    t0 = time.perf_counter()
    for _ in range(passes):
        oracle.inverse_batched(a, n, algo, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": k * passes / dt, "unit": "inversions/s", "cores": cores, "kind": "port",
            "sample": f"{k} SPD {n}x{n} fp64 matrices x {passes} passes, oracle "
                      f"{'Gauss-Jordan partial pivot' if algo_name == 'gj' else 'Cholesky'}"
                      f" (C, OpenMP {cores} threads, schedule(dynamic,8)), {dt:.1f} s"}


def run_mixed(args, api, device, rank, world):
    """BASELINE configs[4]: mixed-size fp32 items n in {32,128,512,1024} through the size-binned multi-queue, full
    add -> inv -> gemv -> dot mean pipeline (fused). Stated mix (items per step and GPU): 32: 16384, 128: 2048, 512: 32,
    1024: 8 -- roughly equal flops per bin is NOT attempted; per-bin rates are reported. One step = submit + flush."""
    bq = importlib.import_module(PKG + ".binqueue")
    mix = {32: 16384, 128: 2048, 512: 32, 1024: 8}
    g = torch.Generator(device=device).manual_seed(0x5EED + rank)
    items, chunks = [], []
    CH = 256  # items arrive in same-size chunks of up to 256 (submit_many); the sizes are interleaved chunk by chunk
    for n, cnt in mix.items():
        r = torch.rand((cnt, n, n), generator=g, dtype=torch.float32, device=device)
        B = r + r.transpose(1, 2)
        B.diagonal(dim1=1, dim2=2).add_(float(n))
        v = torch.rand((3, cnt, n), generator=g, dtype=torch.float32, device=device)
        items += [(v[0, i], B[i].reshape(-1), v[1, i], v[2, i]) for i in range(cnt)]
        chunks += [(n, v[0, i:i + CH].reshape(-1), B[i:i + CH].reshape(-1), v[1, i:i + CH].reshape(-1), v[2, i:i + CH].reshape(-1))
                   for i in range(0, cnt, CH)]
    chunks = [chunks[i] for i in torch.randperm(len(chunks), generator=torch.Generator().manual_seed(1)).tolist()]
    q = bq.SizeBinnedQueue(device=device)

    def step():
        for ch in chunks:
            q.submit_many(*ch)
        return q.flush()[0]

    import torch.distributed as tdist
    multi = world > 1 and tdist.is_initialized()
    for _ in range(max(1, args.warmup)):
        step()
    if multi:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        tdist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:  # max over ranks
        backend_dev = device if tdist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=backend_dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t[0].item())
    per_bin = {}
    for n, cnt in mix.items():  # kernel-only rate of each bin (batched, device resident)
        sel = [it for it in items if it[0].numel() == n]
        A_, B_, C_, D_ = (torch.cat([it[k].reshape(-1) for it in sel]) for k in range(4))
        api.calcluateMean(n, A_, B_, C_, D_)
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        api.calcluateMean(n, A_, B_, C_, D_)
        e_.record()
        torch.cuda.synchronize()
        per_bin[str(n)] = {"items": cnt, "kernel_ms": s_.elapsed_time(e_), "items_per_s": cnt / (s_.elapsed_time(e_) * 1e-3)}
    if rank == 0:
        total = sum(mix.values()) * world * args.steps
        print(json.dumps({
            "metric": "pipeline items/s (mixed sizes)", "value": total / elapsed, "unit": "items/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "mixed: fp32 mean pipeline, size-binned queues 32/128/512/1024 (BASELINE configs[4])",
                       "mix_items_per_step_per_gpu": mix, "includes": "host-side submit_many (same-size chunks of <= 256 items, sizes interleaved) + per-bin batch assembly + kernels"},
            "per_bin": per_bin}), flush=True)


def mfma_flops_per_inversion(algo_name, n):
    """fp64 flops the MFMA tile kernels issue per matrix (None for the families without MFMA): a blocked sweep of 4*NT
    rank-4 steps over NT^2 tiles (Gauss-Jordan: all tiles = 2 n^3 flop) or over the NT(NT+1)/2 lower tiles (SPD sweep),
    2048 flop per v_mfma_f64_16x16x4_f64."""
    if n <= 16 or n > 128 or (algo_name != "gj" and n > 64):
        return None
    nt = (n + 15) // 16
    tiles = nt * nt if algo_name == "gj" else nt * (nt + 1) // 2
    return 4 * nt * tiles * 2048


def rooflines(algo_name, n, batch, kern_ms):
    """(binding roofline dict, the other one or None): HBM at 2 n^2 sizeof(T) algorithmic bytes per inversion, fp64 MFMA at
    the flops above; the binding one is the one with the larger minimum time."""
    bytes_per_inv = 2 * n * n * 8  # read A once + write A^-1 once
    gbs = batch * bytes_per_inv / (kern_ms * 1e-3) / 1e9
    hbm = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": batch * bytes_per_inv}
    fl = mfma_flops_per_inversion(algo_name, n)
    if fl is None:
        return hbm, None
    tf = batch * fl / (kern_ms * 1e-3) / 1e12
    mf = {"bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TFLOPS,
          "algorithmic_flops_per_launch": batch * fl}
    return (hbm, mf) if hbm["frac"] >= mf["frac"] else (mf, hbm)


def load_traffic(kernel_name, n):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None
    try:
        table = json.load(open(p))
    except Exception:
        return None
    return table.get(f"{kernel_name}|n={n}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gj64", choices=sorted(WORKLOADS) + ["mixed"])
    ap.add_argument("--batch", type=int, default=100_000, help="matrices per GPU per step")
    ap.add_argument("--kernel", default="auto", choices=["auto", "lds", "rowlane", "tile", "tilep", "row"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other single-GPU workloads")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # one rank per GPU; MATINV_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed on a single-GPU box (ranks then
    # share the device and the collectives run through host memory)
    backend = os.environ.get("MATINV_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    coll_dev = device if backend == "nccl" else torch.device("cpu")

    api = importlib.import_module(PKG + ".api")
    if args.workload == "mixed":
        run_mixed(args, api, device, rank, world)
        if dist is not None:
            dist.destroy_process_group()
        return
    n, algo_name, desc = WORKLOADS[args.workload]
    algo = api.ALGO_GAUSS_JORDAN if algo_name == "gj" else api.ALGO_CHOLESKY
    kernel = {"auto": api.KERNEL_AUTO, "lds": api.KERNEL_LDS, "rowlane": api.KERNEL_ROWLANE,
              "tile": api.KERNEL_TILE, "tilep": api.KERNEL_TILEP, "row": api.KERNEL_ROW}[args.kernel]
    batch = args.batch

    general = args.workload in GENERAL
    a = (make_general if general else make_spd)(n, batch, 0x5EED + rank, device)
    x = torch.empty_like(a)
    info = torch.empty(batch, dtype=torch.int32, device=device)

    def step():
        api.inverse_batched(a, n, algo, out=x, info=info, kernel=kernel, batch=batch)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    assert int(info.abs().sum()) == 0, "synthetic SPD batch reported singular matrices"

    # HIP events on the stream the kernel is launched on (api passes torch's current stream to the C ABI)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        step()
        e.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))

    gather_ms = None
    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0].item()), float(t[1].item())
        shard = importlib.import_module(PKG + ".shard")
        torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        # result reassembly on every rank (RCCL all-gather over xGMI); host staging only in the gloo rehearsal
        full = shard.all_gather_shards(x if backend == "nccl" else x.cpu(), n, batch * world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert full.numel() == batch * world * n * n
        del full

    # quick correctness guard on the timed output (not timed): residual of a few matrices
    am = a.view(batch, n, n)[:8]
    xm = x.view(batch, n, n)[:8]
    resid = float((torch.bmm(am, xm) - torch.eye(n, dtype=a.dtype, device=device)).abs().max())
    assert resid < (1e-9 if general else 1e-11) * n, f"residual {resid}"

    # what a plain device-to-device copy of the same bytes reaches on this box (SURVEY 8d asks for the fraction against
    # the measured copy bandwidth beside the nominal 8 TB/s); torch's copy kernel, same read+write byte count
    copy_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    x.copy_(a)
    for s_, e_ in copy_ev:
        s_.record()
        x.copy_(a)
        e_.record()
    torch.cuda.synchronize()
    copy_gbs = 2 * a.numel() * a.element_size() / (float(np.median([s_.elapsed_time(e_) for s_, e_ in copy_ev])) * 1e-3) / 1e9

    others = {}
    if rank == 0 and world == 1 and not args.no_others:
        # the other single-GPU configs of BASELINE.json, a few launches each (same timing method), for the record
        del a, x
        for wname in ("gj16", "chol64", "gj32", "gj8", "gj128"):
            if wname == args.workload:
                continue
            n2, algo2_name, _ = WORKLOADS[wname]
            algo2 = api.ALGO_GAUSS_JORDAN if algo2_name == "gj" else api.ALGO_CHOLESKY
            b2 = min(batch, 25_000) if n2 >= 128 else batch
            a2 = make_spd(n2, b2, 0x5EED + 17 * n2, device)
            x2 = torch.empty_like(a2)
            for _ in range(2):
                api.inverse_batched(a2, n2, algo2, out=x2, batch=b2)
            ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for s_, e_ in ev2:
                s_.record()
                api.inverse_batched(a2, n2, algo2, out=x2, batch=b2)
                e_.record()
            torch.cuda.synchronize()
            ms2 = float(np.mean([s_.elapsed_time(e_) for s_, e_ in ev2]))
            r1, r2 = rooflines(algo2_name, n2, b2, ms2)
            others[wname] = {"kernel": api.kernel_name(algo2, api.F64, n2), "batch": b2, "kernel_ms": ms2,
                             "inversions_per_s": b2 / (ms2 * 1e-3), "bound": r1["bound"], "frac": r1["frac"],
                             "achieved": r1["achieved"], "unit": r1["unit"],
                             "other_bound_frac": None if r2 is None else r2["frac"]}
            del a2, x2

    if rank == 0:
        total = batch * world * args.steps
        value = total / elapsed
        kname = api.kernel_name(algo, api.F64, n, kernel)
        roof, roof_other = rooflines(algo_name, n, batch, kern_ms)
        roof.update({"traffic": load_traffic(kname, n), "kernel": kname, "kernel_ms": kern_ms})
        hbm_side = roof if roof["bound"] == "hbm" else roof_other
        hbm_side["measured_copy_GBs"] = copy_gbs
        hbm_side["frac_of_measured_copy"] = hbm_side["achieved"] / copy_gbs
        if roof_other is not None:
            roof["other_bound"] = roof_other
        out = {
            "metric": "matrix inversions/s", "value": value, "unit": "inversions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc.replace('batch', str(batch))}", "n": n,
                       "batch_per_gpu": batch, "algorithm": algo_name, "kernel": kname,
                       "sharding": f"batch block-partitioned over {world} GPU(s), no data-path collective"},
            "roofline": roof,
        }
        if gather_ms is not None:
            out["allgather_ms"] = gather_ms
        if others:
            out["other_workloads"] = others
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, algo_name)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
