#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched inversion hot path on MI355X.

One "step" = one pass of the hot path (one kernel launch) over one device-resident batch of synthetic
matrices. Default workload = BASELINE.json configs[2]: batch 100 000 of 64x64 fp64, Gauss-Jordan, on SPD inputs
generated as R + R^T + n*I (tests/generate_inverse_matrices.m:9-18 of the reference) -- the configuration the
north_star target ("inversions/s of 64x64 fp64 ... fraction of HBM roofline") is quoted on. `--workload` selects
the other single-GPU configs (n16 = configs[1], chol64 = the Cholesky half of configs[2]).

Multi-GPU (`--gpus N`, launched by torch.distributed.run, one rank per GPU) = BASELINE.json configs[3] literally: a TOTAL
batch of 1 000 000 64x64 fp64 matrices (`--total-batch`), block-partitioned over the ranks by `shard.partition`, every rank
generating its own shard from its own seed; no data-path collective. value = total matrices / max-over-ranks time of the
inversions alone. The RCCL all-gather over xGMI that reassembles the result on every rank (the only exchange the path
has) is timed separately and reported as `allgather_ms` / `allgather_GBs` beside the per-link xGMI bound; it is not in
`value`. `--batch B` instead gives every rank B matrices (weak scaling).

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
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32: 2048 flop per 32-cycle issue, same table of the guide (fp32 matrix, dense)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_GUIDE_ACHIEVABLE_GBS = 6290.0  # same table: "6.29 TB/s measured (float4 copy, 79 %)" -- reported beside the nominal fraction

WORKLOADS = {
    # name: (n, algo, description)
    "gj64": (64, "gj", "batch x 64x64 fp64 Gauss-Jordan, SPD inputs (natural-order pivots verified inside the MFMA sweep; a rejected matrix is redone with partial pivoting), BASELINE configs[2]"),
    "chol64": (64, "chol", "batch x 64x64 fp64 Cholesky inverse, SPD inputs, BASELINE configs[2]"),
    "gj16": (16, "gj", "batch x 16x16 fp64 Gauss-Jordan (threshold partial pivoting), SPD inputs, BASELINE configs[1]"),
    "gj32": (32, "gj", "batch x 32x32 fp64 Gauss-Jordan"),
    "gj8": (8, "gj", "batch x 8x8 fp64 Gauss-Jordan"),
    "gj24": (24, "gj", "batch x 24x24 fp64 Gauss-Jordan (two rows per lane, DPP broadcasts; natural-order pivots verified), SPD inputs"),
    "gj50": (50, "gj", "batch x 50x50 fp64 Gauss-Jordan (ragged: 4 x 4 MFMA tiles, 13 of 16 block steps), SPD inputs"),
    "gj128": (128, "gj", "batch x 128x128 fp64 Gauss-Jordan"),
    # GENERAL input: A ~ U(0,1)^(n x n), not symmetric, not dominant (like the reference's tests/square_5_*.mats): every
    # matrix needs row exchanges
    "gj64g": (64, "gj", "batch x 64x64 fp64 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
    "gj32g": (32, "gj", "batch x 32x32 fp64 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
    "gj96g": (96, "gj", "batch x 96x96 fp64 Gauss-Jordan with partial pivoting (three wavefronts per matrix), GENERAL U(0,1) inputs"),
    "gj128g": (128, "gj", "batch x 128x128 fp64 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
    "gj192g": (192, "gj", "batch x 192x192 fp64 Gauss-Jordan with partial pivoting (one wavefront per tile column), GENERAL U(0,1) inputs"),
    "chol192": (192, "chol", "batch x 192x192 fp64 Cholesky inverse (one wavefront per tile column), SPD inputs"),
    "chol130": (130, "chol", "batch x 130x130 fp64 Cholesky inverse (two wavefronts, lower tiles only, one wave per SIMD), SPD inputs"),
    "chol144": (144, "chol", "batch x 144x144 fp64 Cholesky inverse (two wavefronts, lower tiles only, one wave per SIMD), SPD inputs"),
    "chol128": (128, "chol", "batch x 128x128 fp64 Cholesky inverse (two wavefronts, lower tiles only), SPD inputs"),
    "gj256g": (256, "gj", "batch x 256x256 fp64 blocked Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
    "gj1024g": (1024, "gj", "batch x 1024x1024 fp64 blocked Gauss-Jordan (two-level, MFMA update), GENERAL U(0,1) inputs"),
    "chol256": (256, "chol", "batch x 256x256 fp64 blocked Cholesky inverse, SPD inputs"),
    "chol1024": (1024, "chol", "batch x 1024x1024 fp64 blocked Cholesky inverse (panel pairs, rank-128 update, Y Y^T product), SPD inputs"),
    # fp32 = the reference's own DataType (include/types.h:4 there; report.tex:91 "all calculations in single precision")
    "gj64_f32": (64, "gj", "batch x 64x64 fp32 Gauss-Jordan, SPD inputs"),
    "chol64_f32": (64, "chol", "batch x 64x64 fp32 Cholesky inverse, SPD inputs"),
    "gj128_f32": (128, "gj", "batch x 128x128 fp32 Gauss-Jordan, SPD inputs"),
    "gj64g_f32": (64, "gj", "batch x 64x64 fp32 Gauss-Jordan with partial pivoting, GENERAL U(0,1) inputs"),
}
GENERAL = {"gj64g", "gj32g", "gj96g", "gj128g", "gj192g", "gj256g", "gj1024g", "gj64g_f32"}
F32 = {"gj64_f32", "chol64_f32", "gj128_f32", "gj64g_f32"}


def make_spd(n, batch, seed, device, dtype=torch.float64):
    g = torch.Generator(device=device).manual_seed(seed)
    r = torch.rand((batch, n, n), generator=g, dtype=dtype, device=device)
    a = r + r.transpose(1, 2)
    a.diagonal(dim1=1, dim2=2).add_(float(n))
    return a.reshape(-1).contiguous()


def make_general(n, batch, seed, device, dtype=torch.float64):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.rand((batch * n * n,), generator=g, dtype=dtype, device=device)


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


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n, algo_name, target_seconds=9.0):
    """Oracle (oracle/liboracle.so, a C port of the reference's CPU algorithms) timed on this box's host cores,
    OpenMP schedule(dynamic,8) over the batch as src/inverse.c:79 of the reference. Two rows:
    OMP threads = 8 (the reference's README.md:9 protocol) and = every core this process may use. Bounded sample."""
    import oracle
    algo = oracle.ALGO_GJ_PIVOT if algo_name == "gj" else oracle.ALGO_CHOLESKY
    rng = np.random.default_rng(0)

    def sample(k):
        r = rng.random((k, n, n))
        return (r + r.transpose(0, 2, 1) + n * np.eye(n)).reshape(-1)

    def timed(cores):
        probe = sample(max(cores * 16, 256))
        oracle.inverse_batched(probe, n, algo, threads=cores)  # warm-up (thread pool, page faults)
        # calibration over >= 0.3 s (one pass over 256 matrices takes a millisecond and can be off by an order of magnitude)
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.3:
            oracle.inverse_batched(probe, n, algo, threads=cores)
            reps += 1
        rate = reps * (probe.size // (n * n)) / (time.perf_counter() - t0)
        # a sample of at most ~1 GB and about one second per pass, inverted back to back until ~target_seconds of CPU work
        k = int(min(max(rate * 1.0, 1024), 1e9 / (n * n * 8)))
        a = sample(k)
        t0, passes = time.perf_counter(), 0
        while passes == 0 or time.perf_counter() - t0 < target_seconds:
            oracle.inverse_batched(a, n, algo, threads=cores)
            passes += 1
        dt = time.perf_counter() - t0
        return k * passes / dt, f"{k} SPD {n}x{n} fp64 matrices x {passes} passes, {dt:.1f} s"

    allc = usable_cores()
    v_all, s_all = timed(allc)
    out = {"value": v_all, "unit": "inversions/s", "cores": allc, "kind": "port",
           "sample": s_all + f"; oracle {'Gauss-Jordan partial pivot' if algo_name == 'gj' else 'Cholesky'}"
                             f" (C, OpenMP, schedule(dynamic,8))",
           "host": {"model": cpu_model(), "logical_cpus": os.cpu_count(), "usable_cores": allc}}
    if allc != 8:
        v8, s8 = timed(min(8, allc))
        out["omp8"] = {"value": v8, "cores": min(8, allc), "sample": s8}
    return out


def make_mixed_queues(device, count):
    """The C queues of the mixed-size workload, created (with their streams) before anything else in the process uses a stream."""
    bq = importlib.import_module(PKG + ".binqueue")
    qs = [bq.SizeBinnedQueue(device=device) for _ in range(count)]
    for q_ in qs:
        q_.home_stream(torch.float32)
    return qs


def mixed_result(args, api, device, rank, world):
    """BASELINE configs[4]: mixed-size fp32 items n in {32,128,512,1024} through the size-binned multi-queue, full
    add -> inv -> gemv -> dot mean pipeline (fused). Stated mix (items per step and GPU): 32: 16384, 128: 2048, 512: 32,
    1024: 8 -- roughly equal flops per bin is NOT attempted; per-bin rates are reported. One step = submit + flush."""
    bq = importlib.import_module(PKG + ".binqueue")
    mix = {32: 16384, 128: 2048, 512: 32, 1024: 8}
    if os.environ.get("MATINV_MIX"):  # experiments only, e.g. MATINV_MIX=32:16384,128:2048
        mix = {int(k): int(v) for k, v in (kv.split(":") for kv in os.environ["MATINV_MIX"].split(","))}
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
    # `inflight` flushes may be in progress at once (default 4; MATINV_MIX_INFLIGHT overrides it in --workload mixed): consecutive steps
    # alternate between that many queues, so the dependent-launch chains of one step's large bins (8 x 1024^2 = 33 launches that leave
    # most of the chip idle) run beside the next steps'. Every step still submits and flushes ALL of its items, results are complete at
    # the synchronisation that ends the timed region; 1 = strictly one flush after the other.
    # Stream placement (r04): a queue creates one stream when it is created, main() creates the four queues before anything else in the
    # process creates a stream (they land on the four hardware queues), and a flush issued on its queue's own stream stays in it. No
    # selection among stream sets any more (r03 tried four and reported the best).
    inflight = max(1, int(getattr(args, "inflight", 0) or os.environ.get("MATINV_MIX_INFLIGHT", "4")))
    def make_set():
        # the queues: those created at the start of the process when there are enough of them (make_mixed_queues: a queue's streams are
        # placed on hardware queues when they are created, and placement among the first streams of a process is the reproducible one)
        pool = getattr(args, "queues", None) or []
        qs_ = [pool.pop(0) if pool else bq.SizeBinnedQueue(device=device) for _ in range(inflight)]
        # several flushes in flight: each is issued on its queue's OWN stream (matinv_queue_stream) and runs in it from end to end -- no
        # stream of one flush waits for another stream; one flush at a time: on the caller's stream, where the chain of the largest bin
        # forks beside the other bins (the shorter latency). MATINV_MIX_CALLER_STREAMS=1: torch streams also with several in flight (r03).
        if inflight == 1:
            st_ = [torch.cuda.current_stream(device)]
        elif os.environ.get("MATINV_MIX_CALLER_STREAMS") == "1":
            st_ = [torch.cuda.Stream(device=device) for _ in range(inflight)]
        else:
            st_ = [q_.home_stream(torch.float32) for q_ in qs_]
        return {"qs": qs_, "streams": st_, "turn": 0}

    cur = make_set()
    table = cur["qs"][0].chunk_table(chunks)  # the pointer arrays of the C call, built once: the same chunks arrive every step
    host_s = [0.0]

    def step():
        t_ = time.perf_counter()
        k = cur["turn"] % inflight
        cur["turn"] += 1
        with torch.cuda.stream(cur["streams"][k]):
            cur["qs"][k].submit_table(table)  # ONE C call (matinv_queue_submit_chunks) for the step's 74 chunks
            out = cur["qs"][k].flush()[0]
        host_s[0] += time.perf_counter() - t_  # submit + flush return when everything is ENQUEUED: the host share of a step
        return out

    placement = ("caller streams (forking flushes)" if (inflight == 1 or os.environ.get("MATINV_MIX_CALLER_STREAMS") == "1") else
                 "every flush on its queue's own stream (matinv_queue_stream), which it does not leave; queues created first; no trials")
    q = cur["qs"][0]

    import torch.distributed as tdist
    multi = world > 1 and tdist.is_initialized()
    # warm-up: every queue has to have flushed three times before the steady state is reached (the library replays the launch chain of a
    # large-n group as a HIP graph from its third appearance on: first sighting, capture, replay)
    # (no garbage collection inside the timed region: at the end of the whole bench.py run a collection that frees the previous
    # workloads' tensors -- 66 ms in one step of twenty -- otherwise lands in it now and then)
    import gc
    gc.collect()
    gc.disable()
    for _ in range(max(1, args.warmup, 3 * inflight)):
        step()
    if multi:
        tdist.barrier()
    torch.cuda.synchronize()
    host_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        tdist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if multi:  # max over ranks
        backend_dev = device if tdist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=backend_dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t[0].item())
    # host work of one step by itself: submit + flush timed with the device idle (in the timed loop above the host runs ahead
    # of the device and then waits inside flush for the previous step's results, so its share there is not host WORK)
    idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        q.submit_table(table)
        q.flush()
        idle.append(time.perf_counter() - t_)
    torch.cuda.synchronize()
    host_idle_ms = float(np.median(idle)) * 1e3
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
        ms_ = s_.elapsed_time(e_)
        alg = cnt * (n * n + 3 * n + 1) * 4  # fused pipeline: read B, a, c, d, write one scalar, fp32
        per_bin[str(n)] = {"items": cnt, "kernel_ms": ms_, "items_per_s": cnt / (ms_ * 1e-3),
                           "roofline": {"bound": "hbm", "achieved": alg / (ms_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": alg / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg}}
    # the flushed means against the per-bin launches of the same items (the queue adds gather / scatter only)
    means = step()
    torch.cuda.synchronize()
    total = sum(mix.values()) * world * args.steps
    return {
        "metric": "pipeline items/s (mixed sizes)", "value": total / elapsed, "unit": "items/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "mixed: fp32 mean pipeline, size-binned queues 32/128/512/1024 (BASELINE configs[4])",
                   "mix_items_per_step_per_gpu": mix, "queue": "C (matinv_queue_*, csrc/queue.hip)", "flushes_in_flight": inflight,
                   "stream_placement": placement,
                   "includes": "submit of same-size chunks of <= 256 items (sizes interleaved) + flush: segmented gather + one fused launch per size, one stream per bin"},
        "host_ms_per_step": host_idle_ms, "host_share": host_idle_ms / (elapsed / args.steps * 1e3),
        "host_ms_in_calls_back_to_back": host_s[0] / args.steps * 1e3,
        "means_finite": bool(torch.isfinite(means).all()),
        "per_bin": per_bin}


def run_mixed(args, api, device, rank, world):
    out = mixed_result(args, api, device, rank, world)
    if rank == 0:
        print(json.dumps(out), flush=True)


def mfma_flops_per_inversion(algo_name, n):
    """fp64 flops the MFMA tile kernels issue per matrix (None for the families without MFMA): a blocked sweep of 4*NT
    rank-4 steps over NT^2 tiles (Gauss-Jordan: all tiles = 2 n^3 flop) or over the NT(NT+1)/2 lower tiles (SPD sweep on one or
    two or three wavefronts per matrix, n <= 192 in fp64), 2048 flop per MFMA.
    Blocked two-level Gauss-Jordan (n >= 384): 2 n^3. Blocked SPD inverse (n > 192; update and Y Y^T product on the matrix
    cores): n^3 -- factor, triangular inverse and product at n^3 / 3 each (the 64 x 64 tile granularity issues more)."""
    if n > 192 and algo_name == "gj":
        # blocked Gauss-Jordan: 2 n^3 algorithmic flops. n >= 384: rank-128 update on the matrix cores; 192 < n < 384: rank-32
        # update on the vector ALU, priced against the fp64 VECTOR peak, which on gfx950 equals the matrix peak (same datapath)
        return 2 * n ** 3
    if n > 192 and algo_name == "chol":
        return n ** 3
    if n <= 16 or n > 192:
        return None
    nt = (n + 15) // 16
    tiles = nt * nt if algo_name == "gj" else nt * (nt + 1) // 2
    # block steps actually executed: the all-padding 4-column blocks of the last tile column are skipped (fp64: ceil(rem / 4))
    steps = 4 * (nt - 1) + (n - 16 * (nt - 1) + 3) // 4
    return steps * tiles * 2048


def rooflines(algo_name, n, batch, kern_ms, elem=8):
    """(binding roofline dict, the other one or None): HBM at 2 n^2 sizeof(T) algorithmic bytes per inversion, the MFMA pipe of the
    data type at the flops above; the binding one is the one with the larger minimum time."""
    bytes_per_inv = 2 * n * n * elem  # read A once + write A^-1 once
    gbs = batch * bytes_per_inv / (kern_ms * 1e-3) / 1e9
    hbm = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": batch * bytes_per_inv}
    fl = mfma_flops_per_inversion(algo_name, n)
    if fl is None:
        return hbm, None
    tf = batch * fl / (kern_ms * 1e-3) / 1e12
    peak = F64_MFMA_PEAK_TFLOPS if elem == 8 else F32_MFMA_PEAK_TFLOPS
    mf = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "algorithmic_flops_per_launch": batch * fl}
    if algo_name == "gj" and 192 < n < 384:
        mf["pipe"] = "fp64 vector ALU (same peak as the matrix cores on gfx950)"
    return (hbm, mf) if hbm["frac"] >= mf["frac"] else (mf, hbm)


def load_traffic(kernel_name, n, batch):
    """HBM bytes of ONE launch of `batch` matrices, scaled from the bytes per matrix that the committed rocprofv3 --pmc passes
    measured (profiles/traffic.json: FETCH_SIZE / WRITE_SIZE in separate passes, corrected as the guide prescribes), or
    (None, None). It is NOT measured in this run: the source is named in the JSON."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        table = json.load(open(p))
        d = table.get(f"{kernel_name}|n={n}|detail")
        if d is None:
            # rocprofv3 prints every template argument of the instantiation, the library's kernel_name() only the ones a caller can
            # tell apart (matinv_gj_tile_f64<4, true, true> is matinv_gj_tile_f64<4, true, true, false> there: EARLY = false)
            stem = kernel_name[:-1] + ","
            d = next(v for k, v in table.items() if k.endswith(f"|n={n}|detail") and k.startswith(stem))
        per_matrix = (d["read_bytes"] + d["write_bytes"]) / d["batch"]
        return per_matrix * batch, f"profiles/traffic.json ({d['source']}: {per_matrix:.0f} B per matrix x {batch})"
    except Exception:
        return None, None


XGMI_LINK_GBS = 153.6  # per link and direction, 7 links per GPU (/opt/skills/guides/MI355X_MICROARCH.md)


def time_launches(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s_, e_ in ev:
        s_.record()
        fn()
        e_.record()
    torch.cuda.synchronize()
    return [s_.elapsed_time(e_) for s_, e_ in ev]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gj64", choices=sorted(WORKLOADS) + ["mixed"])
    ap.add_argument("--batch", type=int, default=None, help="matrices per GPU per step (default: 100 000 on one GPU; "
                                                             "with --gpus N > 1 the shard of --total-batch)")
    ap.add_argument("--total-batch", type=int, default=1_000_000, help="N > 1: matrices over ALL GPUs (BASELINE configs[3])")
    ap.add_argument("--kernel", default="auto", choices=["auto", "lds", "rowlane", "tile", "tilep", "row", "blocked"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other single-GPU workloads")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the timed all-gather of the results")
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
    # MATINV_BENCH_FORCE_DIST=1: a process group even for one rank (a world-size-1 "nccl" group is how the RCCL branch of this
    # file is exercised on a one-GPU box: tests/test_gpu_multirank.py)
    if world > 1 or os.environ.get("MATINV_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    coll_dev = device if backend == "nccl" else torch.device("cpu")

    api = importlib.import_module(PKG + ".api")
    shard = importlib.import_module(PKG + ".shard")
    # the queues of the mixed-size workload first: where their streams land among the hardware queues is decided when they are created
    mixed_queues = make_mixed_queues(device, 4) if (args.workload == "mixed" or (world == 1 and not args.no_others)) else []
    if args.workload == "mixed":
        args.queues = mixed_queues
        run_mixed(args, api, device, rank, world)
        if dist is not None:
            dist.destroy_process_group()
        return
    n, algo_name, desc = WORKLOADS[args.workload]
    algo = api.ALGO_GAUSS_JORDAN if algo_name == "gj" else api.ALGO_CHOLESKY
    kernel = {"auto": api.KERNEL_AUTO, "lds": api.KERNEL_LDS, "rowlane": api.KERNEL_ROWLANE,
              "tile": api.KERNEL_TILE, "tilep": api.KERNEL_TILEP, "row": api.KERNEL_ROW, "blocked": api.KERNEL_BLOCKED}[args.kernel]
    # which matrices are mine
    if args.batch is not None or world == 1:
        batch = args.batch if args.batch is not None else 100_000
        total_batch, lo, scaling = batch * world, rank * batch, "weak"
        sharding = f"{batch} matrices per GPU x {world} GPU(s), no data-path collective"
    else:
        total_batch = args.total_batch
        parts = shard.partition(total_batch, world, shard.packing_multiple(n))
        lo, hi = parts[rank]
        batch, scaling = hi - lo, "strong"
        sharding = (f"total batch {total_batch} block-partitioned over {world} GPUs by shard.partition "
                    f"(rank 0: {parts[0][1] - parts[0][0]} matrices), each rank generates its shard from seed 0x5EED + rank; "
                    f"no data-path collective")

    general = args.workload in GENERAL
    if general and kernel == api.KERNEL_AUTO and 16 < n <= 192:
        kernel = api.KERNEL_TILEP  # general input: the caller asks for partial pivoting (see other_workloads)
    a = (make_general if general else make_spd)(n, batch, 0x5EED + rank, device)
    x = torch.empty_like(a)
    info = torch.empty(max(batch, 1), dtype=torch.int32, device=device)

    def step():
        api.inverse_batched(a, n, algo, out=x, info=info, kernel=kernel, batch=batch)

    # what a plain device-to-device copy of the same bytes reaches on this box (SURVEY 8d asks for the fraction against
    # the measured copy bandwidth beside the nominal 8 TB/s); torch's copy kernel, same read+write byte count. Measured
    # before the timed region: it belongs to the set-up, and the card has then left its idle clocks when the steps start
    # (no host synchronisation between these copies and the warm-up steps: the events are read after the timed region)
    x.copy_(a)
    copy_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
    for s_, e_ in copy_ev:
        s_.record()
        x.copy_(a)
        e_.record()

    # HIP events on the stream the kernel is launched on (api passes torch's current stream to the C ABI). Created BEFORE the warm-up,
    # and the check of the warm-up's info codes is left until after the timed steps: the card falls back towards its idle clocks
    # within a millisecond or two without work and takes some 20 ms of load to come back (the same 20 steps measured 1.62 ms per step
    # behind a 2 ms gap of host work and 1.48 ms in a long run) -- between warm-up and timed region there is now only the prescribed
    # barrier + synchronisation
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for _ in range(args.warmup):
        step()
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
    step_ms = [s.elapsed_time(e) for s, e in ev]
    kern_ms = float(np.mean(step_ms))
    assert int(info[:batch].abs().sum()) == 0, "synthetic batch reported singular matrices"
    copy_gbs = 2 * a.numel() * a.element_size() / (float(np.median([s_.elapsed_time(e_) for s_, e_ in copy_ev])) * 1e-3) / 1e9

    gather = None
    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0].item()), float(t[1].item())
        ones = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(ones)  # through the data-path backend (RCCL): how many ranks really took part
        ranks_seen = int(ones.item())
        if not args.no_gather:
            # result reassembly on every rank (ONE RCCL all-gather over xGMI); host staging only in the gloo rehearsal. On the nccl
            # backend through the C ABI (matinv_allgather_shards on the library's own RCCL communicator; MATINV_GATHER=torch:
            # torch.distributed instead). The communicator is created and one small gather run BEFORE the timed one, so that the
            # figure is the collective, not the unique-id broadcast + ncclCommInitRank
            gimpl = os.environ.get("MATINV_GATHER", "c") if backend == "nccl" else "torch"
            tiny = torch.zeros(world * shard.packing_multiple(n) * n * n, dtype=x.dtype, device=x.device if backend == "nccl" else "cpu")
            shard.all_gather_shards(tiny[: shard.packing_multiple(n) * n * n], n, world * shard.packing_multiple(n), impl=gimpl)
            del tiny
            torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            full = shard.all_gather_shards(x if backend == "nccl" else x.cpu(), n, total_batch, impl=gimpl)
            torch.cuda.synchronize()
            gms = (time.perf_counter() - g0) * 1e3
            assert full.numel() == total_batch * n * n
            mine_ok = bool(torch.equal(full[lo * n * n:(lo + batch) * n * n].to(x.device), x))
            recv = (total_batch - batch) * n * n * x.element_size()  # bytes this rank received from the others
            tg = torch.tensor([gms], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            gms = float(tg.item())
            gather = {"allgather_ms": gms, "allgather_GBs_per_rank_in": recv / (gms * 1e-3) / 1e9,
                      "gathered_bytes_per_rank": total_batch * n * n * x.element_size(),
                      "xgmi_bound_GBs_per_rank_in": XGMI_LINK_GBS * min(world - 1, 7),
                      "xgmi_per_link_GBs": XGMI_LINK_GBS, "backend": backend, "impl": gimpl,
                      "gather_equals_local_result": mine_ok}
            del full
        else:
            gather = {"backend": backend}
        gather["n_ranks_seen"] = ranks_seen

    # quick correctness guard on the timed output (not timed): residual of a few matrices
    am = a.view(batch, n, n)[:8]
    xm = x.view(batch, n, n)[:8]
    resid = float((torch.bmm(am, xm) - torch.eye(n, dtype=a.dtype, device=device)).abs().max())
    if os.environ.get("MATINV_BENCH_NO_RESIDUAL") != "1":  # (profiling builds that move the bytes without inverting)
        assert resid < (1e-9 if general else 1e-11) * n, f"residual {resid}"

    others, end_to_end, mixed = {}, None, None
    if rank == 0 and world == 1 and not args.no_others:
        # the other single-GPU workloads, a few launches each (same timing method), so that every path -- the weak ones
        # included -- is driver-timed AND residual-checked each round
        del a, x

        def residual(a_, x_, n_, b_, k=64):
            """max_k || A_k X_k - I ||_inf-entry over k matrices spread through the batch (first, last and evenly between);
            the flat batches are column-major, i.e. the views are A^T and X^T: (A^T)(X^T) = (X A)^T -- the left residual"""
            idx = torch.linspace(0, b_ - 1, min(k, b_), device=device).long()
            am_ = a_.view(b_, n_, n_)[idx]
            xm_ = x_.view(b_, n_, n_)[idx]
            return float((torch.bmm(am_, xm_) - torch.eye(n_, dtype=a_.dtype, device=device)).abs().max())

        def one(wname, b2, reps=None, natural_first=False):
            n2, algo2_name, _ = WORKLOADS[wname]
            algo2 = api.ALGO_GAUSS_JORDAN if algo2_name == "gj" else api.ALGO_CHOLESKY
            gen2 = wname in GENERAL
            f32 = wname in F32
            dt2 = torch.float32 if f32 else torch.float64
            a2 = (make_general if gen2 else make_spd)(n2, b2, 0x5EED + 17 * n2, device, dt2)
            x2 = torch.empty_like(a2)
            i2 = torch.empty(b2, dtype=torch.int32, device=device)
            # general input: the caller asks for partial pivoting (MATINV_KERNEL_TILEP; what the reference's inverse_lu_cuda_batched_*
            # names map to) -- the default Gauss-Jordan policy tries the natural order first, per matrix (deterministic), and a
            # general batch would pay both kernels: that figure is the "@natural_first" entry below
            k2 = api.KERNEL_TILEP if (gen2 and 16 < n2 <= 192 and not natural_first) else api.KERNEL_AUTO
            # enough launches for the card to leave its idle clocks (see the headline loop): ~30 ms of warm-up, ~30 ms timed
            reps = reps or (20 if n2 <= 192 else 5)
            for _ in range(max(3, reps)):
                api.inverse_batched(a2, n2, algo2, out=x2, info=i2, batch=b2, kernel=k2)
            # a caller that has seen one batch of this size class complete: the launcher's reject-rate hint is in, and under the default
            # policy a general batch then takes the screening pass (csrc/tile_screen.hpp) -- results do not depend on it, the cost does
            torch.cuda.synchronize()
            ms2 = float(np.mean(time_launches(lambda: api.inverse_batched(a2, n2, algo2, out=x2, batch=b2, kernel=k2), reps)))
            r1, r2 = rooflines(algo2_name, n2, b2, ms2, 4 if f32 else 8)
            kname2 = api.kernel_name(algo2, api.F32 if f32 else api.F64, n2, k2)
            res2 = residual(a2, x2, n2, b2)
            # general U(0,1) matrices have condition numbers of 1e3 .. 1e6 at these sizes: the residual bound scales with it
            tol2 = ((2e-1 if gen2 else 1e-4) if f32 else (1e-7 if gen2 else 1e-11)) * n2
            assert res2 < tol2, f"{wname}: residual {res2} >= {tol2}"
            d = {"kernel": kname2, "dtype": "f32" if f32 else "f64", "batch": b2, "kernel_ms": ms2, "input": "general U(0,1)" if gen2 else "SPD",
                 "inversions_per_s": b2 / (ms2 * 1e-3), "bound": r1["bound"], "frac": r1["frac"],
                 "achieved": r1["achieved"], "unit": r1["unit"],
                 "other_bound_frac": None if r2 is None else r2["frac"], "singular_reported": int((i2 != 0).sum()),
                 "residual_max_64": res2, "residual_tol": tol2}
            del a2, x2
            return d

        for wname in ("gj16", "chol64", "gj32", "gj24", "gj50", "gj8", "gj128", "gj64g", "gj32g", "gj96g", "gj128g", "gj192g", "chol128", "chol130",
                      "chol144", "chol192", "gj256g", "chol256", "gj1024g", "chol1024", "gj64_f32", "chol64_f32", "gj128_f32", "gj64g_f32"):
            if wname == args.workload:
                continue
            n2 = WORKLOADS[wname][0]
            others[wname] = one(wname, 100_000 if n2 <= 64 else (25_000 if n2 <= 128 else (5_000 if n2 <= 192 else (3_000 if n2 <= 256 else 256))))
        # the small sizes again at a batch whose working set is beyond the 256 MiB Infinity Cache (100 k x 16^2 x 2 = 0.4 GB is
        # not: FETCH_SIZE counts those hits, so the 100 k figure may flatter them)
        for wname, b2 in (("gj16", 1_000_000), ("gj8", 2_000_000)):
            others[wname + f"@{b2 // 1_000_000}M"] = one(wname, b2)
        # what a GENERAL batch costs under the default (per-matrix deterministic) policy: natural-order attempt + pivoting kernel
        others["gj64g@natural_first"] = one("gj64g", 100_000, natural_first=True)
        others["gj128g@natural_first"] = one("gj128g", 25_000, natural_first=True)

        # the fused mean pipeline (add -> inv -> gemv -> dot, gauss_bench.cu:127-265 of the reference) at two fp64 sizes around the old
        # 128 -> 130 cliff: kernel-only, device-resident items, checked against torch on a few items
        def one_mean(n2, b2, reps=20):
            g = torch.Generator(device=device).manual_seed(0x5EED + 31 * n2)
            r = torch.rand((b2, n2, n2), generator=g, dtype=torch.float64, device=device)
            B2 = r + r.transpose(1, 2)
            B2.diagonal(dim1=1, dim2=2).add_(float(n2))
            v = torch.rand((3, b2, n2), generator=g, dtype=torch.float64, device=device)
            a2, c2, d2 = v[0].reshape(-1).contiguous(), v[1].reshape(-1).contiguous(), v[2].reshape(-1).contiguous()
            Bf = B2.reshape(-1).contiguous()
            for _ in range(reps):
                m = api.calcluateMean(n2, a2, Bf, c2, d2)
            torch.cuda.synchronize()
            ms2 = float(np.mean(time_launches(lambda: api.calcluateMean(n2, a2, Bf, c2, d2), reps)))
            k = min(8, b2)
            M = B2[:k] + torch.diag_embed(v[1, :k])
            want = torch.einsum("bi,bi->b", v[0, :k], torch.linalg.solve(M, v[2, :k].unsqueeze(-1)).squeeze(-1))
            err = float((m[:k] - want).abs().max())
            assert err < 1e-11, f"mean{n2}: {err}"
            alg = b2 * (n2 * n2 + 3 * n2 + 1) * 8
            return {"kernel": "fused mean pipeline", "dtype": "f64", "batch": b2, "kernel_ms": ms2, "items_per_s": b2 / (ms2 * 1e-3), "bound": "hbm",
                    "achieved": alg / (ms2 * 1e-3) / 1e9, "unit": "GB/s", "frac": alg / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "max_abs_err_8_items": err}

        others["mean128"] = one_mean(128, 25_000)
        others["mean130"] = one_mean(130, 25_000)

        # End to end, as the reference times it (src/inverse_bench.c:187-200: TIMER around the host-pointer call; H2D + kernel + D2H
        # inside, caller's pageable memory in and out). Never `value`.
        n_e, b_e = 64, 100_000
        rng = np.random.default_rng(1)
        r_ = rng.random((2_000, n_e, n_e))
        ha = np.tile((r_ + r_.transpose(0, 2, 1) + n_e * np.eye(n_e)).reshape(-1), b_e // 2_000)
        hx = np.empty_like(ha)
        api.inverse_gauss_batched_gpu(n_e, ha, hx, b_e)  # warm-up: staging pool, page faults of the output
        ts = []
        for _ in range(3):
            t_ = time.perf_counter()
            api.inverse_gauss_batched_gpu(n_e, ha, hx, b_e)
            ts.append(time.perf_counter() - t_)
        t_e = float(np.median(ts))
        k_ = b_e - 1
        he = np.abs(ha[k_ * n_e * n_e:].reshape(n_e, n_e) @ hx[k_ * n_e * n_e:].reshape(n_e, n_e) - np.eye(n_e)).max()
        assert he < 1e-11 * n_e, f"end-to-end residual {he}"
        end_to_end = {"call": "inverse_gauss_batched_gpu(handle, 64, As, aInvs, 100000): host pointers, pageable memory, H2D + kernel + D2H inside (src/inverse_bench.c:187-200)",
                      "ms": t_e * 1e3, "inversions_per_s": b_e / t_e, "host_link_GBs_both_directions": 2 * ha.nbytes / t_e / 1e9,
                      "host_link_spec_GBs_per_direction": 63.0, "devices": int(os.environ.get("MATINV_DEVICES", "1")),
                      "residual_last_matrix": float(he), "calls_timed": len(ts)}
        ndev_e = api.device_count()
        if ndev_e > 1:
            # the same batch block-partitioned over every device of the node, one host thread and one host link per device
            # (matinv_inverse_batched_host_multi; what MATINV_DEVICES=N gives the reference-named entry points)
            api.inverse_batched_host_multi(ha, n_e, api.ALGO_GAUSS_JORDAN, nshards=ndev_e)
            tm = []
            for _ in range(3):
                t_ = time.perf_counter()
                hx_m, _info_m = api.inverse_batched_host_multi(ha, n_e, api.ALGO_GAUSS_JORDAN, nshards=ndev_e)
                tm.append(time.perf_counter() - t_)
            t_m = float(np.median(tm))
            end_to_end["all_devices"] = {"call": f"matinv_inverse_batched_host_multi(..., nshards={ndev_e})", "devices": ndev_e, "ms": t_m * 1e3,
                                         "inversions_per_s": b_e / t_m, "host_link_GBs_both_directions": 2 * ha.nbytes / t_m / 1e9,
                                         "same_bits_as_one_device": bool(np.array_equal(np.asarray(hx_m).reshape(-1), hx))}
            del hx_m
        del ha, hx

        # BASELINE configs[4] on this GPU (the C queue), a short run of the same code path as --workload mixed
        keys = ("value", "unit", "ms_per_step", "steps", "dtype", "host_ms_per_step", "host_share", "means_finite", "per_bin", "config")
        m2 = mixed_result(argparse.Namespace(steps=20, warmup=4, inflight=4, queues=mixed_queues), api, device, rank, world)
        m1 = mixed_result(argparse.Namespace(steps=10, warmup=3, inflight=1, queues=mixed_queues), api, device, rank, world)
        mixed = {k: m2[k] for k in keys}
        mixed["one_flush_at_a_time"] = {k: m1[k] for k in ("value", "ms_per_step", "host_ms_per_step", "host_share")}

    if rank == 0:
        value = total_batch * args.steps / elapsed
        kname = api.kernel_name(algo, api.F64, n, kernel)
        roof, roof_other = rooflines(algo_name, n, batch, kern_ms)
        traffic, traffic_src = load_traffic(kname, n, batch)
        roof.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": kname, "kernel_ms": kern_ms,
                     "kernel_ms_first_min_last": [step_ms[0], min(step_ms), step_ms[-1]],
                     "units_per_launch": batch, "per": "GPU"})
        hbm_side = roof if roof["bound"] == "hbm" else roof_other
        hbm_side["guide_copy_GBs"] = HBM_GUIDE_ACHIEVABLE_GBS
        hbm_side["frac_of_guide_copy"] = hbm_side["achieved"] / HBM_GUIDE_ACHIEVABLE_GBS
        hbm_side["measured_copy_GBs"] = copy_gbs
        hbm_side["frac_of_measured_copy"] = hbm_side["achieved"] / copy_gbs
        if roof_other is not None:
            roof["other_bound"] = roof_other
        out = {
            "metric": "matrix inversions/s", "value": value, "unit": "inversions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc.replace('batch', str(total_batch))}", "n": n,
                       "total_batch": total_batch, "batch_per_gpu": batch, "algorithm": algo_name, "kernel": kname,
                       "input": "general U(0,1)" if general else "SPD: R + R^T + n I", "sharding": sharding},
            "roofline": roof,
        }
        if gather is not None:
            out["gather"] = gather
            if "allgather_ms" in gather:
                out["allgather_ms"] = gather["allgather_ms"]
                # the inversions of one step followed by the reassembly of their results on every rank (never `value`)
                out["value_with_gather"] = total_batch / (elapsed / args.steps + gather["allgather_ms"] * 1e-3)
        if others:
            out["other_workloads"] = others
        if end_to_end is not None:
            out["end_to_end"] = end_to_end
        if mixed is not None:
            out["mixed"] = mixed
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, algo_name)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
