"""Kernel-only rate of the fused mean pipeline for a list of sizes: python tools/time_gp_sizes.py [f64|f32] n1 n2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
api = importlib.import_module("cuda-matrix-inversion_amd.api")
dtype = torch.float32 if sys.argv[1] == "f32" else torch.float64
for n in map(int, sys.argv[2:]):
    batch = max(256, min(100_000, int(1.6e9 / (n * n * 8))))
    g = torch.Generator(device="cuda").manual_seed(n)
    r = torch.rand(batch, n, n, generator=g, device="cuda", dtype=dtype)
    B = (r + r.transpose(1, 2) + n * torch.eye(n, device="cuda", dtype=dtype)).reshape(-1).contiguous()
    a, c, d = (torch.rand(batch * n, generator=g, device="cuda", dtype=dtype) for _ in range(3))
    out = torch.empty(batch, device="cuda", dtype=dtype)
    for _ in range(2):
        api.calcluateMean(n, a, B, c, d, Means=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in ev:
        s.record(); api.calcluateMean(n, a, B, c, d, Means=out); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)[2]
    print(f"n={n:4d} batch={batch:6d} {ms:8.3f} ms  {batch / ms * 1e3:12.4e} items/s  {batch * (n * n + 3 * n + 1) * B.element_size() / ms / 1e6:8.1f} GB/s (alg.)", flush=True)
