#!/usr/bin/env python3
"""Host-side cost of one fused-pipeline call (time until the call returns, nothing waited for) vs its device time."""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
for n, cnt in ((1024, 8), (512, 32), (128, 2048), (32, 16384)):
    dt = torch.float32
    r = torch.rand((cnt, n, n), dtype=dt, device='cuda')
    B = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=dt, device='cuda')).reshape(-1).contiguous()
    a, c, d = (torch.rand(cnt * n, dtype=dt, device='cuda') for _ in range(3))
    out = torch.empty(cnt, dtype=dt, device='cuda')
    for _ in range(3): api.calcluateMean(n, a, B, c, d, Means=out)
    torch.cuda.synchronize()
    host = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); api.calcluateMean(n, a, B, c, d, Means=out); host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): api.calcluateMean(n, a, B, c, d, Means=out)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 10
    print(f"n={n} items={cnt}: host {sorted(host)[5]*1e3:.3f} ms per call (idle device), {tot*1e3:.3f} ms per call back to back")
