#!/usr/bin/env python3
"""A/B: the same launch with and without the info array, per-launch times (ms)."""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
n, batch = 64, 100000
r = torch.rand((batch, n, n), dtype=torch.float64, device='cuda')
a = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=torch.float64, device='cuda')).reshape(-1).contiguous()
x = torch.empty_like(a); info = torch.empty(batch, dtype=torch.int32, device='cuda')
def run(use_info, reps=12):
    for _ in range(3): api.inverse_batched(a, n, 0, out=x, info=info if use_info else None)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); api.inverse_batched(a, n, 0, out=x, info=info if use_info else None); e.record()
    torch.cuda.synchronize()
    return [round(s.elapsed_time(e), 3) for s, e in ev]
for k in range(2):
    print("no info:", run(False)); print("info   :", run(True))
