#!/usr/bin/env python3
"""Time Gauss-Jordan on GENERAL (pivoting-required) U(0,1) matrices: python tools/time_general.py n batch [kernel] [f32]"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
n = int(sys.argv[1]); batch = int(sys.argv[2]); kern = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == 'f32') else torch.float64
a = torch.rand((batch * n * n,), dtype=dt, device='cuda')
x = torch.empty_like(a); info = torch.empty(batch, dtype=torch.int32, device='cuda')
for _ in range(2): api.inverse_batched(a, n, 0, out=x, info=info, kernel=kern)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for s, e in ev:
    s.record(); api.inverse_batched(a, n, 0, out=x, info=info, kernel=kern); e.record()
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in ev)[2]
am = a.view(batch, n, n)[:64].double(); xm = x.view(batch, n, n)[:64].double()
res = float((torch.bmm(am, xm) - torch.eye(n, dtype=torch.float64, device='cuda')).abs().max())
print(f"general n={n} batch={batch} kernel={kern} {dt}: {ms:.3f} ms  {batch/ms*1e3:.3e} inv/s  residual {res:.2e}  info!=0: {int((info != 0).sum())}")
