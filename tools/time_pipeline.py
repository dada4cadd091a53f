#!/usr/bin/env python3
"""Time the fused GP pipeline kernels: python tools/time_pipeline.py n batch [dtype f64|f32]"""
import importlib, sys
import torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
n = int(sys.argv[1]); batch = int(sys.argv[2]); dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == 'f32') else torch.float64
r = torch.rand((batch, n, n), dtype=dt, device='cuda')
B = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=dt, device='cuda')).reshape(-1).contiguous()
a, c, d = (torch.rand(batch * n, dtype=dt, device='cuda') for _ in range(3))
e = torch.rand(batch, dtype=dt, device='cuda')
out = torch.empty(batch, dtype=dt, device='cuda')
for name, fn, last in (("mean", api.calcluateMean, d), ("variance", api.calcluateVariance, e)):
    for _ in range(3): fn(n, a, B, c, last, out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, t in ev:
        s.record(); fn(n, a, B, c, last, out); t.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(t) for s, t in ev)[5]
    byt = (n * n + 3 * n + 1) * B.element_size() * batch
    print(f"{name} n={n} batch={batch} {dt}: {ms:.3f} ms  {batch/ms*1e3:.3e} items/s  {byt/ms/1e6:.0f} GB/s algorithmic")
