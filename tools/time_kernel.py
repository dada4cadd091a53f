#!/usr/bin/env python3
"""Time one kernel launch (no correctness check): python tools/time_kernel.py n batch [algo] [reps] [f32]. Honors MATINV_LIB."""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
n = int(sys.argv[1]); batch = int(sys.argv[2]); algo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == 'f32') else torch.float64
r = torch.rand((batch, n, n), dtype=dt, device='cuda')
a = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=dt, device='cuda')).reshape(-1).contiguous()
x = torch.empty_like(a)
for _ in range(3): api.inverse_batched(a, n, algo, out=x)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for s, e in ev:
    s.record(); api.inverse_batched(a, n, algo, out=x); e.record()
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in ev)[len(ev)//2]
print(f"n={n} batch={batch} algo={algo} median {ms:.3f} ms  {batch/ms*1e3:.3e} inv/s  {2*n*n*a.element_size()*batch/ms/1e6:.0f} GB/s {dt}")
