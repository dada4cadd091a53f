#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU: 1 000 000 x 64x64 fp64 (4.1e9 elements per operand -- beyond the reference's int
indexing, src/gauss/batched_invert.cu:130). Checks the residual of matrices at the very end of the batch."""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
n, batch = 64, 1_000_000
a = torch.empty(batch * n * n, dtype=torch.float64, device='cuda')
chunk = 50_000
g = torch.Generator(device='cuda').manual_seed(1)
for i in range(0, batch, chunk):
    r = torch.rand((chunk, n, n), generator=g, dtype=torch.float64, device='cuda')
    r = r + r.transpose(1, 2); r.diagonal(dim1=1, dim2=2).add_(float(n))
    a[i * n * n:(i + chunk) * n * n] = r.reshape(-1)
x = torch.empty_like(a); info = torch.empty(batch, dtype=torch.int32, device='cuda')
api.inverse_batched(a, n, 0, out=x, info=info); torch.cuda.synchronize()
t0 = time.perf_counter(); api.inverse_batched(a, n, 0, out=x, info=info); torch.cuda.synchronize(); dt = time.perf_counter() - t0
eye = torch.eye(n, dtype=torch.float64, device='cuda')
worst = 0.0
for lo in (0, batch // 2, batch - 1000):
    am = a.view(batch, n, n)[lo:lo + 1000]; xm = x.view(batch, n, n)[lo:lo + 1000]
    worst = max(worst, float((torch.bmm(am, xm) - eye).abs().max()))
print(f"1M x 64x64 f64: {dt*1e3:.1f} ms  {batch/dt:.3e} inv/s  {2*a.numel()*8/dt/1e12:.2f} TB/s  info!=0: {int((info!=0).sum())}  max residual {worst:.2e}")
assert worst < 1e-12 and int((info != 0).sum()) == 0
