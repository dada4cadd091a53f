#!/usr/bin/env python3
"""Kernel-only timing of the pivoting tile kernel on general U(0,1) and on SPD input: python tools/time_tilep.py [n ...]"""
import importlib, sys
import torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
sizes = [int(x) for x in sys.argv[1:]] or [32, 64]
def timeit(a, n, kern):
    x = torch.empty_like(a)
    for _ in range(2): api.inverse_batched(a, n, 0, out=x, kernel=kern)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for s, e in ev:
        s.record(); api.inverse_batched(a, n, 0, out=x, kernel=kern); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[3]
for n in sizes:
    for dt in (torch.float64, torch.float32):
        batch = max(1000, int(100000 * (64 * 64) / (n * n)) // (2 if n < 32 else 1))
        g = torch.rand((batch * n * n,), dtype=dt, device='cuda')
        r = torch.rand((batch, n, n), dtype=dt, device='cuda')
        spd = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=dt, device='cuda')).reshape(-1).contiguous()
        for nm, a in (('general', g), ('spd', spd)):
            for kern, kn in ((api.KERNEL_TILEP, 'tilep'), (api.KERNEL_TILE, 'tile')):
                ms = timeit(a, n, kern)
                print(f"n={n} {str(dt)[6:]} {nm:8s} {kn:6s}: {ms:.3f} ms {batch/ms*1e3:.3e} inv/s frac {2*n*n*a.element_size()*batch/ms/1e6/8000:.3f}", flush=True)
