#!/usr/bin/env python3
"""Pivoting tile kernel vs the CPU oracle on general U(0,1) matrices + timing: python tools/check_tilep.py [n ...]"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
import oracle
api = importlib.import_module('cuda-matrix-inversion_amd.api')
sizes = [int(x) for x in sys.argv[1:]] or [16, 32, 48, 64, 20, 40, 50, 60, 5]
rng = np.random.default_rng(1)
for dt, tol in ((np.float64, 1e-10), (np.float32, 2e-2)):
    for n in sizes:
        batch = 257
        a = rng.random((batch, n, n)).astype(dt)
        d = torch.from_numpy(a.reshape(-1)).cuda()
        info = torch.full((batch,), -7, dtype=torch.int32, device='cuda')
        x = api.inverse_batched(d, n, api.ALGO_GAUSS_JORDAN, info=info, kernel=api.KERNEL_TILEP)
        torch.cuda.synchronize()
        want, _ = oracle.inverse_batched(a.astype(np.float64).reshape(-1), n, oracle.ALGO_GJ_PIVOT)
        got = x.cpu().numpy().astype(np.float64).reshape(batch, n, n)
        want = want.reshape(batch, n, n)
        den = np.maximum(np.abs(want), 1e-3 * np.abs(want).max(axis=(1, 2), keepdims=True))
        err = (np.abs(got - want) / den).max()
        res = np.abs(np.einsum('bij,bjk->bik', a.astype(np.float64), got) - np.eye(n)).max()
        ok = err < tol and int(info.abs().sum()) == 0
        print(f"{np.dtype(dt).name} n={n:3d}: max rel err {err:.2e} residual {res:.2e} info!=0 {int((info != 0).sum())} {'ok' if ok else 'FAIL'}", flush=True)
# singular + NaN handling
n = 32
a = rng.random((8, n, n)); a[3, :, 5] = 0.0; a[6, 2, 2] = np.nan
d = torch.from_numpy(a.reshape(-1)).cuda(); info = torch.zeros(8, dtype=torch.int32, device='cuda')
x = api.inverse_batched(d, n, 0, info=info, kernel=api.KERNEL_TILEP); torch.cuda.synchronize()
print("singular info:", info.cpu().numpy())
for n in ():
    for dt in (torch.float64, torch.float32):
        batch = 100000 if n == 64 else 200000
        a = torch.rand((batch * n * n,), dtype=dt, device='cuda'); x = torch.empty_like(a)
        for kern, nm in ((api.KERNEL_TILEP, 'tilep'), (api.KERNEL_ROW, 'row'), (api.KERNEL_TILE, 'tile')):
            for _ in range(2): api.inverse_batched(a, n, 0, out=x, kernel=kern)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for s, e in ev:
                s.record(); api.inverse_batched(a, n, 0, out=x, kernel=kern); e.record()
            torch.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in ev)[2]
            print(f"general n={n} {dt} {nm}: {ms:.3f} ms {batch/ms*1e3:.3e} inv/s frac {2*n*n*a.element_size()*batch/ms/1e6/8000:.3f}", flush=True)
