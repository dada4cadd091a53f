#!/usr/bin/env python3
"""Gauss-Jordan for 16 < n <= 32 through the TILE family vs the CPU oracle + timing; run once with MATINV_ROWLANE2=2 (natural
pass = two-rows-per-lane kernel) and once with MATINV_ROWLANE2=0 (natural pass = MFMA tile kernel):
MATINV_ROWLANE2=2 python tools/check_rowlane2.py [n ...]"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
import oracle
api = importlib.import_module('cuda-matrix-inversion_amd.api')
sizes = [int(x) for x in sys.argv[1:]] or [17, 20, 23, 24, 25, 31, 32]
rng = np.random.default_rng(1)
for dt, tol in ((np.float64, 1e-10), (np.float32, 2e-2)):
    for n in sizes:
        for kind in ("dominant", "general"):
            batch = 259
            a = rng.random((batch, n, n))
            if kind == "dominant":
                a = a + a.transpose(0, 2, 1) + n * np.eye(n)
            a = a.astype(dt)
            d = torch.from_numpy(a.reshape(-1)).cuda()
            info = torch.full((batch,), -7, dtype=torch.int32, device='cuda')
            x = api.inverse_batched(d, n, api.ALGO_GAUSS_JORDAN, info=info, kernel=api.KERNEL_TILE)
            torch.cuda.synchronize()
            want, _ = oracle.inverse_batched(a.astype(np.float64).reshape(-1), n, oracle.ALGO_GJ_PIVOT)
            got = x.cpu().numpy().astype(np.float64).reshape(batch, n, n)
            want = want.reshape(batch, n, n)
            den = np.maximum(np.abs(want), 1e-3 * np.abs(want).max(axis=(1, 2), keepdims=True))
            err = (np.abs(got - want) / den).max()
            ok = err < tol and int(info.abs().sum()) == 0
            print(f"{np.dtype(dt).name} n={n:3d} {kind:8s}: max rel err {err:.2e} info!=0 {int((info != 0).sum())} {'ok' if ok else 'FAIL'}", flush=True)
# singular + NaN handling: info = k+1 of the first column without a pivot (as the oracle), NaN fill
n = 24
a = rng.random((8, n, n)); a[3, :, 5] = 0.0; a[6, 2, 2] = np.nan
want, winfo = oracle.inverse_batched(a.reshape(-1), n, oracle.ALGO_GJ_PIVOT)
d = torch.from_numpy(a.reshape(-1)).cuda(); info = torch.zeros(8, dtype=torch.int32, device='cuda')
x = api.inverse_batched(d, n, 0, info=info, kernel=api.KERNEL_TILE); torch.cuda.synchronize()
print("singular info:", info.cpu().numpy(), "oracle:", winfo, "nan rows:", torch.isnan(x.reshape(8, -1)).all(dim=1).cpu().numpy())
import os
print('MATINV_ROWLANE2 =', os.environ.get('MATINV_ROWLANE2'), 'natural kernel at n=24:', api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 24, api.KERNEL_TILE))
for n in (17, 20, 24, 25, 28, 32):
    for dt in (torch.float64, torch.float32):
        batch = 200000
        for kind in ("dominant", "general"):
            a = torch.rand((batch, n, n), dtype=dt, device='cuda')
            if kind == "dominant":
                a = a + a.transpose(1, 2) + n * torch.eye(n, dtype=dt, device='cuda')
            a = a.reshape(-1).contiguous(); x = torch.empty_like(a)
            for kern, nm in ((api.KERNEL_TILE, 'tile'), (api.KERNEL_TILEP, 'tilep')):
                for _ in range(3): api.inverse_batched(a, n, 0, out=x, kernel=kern)
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
                for s, e in ev:
                    s.record(); api.inverse_batched(a, n, 0, out=x, kernel=kern); e.record()
                torch.cuda.synchronize()
                ms = sorted(s.elapsed_time(e) for s, e in ev)[3]
                print(f"{kind:8s} n={n} {str(dt)[6:]} {nm:8s}: {ms:.3f} ms {batch/ms*1e3:.3e} inv/s frac {2*n*n*a.element_size()*batch/ms/1e6/8000:.3f}", flush=True)
