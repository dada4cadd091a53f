#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-style host-pointer entry point (alloc + H2D + kernel + D2H + free inside the
call, exactly what inverse_bench times): python tools/time_host_api.py n batch"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
n = int(sys.argv[1]); batch = int(sys.argv[2])
rng = np.random.default_rng(0)
r = rng.random((min(batch, 2000), n, n))
a = np.tile((r + r.transpose(0, 2, 1) + n * np.eye(n)).reshape(-1), batch // r.shape[0])
batch = a.size // (n * n)
out = np.empty_like(a)
api.inverse_gauss_batched_gpu(n, a, out, batch)  # warm-up (device init)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); api.inverse_gauss_batched_gpu(n, a, out, batch); ts.append(time.perf_counter() - t0)
t = min(ts)
print(f"inverse_gauss_batched_gpu n={n} batch={batch}: {t*1e3:.1f} ms  {batch/t:.3e} inv/s  ({2*a.nbytes/t/1e9:.1f} GB/s over the host link, pageable memory)")
