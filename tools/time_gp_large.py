"""Time the fused mean pipeline at large n (blocked multi-launch path vs the one-workgroup GLOBAL kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
import torch
api = importlib.import_module("cuda-matrix-inversion_amd.api")

for dtype in (torch.float32, torch.float64):
    for n, batch in ((128, 2048), (100, 2048), (200, 512), (256, 256), (512, 64), (512, 8), (1024, 32), (1024, 8)):
        g = torch.Generator(device="cuda").manual_seed(n)
        R = torch.rand(batch, n, n, generator=g, device="cuda", dtype=dtype)
        B = (R @ R.transpose(1, 2) + n * torch.eye(n, device="cuda", dtype=dtype)).reshape(-1)
        a, c, d = (torch.rand(batch * n, generator=g, device="cuda", dtype=dtype) for _ in range(3))
        out = torch.empty(batch, device="cuda", dtype=dtype)
        for _ in range(2):
            api.calcluateMean(n, a, B, c, d, Means=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            api.calcluateMean(n, a, B, c, d, Means=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # reference value via torch
        k = min(batch, 4)  # value check on the host (numpy) for a few items
        M = (B.reshape(batch, n, n)[:k] + torch.diag_embed(c.reshape(batch, n)[:k])).double().cpu().numpy()
        want = np.einsum("bi,bi->b", a.reshape(batch, n)[:k].double().cpu().numpy(),
                         np.linalg.solve(M, d.reshape(batch, n)[:k].double().cpu().numpy()[..., None])[..., 0])
        err = (np.abs(out[:k].double().cpu().numpy() - want) / np.abs(want)).max()
        print(f"{str(dtype):14s} n={n:5d} batch={batch:4d}  {dt*1e3:9.3f} ms  {batch/dt:10.1f} items/s  relerr {err:.2e}  "
              f"blocked={os.environ.get('MATINV_GP_BLOCKED','1')}", flush=True)
