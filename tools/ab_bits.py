#!/usr/bin/env python3
"""Do two Gauss-Jordan kernel choices give the same BITS on the same batch?  python tools/ab_bits.py n [f64|f32] [spd|general]
Run once per (policy / kernel) in separate processes (the switches are read once per process); each run writes a hash."""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
api = importlib.import_module("cuda-matrix-inversion_amd.api")
n = int(sys.argv[1])
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
kind = sys.argv[3] if len(sys.argv) > 3 else "spd"
batch = 2000
g = torch.Generator(device="cuda").manual_seed(7 * n)
r = torch.rand(batch, n, n, generator=g, device="cuda", dtype=dtype)
a = (r if kind == "general" else r + r.transpose(1, 2) + n * torch.eye(n, device="cuda", dtype=dtype)).reshape(-1).contiguous()
kernel = {"": api.KERNEL_AUTO, "tile": api.KERNEL_TILE, "tilep": api.KERNEL_TILEP}[os.environ.get("MATINV_TIME_KERNEL", "")]
x = api.inverse_batched(a, n, api.ALGO_GAUSS_JORDAN, batch=batch, kernel=kernel)
torch.cuda.synchronize()
h = hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"n={n} {sys.argv[2] if len(sys.argv) > 2 else 'f64'} {kind} kernel={os.environ.get('MATINV_TIME_KERNEL','auto')} "
      f"policy={os.environ.get('MATINV_GJ_POLICY','adaptive')}: sha {h}  stats {api.tile_stats()}")
