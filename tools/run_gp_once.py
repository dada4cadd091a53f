"""One fused-mean call at a given size (for rocprofv3 kernel traces): python tools/run_gp_once.py n batch [f32|f64] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
api = importlib.import_module("cuda-matrix-inversion_amd.api")
n, batch = int(sys.argv[1]), int(sys.argv[2])
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
g = torch.Generator(device="cuda").manual_seed(n)
R = torch.rand(batch, n, n, generator=g, device="cuda", dtype=dtype)
B = (R @ R.transpose(1, 2) + n * torch.eye(n, device="cuda", dtype=dtype)).reshape(-1)
a, c, d = (torch.rand(batch * n, generator=g, device="cuda", dtype=dtype) for _ in range(3))
for _ in range(reps):
    out = api.calcluateMean(n, a, B, c, d)
torch.cuda.synchronize()
print(out[:2].tolist())
