"""How many matrices of an SPD batch does the first-pass kernel hand to its fallback? (MATINV_DEBUG_REJECTS=1 must be set)
usage: python tools/check_rejects.py f32|f64 chol|mean n..."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
api = importlib.import_module("cuda-matrix-inversion_amd.api")
dt = torch.float32 if sys.argv[1] == "f32" else torch.float64
for n in map(int, sys.argv[3:]):
    batch = 64
    g = torch.Generator(device="cuda").manual_seed(n)
    r = torch.rand(batch, n, n, generator=g, device="cuda", dtype=dt)
    a = (r + r.transpose(1, 2) + n * torch.eye(n, device="cuda", dtype=dt)).reshape(-1).contiguous()
    api.debug_rejects(reset=True)
    if sys.argv[2] == "chol":
        api.inverse_batched(a, n, api.ALGO_CHOLESKY, batch=batch)
    else:
        v = [torch.rand(batch * n, generator=g, device="cuda", dtype=dt) for _ in range(3)]
        api.calcluateMean(n, v[0], a, v[1], v[2])
    torch.cuda.synchronize()
    print(f"n={n} {sys.argv[1]} {sys.argv[2]}: rejected {api.debug_rejects(reset=True)} of {batch}")
