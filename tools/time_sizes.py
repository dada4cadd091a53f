"""Kernel-only rate of the automatic inversion path for a list of sizes: python tools/time_sizes.py [f64|f32] [gj|chol] n1 n2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
api = importlib.import_module("cuda-matrix-inversion_amd.api")
dtype = torch.float32 if sys.argv[1] == "f32" else torch.float64
algo = api.ALGO_CHOLESKY if sys.argv[2] == "chol" else api.ALGO_GAUSS_JORDAN
kernel = {"": api.KERNEL_AUTO, "lds": api.KERNEL_LDS, "row": api.KERNEL_ROW, "tile": api.KERNEL_TILE, "global": api.KERNEL_GLOBAL,
          "blocked": api.KERNEL_BLOCKED, "tilep": api.KERNEL_TILEP}[os.environ.get("MATINV_TIME_KERNEL", "")]
general = os.environ.get("MATINV_TIME_GENERAL", "") == "1"  # U(0,1) non-symmetric input (needs pivoting)
for n in map(int, sys.argv[3:]):
    batch = int(os.environ.get("MATINV_TIME_BATCH", 0)) or max(256, min(100_000, int(1.6e9 / (n * n * 8))))
    g = torch.Generator(device="cuda").manual_seed(n)
    r = torch.rand(batch, n, n, generator=g, device="cuda", dtype=dtype)
    a = (r if general else r + r.transpose(1, 2) + n * torch.eye(n, device="cuda", dtype=dtype)).reshape(-1).contiguous()
    x = torch.empty_like(a)
    if general and kernel == api.KERNEL_AUTO and 16 < n <= (192 if dtype == torch.float64 else 256) and algo == api.ALGO_GAUSS_JORDAN and os.environ.get("MATINV_TIME_NATURAL_FIRST", "") != "1":
        kernel_n = api.KERNEL_TILEP  # general input: ask for partial pivoting (default policy = natural order first, per matrix)
    else:
        kernel_n = kernel
    for _ in range(3):
        api.inverse_batched(a, n, algo, out=x, batch=batch, kernel=kernel_n)
    torch.cuda.synchronize()  # (a caller that has seen one batch complete: the launcher's reject-rate hint of that size class is in)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in ev:
        s.record(); api.inverse_batched(a, n, algo, out=x, batch=batch, kernel=kernel_n); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)[2]
    resid = (torch.bmm(a.view(batch, n, n)[:4], x.view(batch, n, n)[:4]) - torch.eye(n, device="cuda", dtype=dtype)).abs().max().item()
    print(f"n={n:4d} batch={batch:6d} {ms:8.3f} ms  {batch / ms * 1e3:12.4e} inv/s  {batch * 2 * n * n * a.element_size() / ms / 1e6:8.1f} GB/s  resid {resid:.1e}  "
          f"{api.kernel_name(algo, api.F64 if dtype == torch.float64 else api.F32, n, kernel_n)}", flush=True)
