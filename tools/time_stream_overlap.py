#!/usr/bin/env python3
"""Do the large-n pipeline calls of different streams overlap on the device? 1024 x 8 items and 512 x 32 items, alone and together."""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
api = importlib.import_module('cuda-matrix-inversion_amd.api')
def mk(n, cnt):
    dt = torch.float32
    r = torch.rand((cnt, n, n), dtype=dt, device='cuda')
    B = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=dt, device='cuda')).reshape(-1).contiguous()
    a, c, d = (torch.rand(cnt * n, dtype=dt, device='cuda') for _ in range(3))
    return n, a, B, c, d, torch.empty(cnt, dtype=dt, device='cuda')
w = [mk(1024, 8), mk(512, 32), mk(128, 2048)]
st = [torch.cuda.Stream() for _ in w]
def run(idx, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for i in idx:
            n, a, B, c, d, out = w[i]
            with torch.cuda.stream(st[i]): api.calcluateMean(n, a, B, c, d, Means=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for i in range(3): run([i], 3)
print("alone: 1024 %.3f  512 %.3f  128 %.3f ms" % (run([0]), run([1]), run([2])))
print("together on three streams: %.3f ms" % run([0, 1, 2]))
