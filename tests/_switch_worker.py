"""Worker of tests/test_gpu_switches.py: ONE process per environment switch (the library reads each switch once per process). Runs the
checks named on the command line -- kind:dtype:n[:batch] -- through the C ABI and compares with the CPU oracle; prints the kernel that
served each check and `switch-worker ok`, or raises."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import as_mats, general_batch, rel_err, spd_batch  # noqa: E402

api = importlib.import_module("cuda-matrix-inversion_amd.api")


def mixed_batch(n, batch, seed, dtype):
    """a third diagonally dominant, a third mildly non-dominant (some accepted, some rejected by the natural-order kernels), a third general"""
    rng = np.random.default_rng(seed)
    r = rng.random((batch, n, n))
    s = r + r.transpose(0, 2, 1)
    out = r.copy()
    out[0::3] = s[0::3] + n * np.eye(n)
    out[1::3] = s[1::3] + 0.35 * n * np.eye(n)
    return out.reshape(-1).astype(dtype)


def check(kind, dtype_name, n, batch):
    f64 = dtype_name == "f64"
    np_t, t_t = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    if kind in ("gj_spd", "gj_general", "gj_mixed", "chol", "host"):
        a = {"gj_spd": spd_batch, "chol": spd_batch, "host": spd_batch, "gj_general": general_batch}.get(kind, None)
        a = a(n, batch, seed=n, dtype=np_t) if a else mixed_batch(n, batch, n, np_t)
        algo, oalgo = (api.ALGO_CHOLESKY, oracle.ALGO_CHOLESKY) if kind == "chol" else (api.ALGO_GAUSS_JORDAN, oracle.ALGO_GJ_PIVOT)
        want, winfo = oracle.inverse_batched(a.astype(np.float64), n, oalgo)
        assert not winfo.any()
        if kind == "host":
            got = np.empty_like(a)
            api.inverse_gauss_batched_gpu(n, a, got, batch)
            name = "host-pointer entry point"
        else:
            d = torch.from_numpy(a).cuda()
            info = torch.full((batch,), -7, dtype=torch.int32, device="cuda")
            got = api.inverse_batched(d, n, algo, info=info, batch=batch)
            torch.cuda.synchronize()
            assert not info.cpu().numpy().any(), info
            again = api.inverse_batched(d, n, algo, batch=batch)  # e.g. the screening pass switches itself on after a general launch
            if os.environ.get("MATINV_GJ_POLICY") != "adaptive":  # (whose documented price is exactly this)
                assert torch.equal(got, again), "second launch gave other bits"
            got = got.cpu().numpy()
            name = api.kernel_name(algo, api.F64 if f64 else api.F32, n)
        cond = max(np.linalg.cond(m) for m in as_mats(a.astype(np.float64), n)[:: max(1, batch // 8)])
        if f64:
            assert rel_err(got, want, n) < max(1e-10, 1e-15 * cond * n * 8), (kind, n, rel_err(got, want, n))
        else:
            g, w = got.astype(np.float64).reshape(batch, -1), want.reshape(batch, -1)
            fro = (np.linalg.norm(g - w, axis=1) / np.linalg.norm(w, axis=1)).max()
            assert fro < 2e-6 * max(cond, 10.0), (kind, n, fro)
    elif kind in ("mean", "variance"):
        rng = np.random.default_rng(n)
        B = spd_batch(n, batch, seed=n + 1, dtype=np_t)
        av, cv, dv = (rng.random(batch * n).astype(np_t) for _ in range(3))
        ev = rng.random(batch).astype(np_t)
        to = lambda x: torch.from_numpy(x).cuda()
        if kind == "mean":
            got = api.calcluateMean(n, to(av), to(B), to(cv), to(dv)).cpu().numpy()
            want = oracle.mean_batched(av.astype(np.float64), B.astype(np.float64), cv.astype(np.float64), dv.astype(np.float64), n)
        else:
            got = api.calcluateVariance(n, to(av), to(B), to(cv), to(ev)).cpu().numpy()
            want = oracle.variance_batched(av.astype(np.float64), B.astype(np.float64), cv.astype(np.float64), ev.astype(np.float64), n)
        err = np.abs(got - want).max()
        assert err < (1e-10 if f64 else 2e-5), (kind, n, err)
        name = "fused pipeline"
    else:
        raise SystemExit(f"unknown check {kind}")
    print(f"  {kind}:{dtype_name}:{n}:{batch} ok ({name})", flush=True)


def main():
    assert torch.cuda.is_available()
    for spec in sys.argv[1:]:
        f = spec.split(":")
        check(f[0], f[1], int(f[2]), int(f[3]) if len(f) > 3 else 24)
    print("rejects seen by the fallbacks:", api.debug_rejects() if os.environ.get("MATINV_DEBUG_REJECTS") else "(not counted)")
    print("switch-worker ok")


if __name__ == "__main__":
    main()
