"""One-off randomized parity campaign against the CPU oracle (not part of the test suite): random dtype / algorithm / size / input
class per trial; prints every failure and a summary. usage: python tools/fuzz_parity.py [trials] [seed] [nmax]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle
from conftest import spd_batch, general_batch, rel_err, as_mats
api = importlib.import_module("cuda-matrix-inversion_amd.api")

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 320  # sizes beyond 320 (up to 1024) only when asked for: the oracle takes seconds there
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for it in range(trials):
    n = int(rng.choice([rng.integers(1, 33), rng.integers(33, 129), rng.integers(129, 200), rng.integers(200, nmax + 1)], p=[0.3, 0.35, 0.2, 0.15] if nmax <= 320 else [0.05, 0.1, 0.15, 0.7]))
    dt = np.float64 if rng.random() < 0.55 else np.float32
    what = rng.choice(["gj_spd", "gj_gen", "chol", "mean", "var"], p=[0.2, 0.3, 0.2, 0.2, 0.1])
    batch = int(rng.integers(1, 40 if n < 130 else (9 if n <= 320 else 4)))
    seed = int(rng.integers(1 << 30))
    tol32 = 5e-4 if what == "gj_gen" else 5e-5
    detail = ""
    try:
        if what in ("gj_spd", "gj_gen", "chol"):
            a = (general_batch if what == "gj_gen" else spd_batch)(n, batch, seed=seed)
            sing = what == "gj_gen" and batch > 2 and rng.random() < 0.3
            if sing:
                a = a.reshape(batch, n, n).copy(); a[1, n // 2, :] = 0.0; a = a.reshape(-1)
            algo, oalgo = (api.ALGO_CHOLESKY, oracle.ALGO_CHOLESKY) if what == "chol" else (api.ALGO_GAUSS_JORDAN, oracle.ALGO_GJ_PIVOT)
            keep = [i for i in range(batch) if not (sing and i == 1)]
            ak = np.concatenate([a[i * n * n:(i + 1) * n * n] for i in keep])
            want, _ = oracle.inverse_batched(ak, n, oalgo)
            info = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
            got = api.inverse_batched(torch.from_numpy(a.astype(dt)).cuda(), n, algo, batch=batch, info=info).cpu().numpy().astype(np.float64)
            inf = info.cpu().numpy()
            gk = np.concatenate([got[i * n * n:(i + 1) * n * n] for i in keep])
            cond = max(np.linalg.cond(m) for m in as_mats(ak, n))
            tol = max(1e-10, 1e-15 * cond * n) if dt == np.float64 else max(tol32, 2e-6 * cond)  # fp32: eps * cond * O(10); observed up to 1.1e-6 * cond at n = 136
            err = rel_err(gk, want, n)
            ok = err < tol and not inf[keep].any()
            detail = f"err={err:.3g} tol={tol:.3g} cond={cond:.3g} info={inf.tolist()}"
            if sing:
                ok = ok and inf[1] == n // 2 + 1 and np.isnan(got[n * n:2 * n * n]).all()
        else:
            B = spd_batch(n, batch, seed=seed)
            r2 = np.random.default_rng(seed)
            va, vc, vd = (r2.random(batch * n) for _ in range(3))
            ve = r2.random(batch)
            t = [torch.from_numpy(x.astype(dt)).cuda() for x in (va, B, vc, vd, ve)]
            if what == "mean":
                got = api.calcluateMean(n, t[0], t[1], t[2], t[3]).cpu().numpy().astype(np.float64)
                want = oracle.mean_batched(va, B, vc, vd, n)
            else:
                got = api.calcluateVariance(n, t[0], t[1], t[2], t[4]).cpu().numpy().astype(np.float64)
                want = oracle.variance_batched(va, B, vc, ve, n)
            ok = np.abs(got - want).max() < (1e-10 if dt == np.float64 else 5e-5)
    except Exception as e:  # noqa
        ok = False
        print("EXC", what, n, dt.__name__, batch, repr(e)[:200])
    if not ok:
        fails += 1
        print("FAIL", what, "n=", n, dt.__name__, "batch=", batch, "seed=", seed, locals().get("detail", ""))
print(f"fuzz: {trials} trials, {fails} failures")
