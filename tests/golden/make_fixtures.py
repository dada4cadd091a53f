#!/usr/bin/env python3
"""Generate tests/golden/ from the reference's own test DATA files and the CPU oracle.

Run once in the build container (where /root/reference is mounted):
    python tests/golden/make_fixtures.py
What it writes:
  tests/golden/ref/...      DATA files of /root/reference/tests (inputs + the reference's
                            4-digit MATLAB goldens), some truncated to their first K matrices to
                            keep the repository small. Values are copied verbatim (shortest
                            round-trip decimal). No reference source code is copied.
  tests/golden/oracle_fp64.npz
                            fp64 outputs of oracle/ (Gauss-Jordan partial pivot, Cholesky, means,
                            variances) on those inputs -- the 1e-10 parity vectors the GPU tests use.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mats = importlib.import_module("cuda-matrix-inversion_amd.mats")
import oracle  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "ref")

# (relative source, destination name, keep first K matrices or None)
PLAN = [
    ("tests/inverse_100_8x8/a.mats", "inverse_100_8x8/a.mats", None),
    ("tests/inverse_100_8x8/aInv.mats", "inverse_100_8x8/aInv.mats", None),
    ("tests/inverse_100_16x16/a.mats", "inverse_100_16x16/a.mats", None),
    ("tests/inverse_100_16x16/aInv.mats", "inverse_100_16x16/aInv.mats", None),
    ("tests/inverse_100_32x32/a.mats", "inverse_32_32x32/a.mats", 32),
    ("tests/inverse_100_32x32/aInv.mats", "inverse_32_32x32/aInv.mats", 32),
    ("tests/inverse_100_64x64/a.mats", "inverse_12_64x64/a.mats", 12),
    ("tests/square_5_8_8.mats", "square_5_8_8.mats", None),
    ("tests/square_5_16_16.mats", "square_5_16_16.mats", None),
    ("tests/square_5_32_32.mats", "square_5_32_32.mats", None),
    ("tests/square_5_64_64.mats", "square_3_64_64.mats", 3),
    ("tests/square_5_128_128.mats", "square_1_128_128.mats", 1),
    ("tests/simpleMean/chol.mats", "simpleMean/chol.mats", None),
    ("tests/simpleMean/cholinv.mats", "simpleMean/cholinv.mats", None),
    ("src/gauss/batch_3.txt", "batch_3.mats", None),
]
for size, keep in (("8x8", None), ("16x16", None), ("32x32", 32), ("64x64", 12)):
    tag = f"gaussian_{keep or 100}_{size}"
    for f in ("a", "b", "c", "d", "e", "means", "variances"):
        PLAN.append((f"tests/gaussian_100_{size}/{f}.mats", f"{tag}/{f}.mats", keep))


def main():
    for src, dst, keep in PLAN:
        batch, k, m, n = mats.read_mats(os.path.join(REF, src))
        if keep is not None and keep < k:
            batch, k = batch[: keep * m * n], keep
        path = os.path.join(OUT, dst)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        mats.write_mats(path, batch, k, m, n)
        print(f"{dst}: {k} x {m} x {n}")

    gold = {}
    for d in ("inverse_100_8x8", "inverse_100_16x16", "inverse_32_32x32", "inverse_12_64x64"):
        a, k, m, n = mats.read_mats(os.path.join(OUT, d, "a.mats"))
        inv, info = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
        assert not info.any()
        gold[f"{d}/gj"] = inv
        inv, info = oracle.inverse_batched(a, n, oracle.ALGO_CHOLESKY)
        assert not info.any()
        gold[f"{d}/chol"] = inv
    for f in ("square_5_8_8", "square_5_16_16", "square_5_32_32", "square_3_64_64", "square_1_128_128", "batch_3"):
        a, k, m, n = mats.read_mats(os.path.join(OUT, f + ".mats"))
        inv, info = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
        assert not info.any()
        gold[f"{f}/gj"] = inv
    a, k, m, n = mats.read_mats(os.path.join(OUT, "simpleMean/chol.mats"))
    gold["simpleMean/chol"] = oracle.inverse_batched(a, n, oracle.ALGO_CHOLESKY)[0]
    for d in ("gaussian_100_8x8", "gaussian_100_16x16", "gaussian_32_32x32", "gaussian_12_64x64"):
        r = {f: mats.read_mats(os.path.join(OUT, d, f + ".mats")) for f in "abcde"}
        n = r["b"][2]
        gold[f"{d}/means"] = oracle.mean_batched(r["a"][0], r["b"][0], r["c"][0], r["d"][0], n)
        gold[f"{d}/variances"] = oracle.variance_batched(r["a"][0], r["b"][0], r["c"][0], r["e"][0], n)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_fp64.npz"), **gold)
    print("oracle_fp64.npz:", len(gold), "arrays")


if __name__ == "__main__":
    main()
