#!/usr/bin/env python3
"""Synthetic `.mats` fixture directories for the two command lines -- the job of the reference's MATLAB generators
(tests/generate_inverse_matrices.m:9-21, tests/generate_gaussian_matrices.m:15-37 there), whose large outputs (64x64,
128x128) are missing from the reference tree (.MISSING_LARGE_BLOBS).

    generate_fixtures.py inverse  OUTDIR K N [--seed S] [--digits D]   -> a.mats, aInv.mats
    generate_fixtures.py gaussian OUTDIR K N [--seed S] [--digits D]   -> a, b, c, d, e, means, variances .mats

A = R + R^T + N*I with R ~ U(0,1)^(NxN) (SPD, diagonally dominant); a, c, d ~ U(0,1)^N, e ~ U(0,1), B as A;
means = a^T (B + diag c)^-1 d, variances = e - a^T (B + diag c)^-1 a (the documented sign, include/gauss_cpu.h:34).
Expected outputs come from numpy's fp64 LAPACK path, independent of this repository's kernels and of its test oracle.
`--digits 5` reproduces the reference files' short decimal form (dlmwrite default precision); default is full precision.
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mats = importlib.import_module("cuda-matrix-inversion_amd.mats")


def spd(rng, k, n):
    r = rng.random((k, n, n))
    return r + r.transpose(0, 2, 1) + n * np.eye(n)


def colmajor(batch):  # (k, rows, cols) -> flat column-major batch, the layout write_mats takes
    return np.ascontiguousarray(batch.transpose(0, 2, 1)).reshape(-1)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kind", choices=["inverse", "gaussian"])
    ap.add_argument("outdir")
    ap.add_argument("k", type=int, help="number of matrices / items")
    ap.add_argument("n", type=int)
    ap.add_argument("--seed", type=int, default=0x5EED)
    ap.add_argument("--digits", type=int, default=None)
    args = ap.parse_args()
    k, n = args.k, args.n
    rng = np.random.default_rng(args.seed)
    os.makedirs(args.outdir, exist_ok=True)
    put = lambda name, batch, m, c: mats.write_mats(os.path.join(args.outdir, name), batch, k, m, c, digits=args.digits)
    A = spd(rng, k, n)
    if args.kind == "inverse":
        put("a.mats", colmajor(A), n, n)
        put("aInv.mats", colmajor(np.linalg.inv(A)), n, n)
        return
    a, c, d = (rng.random((k, n)) for _ in range(3))
    e = rng.random(k)
    M = A + np.einsum("ki,ij->kij", c, np.eye(n))
    sol_d = np.linalg.solve(M, d[..., None])[..., 0]
    sol_a = np.linalg.solve(M, a[..., None])[..., 0]
    put("a.mats", a.reshape(-1), n, 1)
    put("b.mats", colmajor(A), n, n)
    put("c.mats", c.reshape(-1), n, 1)
    put("d.mats", d.reshape(-1), n, 1)
    put("e.mats", e, 1, 1)
    put("means.mats", np.einsum("ki,ki->k", a, sol_d), 1, 1)
    put("variances.mats", e - np.einsum("ki,ki->k", a, sol_a), 1, 1)


if __name__ == "__main__":
    main()
