"""`.mats` text format I/O and batch replication.

Host-side mirror of the reference helpers
  readMatricesFile   /root/reference/src/helper.cu:15-52
  replicateMatrices  /root/reference/src/helper.cu:54-72
File layout: first line ``numMatrices m n``; then numMatrices*m text rows of n
numbers, written row by row. In memory each matrix is COLUMN-major (element
(i, j) at ``j*m + i``) and matrices are contiguous, so a file with K matrices
becomes a flat array of K*m*n scalars -- the `Array` of include/types.h.
"""
from __future__ import annotations

import numpy as np

# include/helper_cpu.h:4 of the reference caps a file at 64 MiB of scalars.
MAX_MATRIX_BYTE_READ = 67108864


def read_mats(path: str, dtype=np.float64, enforce_cap: bool = False):
    """Return (flat column-major batch, numMatrices, m, n). Raises ValueError on a malformed file
    (the reference calls ensure() -> exit(EXIT_FAILURE), helper.cu:24,46)."""
    with open(path, "r") as fh:
        tokens = fh.read().split()
    if len(tokens) < 3:
        raise ValueError(f"could not read number of matrices from file {path}")
    k, m, n = (int(t) for t in tokens[:3])
    dtype = np.dtype(dtype)
    if enforce_cap and dtype.itemsize * k * m * n > MAX_MATRIX_BYTE_READ:
        raise ValueError(f"cannot read file {path}: array would exceed 0x{MAX_MATRIX_BYTE_READ:X} bytes")
    vals = np.array(tokens[3:3 + k * m * n], dtype=np.float64)
    if vals.size != k * m * n:
        raise ValueError(f"could not read matrix from file {path}: expected {k*m*n} numbers, found {vals.size}")
    # file is row-major per matrix -> memory column-major per matrix
    batch = vals.reshape(k, m, n).transpose(0, 2, 1).astype(dtype, order="C", copy=True).reshape(-1)
    return batch, k, m, n


def write_mats(path: str, batch: np.ndarray, k: int, m: int, n: int, digits: int | None = None) -> None:
    """Inverse of read_mats (the reference only has MATLAB dlmwrite generators,
    tests/generate_inverse_matrices.m:20-21)."""
    arr = np.asarray(batch, dtype=np.float64).reshape(k, n, m).transpose(0, 2, 1)
    with open(path, "w") as fh:
        fh.write(f"{k} {m} {n}\n")
        for mat in arr:
            for row in mat:
                fh.write(" ".join((repr(float(v)) if digits is None else f"{v:.{digits}g}") for v in row) + "\n")


def replicate(batch: np.ndarray, reps: int) -> np.ndarray:
    """replicateMatrices: the whole list repeated `reps` times back to back."""
    return np.tile(np.asarray(batch).reshape(-1), reps)
