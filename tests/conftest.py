import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFDATA = os.path.join(GOLDEN, "ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub: str = ""):
    """Import the product package (its directory name has a hyphen)."""
    name = "cuda-matrix-inversion_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def mats():
    return pkg("mats")


@pytest.fixture(scope="session")
def gold():
    return dict(np.load(os.path.join(GOLDEN, "oracle_fp64.npz")))


def read_ref(rel, dtype=np.float64):
    return pkg("mats").read_mats(os.path.join(REFDATA, rel), dtype=dtype)


def spd_batch(n, batch, seed=0, dtype=np.float64):
    """R + R^T + n*I, the recipe of tests/generate_inverse_matrices.m:9-18 (column-major flat batch)."""
    rng = np.random.default_rng(seed)
    r = rng.random((batch, n, n))
    a = r + r.transpose(0, 2, 1) + n * np.eye(n)
    return np.ascontiguousarray(a.transpose(0, 2, 1)).reshape(-1).astype(dtype)


def general_batch(n, batch, seed=0, dtype=np.float64):
    """U(0,1)^{n x n}, non-symmetric, like tests/square_5_*.mats."""
    rng = np.random.default_rng(seed)
    return rng.random((batch, n, n)).reshape(-1).astype(dtype)


def as_mats(flat, n):
    """flat column-major batch -> (batch, n, n) array indexed [k, row, col]."""
    return np.asarray(flat).reshape(-1, n, n).transpose(0, 2, 1)


def rel_err(x, y, n):
    """max |x-y| / max(|y|, 1e-3*max|Y|) per matrix, max over the batch."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, n * n)
    y = np.asarray(y, dtype=np.float64).reshape(-1, n * n)
    floor = 1e-3 * np.abs(y).max(axis=1, keepdims=True)
    return float((np.abs(x - y) / np.maximum(np.abs(y), floor)).max())
