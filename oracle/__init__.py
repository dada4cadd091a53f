"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and
by the ``cpu_baseline`` leg of ``bench.py`` -- never by the product package.

All arrays are numpy, C-contiguous, batch-major with each matrix column-major
(element (r, c) of matrix k at ``k*n*n + c*n + r``), i.e. exactly the memory the
reference's ``readMatricesFile`` produces (/root/reference/src/helper.cu:38-48).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ALGO_GJ_PIVOT = 0      # Gauss-Jordan, partial pivoting (oracle_gj_pivot)
ALGO_GJ_REFERENCE = 1  # Gauss-Jordan exactly as batched_invert.cu:17-95 (zero-only pivot)
ALGO_CHOLESKY = 2      # inverse_cholesky_cpu.c:17-85 generalised
ALGO_LU = 3            # getrf + getri (inverse.c:63-65)


def build(force: bool = False) -> str:
    """Compile liboracle.so (and oracle/_ref when /root/reference is mounted)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_impl.inc")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    ref_missing = os.path.isdir("/root/reference/src") and not os.path.exists(
        os.path.join(_HERE, "_ref", "inverse_cholesky_cpu"))
    if stale or ref_missing:
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for suf, ct in (("f64", ctypes.c_double), ("f32", ctypes.c_float)):
            p = ctypes.POINTER(ct)
            ip = ctypes.POINTER(ctypes.c_int)
            f = getattr(_LIB, f"oracle_inverse_batched_{suf}")
            f.restype = ctypes.c_long
            f.argtypes = [ctypes.c_int, p, p, ip, ctypes.c_int, ctypes.c_long]
            f = getattr(_LIB, f"oracle_mean_batched_{suf}")
            f.restype = ctypes.c_long
            f.argtypes = [p, p, p, p, p, ctypes.c_int, ctypes.c_long]
            f = getattr(_LIB, f"oracle_variance_batched_{suf}")
            f.restype = ctypes.c_long
            f.argtypes = [p, p, p, p, p, ctypes.c_int, ctypes.c_long, ctypes.c_int]
    return _LIB


def _suf(dtype) -> tuple[str, type]:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", ctypes.c_double
    if dtype == np.float32:
        return "f32", ctypes.c_float
    raise TypeError(f"oracle supports float32/float64, not {dtype}")


def _ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def inverse_batched(As: np.ndarray, n: int, algo: int = ALGO_GJ_PIVOT, threads: int | None = None):
    """Invert a batch. ``As`` has batch*n*n elements. Returns (Ainvs, info[batch])."""
    As = np.ascontiguousarray(As)
    suf, ct = _suf(As.dtype)
    batch = As.size // (n * n)
    assert batch * n * n == As.size
    out = np.empty_like(As)
    info = np.zeros(batch, dtype=np.int32)
    _with_threads(threads, lambda: getattr(lib(), f"oracle_inverse_batched_{suf}")(
        algo, _ptr(As, ct), _ptr(out, ct), info.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n, batch))
    return out, info


def mean_batched(a, B, c, d, n: int, threads: int | None = None) -> np.ndarray:
    """means[k] = a_k^T (B_k + diag c_k)^-1 d_k   (gauss_cpu.c:41-72)."""
    a, B, c, d = (np.ascontiguousarray(x) for x in (a, B, c, d))
    suf, ct = _suf(B.dtype)
    batch = B.size // (n * n)
    out = np.empty(batch, dtype=B.dtype)
    bad = _with_threads(threads, lambda: getattr(lib(), f"oracle_mean_batched_{suf}")(
        _ptr(a, ct), _ptr(B, ct), _ptr(c, ct), _ptr(d, ct), _ptr(out, ct), n, batch))
    if bad:
        raise ArithmeticError(f"{bad} matrices not SPD in oracle mean")
    return out


def variance_batched(a, B, c, e, n: int, ref_sign: bool = False, threads: int | None = None) -> np.ndarray:
    """vars[k] = e_k - a_k^T (B_k + diag c_k)^-1 a_k (documented sign; ref_sign=True -> '+', gauss_cpu.c:198)."""
    a, B, c, e = (np.ascontiguousarray(x) for x in (a, B, c, e))
    suf, ct = _suf(B.dtype)
    batch = B.size // (n * n)
    out = np.empty(batch, dtype=B.dtype)
    bad = _with_threads(threads, lambda: getattr(lib(), f"oracle_variance_batched_{suf}")(
        _ptr(a, ct), _ptr(B, ct), _ptr(c, ct), _ptr(e, ct), _ptr(out, ct), n, batch, int(ref_sign)))
    if bad:
        raise ArithmeticError(f"{bad} matrices not SPD in oracle variance")
    return out


def _with_threads(threads, fn):
    """Run fn() with OMP_NUM_THREADS-like control via omp_set_num_threads of libgomp."""
    if threads is None:
        return fn()
    gomp = ctypes.CDLL("libgomp.so.1")
    gomp.omp_get_max_threads.restype = ctypes.c_int
    old = gomp.omp_get_max_threads()
    gomp.omp_set_num_threads(int(threads))
    try:
        return fn()
    finally:
        gomp.omp_set_num_threads(old)


def ref_cholesky_4x4(a4: np.ndarray) -> np.ndarray | None:
    """Run the reference's own scalar Cholesky inverse (oracle/_ref/inverse_cholesky_cpu,
    built from /root/reference/src/inverse_cholesky_cpu.c, fixed N=4, fp32, row-major stdin)
    on one 4x4 matrix given as a (4,4) array [row][col]. Returns None when the binary is absent."""
    exe = os.path.join(_HERE, "_ref", "inverse_cholesky_cpu")
    if not os.path.exists(exe):
        return None
    text = "\n".join(" ".join(repr(float(v)) for v in row) for row in np.asarray(a4, dtype=np.float64)) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout
    tail = out.split("Inverse is:")[1].split()
    return np.array([float(x) for x in tail[:16]], dtype=np.float64).reshape(4, 4)
