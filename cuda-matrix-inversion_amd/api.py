"""Host-side mirror of the reference's operator interface for the inversion hot path, over the C ABI.

Names, argument order and meaning follow /root/reference/include/inverse_gpu.h:7-31 (``handle`` dropped: the
reference's hand-written kernels ignore it) and src/gauss_bench.cu:127,275 (calcluateMean / calcluateVariance --
the reference's spelling is kept). Two layers:

* host-pointer family ``*_batched_gpu(n, As, aInvs, batchSize)`` on numpy arrays: calls the identically named
  C symbol (H2D + kernel + D2H inside, synchronous), exactly what ``inverse_bench`` times;
* device family on torch CUDA tensors: ``inverse_batched`` / ``mean_batched`` / ``variance_batched`` call the
  native ``matinv_*`` entry points on torch's current stream with no copies.

A batch is flat memory: matrix k occupies ``[k*n*n, (k+1)*n*n)``, column-major (element (r, c) at c*n + r).
torch is used for device memory and streams only; all arithmetic runs in libmatinv_hip.so.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import (GJ_ADAPTIVE, GJ_NATURAL_FIRST, GJ_PIVOT,  # noqa: F401
                   ALGO_CHOLESKY, ALGO_GAUSS_JORDAN, F32, F64, KERNEL_AUTO, KERNEL_BLOCKED, KERNEL_GLOBAL, KERNEL_LDS,  # noqa: F401
                   KERNEL_ROW, KERNEL_ROWLANE, KERNEL_TILE, KERNEL_TILEP, MatinvError)


def _np_dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return F64
    if dtype == np.float32:
        return F32
    raise TypeError(f"only float32/float64 batches are supported, got {dtype}")


def _torch_dtype_code(t) -> int:
    import torch
    if t.dtype == torch.float64:
        return F64
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"only float32/float64 batches are supported, got {t.dtype}")


# ---------------------------------------------------------------------------------------------- host family
def _call_reference_gpu(name: str, n: int, As: np.ndarray, aInvs: np.ndarray, batchSize: int) -> None:
    if not (isinstance(As, np.ndarray) and isinstance(aInvs, np.ndarray)):
        raise TypeError("host-pointer family takes numpy arrays")
    if As.dtype != aInvs.dtype or not As.flags.c_contiguous or not aInvs.flags.c_contiguous:
        raise ValueError("As and aInvs must be C-contiguous arrays of one dtype")
    if As.size < batchSize * n * n or aInvs.size < batchSize * n * n:
        raise ValueError("array smaller than batchSize*n*n")
    suffix = "" if _np_dtype_code(As.dtype) == F64 else "_f32"
    fn = getattr(_lib.lib(), name + suffix)
    fn(None, int(n), As.ctypes.data_as(ctypes.c_void_p), aInvs.ctypes.data_as(ctypes.c_void_p), int(batchSize))


def inverse_gauss_batched_gpu(n, As, aInvs, batchSize):
    """inverse_gpu.h:7 / src/gauss/batched_invert.cu:99."""
    _call_reference_gpu("inverse_gauss_batched_gpu", n, As, aInvs, batchSize)


def inverse_lu_cuda_batched_gpu(n, As, aInvs, batchSize):
    """inverse_gpu.h:8 / src/gauss/inverse_gpu.cu:60 (served by the pivoted Gauss-Jordan kernel)."""
    _call_reference_gpu("inverse_lu_cuda_batched_gpu", n, As, aInvs, batchSize)


def inverse_cholesky_batched_gpu(n, As, aInvs, batchSize):
    """inverse_gpu.h:27 / src/inverse_cholesky_gpu.cu:397. As is NOT clobbered (the reference does, :442)."""
    _call_reference_gpu("inverse_cholesky_batched_gpu", n, As, aInvs, batchSize)


def inverse_cholesky_mm_batched_gpu(n, As, aInvs, batchSize):
    _call_reference_gpu("inverse_cholesky_mm_batched_gpu", n, As, aInvs, batchSize)


def inverse_cholesky_mm2_batched_gpu(n, As, aInvs, batchSize):
    _call_reference_gpu("inverse_cholesky_mm2_batched_gpu", n, As, aInvs, batchSize)


def inverse_cholesky_stride_batched_gpu(n, As, aInvs, batchSize):
    _call_reference_gpu("inverse_cholesky_stride_batched_gpu", n, As, aInvs, batchSize)


def inverse_batched_host(As: np.ndarray, n: int, algo: int = ALGO_GAUSS_JORDAN):
    """matinv_inverse_batched_host: returns (aInvs, info) for a numpy batch."""
    As = np.ascontiguousarray(As)
    batch = As.size // (n * n)
    out = np.empty_like(As)
    info = np.zeros(batch, dtype=np.int32)
    _lib.check(_lib.lib().matinv_inverse_batched_host(
        algo, _np_dtype_code(As.dtype), n, As.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
        batch, info.ctypes.data_as(ctypes.c_void_p)))
    return out, info


def inverse_batched_host_multi(As: np.ndarray, n: int, algo: int = ALGO_GAUSS_JORDAN, nshards: int = 0):
    """matinv_inverse_batched_host_multi: the batch block-partitioned over `nshards` shards (0 = one per visible device),
    one host thread and one device per shard, results back in host memory; no collective. Returns (aInvs, info)."""
    As = np.ascontiguousarray(As)
    batch = As.size // (n * n)
    out = np.empty_like(As)
    info = np.zeros(batch, dtype=np.int32)
    _lib.check(_lib.lib().matinv_inverse_batched_host_multi(
        algo, _np_dtype_code(As.dtype), n, As.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
        batch, info.ctypes.data_as(ctypes.c_void_p), int(nshards)))
    return out, info


def device_count() -> int:
    k = _lib.lib().matinv_device_count()
    if k < 0:
        _lib.check(k)
    return k


def debug_rejects(reset: bool = False) -> int:
    """matinv_debug_rejects: matrices the first-pass kernels handed to a fallback since the last reset (needs
    MATINV_DEBUG_REJECTS=1 in the environment before the library is loaded; otherwise 0)."""
    return int(_lib.lib().matinv_debug_rejects(1 if reset else 0))


def set_gj_policy(policy: int) -> int:
    """matinv_set_gj_policy (GJ_NATURAL_FIRST / GJ_PIVOT / GJ_ADAPTIVE); returns the previous policy."""
    old = _lib.lib().matinv_set_gj_policy(int(policy))
    if old < 0:
        _lib.check(old)
    return old


# -------------------------------------------------------------------------------------------- device family
def _stream_ptr(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise ValueError("device family takes contiguous CUDA tensors")


def inverse_batched(As, n: int, algo: int = ALGO_GAUSS_JORDAN, out=None, info=None, kernel: int = KERNEL_AUTO,
                    batch: int | None = None, stride: int | None = None):
    """Invert a device-resident batch on torch's current stream (asynchronous).

    As: CUDA tensor holding `batch` matrices, matrix k at element offset k*stride (default n*n).
    out: optional result tensor of the same layout (allocated when None). info: optional int32[batch] tensor.
    Returns out.
    """
    import torch
    _require_cuda(As, out, info)
    stride = n * n if stride is None else int(stride)
    if batch is None:
        batch = As.numel() // stride
    if out is None:
        out = torch.empty_like(As)
    if out.dtype != As.dtype:
        raise TypeError("out dtype differs from input dtype")
    if info is not None and (info.dtype != torch.int32 or info.numel() < batch):
        raise ValueError("info must be an int32 tensor with at least `batch` elements")
    with torch.cuda.device(As.device):
        _lib.check(_lib.lib().matinv_inverse_batched_ex(
            algo, _torch_dtype_code(As), n, ctypes.c_void_p(As.data_ptr()), stride, ctypes.c_void_p(out.data_ptr()),
            stride, batch, ctypes.c_void_p(info.data_ptr()) if info is not None else None, _stream_ptr(As), kernel))
    return out


def inverse_gauss_batched_device(n, devAs, devAInvs, batchSize):
    """inverse_gpu.h:10 (declared there, never defined in the reference): Gauss-Jordan on device-resident batches."""
    return inverse_batched(devAs, n, ALGO_GAUSS_JORDAN, out=devAInvs, batch=batchSize)


def inverse_lu_cuda_batched_device(n, devAs, devAInvs, batchSize):
    """inverse_gpu.h:11 / src/gauss/inverse_gpu.cu:16 -- the dispatch point of gauss_bench's batchedInverse (:68-78)."""
    return inverse_batched(devAs, n, ALGO_GAUSS_JORDAN, out=devAInvs, batch=batchSize)


def inverse_cholesky_batched_device(n, devAs, devAInvs, batchSize):
    """inverse_gpu.h:20 / src/inverse_cholesky_gpu.cu:323."""
    return inverse_batched(devAs, n, ALGO_CHOLESKY, out=devAInvs, batch=batchSize)


def calcluateMean(n, As, Bs, Cs, Ds, Means=None, batchSize=None, info=None):
    """means[k] = a_k^T (B_k + diag c_k)^-1 d_k on device tensors (src/gauss_bench.cu:127-265; reference spelling).
    Unlike the reference CPU path (gauss_cpu.h:42) no input is destroyed."""
    import torch
    _require_cuda(As, Bs, Cs, Ds, Means, info)
    if batchSize is None:
        batchSize = Bs.numel() // (n * n)
    if Means is None:
        Means = torch.empty(batchSize, dtype=Bs.dtype, device=Bs.device)
    with torch.cuda.device(Bs.device):
        _lib.check(_lib.lib().matinv_mean_batched(
            _torch_dtype_code(Bs), n, ctypes.c_void_p(As.data_ptr()), ctypes.c_void_p(Bs.data_ptr()),
            ctypes.c_void_p(Cs.data_ptr()), ctypes.c_void_p(Ds.data_ptr()), ctypes.c_void_p(Means.data_ptr()),
            batchSize, ctypes.c_void_p(info.data_ptr()) if info is not None else None, _stream_ptr(Bs)))
    return Means


def calcluateVariance(n, As, Bs, Cs, Es, Variances=None, batchSize=None, info=None):
    """vars[k] = e_k - a_k^T (B_k + diag c_k)^-1 a_k (src/gauss_bench.cu:275-409; documented sign, gauss_cpu.h:34)."""
    import torch
    _require_cuda(As, Bs, Cs, Es, Variances, info)
    if batchSize is None:
        batchSize = Bs.numel() // (n * n)
    if Variances is None:
        Variances = torch.empty(batchSize, dtype=Bs.dtype, device=Bs.device)
    with torch.cuda.device(Bs.device):
        _lib.check(_lib.lib().matinv_variance_batched(
            _torch_dtype_code(Bs), n, ctypes.c_void_p(As.data_ptr()), ctypes.c_void_p(Bs.data_ptr()),
            ctypes.c_void_p(Cs.data_ptr()), ctypes.c_void_p(Es.data_ptr()), ctypes.c_void_p(Variances.data_ptr()),
            batchSize, ctypes.c_void_p(info.data_ptr()) if info is not None else None, _stream_ptr(Bs)))
    return Variances


mean_batched = calcluateMean
variance_batched = calcluateVariance


def tile_stats() -> dict:
    """matinv_tile_stats: how the adaptive natural-order / pivoting dispatch of the tile family went since load."""
    v = [ctypes.c_ulonglong(0) for _ in range(4)]
    _lib.check(_lib.lib().matinv_tile_stats(*[ctypes.byref(x) for x in v]))
    return dict(zip(("natural_launches", "pivot_launches", "last_rejected", "last_batch"), (int(x.value) for x in v)))


def select_kernel(algo: int, dtype, n: int) -> int:
    code = dtype if isinstance(dtype, int) else _np_dtype_code(dtype)
    k = _lib.lib().matinv_select_kernel(algo, code, n)
    if k < 0:
        _lib.check(k)
    return k


def kernel_name(algo: int, dtype, n: int, kernel: int = KERNEL_AUTO) -> str:
    code = dtype if isinstance(dtype, int) else _np_dtype_code(dtype)
    return _lib.lib().matinv_kernel_name(algo, code, n, kernel).decode()
