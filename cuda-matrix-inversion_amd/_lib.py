"""ctypes binding of libmatinv_hip.so (the C ABI declared in include/matinv.h and include/inverse_gpu.h).

There is NO fallback: if the shared library is missing or fails to load, importing a compute entry point
raises. The library is built in-tree by ``make -C cuda-matrix-inversion_amd`` (``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MATINV_LIB", os.path.join(_HERE, "libmatinv_hip.so"))  # override: profiling builds only

# include/matinv.h enums
OK, ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE = 0, -1, -2, -3, -4
F64, F32 = 0, 1
ALGO_GAUSS_JORDAN, ALGO_CHOLESKY = 0, 1
KERNEL_AUTO, KERNEL_LDS, KERNEL_ROWLANE, KERNEL_TILE, KERNEL_ROW, KERNEL_GLOBAL, KERNEL_BLOCKED, KERNEL_TILEP = 0, 1, 2, 3, 4, 5, 6, 7

REFERENCE_GPU_NAMES = [
    "inverse_gauss_batched_gpu", "inverse_lu_cuda_batched_gpu", "inverse_cholesky_stride_batched_gpu",
    "inverse_cholesky_batched_gpu", "inverse_cholesky_mm_batched_gpu", "inverse_cholesky_mm2_batched_gpu",
]
REFERENCE_DEVICE_NAMES = [
    "inverse_gauss_batched_device", "inverse_lu_cuda_batched_device", "inverse_cholesky_stride_batched_device",
    "decompose_cholesky_stride_batched_device", "inverse_upper_stride_batched_device",
    "multiply_upper_stride_batched_device", "inverse_cholesky_batched_device", "decompose_cholesky_batched_device",
    "inverse_cholesky_mm_batched_device", "decompose_cholesky_mm_batched_device",
    "inverse_cholesky_mm2_batched_device",
]
NATIVE_NAMES = [
    "matinv_inverse_batched", "matinv_inverse_batched_ex", "matinv_select_kernel", "matinv_kernel_name",
    "matinv_mean_batched", "matinv_variance_batched", "matinv_inverse_batched_host", "matinv_mean_batched_host",
    "matinv_variance_batched_host", "matinv_last_error",
    "matinv_abi_version", "matinv_release_cache", "matinv_stream_retire", "matinv_batched_malloc", "matinv_batched_free", "matinv_memcpy_2d",
    "matinv_device_synchronize", "matinv_tile_stats", "matinv_queue_create", "matinv_queue_submit", "matinv_queue_submit_chunks", "matinv_queue_pending",
    "matinv_queue_bins", "matinv_queue_flush", "matinv_queue_destroy", "matinv_queue_last_error", "matinv_queue_stream",
    "matinv_set_gj_policy", "matinv_device_count", "matinv_shard_range", "matinv_inverse_batched_host_multi", "matinv_comm_unique_id",
    "matinv_comm_init_rank", "matinv_comm_destroy", "matinv_allgather_shards", "matinv_allgather_local", "matinv_allgather_local_after", "matinv_debug_rejects",
]
GJ_NATURAL_FIRST, GJ_PIVOT, GJ_ADAPTIVE = 0, 1, 2


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the shared library (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "-s", "clean"], check=True)
    subprocess.run(["make", "-C", _HERE, "-s", "-j4"], check=True)
    return LIB_PATH


class MatinvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmatinv_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run __graft_entry__.build() "
            f"(make -C cuda-matrix-inversion_amd). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.matinv_inverse_batched.restype = ci
    L.matinv_inverse_batched.argtypes = [ci, ci, ci, vp, sz, vp, sz, sz, vp, vp]
    L.matinv_inverse_batched_ex.restype = ci
    L.matinv_inverse_batched_ex.argtypes = [ci, ci, ci, vp, sz, vp, sz, sz, vp, vp, ci]
    L.matinv_select_kernel.restype = ci
    L.matinv_select_kernel.argtypes = [ci, ci, ci]
    L.matinv_kernel_name.restype = ctypes.c_char_p
    L.matinv_kernel_name.argtypes = [ci, ci, ci, ci]
    for f in (L.matinv_mean_batched, L.matinv_variance_batched):
        f.restype = ci
        f.argtypes = [ci, ci, vp, vp, vp, vp, vp, sz, vp, vp]
    L.matinv_inverse_batched_host.restype = ci
    L.matinv_inverse_batched_host.argtypes = [ci, ci, ci, vp, vp, sz, vp]
    for f in (L.matinv_mean_batched_host, L.matinv_variance_batched_host):
        f.restype = ci
        f.argtypes = [ci, ci, vp, vp, vp, vp, vp, sz, vp]
    L.matinv_inverse_batched_host_multi.restype = ci
    L.matinv_inverse_batched_host_multi.argtypes = [ci, ci, ci, vp, vp, sz, vp, ci]
    L.matinv_set_gj_policy.restype = ci
    L.matinv_set_gj_policy.argtypes = [ci]
    L.matinv_device_count.restype = ci
    L.matinv_shard_range.restype = ci
    L.matinv_shard_range.argtypes = [sz, ci, ci, ci, vp, vp]
    L.matinv_comm_unique_id.restype = ci
    L.matinv_comm_unique_id.argtypes = [vp]
    L.matinv_comm_init_rank.restype = ci
    L.matinv_comm_init_rank.argtypes = [vp, ci, vp, ci]
    L.matinv_comm_destroy.restype = ci
    L.matinv_comm_destroy.argtypes = [vp]
    L.matinv_allgather_shards.restype = ci
    L.matinv_allgather_shards.argtypes = [vp, ci, vp, vp, sz, vp]
    L.matinv_allgather_local.restype = ci
    L.matinv_allgather_local.argtypes = [ci, vp, ci, vp, vp, sz]
    L.matinv_allgather_local_after.restype = ci
    L.matinv_allgather_local_after.argtypes = [ci, vp, ci, vp, vp, sz, vp]
    L.matinv_tile_stats.restype = ci
    L.matinv_tile_stats.argtypes = [vp, vp, vp, vp]
    L.matinv_batched_malloc.restype = ci
    L.matinv_batched_malloc.argtypes = [vp, vp, sz, ci]
    L.matinv_batched_free.restype = ci
    L.matinv_batched_free.argtypes = [vp]
    L.matinv_memcpy_2d.restype = ci
    L.matinv_memcpy_2d.argtypes = [vp, sz, vp, sz, sz, sz, ci]
    L.matinv_device_synchronize.restype = ci
    L.matinv_queue_create.restype = ci
    L.matinv_queue_create.argtypes = [vp, ci, vp, ci]
    L.matinv_queue_submit.restype = ci
    L.matinv_queue_submit.argtypes = [vp, ci, vp, vp, vp, vp, vp, sz, vp]
    L.matinv_queue_submit_chunks.restype = ci
    L.matinv_queue_submit_chunks.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.matinv_queue_pending.restype = ci
    L.matinv_queue_pending.argtypes = [vp, vp, vp]
    L.matinv_queue_bins.restype = ci
    L.matinv_queue_bins.argtypes = [vp, vp, ci]
    L.matinv_queue_flush.restype = ci
    L.matinv_queue_flush.argtypes = [vp, vp, vp, vp]
    L.matinv_queue_stream.restype = vp
    L.matinv_queue_stream.argtypes = [vp]
    L.matinv_queue_destroy.restype = ci
    L.matinv_queue_destroy.argtypes = [vp]
    L.matinv_queue_last_error.restype = ctypes.c_char_p
    L.matinv_queue_last_error.argtypes = [vp]
    L.matinv_last_error.restype = ctypes.c_char_p
    L.matinv_abi_version.restype = ci
    L.matinv_release_cache.restype = ci
    L.matinv_debug_rejects.restype = ctypes.c_longlong
    L.matinv_debug_rejects.argtypes = [ci]
    for suffix in ("", "_f32"):
        for name in REFERENCE_GPU_NAMES + REFERENCE_DEVICE_NAMES:
            f = getattr(L, name + suffix)
            f.restype = None
            f.argtypes = [vp, ci, vp, vp, ci]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != OK:
        raise MatinvError(rc, lib().matinv_last_error().decode())
