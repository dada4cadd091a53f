"""MI355X-native batched small-matrix inversion: HIP kernels behind the reference's inverse_gpu.h boundary.

Package layout (only what the hot path needs):
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (libmatinv_hip.so)
  _lib.py    ctypes binding of that ABI (no fallback: fails loudly when the .so is missing)
  api.py     host-side mirror of the reference's operator interface
  mats.py    `.mats` file I/O (readMatricesFile / replicateMatrices)
  shard.py   batch partitioning across GPUs and result reassembly (RCCL all-gather)
  binqueue.py  size-binned multi-queue for mixed-size pipeline items (bins 32/128/512/1024)
"""
from . import mats  # noqa: F401
from ._lib import (ALGO_CHOLESKY, ALGO_GAUSS_JORDAN, F32, F64, KERNEL_AUTO, KERNEL_LDS, KERNEL_ROW,  # noqa: F401
                   KERNEL_ROWLANE, KERNEL_TILE, MatinvError, build)
