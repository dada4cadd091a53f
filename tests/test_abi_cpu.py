"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports every symbol
that include/matinv.h and include/inverse_gpu.h declare. No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, pkg


def _declared(header, pattern):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(pattern, text)))


def test_library_loads_and_exports_native_abi():
    L = pkg("_lib").lib()
    names = _declared("matinv.h", r"\b(matinv_[a-z_0-9]+)\s*\(")
    assert "matinv_inverse_batched" in names and "matinv_mean_batched" in names
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/matinv.h but not exported"
    assert sorted(names) == sorted(pkg("_lib").NATIVE_NAMES)
    assert L.matinv_abi_version() == 2


def test_library_exports_17_reference_names_both_precisions():
    L = pkg("_lib").lib()
    names = _declared("inverse_gpu.h", r"\bvoid\s+([a-z_0-9]+)\s*\(cublasHandle_t")
    assert len(names) == 17
    lib = pkg("_lib")
    assert sorted(names) == sorted(lib.REFERENCE_GPU_NAMES + lib.REFERENCE_DEVICE_NAMES)
    for n in names:
        assert hasattr(L, n), n
        assert hasattr(L, n + "_f32"), n + "_f32"


def test_kernel_selection_is_pure_host_logic():
    api = pkg("api")
    for n in (1, 8, 16, 32, 64, 128, 192):
        k = api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F64, n)
        assert k in (api.KERNEL_ROWLANE, api.KERNEL_TILE)
        assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, n).startswith("matinv_")
        assert api.select_kernel(api.ALGO_CHOLESKY, api.F64, n) == (api.KERNEL_ROWLANE if n <= 16 else api.KERNEL_TILE)
        assert api.select_kernel(api.ALGO_CHOLESKY, api.F32, n) == (api.KERNEL_ROWLANE if n <= 16 else api.KERNEL_TILE)
    assert api.select_kernel(api.ALGO_CHOLESKY, api.F64, 129) == api.KERNEL_TILE   # one wavefront per tile column up to 192 / 256
    assert api.select_kernel(api.ALGO_CHOLESKY, api.F64, 192) == api.KERNEL_TILE
    assert api.select_kernel(api.ALGO_CHOLESKY, api.F64, 193) == api.KERNEL_BLOCKED
    assert api.select_kernel(api.ALGO_CHOLESKY, api.F32, 256) == api.KERNEL_TILE
    assert api.select_kernel(api.ALGO_CHOLESKY, api.F32, 257) == api.KERNEL_BLOCKED
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F64, 192) == "matinv_spd_tile3w_f64<12, false>"  # r04: three waves, lower tiles
    assert api.select_kernel(api.ALGO_CHOLESKY, api.F32, 1024) == api.KERNEL_BLOCKED
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F32, 128) == "matinv_gj_tile4_f32<8, true, 4, false>"
    # r03: one wavefront per matrix on VGPRs + AGPRs up to 7 x 7 lower tiles (fp64 Cholesky), 8 x 8 in fp32
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F64, 100) == "matinv_spd_tile_f64<7, false>"
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F64, 120) == "matinv_spd_tile2_f64<false>"  # two waves, lower tiles only
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F64, 144) == "matinv_spd_tile2w_f64<9, false>"  # r04: the same beyond 128, one wave per SIMD
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F64, 176) == "matinv_spd_tile2w_f64<11, false>"
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F32, 128) == "matinv_spd_tile_f32<8, false>"
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 96) == "matinv_gj_tile4_f64<6, true, 2, false>"
    assert api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F64, 512) == api.KERNEL_BLOCKED
    assert api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F32, 150) == api.KERNEL_TILE   # one wavefront per tile column
    assert api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F64, 192) == api.KERNEL_TILE
    assert api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F64, 193) == api.KERNEL_BLOCKED
    assert api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F32, 257) == api.KERNEL_BLOCKED
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 160) == "matinv_gj_tile4_f64<10, false, 10, false>"
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 160, api.KERNEL_TILEP) == "matinv_gj_tileqw_f64<10, false>"  # r04: pivot columns searched
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 128, api.KERNEL_TILEP) == "matinv_gj_tilep4_f64<8, true>"
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F32, 96, api.KERNEL_TILEP) == "matinv_gj_tilep3_f32<6, true>"
    # 16 < n <= 25: the natural-order pass of the tile family is the two-rows-per-lane kernel (csrc/rowlane2_kernels.hip)
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 24) == "matinv_gj_rowlane2<double, 24, true, 0>"
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F32, 20) == "matinv_gj_rowlane2<float, 24, false, 0>"
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 25) == "matinv_gj_rowlane2<double, 32, false, 0>"
    assert api.kernel_name(api.ALGO_GAUSS_JORDAN, api.F64, 26) == "matinv_gj_tile_f64<2, false, true>"
    assert api.kernel_name(api.ALGO_CHOLESKY, api.F64, 24) == "matinv_spd_tile_f64<2, false>"
    with pytest.raises(pkg("_lib").MatinvError):
        api.select_kernel(api.ALGO_GAUSS_JORDAN, api.F64, 4096)


def test_argument_errors_do_not_need_a_gpu():
    lib = pkg("_lib")
    L = lib.lib()
    assert L.matinv_inverse_batched(0, 0, 0, None, 0, None, 0, 1, None, None) == lib.ERR_ARG
    assert b"n must be" in L.matinv_last_error()
    assert L.matinv_inverse_batched(0, 7, 8, None, 64, None, 64, 0, None, None) == lib.ERR_ARG
    # empty batch is a no-op, before any device is touched
    assert L.matinv_inverse_batched(0, 0, 8, None, 64, None, 64, 0, None, None) == lib.OK
    assert L.matinv_inverse_batched(9, 0, 8, None, 64, None, 64, 0, None, None) == lib.ERR_ARG


def test_compute_call_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = pkg("_lib")
    buf = (ctypes.c_double * 64)()
    rc = lib.lib().matinv_inverse_batched_host(0, 0, 8, buf, buf, 1, None)
    assert rc in (lib.ERR_NO_DEVICE, lib.ERR_HIP)
    assert lib.lib().matinv_last_error()


def test_mats_roundtrip(tmp_path, mats):
    import numpy as np
    rng = np.random.default_rng(0)
    batch = rng.standard_normal(3 * 4 * 5)
    p = tmp_path / "x.mats"
    mats.write_mats(str(p), batch, 3, 4, 5)
    back, k, m, n = mats.read_mats(str(p))
    assert (k, m, n) == (3, 4, 5)
    assert np.array_equal(back, batch)
    # row-major text -> column-major memory (helper.cu:38-48)
    p.write_text("1 2 3\n1 2 3\n4 5 6\n")
    b, k, m, n = mats.read_mats(str(p))
    assert b.tolist() == [1, 4, 2, 5, 3, 6]
    assert np.array_equal(mats.replicate(b, 2), np.concatenate([b, b]))
    p.write_text("2 2 2\n1 2 3\n")
    with pytest.raises(ValueError):
        mats.read_mats(str(p))
