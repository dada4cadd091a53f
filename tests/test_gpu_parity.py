"""Parity tests proper: HIP path (through the C ABI of libmatinv_hip.so) vs the CPU oracle on the same inputs,
vs the committed golden vectors, and -- at BASELINE.json's full sizes -- through size-independent properties.

Tolerances (BASELINE.json north_star "max element-wise rel. error < 1e-10"):
  fp64 GPU vs fp64 oracle : max |x-y| / max(|y|, 1e-3*max|Y|) < 1e-10        (rel_err below)
  fp32 GPU vs fp64 oracle : ||X-Y||_F / ||Y||_F < 1e-5 * cond
  vs 4-digit reference goldens : sum|err| per matrix at the rounding floor (< 5e-4), means/variances < 5e-5
"""
import ctypes
import os

import numpy as np
import pytest

import oracle
from conftest import as_mats, general_batch, pkg, read_ref, rel_err, spd_batch

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
api = pkg("api")
GJ, CH = api.ALGO_GAUSS_JORDAN, api.ALGO_CHOLESKY
FAMILIES = {"auto": api.KERNEL_AUTO, "lds": api.KERNEL_LDS, "rowlane": api.KERNEL_ROWLANE, "tile": api.KERNEL_TILE,
            "row": api.KERNEL_ROW, "tilep": api.KERNEL_TILEP}


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def gpu_inverse(a_np, n, algo=GJ, kernel=api.KERNEL_AUTO, want_info=False):
    a = dev(a_np)
    batch = a.numel() // (n * n)
    info = torch.full((max(batch, 1),), -7, dtype=torch.int32, device="cuda")
    out = api.inverse_batched(a, n, algo, info=info, kernel=kernel, batch=batch)
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), torch.from_numpy(np.ascontiguousarray(a_np))), "input batch was modified"
    res = out.cpu().numpy()
    return (res, info[:batch].cpu().numpy()) if want_info else res


def family_or_skip(name, algo, dtype, n):
    k = FAMILIES[name]
    if k == api.KERNEL_AUTO:
        return k
    lib = pkg("_lib")
    probe = torch.zeros(n * n, dtype=dtype, device="cuda") + torch.eye(n, dtype=dtype, device="cuda").reshape(-1)
    try:
        api.inverse_batched(probe, n, algo, kernel=k, batch=1)
    except lib.MatinvError as e:
        if e.code == lib.ERR_UNSUPPORTED:
            pytest.skip(f"{name} family does not serve n={n}")
        raise
    return k


SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 100, 128]


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("n", SIZES)
def test_gj_spd_fp64_vs_oracle(n, family):
    k = family_or_skip(family, GJ, torch.float64, n)
    a = spd_batch(n, 37, seed=n)
    want, info = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    assert not info.any()
    got, ginfo = gpu_inverse(a, n, GJ, k, want_info=True)
    assert not ginfo.any()
    assert rel_err(got, want, n) < 1e-10


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("n", [2, 3, 5, 8, 16, 19, 32, 48, 64, 128])
def test_gj_general_needs_pivoting_fp64(n, family):
    """U(0,1) non-symmetric matrices (like tests/square_5_*): correct only with row pivoting."""
    k = family_or_skip(family, GJ, torch.float64, n)
    a = general_batch(n, 21, seed=1000 + n)
    want, info = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    assert not info.any()
    cond = max(np.linalg.cond(m) for m in as_mats(a, n))
    got = gpu_inverse(a, n, GJ, k)
    assert rel_err(got, want, n) < max(1e-10, 1e-15 * cond * n)


@pytest.mark.parametrize("family", ["auto", "lds", "tile", "rowlane"])
@pytest.mark.parametrize("n", SIZES)
def test_cholesky_spd_fp64_vs_oracle(n, family):
    k = family_or_skip(family, CH, torch.float64, n)
    a = spd_batch(n, 19, seed=50 + n)
    want, info = oracle.inverse_batched(a, n, oracle.ALGO_CHOLESKY)
    assert not info.any()
    got, ginfo = gpu_inverse(a, n, CH, k, want_info=True)
    assert not ginfo.any()
    assert rel_err(got, want, n) < 1e-10
    g = as_mats(got, n)
    assert np.array_equal(g, g.transpose(0, 2, 1)) or rel_err(g.transpose(0, 2, 1).reshape(-1), got, n) < 1e-14


@pytest.mark.parametrize("n", [3, 5, 8, 12, 16, 33, 64, 70, 96, 128])
def test_cholesky_reads_only_the_lower_triangle(n):
    """Garbage in the strict upper triangle must not change the result (both Cholesky families)."""
    a = spd_batch(n, 9, seed=7 + n).reshape(9, n, n)      # memory [k, col, row]
    dirty = a.copy()
    iu = np.triu_indices(n, 1)                              # (col, row) with col < row ... careful: memory is [col,row]
    dirty[:, iu[1], iu[0]] = 1e30                          # element (row=iu[0], col=iu[1]) with row < col: upper
    want, _ = oracle.inverse_batched(a.reshape(-1), n, oracle.ALGO_CHOLESKY)
    for fam in ("lds", "tile", "auto"):
        k = family_or_skip(fam, CH, torch.float64, n)
        got, info = gpu_inverse(dirty.reshape(-1), n, CH, k, want_info=True)
        assert not info.any()
        assert rel_err(got, want, n) < 1e-10, fam


@pytest.mark.parametrize("n", [8, 32, 64, 100, 128])
def test_spd_general_not_diagonally_dominant(n):
    """SPD but far from diagonally dominant (Wishart-like G G^T + eps I, cond ~1e3-1e5): no pivoting needed or wanted."""
    rng = np.random.default_rng(n)
    g_ = rng.standard_normal((11, n, n + 3))
    A = g_ @ g_.transpose(0, 2, 1) + 1e-2 * np.eye(n)
    a = np.ascontiguousarray(A.transpose(0, 2, 1)).reshape(-1)
    want, info = oracle.inverse_batched(a, n, oracle.ALGO_CHOLESKY)
    assert not info.any()
    cond = max(np.linalg.cond(m) for m in A)
    for fam in ("lds", "tile", "auto"):
        k = family_or_skip(fam, CH, torch.float64, n)
        got, ginfo = gpu_inverse(a, n, CH, k, want_info=True)
        assert not ginfo.any()
        assert rel_err(got, want, n) < max(1e-10, 1e-14 * cond), (fam, cond)


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("d,n", [("inverse_100_8x8", 8), ("inverse_100_16x16", 16), ("inverse_32_32x32", 32),
                                 ("inverse_12_64x64", 64)])
def test_reference_fixtures_and_goldens(d, n, family, gold):
    k = family_or_skip(family, GJ, torch.float64, n)
    a, cnt, _, _ = read_ref(f"{d}/a.mats")
    got = gpu_inverse(a, n, GJ, k)
    assert rel_err(got, gold[f"{d}/gj"], n) < 1e-10
    got_c = gpu_inverse(a, n, CH)
    assert rel_err(got_c, gold[f"{d}/chol"], n) < 1e-10
    if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "ref", d, "aInv.mats")):
        g, _, _, _ = read_ref(f"{d}/aInv.mats")  # the reference's own 4-digit MATLAB golden
        for x in (got, got_c):
            assert np.abs(x - g).sum() / cnt < 5e-4  # inverse_bench's error metric (inverse_bench.c:49-51,58)


@pytest.mark.parametrize("f,n", [("square_5_8_8", 8), ("square_5_16_16", 16), ("square_5_32_32", 32),
                                 ("square_3_64_64", 64), ("square_1_128_128", 128), ("batch_3", 3)])
def test_square_fixtures_pivoting(f, n, gold):
    a, _, _, _ = read_ref(f + ".mats")
    assert rel_err(gpu_inverse(a, n, GJ), gold[f + "/gj"], n) < 1e-10
    if n <= 128:  # the pivoting MFMA tile kernels themselves (the reference's general fixtures, tests/square_5_*.mats)
        assert rel_err(gpu_inverse(a, n, GJ, api.KERNEL_TILEP), gold[f + "/gj"], n) < 1e-10


@pytest.mark.parametrize("n", list(range(17, 33)))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_natural_pass_two_rows_per_lane(n, dtype):
    """16 < n <= 32: the natural-order pass of the tile family is the two-rows-per-lane kernel where it is the faster one
    (csrc/rowlane2_kernels.hip). A mixed batch -- SPD matrices it finishes, general ones it must reject for the
    pivoting kernel, a singular one and one with a NaN -- whose size is not a multiple of the 4 matrices per wavefront."""
    name = api.kernel_name(GJ, api.F64 if dtype == np.float64 else api.F32, n, api.KERNEL_TILE)
    assert ("rowlane2" in name) == (n <= 25), name
    batch = 203
    a = spd_batch(n, batch, seed=900 + n).reshape(batch, n, n).copy()
    g = general_batch(n, batch, seed=901 + n).reshape(batch, n, n)
    a[1::3] = g[1::3]            # every third matrix needs row exchanges
    a[7, :, n // 2] = 0.0        # memory [k, col, row]: row n/2 of matrix 7 is zero -> singular
    a[12, 3, 3] = np.nan
    a = a.astype(dtype)
    want, winfo = oracle.inverse_batched(a.astype(np.float64).reshape(-1), n, oracle.ALGO_GJ_PIVOT)
    assert winfo[7] != 0 and winfo[12] != 0 and np.count_nonzero(winfo) == 2
    for policy_rep in range(3):  # natural order first, then whatever the adaptive dispatch chooses
        d = dev(a.reshape(-1))  # (not gpu_inverse: its input-unchanged check cannot hold a NaN)
        dinfo = torch.full((batch,), -7, dtype=torch.int32, device="cuda")
        got = api.inverse_batched(d, n, GJ, info=dinfo, kernel=api.KERNEL_TILE, batch=batch).cpu().numpy()
        info = dinfo.cpu().numpy()
        assert np.array_equal(info != 0, winfo != 0)
        assert info[7] == winfo[7] and info[12] == winfo[12]
        gm, wm = as_mats(got.astype(np.float64), n), as_mats(want, n)
        assert np.isnan(gm[7]).all() and np.isnan(gm[12]).all()
        ok = winfo == 0
        cond = max(np.linalg.cond(m) for m in as_mats(a.astype(np.float64).reshape(-1), n)[ok])
        tol = max(1e-10, 1e-15 * cond * n) if dtype == np.float64 else 2e-6 * cond
        err = np.abs(gm[ok] - wm[ok]).max(axis=(1, 2)) / np.abs(wm[ok]).max(axis=(1, 2))
        assert err.max() < tol, (policy_rep, err.max(), tol)


@pytest.mark.parametrize("n", [20, 32, 50, 64, 72, 100, 128, 150, 192])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_family_pivots_inside_the_kernel(n, dtype):
    """General batches through the tile family: the first launch tries the natural order, every matrix it rejects is
    redone by the PIVOTING MFMA kernel in the same stream (not by the lane-per-row kernel), and from the next launch on
    the batch goes to the pivoting kernel directly (the opt-in MATINV_GJ_ADAPTIVE policy, matinv_tile_stats)."""
    old_policy = api.set_gj_policy(api.GJ_ADAPTIVE)
    try:
        _adaptive_dispatch_body(n, dtype)
    finally:
        api.set_gj_policy(old_policy)


def _adaptive_dispatch_body(n, dtype):
    a = general_batch(n, 300, seed=77 + n).astype(dtype)
    want, _ = oracle.inverse_batched(a.astype(np.float64), n, oracle.ALGO_GJ_PIVOT)
    cond = max(np.linalg.cond(m) for m in as_mats(a.astype(np.float64), n))
    tol = max(1e-10, 1e-15 * cond * n) if dtype == np.float64 else 1e-5 * cond
    lib = pkg("_lib").lib()
    b = spd_batch(n, 64, seed=5).astype(dtype)
    for rep in range(34):  # whatever earlier tests left behind: at most 32 launches later the natural order is probed again
        gpu_inverse(b, n, GJ, api.KERNEL_TILE)
    assert lib.matinv_device_synchronize() == 0
    s0 = api.tile_stats()
    assert s0["last_rejected"] == 0
    for rep in range(3):
        got, info = gpu_inverse(a, n, GJ, api.KERNEL_TILE, want_info=True)
        assert not info.any()
        err = rel_err(got.astype(np.float64), want, n) if dtype == np.float64 else \
            np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want)
        assert err < tol, (rep, err, tol)
        assert lib.matinv_device_synchronize() == 0
    s1 = api.tile_stats()
    assert s1["natural_launches"] == s0["natural_launches"] + 1, (s0, s1)  # the first launch tries the natural order
    assert s1["pivot_launches"] == s0["pivot_launches"] + 2, (s0, s1)  # ... then straight to the pivoting kernel
    assert s1["last_rejected"] >= 0.9 * 300 and s1["last_batch"] == 300
    # an SPD batch afterwards flips the guess back after one natural-order launch
    wantb, _ = oracle.inverse_batched(b.astype(np.float64), n, oracle.ALGO_GJ_PIVOT)
    for rep in range(34):
        gotb = gpu_inverse(b, n, GJ, api.KERNEL_TILE)
    assert lib.matinv_device_synchronize() == 0
    s2 = api.tile_stats()
    assert s2["last_rejected"] == 0 and s2["last_batch"] == 64
    errb = np.linalg.norm(gotb.astype(np.float64) - wantb) / np.linalg.norm(wantb)
    assert errb < (1e-12 if dtype == np.float64 else 1e-4)


def mild_batch(n, batch, seed=0, dtype=np.float64):
    """R + R^T + 0.35 n I: symmetric, positive definite for n >= 8 but NOT diagonally dominant -- the natural-order kernels
    accept some of these (multipliers between 1 and 4, where partial pivoting would have moved rows) and reject others."""
    rng = np.random.default_rng(seed)
    r = rng.random((batch, n, n))
    return (r + r.transpose(0, 2, 1) + 0.35 * n * np.eye(n)).reshape(-1).astype(dtype)


@pytest.mark.parametrize("n", [20, 24, 32, 48, 64, 80, 100, 128, 144, 192])
def test_default_policy_is_a_function_of_the_matrix_alone(n):
    """with the r02 adaptive dispatch a matrix that the natural-order kernel accepts but partial pivoting
    would treat differently got either kernel's bits, depending on what had run before. The default policy since r03
    (MATINV_GJ_NATURAL_FIRST) decides per matrix: the same batch gives the same bits (i) again, (ii) after general batches of
    the same size have gone through, (iii) cut into shards of odd sizes, (iv) with its matrices in another order; and the
    PIVOT policy agrees with it within the tolerance (different pivot sequences, same inverse)."""
    assert api.set_gj_policy(api.GJ_NATURAL_FIRST) in (api.GJ_NATURAL_FIRST, api.GJ_PIVOT, api.GJ_ADAPTIVE)
    batch = 301
    a = mild_batch(n, batch, seed=31 + n)
    d = dev(a)
    first = api.inverse_batched(d, n, GJ, batch=batch).clone()
    g = dev(general_batch(n, 64, seed=n))
    for _ in range(3):
        api.inverse_batched(g, n, GJ, batch=64)
    again = api.inverse_batched(d, n, GJ, batch=batch)
    assert torch.equal(first, again), "launch history changed the bits"
    parts = [(0, 7), (7, 130), (130, 131), (131, batch)]
    cut = torch.cat([api.inverse_batched(d[lo * n * n:hi * n * n], n, GJ, batch=hi - lo) for lo, hi in parts])
    assert torch.equal(first, cut), "sharding changed the bits"
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(n)).cuda()
    shuffled = api.inverse_batched(d.view(batch, n * n)[perm].reshape(-1).contiguous(), n, GJ, batch=batch)
    assert torch.equal(first.view(batch, n * n)[perm].reshape(-1), shuffled), "batch mates changed the bits"
    want, winfo = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    assert not winfo.any()
    cond = max(np.linalg.cond(m) for m in as_mats(a, n))
    tol = max(1e-10, 1e-15 * cond * n)
    assert rel_err(first.cpu().numpy(), want, n) < tol
    api.set_gj_policy(api.GJ_PIVOT)
    try:
        piv = api.inverse_batched(d, n, GJ, batch=batch)
        assert rel_err(piv.cpu().numpy(), want, n) < tol
        assert torch.equal(piv, api.inverse_batched(d, n, GJ, batch=batch))
    finally:
        api.set_gj_policy(api.GJ_NATURAL_FIRST)


def test_lu_names_take_the_pivoting_kernel():
    """inverse_lu_cuda_batched_gpu (the reference's cuBLAS getrf/getri entry, src/gauss/inverse_gpu.cu:60-123: partial pivoting
    is its contract) goes straight to the pivoting MFMA kernel: no natural-order launch is counted for it."""
    n, batch = 64, 50
    a = general_batch(n, batch, seed=3)
    want, _ = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    s0 = api.tile_stats()
    out = np.empty_like(a)
    api.inverse_lu_cuda_batched_gpu(n, a, out, batch)
    s1 = api.tile_stats()
    assert s1["natural_launches"] == s0["natural_launches"]
    cond = max(np.linalg.cond(m) for m in as_mats(a, n))
    assert rel_err(out, want, n) < max(1e-10, 1e-15 * cond * n)
    api.inverse_gauss_batched_gpu(n, a, out, batch)  # the Gauss-Jordan name: natural order first, rejects to the pivoting kernel
    s2 = api.tile_stats()
    assert s2["natural_launches"] == s1["natural_launches"] + 1
    assert rel_err(out, want, n) < max(1e-10, 1e-15 * cond * n)


def test_simplemean_cholesky_golden(gold):
    a, _, _, n = read_ref("simpleMean/chol.mats")
    g, _, _, _ = read_ref("simpleMean/cholinv.mats")
    got = gpu_inverse(a, n, CH)
    assert np.abs(got - g).max() < 1e-5
    assert rel_err(got, gold["simpleMean/chol"], n) < 1e-9  # cond ~1185


@pytest.mark.parametrize("n", [3, 8, 16, 32, 64, 90, 128])
def test_singular_and_not_spd_report_info(n):
    a = spd_batch(n, 8, seed=3).reshape(8, n, n)
    # structurally singular inputs are detected exactly by every kernel family (an exactly zero column stays
    # exactly zero under elimination); merely rank-deficient ones depend on rounding and are not asserted
    a[2, 1, :] = 0.0  # column 1 of matrix 2 (memory is [k, col, row])
    a[5] = 0.0
    got, info = gpu_inverse(a.reshape(-1), n, GJ, want_info=True)
    assert info[2] == 2 and info[5] == 1
    assert np.isnan(as_mats(got, n)[2]).all()
    assert (info[[0, 1, 3, 4, 6, 7]] == 0).all()
    assert np.isnan(as_mats(got, n)[5]).all()
    want, _ = oracle.inverse_batched(a.reshape(-1), n, oracle.ALGO_GJ_PIVOT)
    ok = [0, 1, 3, 4, 6, 7]
    assert rel_err(as_mats(got, n)[ok].reshape(-1), as_mats(want, n)[ok].reshape(-1), n) < 1e-10
    b = spd_batch(n, 4, seed=4).reshape(4, n, n)
    b[1, n - 1, n - 1] = -1.0
    got, info = gpu_inverse(b.reshape(-1), n, CH, want_info=True)
    _, oinfo = oracle.inverse_batched(b.reshape(-1), n, oracle.ALGO_CHOLESKY)
    assert info.tolist() == oinfo.tolist() and info[1] == n
    assert np.isnan(as_mats(got, n)[1]).all()


@pytest.mark.parametrize("n,batch", [(142, 5), (200, 4), (257, 3), (512, 2), (1024, 1)])
def test_large_n_global_family(n, batch):
    """Sizes that no longer fit on chip (the reference accepts n <= 1024: one thread per row, batched_invert.cu:87-93)."""
    a = spd_batch(n, batch, seed=n)
    want, _ = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    got, info = gpu_inverse(a, n, GJ, want_info=True)
    assert not info.any() and rel_err(got, want, n) < 1e-10
    gotc, info = gpu_inverse(a, n, CH, want_info=True)
    assert not info.any() and rel_err(gotc, want, n) < 1e-10
    if n <= 512:
        g = general_batch(n, 2, seed=7 * n)
        wg, _ = oracle.inverse_batched(g, n, oracle.ALGO_GJ_PIVOT)
        cond = max(np.linalg.cond(m) for m in as_mats(g, n))
        assert rel_err(gpu_inverse(g, n, GJ), wg, n) < max(1e-10, 1e-15 * cond * n)
        rng = np.random.default_rng(n)
        va, vc, vd = (rng.random(batch * n) for _ in range(3))
        ve = rng.random(batch)
        t = [dev(x) for x in (va, a, vc, vd, ve)]
        m = api.calcluateMean(n, t[0], t[1], t[2], t[3]).cpu().numpy()
        v = api.calcluateVariance(n, t[0], t[1], t[2], t[4]).cpu().numpy()
        assert np.abs(m - oracle.mean_batched(va, a, vc, vd, n)).max() < 1e-10
        assert np.abs(v - oracle.variance_batched(va, a, vc, ve, n)).max() < 1e-10
    with pytest.raises(pkg("_lib").MatinvError):
        api.inverse_batched(torch.zeros(4, dtype=torch.float64, device="cuda"), 1025, GJ, batch=0 + 1)


@pytest.mark.parametrize("forced", [False, True])
@pytest.mark.parametrize("n,batch,dtype", [(129, 9, "f64"), (144, 6, "f64"), (177, 5, "f64"), (192, 5, "f64"), (200, 5, "f64"),
                                           (333, 3, "f64"), (130, 7, "f32"), (144, 6, "f32"), (159, 5, "f32"), (160, 5, "f32"),
                                           (161, 5, "f32"), (200, 5, "f32"), (241, 4, "f32"), (256, 4, "f32"),
                                           (512, 2, "f32"), (1000, 2, "f32")])
def test_cholesky_large_n_blocked(n, batch, dtype, forced):
    """SPD inverse beyond the four-wave kernel: up to 12 x 12 (f64) / 16 x 16 (f32) tiles one wavefront per tile column
    (tile4_impl.hpp, r02), beyond that -- or forced -- blocked Cholesky with an identity border + symmetric product
    (blocked_gp_kernels.hip). Ragged panels / tiles, garbage in the strict upper triangle (must not be read), an item that
    is not positive definite (info = failing column, NaN result, neighbours untouched), both precisions."""
    a = spd_batch(n, batch, seed=40 + n)
    want, _ = oracle.inverse_batched(a, n, oracle.ALGO_CHOLESKY)
    dirty = a.reshape(batch, n, n).copy()
    iu = np.triu_indices(n, 1)
    dirty[:, iu[1], iu[0]] = 1e30                      # strict upper triangle (memory is [k, col, row])
    bad = batch - 1
    dirty[bad, n // 3, n // 3] = -1.0                   # pivot n/3 turns negative
    np_t = np.float64 if dtype == "f64" else np.float32
    wide = n <= (192 if dtype == "f64" else 256)
    assert api.select_kernel(CH, api.F64 if dtype == "f64" else api.F32, n) == (api.KERNEL_TILE if wide else api.KERNEL_BLOCKED)
    if forced and not wide:
        pytest.skip("the automatic path is the blocked one already")
    got, info = gpu_inverse(dirty.reshape(-1).astype(np_t), n, CH, api.KERNEL_BLOCKED if forced else api.KERNEL_AUTO, want_info=True)
    got = got.astype(np.float64)
    assert info.tolist() == [0] * (batch - 1) + [n // 3 + 1]
    assert np.isnan(as_mats(got, n)[bad]).all()
    g, w_ = as_mats(got, n)[:bad], as_mats(want, n)[:bad]
    err = (np.linalg.norm((g - w_).reshape(bad, -1), axis=1) / np.linalg.norm(w_.reshape(bad, -1), axis=1)).max()
    assert err < (1e-12 if dtype == "f64" else 2e-5)
    if forced or not wide:
        assert np.array_equal(g, g.transpose(0, 2, 1))     # blocked path: mirrored on write, exactly symmetric
    else:
        assert np.abs(g - g.transpose(0, 2, 1)).max() < (1e-14 if dtype == "f64" else 1e-6) * np.abs(g).max()


@pytest.mark.parametrize("forced", [False, True])
@pytest.mark.parametrize("n,batch,dtype", [(129, 7, "f64"), (138, 6, "f64"), (161, 5, "f64"), (192, 5, "f64"), (200, 4, "f64"),
                                           (333, 3, "f64"), (130, 7, "f32"), (177, 5, "f32"), (230, 4, "f32"), (256, 4, "f32"),
                                           (512, 2, "f32"), (1000, 2, "f32"), (1024, 1, "f64")])
def test_gauss_jordan_large_n_blocked_general(n, batch, dtype, forced):
    """General (non-symmetric, needs row exchanges) matrices beyond the four-wave kernels: up to 12 x 12 (f64) / 16 x 16 (f32)
    tiles the pivoting MFMA tile kernel with one wavefront per tile column (tilepw_impl.hpp, r02); beyond that -- or forced --
    blocked Gauss-Jordan with partial pivoting over global-memory working copies (blocked_gj_kernels.hip); ragged panels and
    tiles, a structurally singular item (zero column: info = that column, NaN result, neighbours untouched), both precisions,
    and the explicit GLOBAL family as a second opinion."""
    a = general_batch(n, batch, seed=70 + n)
    want, _ = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    cond = max(np.linalg.cond(m) for m in as_mats(a, n))
    bad = batch - 1
    sing = a.reshape(batch, n, n).copy()
    if batch > 1:
        sing[bad, n // 2, :] = 0.0                        # column n/2 of the last item (memory is [k, col, row])
    np_t = np.float64 if dtype == "f64" else np.float32
    code = api.F64 if dtype == "f64" else api.F32
    wide = n <= (192 if dtype == "f64" else 256)
    assert api.select_kernel(GJ, code, n) == (api.KERNEL_TILE if wide else api.KERNEL_BLOCKED)
    if forced and not wide:
        pytest.skip("the automatic path is the blocked one already")
    got, info = gpu_inverse(sing.reshape(-1).astype(np_t), n, GJ, api.KERNEL_BLOCKED if forced else api.KERNEL_AUTO, want_info=True)
    got = got.astype(np.float64)
    ok = batch - 1 if batch > 1 else 1
    if batch > 1:
        assert info.tolist() == [0] * (batch - 1) + [n // 2 + 1]
        assert np.isnan(as_mats(got, n)[bad]).all()
    g, w_ = as_mats(got, n)[:ok], as_mats(want, n)[:ok]
    err = (np.linalg.norm((g - w_).reshape(ok, -1), axis=1) / np.linalg.norm(w_.reshape(ok, -1), axis=1)).max()
    assert err < (max(1e-10, 1e-15 * cond * n) if dtype == "f64" else max(1e-3, 1e-6 * cond))
    if n <= 333 and dtype == "f64":
        ref = gpu_inverse(a[: n * n].copy(), n, GJ, api.KERNEL_GLOBAL)
        assert rel_err(ref, got[: n * n], n) < max(1e-10, 1e-15 * cond * n)


def test_blocked_paths_chunk_the_batch_when_the_workspace_is_capped():
    """The blocked families cut a batch so that their working copies stay under a cap (4 GiB; MATINV_BLOCKED_WS_MB for
    this test: 2 MiB -> chunks of 5 / 5 matrices at n = 150). Runs in a child process because the cap is read once."""
    import subprocess
    import sys
    code = r"""
import sys, importlib, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle
from conftest import spd_batch, general_batch, rel_err
api = importlib.import_module("cuda-matrix-inversion_amd.api")
n, batch = 150, 13
for algo, oalgo, a in ((api.ALGO_GAUSS_JORDAN, oracle.ALGO_GJ_PIVOT, general_batch(n, batch, seed=1)),
                       (api.ALGO_CHOLESKY, oracle.ALGO_CHOLESKY, spd_batch(n, batch, seed=2))):
    want, _ = oracle.inverse_batched(a, n, oalgo)
    info = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    got = api.inverse_batched(torch.from_numpy(a).cuda(), n, algo, batch=batch, info=info, kernel=api.KERNEL_BLOCKED).cpu().numpy()
    assert not info.cpu().numpy().any()
    cond = max(np.linalg.cond(m) for m in a.reshape(batch, n, n))
    assert rel_err(got, want, n) < max(1e-10, 1e-15 * cond * n), algo
n, batch = 210, 11  # pipeline beyond the LDS limit: (n + 2) n doubles per item = 0.36 MB -> chunks of 5
rng = np.random.default_rng(3)
B = spd_batch(n, batch, seed=4)
va, vc, vd = (rng.random(batch * n) for _ in range(3))
ve = rng.random(batch)
t = [torch.from_numpy(x).cuda() for x in (va, B, vc, vd, ve)]
m = api.calcluateMean(n, t[0], t[1], t[2], t[3]).cpu().numpy()
v = api.calcluateVariance(n, t[0], t[1], t[2], t[4]).cpu().numpy()
assert np.abs(m - oracle.mean_batched(va, B, vc, vd, n)).max() < 1e-10
assert np.abs(v - oracle.variance_batched(va, B, vc, ve, n)).max() < 1e-10
print("CHUNKED-OK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MATINV_BLOCKED_WS_MB="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHUNKED-OK" in r.stdout, r.stdout + r.stderr


def test_first_pass_kernels_accept_every_well_conditioned_matrix():
    """The fast kernels hand what they cannot finish (needs row exchanges, not positive definite) to a second kernel, so a fast
    kernel that wrongly rejected EVERYTHING would still pass every parity test -- at a tenth of the speed (r03: a build of the fp32
    symmetric sweep with 9 x 9 tiles did). MATINV_DEBUG_REJECTS=1 (matinv.h: matinv_debug_rejects) counts the hand-overs: zero for
    diagonally dominant SPD batches through Gauss-Jordan, Cholesky and the fused pipeline at every size class and both precisions,
    and non-zero for a general batch (the counter itself works). Child process: the switch is read once."""
    import subprocess
    import sys
    code = r"""
import sys, importlib, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import spd_batch, general_batch
api = importlib.import_module("cuda-matrix-inversion_amd.api")
assert api.debug_rejects(reset=True) == 0
batch = 37
sizes = [3, 8, 16, 17, 24, 25, 31, 32, 40, 48, 63, 64, 65, 72, 80, 81, 96, 97, 100, 112, 113, 120, 128, 129, 144, 160, 161, 176, 192, 200, 256]
for dt, code_ in ((np.float64, api.F64), (np.float32, api.F32)):
    for n in sizes:
        a = torch.from_numpy(spd_batch(n, batch, seed=n, dtype=dt)).cuda()
        rng = np.random.default_rng(n)
        v = [torch.from_numpy(rng.random(batch * n).astype(dt)).cuda() for _ in range(3)]
        for what in ("gj", "chol", "mean"):
            if what == "gj":
                api.inverse_batched(a, n, api.ALGO_GAUSS_JORDAN, batch=batch)
            elif what == "chol":
                api.inverse_batched(a, n, api.ALGO_CHOLESKY, batch=batch)
            else:
                api.calcluateMean(n, v[0], a, v[1], v[2])
            torch.cuda.synchronize()
            r = api.debug_rejects(reset=True)
            assert r == 0, (what, n, dt.__name__, r, api.kernel_name(api.ALGO_CHOLESKY if what != "gj" else api.ALGO_GAUSS_JORDAN, code_, n))
g = torch.from_numpy(general_batch(64, 50, seed=1)).cuda()
api.inverse_batched(g, 64, api.ALGO_GAUSS_JORDAN, batch=50)
torch.cuda.synchronize()
assert api.debug_rejects(reset=True) > 40
print("REJECTS-OK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MATINV_DEBUG_REJECTS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REJECTS-OK" in r.stdout, r.stdout + r.stderr


def test_large_n_singular_and_fp32():
    n = 160
    a = spd_batch(n, 3, seed=1).reshape(3, n, n)
    a[1, 5, :] = 0.0
    got, info = gpu_inverse(a.reshape(-1), n, GJ, want_info=True)
    assert info.tolist() == [0, 6, 0] and np.isnan(as_mats(got, n)[1]).all()
    b = spd_batch(n, 2, seed=2)
    want, _ = oracle.inverse_batched(b, n, oracle.ALGO_GJ_PIVOT)
    n32 = 256
    c = spd_batch(n32, 2, seed=3)
    want32, _ = oracle.inverse_batched(c, n32, oracle.ALGO_GJ_PIVOT)
    got32 = gpu_inverse(c.astype(np.float32), n32, GJ).astype(np.float64)
    x, y = got32.reshape(-1, n32 * n32), want32.reshape(-1, n32 * n32)
    assert (np.linalg.norm(x - y, axis=1) / np.linalg.norm(y, axis=1)).max() < 1e-4
    assert rel_err(gpu_inverse(b, n, GJ), want, n) < 1e-10


@pytest.mark.parametrize("n,batch,dtype", [(142, 7, "f64"), (201, 5, "f64"), (257, 3, "f64"), (500, 3, "f64"), (512, 2, "f32"),
                                           (1000, 2, "f32"), (1024, 2, "f64")])
def test_pipeline_large_n_blocked(n, batch, dtype):
    """Fused mean / variance beyond the LDS limit: blocked multi-launch Cholesky with the vectors as border rows
    (blocked_gp_kernels.hip); ragged last panel / tile (n not a multiple of 32 or 64), both precisions, and an item
    that is not positive definite (info = failing column, NaN result, neighbours untouched)."""
    rng = np.random.default_rng(n)
    B = spd_batch(n, batch, seed=n + 1)
    va, vc, vd = (rng.random(batch * n) for _ in range(3))
    ve = rng.random(batch)
    wm = oracle.mean_batched(va, B, vc, vd, n)
    wv = oracle.variance_batched(va, B, vc, ve, n)
    bad = batch - 1
    B, vc = B.copy(), vc.copy()
    B.reshape(batch, n, n)[bad, n // 2, n // 2] = -1e3  # pivot n/2 turns negative (the oracle refuses such input)
    vc.reshape(batch, n)[bad, n // 2] = 0.0
    np_t = np.float64 if dtype == "f64" else np.float32
    t = [dev(x.astype(np_t)) for x in (va, B, vc, vd, ve)]
    info_m = torch.zeros(batch, dtype=torch.int32, device="cuda")
    info_v = torch.zeros(batch, dtype=torch.int32, device="cuda")
    m = api.calcluateMean(n, t[0], t[1], t[2], t[3], info=info_m).cpu().numpy().astype(np.float64)
    v = api.calcluateVariance(n, t[0], t[1], t[2], t[4], info=info_v).cpu().numpy().astype(np.float64)
    good = np.arange(batch) != bad
    tol = 1e-10 if dtype == "f64" else 2e-4
    assert np.abs(m[good] - wm[good]).max() < tol * max(1.0, np.abs(wm[good]).max())
    assert np.abs(v[good] - wv[good]).max() < tol * max(1.0, np.abs(wv[good]).max())
    assert np.isnan(m[bad]) and np.isnan(v[bad])
    assert info_m.cpu().tolist() == [0] * (batch - 1) + [n // 2 + 1] == info_v.cpu().tolist()


@pytest.mark.parametrize("n,batch,dtype", [(200, 5, "f64"), (512, 3, "f32")])
def test_blocked_pipeline_graph_replay_follows_the_data(n, batch, dtype):
    """A latency-bound call of the blocked fused pipeline that repeats its pointers is replayed as a HIP graph from the third call on
    (blocked_gp_kernels.hip): the replay must read the buffers as they are NOW (values changed in place between the calls), report a
    matrix that stopped being positive definite, and agree bit for bit with the direct launches of the first call."""
    rng = np.random.default_rng(n)
    np_t = np.float64 if dtype == "f64" else np.float32
    B = spd_batch(n, batch, seed=n + 7)
    va, vc, vd = (rng.random(batch * n) for _ in range(3))
    t = [dev(x.astype(np_t)) for x in (va, B, vc, vd)]
    out = torch.empty(batch, dtype=t[0].dtype, device="cuda")
    info = torch.zeros(batch, dtype=torch.int32, device="cuda")
    tol = 1e-10 if dtype == "f64" else 2e-4
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        first = None
        for call in range(5):
            api.calcluateMean(n, t[0], t[1], t[2], t[3], Means=out, info=info)
            st.synchronize()
            if call == 0:
                first = out.clone()
            assert torch.equal(out, first) and int(info.abs().sum()) == 0
        want = oracle.mean_batched(va, B, vc, vd, n)
        assert np.abs(out.cpu().numpy().astype(np.float64) - want).max() < tol * max(1.0, np.abs(want).max())
        # new values in the same buffers
        va2 = rng.random(batch * n)
        B2 = spd_batch(n, batch, seed=n + 8)
        t[0].copy_(dev(va2.astype(np_t)))
        t[1].copy_(dev(B2.astype(np_t)))
        api.calcluateMean(n, t[0], t[1], t[2], t[3], Means=out, info=info)
        st.synchronize()
        want2 = oracle.mean_batched(va2, B2, vc, vd, n)
        assert np.abs(out.cpu().numpy().astype(np.float64) - want2).max() < tol * max(1.0, np.abs(want2).max())
        # and an item that is no longer positive definite
        t[1].view(batch, n, n)[1, n // 3, n // 3] = -1e3
        t[2].view(batch, n)[1, n // 3] = 0.0
        api.calcluateMean(n, t[0], t[1], t[2], t[3], Means=out, info=info)
        st.synchronize()
        assert info.cpu().tolist() == [0, n // 3 + 1] + [0] * (batch - 2) and bool(torch.isnan(out[1]))


@pytest.mark.parametrize("n,batch,dtype", [(200, 530, "f64"), (257, 340, "f32"), (333, 240, "f64")])
def test_blocked_cholesky_paths_panel_pairs(n, batch, dtype):
    """Batches large enough for the blocked Cholesky path to apply its 64-column panels in PAIRS (narrow update, second panel,
    one rank-128 update: blocked_gp_kernels.hip, bgp_pairs_pay) -- the small batches of the tests above take one update per
    panel. SPD inverse against the oracle and the fused mean against a float64 solve; ragged n (last panel, last pair and
    last tile incomplete), a not-positive-definite item in the middle whose neighbours must be untouched."""
    np_t = np.float64 if dtype == "f64" else np.float32
    tol = 1e-11 if dtype == "f64" else 5e-5
    a = spd_batch(n, batch, seed=7000 + n)
    bad = batch // 2
    dirty = a.reshape(batch, n, n).copy()
    dirty[bad, 70, 70] = -1.0  # second panel of the first pair
    got, info = gpu_inverse(dirty.reshape(-1).astype(np_t), n, CH, api.KERNEL_BLOCKED, want_info=True)
    assert info[bad] == 71 and np.count_nonzero(info) == 1
    g = as_mats(got.astype(np.float64), n)
    assert np.isnan(g[bad]).all()
    ok = np.arange(batch) != bad
    A = as_mats(a, n)[ok]                      # symmetric: memory order is irrelevant
    res = np.abs(np.einsum("bij,bjk->bik", A, g[ok]) - np.eye(n)).max()
    assert res < tol * n, res
    assert np.array_equal(g[ok], g[ok].transpose(0, 2, 1))
    want, _ = oracle.inverse_batched(a.reshape(batch, -1)[:6].reshape(-1), n, oracle.ALGO_CHOLESKY)
    w6 = as_mats(want, n)
    assert np.abs(g[:6] - w6).max() < tol * np.abs(w6).max() * 10
    # fused mean on the same matrices: a^T (B + diag c)^-1 d
    rng = np.random.default_rng(n)
    va, vc, vd = (rng.random((batch, n)) for _ in range(3))
    t = [dev(x.reshape(-1).astype(np_t)) for x in (va, a, vc, vd)]
    info_m = torch.zeros(batch, dtype=torch.int32, device="cuda")
    m = api.calcluateMean(n, t[0], t[1], t[2], t[3], info=info_m).cpu().numpy().astype(np.float64)
    assert not info_m.cpu().numpy().any()
    M = as_mats(a, n) + np.einsum("bi,ij->bij", vc, np.eye(n))
    wm = np.einsum("bi,bi->b", va, np.linalg.solve(M, vd[:, :, None])[:, :, 0])
    assert np.abs(m - wm).max() < (1e-10 if dtype == "f64" else 2e-4) * max(1.0, np.abs(wm).max())


def test_randomized_sizes_batches_dtypes():
    """Seeded sweep over odd sizes / batch counts (ragged last wavefront, identity padding, every dispatch boundary):
    n in 1..140, batch in 1..260, both precisions, both algorithms, SPD and general inputs."""
    rng = np.random.default_rng(20261003)
    for trial in range(60):
        n = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 23, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 97, 127, 128, 129, 140]))
        batch = int(rng.integers(1, 261)) if n <= 64 else int(rng.integers(1, 12))
        f32 = bool(rng.integers(0, 2))
        general = bool(rng.integers(0, 3) == 0) and n <= 64
        algo = GJ if (general or rng.integers(0, 2)) else CH
        a = general_batch(n, batch, seed=trial) if general else spd_batch(n, batch, seed=trial)
        want, oinfo = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT if algo == GJ else oracle.ALGO_CHOLESKY)
        assert not oinfo.any()
        got, info = gpu_inverse(a.astype(np.float32) if f32 else a, n, algo, want_info=True)
        assert not info.any(), (trial, n, batch, f32, general, algo)
        cond = max(np.linalg.cond(m) for m in as_mats(a, n)[: min(batch, 8)]) if general else 3.0
        if f32:
            x, y = got.astype(np.float64).reshape(-1, n * n), want.reshape(-1, n * n)
            fro = np.linalg.norm(x - y, axis=1) / np.linalg.norm(y, axis=1)
            assert fro.max() < 2e-5 * max(cond, 10.0), (trial, n, batch, general, algo, fro.max())
        else:
            assert rel_err(got, want, n) < max(1e-10, 1e-14 * cond * n), (trial, n, batch, general, algo)


def test_empty_and_single_batch():
    e = torch.empty(0, dtype=torch.float64, device="cuda")
    out = api.inverse_batched(e, 8, GJ, batch=0)
    assert out.numel() == 0
    a = spd_batch(16, 1, seed=9)
    assert rel_err(gpu_inverse(a, 16), oracle.inverse_batched(a, 16)[0], 16) < 1e-10


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("n", [4, 8, 16, 32, 64, 128])
def test_fp32_vs_fp64_oracle(n, family):
    k = family_or_skip(family, GJ, torch.float32, n)
    a = spd_batch(n, 16, seed=n)
    want, _ = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    cond = max(np.linalg.cond(m) for m in as_mats(a, n))
    for algo in (GJ, CH):
        # (the Cholesky side always goes through the automatic choice: SPD tile kernel for n <= 64, LDS beyond)
        got = gpu_inverse(a.astype(np.float32), n, algo, k if algo == GJ else api.KERNEL_AUTO).astype(np.float64)
        x, y = got.reshape(-1, n * n), want.reshape(-1, n * n)
        fro = np.linalg.norm(x - y, axis=1) / np.linalg.norm(y, axis=1)
        assert fro.max() < 1e-5 * cond


def test_strided_batch_and_in_place():
    n, batch, stride = 16, 9, 16 * 16 + 40
    a = spd_batch(n, batch, seed=11).reshape(batch, n * n)
    buf = np.full((batch, stride), 123.0)
    buf[:, : n * n] = a
    d_in = dev(buf.reshape(-1))
    d_out = torch.full_like(d_in, -5.0)
    api.inverse_batched(d_in, n, GJ, out=d_out, batch=batch, stride=stride)
    res = d_out.cpu().numpy().reshape(batch, stride)
    want, _ = oracle.inverse_batched(a.reshape(-1), n)
    assert rel_err(res[:, : n * n].reshape(-1), want, n) < 1e-10
    assert (res[:, n * n:] == -5.0).all(), "padding between matrices was written"
    api.inverse_batched(d_in, n, GJ, out=d_in, batch=batch, stride=stride)  # exact aliasing is allowed
    assert rel_err(d_in.cpu().numpy().reshape(batch, stride)[:, : n * n].reshape(-1), want, n) < 1e-10


@pytest.mark.parametrize("n", [24, 50, 64, 100, 128, 150, 192, 200, 300])
def test_strided_batch_and_in_place_every_family(n):
    """The same contract (matinv.h: matrix k at k * stride, padding never written, exact aliasing of input and output allowed) at a
    size of every kernel family, general input through Gauss-Jordan and SPD input through Cholesky. The blocked Gauss-Jordan reads a
    strided batch in place during its first block and writes the caller's buffer from its last block-level update (r03)."""
    batch, stride = 5, n * n + 40
    for algo, oalgo, a in ((GJ, oracle.ALGO_GJ_PIVOT, general_batch(n, batch, seed=n)), (CH, oracle.ALGO_CHOLESKY, spd_batch(n, batch, seed=n))):
        want, _ = oracle.inverse_batched(a, n, oalgo)
        cond = max(np.linalg.cond(m) for m in as_mats(a, n))
        tol = max(1e-10, 1e-15 * cond * n)
        buf = np.full((batch, stride), 123.0)
        buf[:, : n * n] = a.reshape(batch, n * n)
        d_in = dev(buf.reshape(-1))
        d_out = torch.full_like(d_in, -5.0)
        api.inverse_batched(d_in, n, algo, out=d_out, batch=batch, stride=stride)
        res = d_out.cpu().numpy().reshape(batch, stride)
        assert rel_err(res[:, : n * n].reshape(-1), want, n) < tol, (algo, n)
        assert (res[:, n * n:] == -5.0).all(), "padding between matrices was written"
        assert np.array_equal(d_in.cpu().numpy(), buf.reshape(-1)), "input batch was modified"
        api.inverse_batched(d_in, n, algo, out=d_in, batch=batch, stride=stride)
        back = d_in.cpu().numpy().reshape(batch, stride)
        assert rel_err(back[:, : n * n].reshape(-1), want, n) < tol, (algo, n, "in place")
        assert (back[:, n * n:] == 123.0).all()


@pytest.mark.parametrize("n", [8, 16, 32, 64, 100, 128, 200])
def test_nan_and_inf_inputs_are_reported_not_propagated(n):
    """A NaN or an Inf inside one matrix: that matrix is reported (info != 0, result all NaN) by every family on the
    automatic path, the other matrices of the batch are inverted as usual."""
    batch = 6
    a = spd_batch(n, batch, seed=17).reshape(batch, n, n)
    want, _ = oracle.inverse_batched(a.reshape(-1), n)
    a[1, n // 2, n // 3] = np.nan
    a[1, n // 3, n // 2] = np.nan
    a[4, 0, 0] = np.inf
    ok = [0, 2, 3, 5]
    for algo in (GJ, CH):
        d_in = dev(a.reshape(-1))
        t_info = torch.full((batch,), -7, dtype=torch.int32, device="cuda")
        got = api.inverse_batched(d_in, n, algo, info=t_info, batch=batch).cpu().numpy()
        info = t_info.cpu().numpy()
        assert np.array_equal(d_in.cpu().numpy(), a.reshape(-1), equal_nan=True), "input batch was modified"
        assert info[1] != 0 and info[4] != 0 and not info[ok].any(), (algo, info)
        g = as_mats(got, n)
        assert np.isnan(g[1]).all() and np.isnan(g[4]).all()
        assert rel_err(g[ok].reshape(-1), as_mats(want, n)[ok].reshape(-1), n) < 1e-10


@pytest.mark.parametrize("n,dtype", [(64, np.float64), (64, np.float32), (32, np.float64), (128, np.float64), (128, np.float32)])
def test_odd_stride_breaks_vector_alignment(n, dtype):
    """Strides that are odd in elements put every other matrix on an address that is only element-aligned: the kernels'
    8- and 16-byte vector accesses must still be correct (and must not touch the gap), for both algorithms."""
    batch, stride = 7, n * n + 1
    a = spd_batch(n, batch, seed=13).astype(dtype).reshape(batch, n * n)
    buf = np.full((batch, stride), 123.0, dtype=dtype)
    buf[:, : n * n] = a
    want, _ = oracle.inverse_batched(a.astype(np.float64).reshape(-1), n)
    tol = 1e-10 if dtype == np.float64 else 1e-4
    for algo in (GJ, CH):
        d_in = dev(buf.reshape(-1))
        d_out = torch.full_like(d_in, -5.0)
        api.inverse_batched(d_in, n, algo, out=d_out, batch=batch, stride=stride)
        res = d_out.cpu().numpy().reshape(batch, stride)
        assert rel_err(res[:, : n * n].reshape(-1).astype(np.float64), want, n) < tol
        assert (res[:, n * n:] == -5.0).all(), "padding between matrices was written"
        assert torch.equal(d_in.cpu(), torch.from_numpy(buf.reshape(-1)))


# ------------------------------------------------------------------ reference-named entry points (C symbols)
@pytest.mark.parametrize("name", ["inverse_gauss_batched_gpu", "inverse_lu_cuda_batched_gpu",
                                  "inverse_cholesky_batched_gpu", "inverse_cholesky_mm_batched_gpu",
                                  "inverse_cholesky_mm2_batched_gpu", "inverse_cholesky_stride_batched_gpu"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,batch", [(32, 50), (128, 9), (150, 5)])
def test_reference_host_pointer_family(name, dtype, n, batch):
    """inverse_bench's call sites (src/inverse_bench.c:144,168,191,214): host arrays in, host arrays out; the reference's
    sweep sizes up to 128 (`make run-inverse-bench`) and one size beyond (blocked families)."""
    a = spd_batch(n, batch, seed=21).astype(dtype)
    keep = a.copy()
    out = np.zeros_like(a)
    getattr(api, name)(n, a, out, batch)
    assert np.array_equal(a, keep), "As must not be clobbered (the reference's Cholesky paths do, :442)"
    want, _ = oracle.inverse_batched(keep.astype(np.float64), n)
    assert rel_err(out.astype(np.float64), want, n) < (1e-10 if dtype == np.float64 else 1e-4)


def test_staging_cache_is_reused_and_can_be_released():
    """Repeated host-pointer calls reuse the pooled staging buffers (same results), and matinv_release_cache() hands the
    pool back to the driver without affecting later calls."""
    L = pkg("_lib").lib()
    n, batch = 48, 300
    a = spd_batch(n, batch, seed=5)
    want, _ = oracle.inverse_batched(a, n)
    outs = []
    for _ in range(3):
        out = np.zeros_like(a)
        api.inverse_gauss_batched_gpu(n, a, out, batch)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    assert rel_err(outs[0], want, n) < 1e-10
    free0 = torch.cuda.mem_get_info()[0]
    assert L.matinv_release_cache() == 0
    assert torch.cuda.mem_get_info()[0] >= free0
    out = np.zeros_like(a)
    api.inverse_gauss_batched_gpu(n, a, out, batch)
    assert np.array_equal(out, outs[0])


def _table(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    return arr


@pytest.mark.parametrize("uniform", [True, False])
def test_reference_device_table_family(uniform):
    """gauss_bench's dispatch point (src/gauss_bench.cu:77): host tables of device pointers, async on stream 0."""
    n, batch = 16, 12
    L = pkg("_lib").lib()
    a = spd_batch(n, batch, seed=31)
    pitch = n * n + (24 if uniform else 0)
    d_in = torch.zeros(batch * (n * n + 64), dtype=torch.float64, device="cuda")
    d_out = torch.zeros_like(d_in)
    offs = [i * pitch for i in range(batch)] if uniform else [((i * 7) % batch) * (n * n + 64) for i in range(batch)]
    for i, o in enumerate(offs):
        d_in[o:o + n * n] = torch.from_numpy(a[i * n * n:(i + 1) * n * n]).cuda()
    tin = _table([d_in.data_ptr() + 8 * o for o in offs])
    tout = _table([d_out.data_ptr() + 8 * o for o in offs])
    want, _ = oracle.inverse_batched(a, n)
    torch.cuda.synchronize()
    for name in ("inverse_gauss_batched_device", "inverse_lu_cuda_batched_device", "inverse_cholesky_batched_device",
                 "inverse_cholesky_mm_batched_device", "inverse_cholesky_mm2_batched_device"):
        d_out.zero_()
        torch.cuda.synchronize()
        getattr(L, name)(None, n, tin, tout, batch)
        torch.cuda.synchronize()
        res = d_out.cpu().numpy()
        got = np.concatenate([res[o:o + n * n] for o in offs])
        assert rel_err(got, want, n) < 1e-10, name


@pytest.mark.parametrize("uniform", [True, False])
def test_reference_device_table_family_blocked_sizes(uniform):
    """The same dispatch point at a size the blocked paths serve (n = 200: two 128-column blocks): a GENERAL batch through
    inverse_gauss_batched_device -- uniform pitch > n^2 = strided input read in place by the first block, scattered pointers = the
    gathering copy; either way the last block-level update writes the permuted columns straight into the caller's buffers (r03) --
    and an SPD batch through inverse_cholesky_batched_device. One singular item: info is not reported through this family, its
    result is NaN and its neighbours are untouched."""
    n, batch = 200, 6
    L = pkg("_lib").lib()
    g = general_batch(n, batch, seed=77).reshape(batch, n, n)
    g[4, 11, :] = 0.0                      # column 11 of item 4 (memory is [k, col, row])
    spd = spd_batch(n, batch, seed=78)
    pitch = n * n + (40 if uniform else 0)
    slot = n * n + 128
    d_in = torch.zeros(batch * slot, dtype=torch.float64, device="cuda")
    d_out = torch.zeros_like(d_in)
    offs = [i * pitch for i in range(batch)] if uniform else [((i * 5) % batch) * slot for i in range(batch)]
    tin = _table([d_in.data_ptr() + 8 * o for o in offs])
    tout = _table([d_out.data_ptr() + 8 * o for o in offs])
    for name, a, oalgo in (("inverse_gauss_batched_device", g.reshape(-1), oracle.ALGO_GJ_PIVOT),
                           ("inverse_cholesky_batched_device", spd, oracle.ALGO_CHOLESKY)):
        d_in.zero_()
        d_out.zero_()
        for i, o in enumerate(offs):
            d_in[o:o + n * n] = torch.from_numpy(a[i * n * n:(i + 1) * n * n]).cuda()
        torch.cuda.synchronize()
        getattr(L, name)(None, n, tin, tout, batch)
        torch.cuda.synchronize()
        res = d_out.cpu().numpy()
        got = np.concatenate([res[o:o + n * n] for o in offs])
        keep = [i for i in range(batch) if not (name.startswith("inverse_gauss") and i == 4)]
        want, _ = oracle.inverse_batched(np.concatenate([a[i * n * n:(i + 1) * n * n] for i in keep]), n, oalgo)
        gk = np.concatenate([got[i * n * n:(i + 1) * n * n] for i in keep])
        cond = max(np.linalg.cond(m) for m in as_mats(np.concatenate([a[i * n * n:(i + 1) * n * n] for i in keep]), n))
        assert rel_err(gk, want, n) < max(1e-10, 1e-15 * cond * n), name
        if len(keep) < batch:
            assert np.isnan(got[4 * n * n:5 * n * n]).all()


def test_cholesky_subphase_entry_points():
    """decompose / inverse_upper / multiply_upper (inverse_gpu.h:15-24): in place, lower triangle, upper zeroed."""
    n, batch = 24, 6
    L = pkg("_lib").lib()
    a = spd_batch(n, batch, seed=41)
    d = dev(a)
    tab = _table([d.data_ptr() + 8 * i * n * n for i in range(batch)])
    L.decompose_cholesky_batched_device(None, n, tab, tab, batch)
    torch.cuda.synchronize()
    Lf = as_mats(d.cpu().numpy(), n)
    assert np.allclose(np.triu(Lf, 1), 0)
    A = as_mats(a, n)
    assert np.abs(Lf @ Lf.transpose(0, 2, 1) - A).max() < 1e-12 * n * np.abs(A).max()
    L.inverse_upper_stride_batched_device(None, n, tab, tab, batch)
    torch.cuda.synchronize()
    Li = as_mats(d.cpu().numpy(), n)
    assert np.abs(Li @ Lf - np.eye(n)).max() < 1e-12
    L.multiply_upper_stride_batched_device(None, n, tab, tab, batch)
    torch.cuda.synchronize()
    want, _ = oracle.inverse_batched(a, n, oracle.ALGO_CHOLESKY)
    assert rel_err(d.cpu().numpy(), want, n) < 1e-10


# ---------------------------------------------------------------------------------------------- pipeline
@pytest.mark.parametrize("d,n", [("gaussian_100_8x8", 8), ("gaussian_100_16x16", 16), ("gaussian_32_32x32", 32),
                                 ("gaussian_12_64x64", 64)])
def test_pipeline_fixtures(d, n, gold):
    r = {f: read_ref(f"{d}/{f}.mats")[0] for f in ("a", "b", "c", "d", "e", "means", "variances")}
    t = {k: dev(v) for k, v in r.items()}
    keep_b = t["b"].clone()
    means = api.calcluateMean(n, t["a"], t["b"], t["c"], t["d"])
    var = api.calcluateVariance(n, t["a"], t["b"], t["c"], t["e"])
    torch.cuda.synchronize()
    assert torch.equal(keep_b, t["b"])
    m, v = means.cpu().numpy(), var.cpu().numpy()
    assert np.abs(m - gold[f"{d}/means"]).max() < 1e-10 * max(1.0, np.abs(m).max())
    assert np.abs(v - gold[f"{d}/variances"]).max() < 1e-10 * max(1.0, np.abs(v).max())
    assert np.abs(m - r["means"]).mean() < 5e-5       # the reference's 4-digit goldens
    assert np.abs(v - r["variances"]).mean() < 5e-5


@pytest.mark.parametrize("n", [1, 2, 5, 7, 8, 9, 12, 16, 17, 33, 64, 65, 80, 81, 88, 96, 97, 100, 104, 112, 113, 127, 128, 129,
                               130, 144, 145, 159, 160, 161])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pipeline_synthetic(n, dtype):
    rng = np.random.default_rng(n)
    batch = 23
    B = spd_batch(n, batch, seed=n)
    a, c, d_ = (rng.random(batch * n) for _ in range(3))
    e = rng.random(batch)
    wm = oracle.mean_batched(a, B, c, d_, n)
    wv = oracle.variance_batched(a, B, c, e, n)
    t = [dev(x.astype(dtype)) for x in (a, B, c, d_, e)]
    m = api.calcluateMean(n, t[0], t[1], t[2], t[3]).cpu().numpy().astype(np.float64)
    v = api.calcluateVariance(n, t[0], t[1], t[2], t[4]).cpu().numpy().astype(np.float64)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert np.abs(m - wm).max() < tol and np.abs(v - wv).max() < tol


@pytest.mark.parametrize("n", [120, 130, 150, 160])
def test_pipeline_not_spd_f32_wide_one_wave(n):
    """fp32 fused pipeline on the one-wavefront symmetric sweep (8 x 8 .. 10 x 10 lower tiles, r03): an item whose matrix is not
    positive definite goes to the LDS pipeline through the work list and comes back as info = failing column + NaN."""
    batch = 9
    rng = np.random.default_rng(n)
    B = spd_batch(n, batch, seed=n).reshape(batch, n, n)
    B[4, n - 3, n - 3] = -50.0 * n
    a, c, d_ = (rng.random(batch * n) for _ in range(3))
    t = [dev(x.astype(np.float32)) for x in (a, B.reshape(-1), c, d_)]
    info = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    m = api.calcluateMean(n, t[0], t[1], t[2], t[3], info=info).cpu().numpy().astype(np.float64)
    inf = info.cpu().numpy()
    assert inf[4] == n - 2 and (np.delete(inf, 4) == 0).all()
    assert np.isnan(m[4])
    keep = lambda x, w: np.delete(x.reshape(batch, w), 4, 0).reshape(-1)
    want = oracle.mean_batched(keep(a, n), keep(B, n * n), keep(c, n), keep(d_, n), n)
    assert np.abs(np.delete(m, 4) - want).max() < 2e-5


@pytest.mark.parametrize("n", [8, 13, 32, 72, 90, 100])
def test_pipeline_not_spd_reports_info_and_nan(n):
    batch = 6
    rng = np.random.default_rng(5)
    B = spd_batch(n, batch, seed=5).reshape(batch, n, n)
    B[3, 7, 7] = -50.0 * n                   # breaks positive definiteness of item 3
    a, c, d_ = (rng.random(batch * n) for _ in range(3))
    t = [dev(x) for x in (a, B.reshape(-1), c, d_)]
    info = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    m = api.calcluateMean(n, t[0], t[1], t[2], t[3], info=info).cpu().numpy()
    inf = info.cpu().numpy()
    assert inf[3] == 8 and (np.delete(inf, 3) == 0).all()
    assert np.isnan(m[3])
    want = oracle.mean_batched(np.delete(a.reshape(batch, n), 3, 0).reshape(-1), np.delete(B, 3, 0).reshape(-1),
                               np.delete(c.reshape(batch, n), 3, 0).reshape(-1),
                               np.delete(d_.reshape(batch, n), 3, 0).reshape(-1), n)
    assert np.abs(np.delete(m, 3) - want).max() < 1e-10


@pytest.mark.parametrize("n,batch", [(16, 200_000), (64, 100_000)])
def test_pipeline_full_size_properties(n, batch):
    """Size-independent checks at BASELINE batch sizes: mean is bilinear in (a, d); var(a) = e - mean(a, a);
    scaling B and c by 4 divides the quadratic form by 4 exactly."""
    g = torch.Generator(device="cuda").manual_seed(99 + n)
    r = torch.rand((batch, n, n), generator=g, dtype=torch.float64, device="cuda")
    B = (r + r.transpose(1, 2) + n * torch.eye(n, dtype=torch.float64, device="cuda")).reshape(-1)
    del r
    a, c, d_ = (torch.rand(batch * n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3))
    e = torch.rand(batch, generator=g, dtype=torch.float64, device="cuda")
    m_ad = api.calcluateMean(n, a, B, c, d_)
    m_da = api.calcluateMean(n, d_, B, c, a)
    assert float((m_ad - m_da).abs().max()) < 1e-13                       # symmetry of M^-1
    m_aa = api.calcluateMean(n, a, B, c, a)
    v = api.calcluateVariance(n, a, B, c, e)
    assert float((v - (e - m_aa)).abs().max()) < 1e-13
    m_2 = api.calcluateMean(n, a, B, c, d_ * 2.0)
    assert torch.equal(m_2, m_ad * 2.0)                                    # linear in d, exact for powers of two
    m_s = api.calcluateMean(n, a, B * 4.0, c * 4.0, d_)
    assert torch.equal(m_s * 4.0, m_ad)
    idx = torch.arange(0, batch, batch // 50, device="cuda")[:50].cpu().numpy()
    sel = lambda t, w: t.view(batch, w)[idx].reshape(-1).cpu().numpy()
    want = oracle.mean_batched(sel(a, n), sel(B, n * n), sel(c, n), sel(d_, n), n)
    assert np.abs(m_ad[idx].cpu().numpy() - want).max() < 1e-10


# ------------------------------------------------------------- full-size, size-independent properties
@pytest.mark.parametrize("n,batch", [(16, 100_000), (64, 100_000)])
@pytest.mark.parametrize("algo", [GJ, CH])
def test_full_size_properties(n, batch, algo):
    """BASELINE.json configs 2 and 3 at full size. The oracle would take minutes here, so check properties:
    residual A*X = I, involution inv(inv(A)) = A, homogeneity inv(4A) = inv(A)/4 (exact in binary fp, sqrt included),
    and spot-check 64 matrices against the oracle."""
    g = torch.Generator(device="cuda").manual_seed(1234 + n)
    r = torch.rand((batch, n, n), generator=g, dtype=torch.float64, device="cuda")
    a = r + r.transpose(1, 2) + n * torch.eye(n, dtype=torch.float64, device="cuda")
    del r
    flat = a.reshape(-1)
    info = torch.empty(batch, dtype=torch.int32, device="cuda")
    x = api.inverse_batched(flat, n, algo, info=info)
    assert int(info.abs().sum()) == 0
    xm = x.view(batch, n, n)
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    res = (torch.bmm(a, xm) - eye).abs().amax()
    assert float(res) < 1e-13 * n
    back = api.inverse_batched(x, n, algo)
    assert float((back.view(batch, n, n) - a).abs().amax() / a.abs().amax()) < 1e-12
    quarter = api.inverse_batched(flat * 4.0, n, algo)
    assert torch.equal(quarter * 4.0, x), "inv(4A) must equal inv(A)/4 bit for bit (power-of-4 scaling, exact in sqrt too)"
    idx = torch.arange(0, batch, batch // 64, device="cuda")[:64]
    sub = a[idx].reshape(-1).cpu().numpy()
    want, _ = oracle.inverse_batched(sub, n, oracle.ALGO_GJ_PIVOT if algo == GJ else oracle.ALGO_CHOLESKY)
    assert rel_err(xm[idx].reshape(-1).cpu().numpy(), want, n) < 1e-10


@pytest.mark.parametrize("n,batch", [(64, 100_000), (128, 25_000)])
def test_full_size_properties_general(n, batch):
    """The same at full size on GENERAL input (A ~ U(0,1)^(n x n), like the reference's tests/square_5_*.mats: every matrix
    needs row exchanges): residual, info == 0, homogeneity, 64 oracle spot checks, and the launches the adaptive dispatch
    makes one after the other (natural-order attempt + work list, then the pivoting kernel directly) bit-identical."""
    g = torch.Generator(device="cuda").manual_seed(4321 + n)
    a = torch.rand((batch, n, n), generator=g, dtype=torch.float64, device="cuda")
    flat = a.reshape(-1)
    info = torch.full((batch,), -7, dtype=torch.int32, device="cuda")
    runs = []
    for _ in range(3):  # the default policy (natural order first, rejects to the pivoting kernel) launch after launch
        runs.append(api.inverse_batched(flat, n, GJ, info=info).clone())
        torch.cuda.synchronize()
        assert int((info != 0).sum()) == 0
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2]), "launch history changed the result bits"
    x = runs[0]
    xm = x.view(batch, n, n)
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    # cond(U(0,1)^(n x n)) reaches 1e5 .. 1e6 somewhere in 100 k draws: scale the bound per matrix with |X|
    res = (torch.bmm(a, xm) - eye).abs().amax(dim=(1, 2))
    scale = xm.abs().amax(dim=(1, 2)) * a.abs().amax(dim=(1, 2)) * n
    assert float((res / scale).max()) < 1e-13, "residual |A X - I| beyond n * |A| * |X| * 1e-13"
    quarter = api.inverse_batched(flat * 4.0, n, GJ)
    assert torch.equal(quarter * 4.0, x), "inv(4A) must equal inv(A)/4 bit for bit"
    idx = torch.arange(0, batch, batch // 64, device="cuda")[:64]
    sub = a[idx].reshape(-1).cpu().numpy()
    want, winfo = oracle.inverse_batched(sub, n, oracle.ALGO_GJ_PIVOT)
    assert not winfo.any()
    cond = max(np.linalg.cond(m) for m in as_mats(sub, n))
    assert rel_err(xm[idx].reshape(-1).cpu().numpy(), want, n) < max(1e-10, 1e-15 * cond * n)


@pytest.mark.parametrize("n,batch", [(8, 1003), (16, 1001), (32, 517), (64, 300), (100, 67), (128, 41), (200, 9)])
def test_sharded_results_bit_identical_to_unsharded(n, batch):
    """SURVEY 8(e) determinism: what rank g of G computes for its block of the batch is bit-identical to the same
    matrices inverted in one launch (no atomics, no dependence on the position inside a wavefront or workgroup), for the
    inversion (both algorithms) and the fused pipeline, G = 1 vs 2, 3, 8 with ragged tails."""
    shard = pkg("shard")
    a = dev(spd_batch(n, batch, seed=900 + n))
    rng = np.random.default_rng(n)
    va, vc, vd = (dev(rng.random(batch * n)) for _ in range(3))
    full = {algo: api.inverse_batched(a, n, algo, batch=batch).clone() for algo in (GJ, CH)}
    full_m = api.calcluateMean(n, va, a, vc, vd).clone()
    for world in (2, 3, 8):
        parts = shard.partition(batch, world, shard.packing_multiple(n))
        assert parts[0][0] == 0 and max(hi for _, hi in parts) == batch
        for algo in (GJ, CH):
            got = torch.cat([api.inverse_batched(a[lo * n * n:hi * n * n], n, algo, batch=hi - lo)
                             for lo, hi in parts if hi > lo])
            assert torch.equal(got, full[algo]), (n, world, algo)
        got_m = torch.cat([api.calcluateMean(n, va[lo * n:hi * n], a[lo * n * n:hi * n * n], vc[lo * n:hi * n],
                                             vd[lo * n:hi * n], batchSize=hi - lo) for lo, hi in parts if hi > lo])
        assert torch.equal(got_m, full_m), (n, world)


def test_batch_beyond_32bit_element_offsets():
    """BASELINE configs[3] (1 M x 64x64 fp64) on one GPU, slightly enlarged so that ELEMENT offsets exceed 2^32 (the
    reference indexes with int, src/gauss/batched_invert.cu:130, and overflows at 2^31). 36 GB per operand."""
    n, batch, chunk = 64, 1_100_000, 50_000
    assert batch * n * n > 2 ** 32
    free, _ = torch.cuda.mem_get_info()
    if free < 2.2 * batch * n * n * 8:
        pytest.skip("not enough free device memory")
    a = torch.empty(batch * n * n, dtype=torch.float64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for i in range(0, batch, chunk):
        r = torch.rand((chunk, n, n), generator=g, dtype=torch.float64, device="cuda")
        r = r + r.transpose(1, 2)
        r.diagonal(dim1=1, dim2=2).add_(float(n))
        a[i * n * n:(i + chunk) * n * n] = r.reshape(-1)
    info = torch.empty(batch, dtype=torch.int32, device="cuda")
    x = api.inverse_batched(a, n, GJ, info=info)
    assert int((info != 0).sum()) == 0
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    for lo in (0, batch // 2, (2 ** 32) // (n * n) - 500, batch - 1000):
        am, xm = a.view(batch, n, n)[lo:lo + 1000], x.view(batch, n, n)[lo:lo + 1000]
        assert float((torch.bmm(am, xm) - eye).abs().max()) < 1e-12
    idx = torch.tensor([0, batch - 1], device="cuda")
    want, _ = oracle.inverse_batched(a.view(batch, n * n)[idx].reshape(-1).cpu().numpy(), n)
    assert rel_err(x.view(batch, n * n)[idx].reshape(-1).cpu().numpy(), want, n) < 1e-10
    del a, x
    torch.cuda.empty_cache()
