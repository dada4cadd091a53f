"""Pin the CPU oracle (oracle/) before trusting it: against every golden the reference ships
for this path, against the reference's own scalar Cholesky compiled as it lies
(oracle/_ref), and against numpy fp64 as an independent cross-check."""
import os

import numpy as np
import pytest

import oracle
from conftest import as_mats, general_batch, read_ref, rel_err, spd_batch

ALGOS = [oracle.ALGO_GJ_PIVOT, oracle.ALGO_GJ_REFERENCE, oracle.ALGO_CHOLESKY, oracle.ALGO_LU]


@pytest.mark.parametrize("d,n", [("inverse_100_8x8", 8), ("inverse_100_16x16", 16), ("inverse_32_32x32", 32)])
@pytest.mark.parametrize("algo", ALGOS)
def test_reference_inverse_goldens(d, n, algo):
    """aInv.mats is MATLAB inv() rounded to 4 significant digits (generate_inverse_matrices.m:20-21):
    the reference's own error metric, sum|computed - aInv| / batch (inverse_bench.c:49-51,58), sits at the
    rounding floor ~3.4e-4 (8x8) for a correct inverse."""
    a, k, _, _ = read_ref(f"{d}/a.mats")
    g, _, _, _ = read_ref(f"{d}/aInv.mats")
    inv, info = oracle.inverse_batched(a, n, algo)
    assert not info.any()
    err = np.abs(inv - g).sum() / k
    assert err < 5e-4, err
    # element-wise: a.mats is rounded to 4 digits too (diagonal ~9.xxx -> +-5e-4), which moves
    # inv(A) by ~ |inv| * 5e-4 * |inv| ~ 1e-5 on top of the golden's own 4-digit rounding
    assert np.abs(inv - g).max() < 1e-4  # 0.1xxx rounded to 4 digits = +-5e-5, plus the input rounding


@pytest.mark.parametrize("d,n", [("gaussian_100_8x8", 8), ("gaussian_100_16x16", 16),
                                 ("gaussian_32_32x32", 32), ("gaussian_12_64x64", 64)])
def test_reference_pipeline_goldens(d, n):
    """means.mats / variances.mats (generate_gaussian_matrices.m:30-37), 4 significant digits."""
    r = {f: read_ref(f"{d}/{f}.mats")[0] for f in ("a", "b", "c", "d", "e", "means", "variances")}
    m = oracle.mean_batched(r["a"], r["b"], r["c"], r["d"], n)
    v = oracle.variance_batched(r["a"], r["b"], r["c"], r["e"], n)
    assert np.abs(m - r["means"]).mean() < 5e-5
    assert np.abs(v - r["variances"]).mean() < 5e-5
    # the reference CPU code adds instead of subtracting (gauss_cpu.c:198): reproduce and show the gap
    v_ref = oracle.variance_batched(r["a"], r["b"], r["c"], r["e"], n, ref_sign=True)
    assert np.abs(v_ref - r["variances"]).mean() > 1e-2


def test_reference_simplemean_cholesky():
    """tests/simpleMean/chol.mats -> cholinv.mats (6 decimals; cond ~1185); Makefile:229-235 feeds the same matrix."""
    a, _, _, n = read_ref("simpleMean/chol.mats")
    g, _, _, _ = read_ref("simpleMean/cholinv.mats")
    for algo in ALGOS:
        inv, info = oracle.inverse_batched(a, n, algo)
        assert not info.any()
        assert np.abs(inv - g).max() < 1e-5  # the golden itself carries fp32 round-off (2.515631 vs exact 2.515625)


def test_against_reference_binary_4x4():
    """oracle/_ref/inverse_cholesky_cpu is /root/reference/src/inverse_cholesky_cpu.c compiled unmodified
    (fp32, N=4). Its output must match the fp32 restatement to fp32 round-off and the fp64 one to ~cond*eps32."""
    rng = np.random.default_rng(7)
    mats4 = [np.array([[18, 22, 54, 42], [22, 70, 86, 62], [54, 86, 174, 134], [42, 62, 134, 106]], dtype=np.float64)]
    for _ in range(5):
        r = rng.random((4, 4))
        mats4.append(r + r.T + 4 * np.eye(4))
    ran = 0
    for a in mats4:
        ref = oracle.ref_cholesky_4x4(a)
        if ref is None:
            pytest.skip("oracle/_ref not built (reference mount absent and no prebuilt binary)")
        ran += 1
        flat = np.ascontiguousarray(a.T).reshape(-1)
        inv32, _ = oracle.inverse_batched(flat.astype(np.float32), 4, oracle.ALGO_CHOLESKY)
        inv64, _ = oracle.inverse_batched(flat, 4, oracle.ALGO_CHOLESKY)
        cond = np.linalg.cond(a)
        scale = np.abs(inv64).max()
        # printed with %f (6 decimals) -> 5e-7 print rounding on top of fp32 round-off
        assert np.abs(as_mats(inv32, 4)[0] - ref).max() <= 4 * cond * 6e-8 * scale + 1e-6
        assert np.abs(as_mats(inv64, 4)[0] - ref).max() <= 4 * cond * 6e-8 * scale + 1e-6
    assert ran == 6


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 16, 31, 32, 64, 100, 128])
def test_algorithms_agree_spd_fp64(n):
    a = spd_batch(n, 6, seed=n)
    want = np.linalg.inv(as_mats(a, n)).transpose(0, 2, 1).reshape(-1)
    for algo in ALGOS:
        inv, info = oracle.inverse_batched(a, n, algo)
        assert not info.any()
        assert rel_err(inv, want, n) < 1e-12, (algo, n)


@pytest.mark.parametrize("n", [2, 3, 8, 16, 32, 64, 128])
def test_pivoting_general_fp64(n):
    """Non-symmetric U(0,1) matrices need row pivoting; GJ-pivot and LU must agree with numpy."""
    a = general_batch(n, 4, seed=100 + n)
    A = as_mats(a, n)
    want = np.linalg.inv(A).transpose(0, 2, 1).reshape(-1)
    cond = max(np.linalg.cond(m) for m in A)
    for algo in (oracle.ALGO_GJ_PIVOT, oracle.ALGO_LU):
        inv, info = oracle.inverse_batched(a, n, algo)
        assert not info.any()
        assert rel_err(inv, want, n) < 1e-14 * cond * n + 1e-13, (algo, n, cond)


def test_square_fixtures_and_gold(gold):
    for f, n in (("square_5_8_8", 8), ("square_5_16_16", 16), ("square_5_32_32", 32),
                 ("square_3_64_64", 64), ("square_1_128_128", 128)):
        a, k, _, _ = read_ref(f + ".mats")
        want = np.linalg.inv(as_mats(a, n)).transpose(0, 2, 1).reshape(-1)
        assert rel_err(gold[f + "/gj"], want, n) < 1e-10


def test_batch3_zero_pivot():
    """src/gauss/batch_3.txt matrix 3 (3 0 2 / 2 0 -2 / 0 1 1) hits an exact zero pivot at step 2:
    the only shipped input that takes pivotRow's swap branch (batched_invert.cu:19-35)."""
    a, k, _, n = read_ref("batch_3.mats")
    assert (k, n) == (6, 3)
    want = np.linalg.inv(as_mats(a, n)).transpose(0, 2, 1).reshape(-1)
    for algo in (oracle.ALGO_GJ_PIVOT, oracle.ALGO_GJ_REFERENCE, oracle.ALGO_LU):
        inv, info = oracle.inverse_batched(a, n, algo)
        assert not info.any()
        assert rel_err(inv, want, n) < 1e-13


def test_singular_and_not_spd_info():
    sing = np.array([1.0, 2, 3, 2, 4, 6, 1, 0, 1])  # col1 = 2*col0 (column-major 3x3)
    for algo in (oracle.ALGO_GJ_PIVOT, oracle.ALGO_GJ_REFERENCE, oracle.ALGO_LU):
        _, info = oracle.inverse_batched(sing, 3, algo)
        assert info[0] != 0
    notspd = np.array([1.0, 2, 2, 1])
    _, info = oracle.inverse_batched(notspd, 2, oracle.ALGO_CHOLESKY)
    assert info[0] == 2


def test_fp32_vs_fp64():
    for n in (8, 16, 64):
        a = spd_batch(n, 4, seed=n)
        i64, _ = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
        for algo in ALGOS:
            i32, info = oracle.inverse_batched(a.astype(np.float32), n, algo)
            assert not info.any()
            x = i32.astype(np.float64).reshape(-1, n * n)
            y = i64.reshape(-1, n * n)
            fro = np.linalg.norm(x - y, axis=1) / np.linalg.norm(y, axis=1)
            assert fro.max() < 1e-5, (n, algo, fro.max())


def test_empty_batch():
    inv, info = oracle.inverse_batched(np.zeros(0), 8)
    assert inv.size == 0 and info.size == 0


def test_committed_gold_matches_live_oracle(gold):
    a, _, _, n = read_ref("inverse_100_16x16/a.mats")
    inv, _ = oracle.inverse_batched(a, n, oracle.ALGO_GJ_PIVOT)
    assert rel_err(inv, gold["inverse_100_16x16/gj"], n) < 1e-14
