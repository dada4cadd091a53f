"""The reference's two command lines, end to end on an MI355X: same argv, same output columns, errors at the
fixtures' rounding floor (SURVEY.md 8b 'CLI contract')."""
import os
import subprocess

import pytest

from conftest import REFDATA, ROOT

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "cuda-matrix-inversion_amd", "host")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", os.path.join(ROOT, "cuda-matrix-inversion_amd"), "-s", "-j4"], check=True)
    subprocess.run(["make", "-C", HOST, "-s"], check=True)
    subprocess.run(["make", "-C", HOST, "-s", "DTYPE=float"], check=True)


def run(exe, *args, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="8")
    e.update(env or {})
    p = subprocess.run([os.path.join(HOST, exe), *args], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode == 0, p.stderr
    return p.stdout.strip().splitlines()


@pytest.mark.parametrize("exe,tol", [("inverse_bench", 5e-4), ("inverse_bench_f32", 6e-4)])
@pytest.mark.parametrize("d,n", [("inverse_100_8x8", 8), ("inverse_100_16x16", 16), ("inverse_32_32x32", 32)])
def test_inverse_bench(exe, tol, d, n):
    lines = run(exe, os.path.join(REFDATA, d), "3", "4", "-csv")
    names = [ln.split()[3] for ln in lines]
    assert names == ["lu_blas_cpu", "lu_blas_omp_cpu", "chol_gpu", "chol_mm2_gpu", "gauss_batched_gpu", "lu_cuda_batched_gpu"]
    for ln in lines:
        f = ln.split()
        assert int(f[1]) == n and int(f[2]) == 3 and len(f) == 8
        assert float(f[4]) > 0
        assert float(f[7]) < tol, ln  # sum|inv - aInv.mats| per matrix: the 4-digit rounding floor (3.4e-4 at 8x8)


@pytest.mark.parametrize("exe", ["gauss_bench", "gauss_bench_f32"])
@pytest.mark.parametrize("d", ["gaussian_100_8x8", "gaussian_100_16x16", "gaussian_32_32x32", "gaussian_12_64x64"])
def test_gauss_bench(exe, d):
    lines = run(exe, os.path.join(REFDATA, d), "2", "3", "-csv")
    names = [ln.split()[3] for ln in lines]
    assert names == ["means_cpu", "variances_cpu", "means_gpu", "variances_gpu"]
    for ln in lines:
        assert float(ln.split()[7]) < 6e-5, ln


def test_detailed_logging_lines():
    """log=1 build of the reference: name,batch,n,ms,ns CRLF lines with the reference's key names (timer.h:8-9)."""
    p = subprocess.run([os.path.join(HOST, "inverse_bench"), os.path.join(REFDATA, "inverse_100_8x8"), "1", "1"],
                       capture_output=True, text=True, env=dict(os.environ, MATINV_DETAILED_LOGGING="1"))
    assert p.returncode == 0, p.stderr
    keys = [ln.split(",")[0] for ln in p.stdout.splitlines() if "," in ln]
    for k in ("lu_blas_cpu", "lu_blas_omp_cpu", "decompose_cholesky_batched_gpu_mem_htod", "decompose_cholesky_batched_gpu_ker",
              "cholesky_mm2_batched_gpu_ker", "inverse_gauss_batched_gpu_mem_htod", "inverse_gauss_batched_gpu_ker",
              "inverse_gauss_batched_gpu_mem_dtoh", "inverse_lu_cuda_batched_gpu_ker", "gauss_batched_gpu"):
        assert k in keys, (k, keys)
    f = [ln for ln in p.stdout.splitlines() if ln.startswith("inverse_gauss_batched_gpu_ker,")][0].strip().split(",")
    assert f[1] == "100" and f[2] == "8" and float(f[3]) >= 0 and int(f[4]) >= 0


def test_pipeline_logging_keys_are_the_references():
    """src/gauss_bench.cu:151-156,250-255,298-303,394-399: the six phase keys results/generate_plots.m:69-76 sums."""
    p = subprocess.run([os.path.join(HOST, "gauss_bench"), os.path.join(REFDATA, "gaussian_100_8x8"), "1", "1"],
                       capture_output=True, text=True, env=dict(os.environ, MATINV_DETAILED_LOGGING="1"))
    assert p.returncode == 0, p.stderr
    keys = [ln.split(",")[0] for ln in p.stdout.splitlines() if "," in ln]
    for what in ("mean", "variance"):
        for phase in ("mem_htod", "add", "inv", "mul", "dot", "mem_dtoh"):
            assert f"calculate_{what}_gpu_{phase}" in keys, (what, phase, keys)
    for k in ("means_cpu", "variances_cpu", "means_gpu", "variances_gpu"):
        assert k in keys, (k, keys)


@pytest.mark.parametrize("exe", ["device_table_test", "device_table_test_f32"])
@pytest.mark.parametrize("n,batch", [(8, 100), (32, 257), (64, 1000), (100, 50), (150, 9)])
def test_device_table_call_sequence_in_c(exe, n, batch):
    """batchedCudaMalloc + cudaMemcpy2D + inverse_lu_cuda_batched_device exactly as src/gauss_bench.cu:68-78,160-170 does,
    in plain C, on general matrices, against the host LU."""
    lines = run(exe, str(n), str(batch))
    assert lines[-1].startswith(f"device_table_test n={n} batch={batch} pitch="), lines
    pitch = int(lines[-1].split("pitch=")[1].split()[0])
    esz = 4 if exe.endswith("f32") else 8
    assert pitch >= n * n * esz and pitch % 256 == 0


@pytest.mark.parametrize("exe", ["queue_test", "queue_test_f32"])
def test_size_binned_queue_from_c(exe):
    """matinv_queue_create / submit / flush driven from plain C (README.md:41-44 of the reference; BASELINE configs[4]),
    checked against the host pipeline (host/gauss_cpu.c)."""
    lines = run(exe, "12345")
    assert lines[-1].startswith("queue_test items=120 "), lines


@pytest.mark.parametrize("n,k", [(64, 40), (128, 12)])
def test_sweep_sizes_missing_from_the_reference_tree(tmp_path, n, k):
    """The reference's `make run-inverse-bench` / `run-gauss-bench` sweeps go up to 128x128, but its 64 / 128 fixtures are
    absent (.MISSING_LARGE_BLOBS): generate them (tools/generate_fixtures.py) and run both CLIs, both DataTypes."""
    import sys
    gen = os.path.join(ROOT, "tools", "generate_fixtures.py")
    inv_dir, gp_dir = str(tmp_path / f"inverse_{k}_{n}x{n}"), str(tmp_path / f"gaussian_{k}_{n}x{n}")
    subprocess.run([sys.executable, gen, "inverse", inv_dir, str(k), str(n)], check=True)
    subprocess.run([sys.executable, gen, "gaussian", gp_dir, str(k), str(n)], check=True)
    for exe, tol in (("inverse_bench", 1e-12), ("inverse_bench_f32", 2e-4)):
        lines = run(exe, inv_dir, "2", "2", "-csv")
        assert [ln.split()[3] for ln in lines] == ["lu_blas_cpu", "lu_blas_omp_cpu", "chol_gpu", "chol_mm2_gpu",
                                                    "gauss_batched_gpu", "lu_cuda_batched_gpu"]
        for ln in lines:
            assert ln.split()[:3] == [str(2 * k), str(n), "2"] and float(ln.split()[7]) < tol, ln
    for exe, tol in (("gauss_bench", 1e-12), ("gauss_bench_f32", 2e-5)):
        lines = run(exe, gp_dir, "2", "2", "-csv")
        assert [ln.split()[3] for ln in lines] == ["means_cpu", "variances_cpu", "means_gpu", "variances_gpu"]
        for ln in lines:
            assert float(ln.split()[7]) < tol, ln


@pytest.mark.parametrize("exe", ["multi_test", "multi_test_f32"])
@pytest.mark.parametrize("n,batch,shards", [(8, 1003, 3), (16, 1001, 2), (24, 500, 5), (64, 300, 4), (100, 67, 3), (150, 20, 2),
                                            (64, 9000, 3)])
def test_multi_device_host_path_in_c(exe, n, batch, shards):
    """matinv_inverse_batched_host_multi from plain C: `shards` shards (virtual ones on a one-GPU box: round robin over the
    devices there are), one host thread + hipSetDevice per shard, bit-identical to the single-device call on a mix of
    dominant, mildly non-dominant and general matrices (SURVEY 8e determinism); 64 x 9000 f64 takes the page-locked,
    pipelined three-stream path per shard."""
    lines = run(exe, str(n), str(batch), str(shards))
    assert lines[-1].startswith(f"multi_test n={n} batch={batch} shards={shards} ") and lines[-1].endswith("identical"), lines


def test_reference_names_honour_matinv_devices():
    """MATINV_DEVICES=3: inverse_gauss_batched_gpu itself shards over three (virtual) devices -- same bits (multi_test compares
    it with matinv_inverse_batched_host); and inverse_bench runs unchanged under MATINV_DEVICES=1 and =2."""
    lines = run("multi_test", "64", "500", "2", env={"MATINV_DEVICES": "3"})
    assert lines[-1].endswith("identical"), lines
    for nd in ("1", "2"):
        lines = run("inverse_bench", os.path.join(REFDATA, "inverse_100_16x16"), "2", "4", "-csv", env={"MATINV_DEVICES": nd})
        assert len(lines) == 6 and all(float(ln.split()[7]) < 5e-4 for ln in lines), lines
