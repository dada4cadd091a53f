/*
 * inverse_bench.c -- `inverse_bench DIR REPS DUPS [-csv]`: raw batched-inversion benchmark.
 *
 * Command line, input files (DIR/a.mats, DIR/aInv.mats), algorithm names, output columns and error metric follow
 * /root/reference/src/inverse_bench.c:76-303: two CPU algorithms (lu_blas_cpu, lu_blas_omp_cpu) and four GPU entry
 * points (chol_gpu, chol_mm2_gpu, gauss_batched_gpu, lu_cuda_batched_gpu), each timed REPS times around the whole
 * call (allocation + H2D + kernel + D2H, as the reference does, report.tex:104); the error of the LAST repetition
 * against aInv.mats is reported as sum|diff| / numMatrices (:49-51,58). With MATINV_DETAILED_LOGGING=1 the per-call
 * `name,batch,n,ms,ns` lines of the reference's log=1 build are printed instead of the summary.
 * The GPU side is libmatinv_hip.so through include/inverse_gpu.h -- plain C, no CUDA.
 */
#include "bench_common.h"

#include "../../include/inverse_cpu.h"
#include "../../include/inverse_gpu.h"

typedef void (*gpu_entry)(cublasHandle_t, int, Array, Array, int);

static void read_test(const char *dir, int *numMatrices, int *n, Array *a, Array *aInv)
{
    char path[1024];
    int kA, mA, nA, kI, mI, nI;
    snprintf(path, sizeof path, "%s/a.mats", dir);
    readMatricesFile(path, &kA, &mA, &nA, a);
    snprintf(path, sizeof path, "%s/aInv.mats", dir);
    readMatricesFile(path, &kI, &mI, &nI, aInv);
    ensure(kA == kI, "test in directory %s invalid, number of matrices in files not matching: a(%d) aInv(%d)", dir, kA, kI);
    ensure(mA == nA && mI == nI && mA == mI, "test in directory %s invalid, dimensions not matching: a(%dx%d) aInv(%dx%d)",
           dir, mA, nA, mI, nI);
    *numMatrices = kA;
    *n = mA;
}

int main(int argc, char const *argv[])
{
    ensure(argc >= 4, "Usage: inverse_bench TEST_FOLDER TEST_REPLICATIONS MATRIX_DUPLICATES [-csv]");
    const bool csv = (argc >= 5) && !strncmp("-csv", argv[4], 4);
    const int numReps = atoi(argv[2]), numDups = atoi(argv[3]);
    ensure(numReps >= 1 && numDups >= 1, "TEST_REPLICATIONS and MATRIX_DUPLICATES must be >= 1");
    const bool log = detailed_logging();

    int numMatrices, N;
    Array a, aInv;
    read_test(argv[1], &numMatrices, &N, &a, &aInv);
    replicateMatrices(&a, N, N, numMatrices, numDups);
    replicateMatrices(&aInv, N, N, numMatrices, numDups);
    numMatrices *= numDups;

    const size_t count = (size_t)numMatrices * N * N;
    Array inv = (Array)malloc(count * sizeof(DataType));
    Array workspace = (Array)malloc(count * sizeof(DataType));
    ensure(inv && workspace, "Could not allocate result buffers");

    /* ---- CPU, one thread: the reference loops inverse_lu_blas over the batch (:95-109) ---- */
    {
        TIMER_INIT(lu_blas_cpu) TIMER_ACC_INIT(lu_blas_cpu)
        for (int rep = 0; rep < numReps; ++rep) {
            memcpy(inv, a, count * sizeof(DataType));
            TIMER_START(lu_blas_cpu)
            for (int i = 0; i < numMatrices; ++i) inverse_lu_blas(inv + (size_t)i * N * N, workspace, N);
            TIMER_STOP(lu_blas_cpu)
            if (log) { TIMER_LOG(lu_blas_cpu, numMatrices, N) }
            TIMER_ACC(lu_blas_cpu)
        }
        if (!log)
            report_line(csv, "lu_blas_cpu", numMatrices, N, numReps, TIMER_TOTAL(lu_blas_cpu), TIMER_MEAN(lu_blas_cpu),
                        TIMER_VARIANCE(lu_blas_cpu), abs_diff_sum(inv, aInv, count) / numMatrices);
    }
    /* ---- CPU, OpenMP (:117-127) ---- */
    {
        TIMER_INIT(lu_blas_omp_cpu) TIMER_ACC_INIT(lu_blas_omp_cpu)
        for (int rep = 0; rep < numReps; ++rep) {
            memcpy(inv, a, count * sizeof(DataType));
            TIMER_START(lu_blas_omp_cpu)
            inverse_lu_blas_omp(inv, N, numMatrices);
            TIMER_STOP(lu_blas_omp_cpu)
            if (log) { TIMER_LOG(lu_blas_omp_cpu, numMatrices, N) }
            TIMER_ACC(lu_blas_omp_cpu)
        }
        if (!log)
            report_line(csv, "lu_blas_omp_cpu", numMatrices, N, numReps, TIMER_TOTAL(lu_blas_omp_cpu),
                        TIMER_MEAN(lu_blas_omp_cpu), TIMER_VARIANCE(lu_blas_omp_cpu),
                        abs_diff_sum(inv, aInv, count) / numMatrices);
    }
    /* ---- GPU entry points, in the reference's order (:140-223). The input is never clobbered here, so every
     *      algorithm sees the original `a` (the reference hands its master copy to the Cholesky paths, :144,168). ---- */
    static const struct { const char *name; gpu_entry fn; } gpu[] = {
        {"chol_gpu", inverse_cholesky_batched_gpu},
        {"chol_mm2_gpu", inverse_cholesky_mm2_batched_gpu},
        {"gauss_batched_gpu", inverse_gauss_batched_gpu},
        {"lu_cuda_batched_gpu", inverse_lu_cuda_batched_gpu},
    };
    cublasHandle_t handle = NULL; /* was cublasCreate(&handle), :132 */
    const char *skip = getenv("MATINV_SKIP_GPU"); /* CPU lines only, for a host without an MI355X (like gauss_bench) */
    const bool skip_gpu = skip && *skip && *skip != '0';
    for (size_t g = 0; !skip_gpu && g < sizeof gpu / sizeof gpu[0]; ++g) {
        TIMER_INIT(gpu_call) TIMER_ACC_INIT(gpu_call)
        for (int rep = 0; rep < numReps; ++rep) {
            memcpy(workspace, a, count * sizeof(DataType));
            TIMER_START(gpu_call)
            gpu[g].fn(handle, N, workspace, inv, numMatrices);
            TIMER_STOP(gpu_call)
            if (log) printf("%s,%d,%d,%.4f,%lu\r\n", gpu[g].name, numMatrices, N, TIMER_ELAPSED(gpu_call), TIMER_ELAPSED_NS(gpu_call));
            TIMER_ACC(gpu_call)
        }
        if (!log)
            report_line(csv, gpu[g].name, numMatrices, N, numReps, TIMER_TOTAL(gpu_call), TIMER_MEAN(gpu_call),
                        TIMER_VARIANCE(gpu_call), abs_diff_sum(inv, aInv, count) / numMatrices);
    }

    free(inv);
    free(workspace);
    free(a);
    free(aInv);
    return 0;
}
