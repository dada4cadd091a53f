/*
 * gauss_bench.c -- `gauss_bench DIR REPS DUPS [-csv]`: Gaussian-process mean / variance benchmark.
 *
 * Command line, the seven input files (a, b, c, d, e, means, variances .mats), the shape checks, the CPU/GPU order
 * and the output columns follow /root/reference/src/gauss_bench.cu:411-467 (readTest) and :577-702 (main):
 *   means_cpu / variances_cpu : calcluateMeanCPU / calcluateVarianceCPU (host/gauss_cpu.c), OpenMP
 *   means_gpu / variances_gpu : ONE fused kernel each in libmatinv_hip.so (matinv_mean_batched_host /
 *                               matinv_variance_batched_host) instead of addDiagonal + batched LU + 2 gemmBatched
 *                               (:127-265, :275-409) -- (B+diag c)^-1 is never formed.
 * Errors are sum|out - golden| / numMatrices / numReps as in BENCH_ERROR_* (:493-499). The reference prints
 * TIMER_VARIANCE(means_*) on the variances_* line (copy-paste bug, :510); here each line carries its own variance.
 * BASELINE.json configs[0] is this program on tests/gaussian_100_8x8 with OMP_NUM_THREADS=8 (CPU lines; the GPU lines
 * need an MI355X -- pass MATINV_SKIP_GPU=1 to print the CPU lines only on a host without one).
 */
#include "bench_common.h"

#include "../../include/gauss_cpu.h"
#include "../../include/matinv.h"

#ifdef MATINV_DATATYPE_FLOAT
#define BENCH_DTYPE MATINV_F32
#else
#define BENCH_DTYPE MATINV_F64
#endif

static void read_one(const char *dir, const char *file, int *k, int *m, int *n, Array *out)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.mats", dir, file);
    readMatricesFile(path, k, m, n, out);
}

static void read_test(const char *dir, int *numMatrices, int *n, Array *a, Array *b, Array *c, Array *d, Array *e,
                      Array *means, Array *variances)
{
    int k[7], m[7], nn[7];
    read_one(dir, "a", &k[0], &m[0], &nn[0], a);
    read_one(dir, "b", &k[1], &m[1], &nn[1], b);
    read_one(dir, "c", &k[2], &m[2], &nn[2], c);
    read_one(dir, "d", &k[3], &m[3], &nn[3], d);
    read_one(dir, "e", &k[4], &m[4], &nn[4], e);
    read_one(dir, "means", &k[5], &m[5], &nn[5], means);
    read_one(dir, "variances", &k[6], &m[6], &nn[6], variances);
    for (int i = 1; i < 7; ++i)
        ensure(k[i] == k[0], "test in directory %s invalid, number of matrices in files not matching (file %d: %d vs %d)",
               dir, i, k[i], k[0]);
    ensure(m[0] == m[1] && m[1] == m[2] && m[2] == m[3] && m[4] == 1 && m[5] == 1 && m[6] == 1 && nn[0] == 1 &&
               nn[1] == m[1] && nn[2] == 1 && nn[3] == 1 && nn[4] == 1 && nn[5] == 1 && nn[6] == 1,
           "test in directory %s invalid, dimensions not matching\r\nmA(%d) mB(%d) mC(%d) mD(%d) mE(%d) mMeans(%d) "
           "mVariances(%d)\r\nnA(%d) nB(%d) nC(%d) nD(%d) nE(%d) nMeans(%d) nVariances(%d)",
           dir, m[0], m[1], m[2], m[3], m[4], m[5], m[6], nn[0], nn[1], nn[2], nn[3], nn[4], nn[5], nn[6]);
    *numMatrices = k[0];
    *n = m[0];
}

int main(int argc, char const *argv[])
{
    ensure(argc >= 4, "Usage: gauss_bench TEST_FOLDER TEST_REPLICATIONS MATRIX_DUPLICATES [-csv]");
    const bool csv = (argc >= 5) && !strncmp("-csv", argv[4], 4);
    const int numReps = atoi(argv[2]), numDups = atoi(argv[3]);
    ensure(numReps >= 1 && numDups >= 1, "TEST_REPLICATIONS and MATRIX_DUPLICATES must be >= 1");
    const bool log = detailed_logging();
    const char *skip = getenv("MATINV_SKIP_GPU");
    const bool skip_gpu = skip && *skip && *skip != '0';

    int numMatrices, n;
    Array _a, _b, _c, _d, _e, _means, _variances;
    read_test(argv[1], &numMatrices, &n, &_a, &_b, &_c, &_d, &_e, &_means, &_variances);
    replicateMatrices(&_a, n, 1, numMatrices, numDups);
    replicateMatrices(&_b, n, n, numMatrices, numDups);
    replicateMatrices(&_c, n, 1, numMatrices, numDups);
    replicateMatrices(&_d, n, 1, numMatrices, numDups);
    replicateMatrices(&_e, 1, 1, numMatrices, numDups);
    replicateMatrices(&_means, 1, 1, numMatrices, numDups);
    replicateMatrices(&_variances, 1, 1, numMatrices, numDups);
    numMatrices *= numDups;

    const size_t vec = (size_t)numMatrices * n, mat = vec * n, one = (size_t)numMatrices;
    Array a = (Array)malloc(vec * sizeof(DataType)), b = (Array)malloc(mat * sizeof(DataType));
    Array c = (Array)malloc(vec * sizeof(DataType)), d = (Array)malloc(vec * sizeof(DataType));
    Array e = (Array)malloc(one * sizeof(DataType));
    Array means_out = (Array)malloc(one * sizeof(DataType)), variances_out = (Array)malloc(one * sizeof(DataType));
    ensure(a && b && c && d && e && means_out && variances_out, "Could not allocate memory for the working copies");

#define RESTORE_INPUTS()                                  \
    memcpy(a, _a, vec * sizeof(DataType));                \
    memcpy(b, _b, mat * sizeof(DataType));                \
    memcpy(c, _c, vec * sizeof(DataType));                \
    memcpy(d, _d, vec * sizeof(DataType));                \
    memcpy(e, _e, one * sizeof(DataType));

    for (int side = 0; side < 2; ++side) { /* 0 = cpu, 1 = gpu */
        if (side == 1 && skip_gpu) break;
        const char *mname = side ? "means_gpu" : "means_cpu", *vname = side ? "variances_gpu" : "variances_cpu";
        double err_m = 0, err_v = 0;
        TIMER_INIT(means) TIMER_ACC_INIT(means) TIMER_INIT(variances) TIMER_ACC_INIT(variances)
        for (int rep = 0; rep < numReps; ++rep) {
            RESTORE_INPUTS()
            TIMER_START(means)
            if (side == 0) calcluateMeanCPU(n, a, b, c, d, means_out, numMatrices);
            else ensure(matinv_mean_batched_host(BENCH_DTYPE, n, a, b, c, d, means_out, (size_t)numMatrices, NULL) == MATINV_OK,
                        "calcluateMean on the GPU failed: %s", matinv_last_error());
            TIMER_STOP(means)
            TIMER_ACC(means)
            if (log) printf("%s,%d,%d,%.4f,%lu\r\n", mname, numMatrices, n, TIMER_ELAPSED(means), TIMER_ELAPSED_NS(means));
            err_m += abs_diff_sum(means_out, _means, one);

            RESTORE_INPUTS()
            TIMER_START(variances)
            if (side == 0) calcluateVarianceCPU(n, a, b, c, e, variances_out, numMatrices);
            else ensure(matinv_variance_batched_host(BENCH_DTYPE, n, a, b, c, e, variances_out, (size_t)numMatrices, NULL) == MATINV_OK,
                        "calcluateVariance on the GPU failed: %s", matinv_last_error());
            TIMER_STOP(variances)
            TIMER_ACC(variances)
            if (log) printf("%s,%d,%d,%.4f,%lu\r\n", vname, numMatrices, n, TIMER_ELAPSED(variances), TIMER_ELAPSED_NS(variances));
            err_v += abs_diff_sum(variances_out, _variances, one);
        }
        if (!log) {
            report_line(csv, mname, numMatrices, n, numReps, TIMER_TOTAL(means), TIMER_MEAN(means), TIMER_VARIANCE(means),
                        err_m / numMatrices / numReps);
            report_line(csv, vname, numMatrices, n, numReps, TIMER_TOTAL(variances), TIMER_MEAN(variances),
                        TIMER_VARIANCE(variances), err_v / numMatrices / numReps);
        }
    }

    free(a); free(b); free(c); free(d); free(e); free(means_out); free(variances_out);
    free(_a); free(_b); free(_c); free(_d); free(_e); free(_means); free(_variances);
    return 0;
}
