/*
 * multi_test.c -- the multi-device host path of the C ABI from plain C ("batches shard trivially over the 8
 * GPUs of one node ... one host thread per GPU, hipSetDevice per shard"; the reference itself is single-device,
 * /root/reference/src/gauss_bench.cu:558-575 only enumerates devices):
 *
 *   multi_test N BATCH NSHARDS     inverts BATCH matrices (a mix of dominant, mildly non-dominant and general ones) once with
 *                                  the single-device host-pointer call (inverse_gauss_batched_gpu / matinv_inverse_batched_host)
 *                                  and once as NSHARDS shards (matinv_inverse_batched_host_multi; shards beyond the device
 *                                  count share devices round robin) and compares the two results BIT FOR BIT, info included;
 *                                  also the Cholesky entry point on the SPD third of the batch.
 *   MATINV_DEVICES=K multi_test .. additionally routes the reference-named call through K shards (read at first use).
 *
 * prints "multi_test n=.. batch=.. shards=.. devices=.. identical" and exits 0, or the first differing matrix and exits 1.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/types.h"
#include "../../include/helper_cpu.h"
#include "../../include/helper_gpu.h"
#include "../../include/inverse_gpu.h"

#ifdef MATINV_DATATYPE_FLOAT
#define REF_GJ inverse_gauss_batched_gpu_f32
#define DT MATINV_F32
#else
#define REF_GJ inverse_gauss_batched_gpu
#define DT MATINV_F64
#endif

static int first_difference(const DataType *x, const DataType *y, size_t batch, size_t mat)
{
    int first = -1, count = 0;
    for (size_t k = 0; k < batch; ++k)
        if (memcmp(x + k * mat, y + k * mat, mat * sizeof(DataType)) != 0) {
            double worst = 0, scale = 0;
            for (size_t e = 0; e < mat; ++e) {
                worst = fmax(worst, fabs((double)x[k * mat + e] - (double)y[k * mat + e]));
                scale = fmax(scale, fabs((double)y[k * mat + e]));
            }
            if (count < 8) fprintf(stderr, "  matrix %zu (class %zu) differs: max |dx| %.3e of max |y| %.3e\n", k, k % 3, worst, scale);
            if (first < 0) first = (int)k;
            ++count;
        }
    if (count) fprintf(stderr, "  %d of %zu matrices differ\n", count, batch);
    return first;
}

int main(int argc, char const *argv[])
{
    ensure(argc >= 4, "Usage: multi_test N BATCH NSHARDS");
    const int n = atoi(argv[1]), batch = atoi(argv[2]), nshards = atoi(argv[3]);
    ensure(n >= 1 && batch >= 1, "N and BATCH must be positive");
    const size_t mat = (size_t)n * n, bytes = sizeof(DataType) * mat * (size_t)batch;
    Array As = (Array)malloc(bytes), One = (Array)malloc(bytes), Many = (Array)malloc(bytes), Ref = (Array)malloc(bytes);
    int *info1 = (int *)malloc(sizeof(int) * batch), *infoM = (int *)malloc(sizeof(int) * batch);
    ensure(As && One && Many && Ref && info1 && infoM, "out of memory");
    unsigned long long s = 0x5EEDull;
    for (int k = 0; k < batch; ++k) {
        /* k % 3 == 0: R + R^T + n I (SPD, dominant); 1: R + R^T + 0.35 n I (symmetric, NOT dominant: natural pivots are
           accepted with multipliers above 1 or rejected, matrix by matrix); 2: U(0,1) general (needs row exchanges) */
        const double diag = k % 3 == 0 ? (double)n : (k % 3 == 1 ? 0.35 * n : 0.0);
        for (int c = 0; c < n; ++c)
            for (int r = 0; r <= c; ++r) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const double u = (double)(s >> 11) / 9007199254740992.0;
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const double v = (double)(s >> 11) / 9007199254740992.0;
                const size_t i = (size_t)k * mat + (size_t)c * n + r, j = (size_t)k * mat + (size_t)r * n + c;
                if (k % 3 == 2) {
                    As[i] = (DataType)u;
                    As[j] = (DataType)v;
                } else {
                    As[i] = As[j] = (DataType)(u + v + (r == c ? diag : 0.0));
                }
            }
    }
    const int devices = matinv_device_count();
    ensure(devices >= 1, "no gfx950 device");
    gpuErrchk(matinv_inverse_batched_host(MATINV_ALGO_GAUSS_JORDAN, DT, n, As, One, (size_t)batch, info1));
    gpuErrchk(matinv_inverse_batched_host_multi(MATINV_ALGO_GAUSS_JORDAN, DT, n, As, Many, (size_t)batch, infoM, nshards));
    int bad = first_difference(One, Many, (size_t)batch, mat);
    if (bad >= 0 || memcmp(info1, infoM, sizeof(int) * batch) != 0) {
        fprintf(stderr, "multi_test: Gauss-Jordan, %d shards differ from one call (first differing matrix %d)\n", nshards, bad);
        return 1;
    }
    /* the reference-named entry point (takes MATINV_DEVICES shards when that is set) against the same bits */
    REF_GJ(NULL, n, As, Ref, batch);
    bad = first_difference(One, Ref, (size_t)batch, mat);
    if (bad >= 0) {
        fprintf(stderr, "multi_test: inverse_gauss_batched_gpu differs from matinv_inverse_batched_host (matrix %d)\n", bad);
        return 1;
    }
    /* residual of a few matrices, so that "identical" is not "identically wrong" */
    double worst = 0;
    for (int k = 0; k < batch; k += (batch / 7 > 0 ? batch / 7 : 1)) {
        if (info1[k] != 0) continue;
        double scale = 0;
        for (size_t e = 0; e < mat; ++e) scale = fmax(scale, fabs((double)One[(size_t)k * mat + e]));
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                double acc = 0;
                for (int t = 0; t < n; ++t) acc += (double)As[(size_t)k * mat + (size_t)t * n + r] * (double)One[(size_t)k * mat + (size_t)c * n + t];
                worst = fmax(worst, fabs(acc - (r == c ? 1.0 : 0.0)) / (scale * n));
            }
    }
    ensure(worst < (sizeof(DataType) == 8 ? 1e-10 : 1e-3), "residual too large");
    /* Cholesky on the SPD matrices (every third), gathered to the front */
    int nspd = 0;
    for (int k = 0; k < batch; k += 3) memcpy(Ref + (size_t)nspd++ * mat, As + (size_t)k * mat, mat * sizeof(DataType));
    gpuErrchk(matinv_inverse_batched_host(MATINV_ALGO_CHOLESKY, DT, n, Ref, One, (size_t)nspd, info1));
    gpuErrchk(matinv_inverse_batched_host_multi(MATINV_ALGO_CHOLESKY, DT, n, Ref, Many, (size_t)nspd, infoM, nshards));
    bad = first_difference(One, Many, (size_t)nspd, mat);
    if (bad >= 0 || memcmp(info1, infoM, sizeof(int) * nspd) != 0) {
        fprintf(stderr, "multi_test: Cholesky, %d shards differ from one call (first differing matrix %d)\n", nshards, bad);
        return 1;
    }
    printf("multi_test n=%d batch=%d shards=%d devices=%d residual=%.2e identical\n", n, batch, nshards, devices, worst);
    free(As), free(One), free(Many), free(Ref), free(info1), free(infoM);
    return 0;
}
