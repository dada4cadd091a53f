/*
 * queue_test.c -- the size-binned multi-queue of libmatinv_hip.so driven from plain C (no torch, no HIP headers): mixed-size
 * Gaussian-process items (the queues the reference's README sketches, /root/reference/README.md:41-44; BASELINE configs[4])
 * are submitted as chunks, flushed, and every mean / variance is checked against the host path (calcluateMeanCPU /
 * calcluateVarianceCPU, host/gauss_cpu.c, the functions gauss_bench reports as means_cpu / variances_cpu).
 *
 *   queue_test [seed]        prints "queue_test items=.. max_err_mean=.. max_err_var=.." and exits 0 within the tolerance
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/types.h"
#include "../../include/gauss_cpu.h"
#include "../../include/helper_cpu.h"
#include "../../include/matinv.h"

#ifdef MATINV_DATATYPE_FLOAT
#define QT_DTYPE MATINV_F32
#define TOL 2e-4
#else
#define QT_DTYPE MATINV_F64
#define TOL 1e-10
#endif

static unsigned long long rng_state;
static double urand(void)
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

static void *to_device(const void *h, size_t bytes)
{
    void *tab[1];
    size_t pitch;
    ensure(matinv_batched_malloc(tab, &pitch, bytes, 1) == 0, "device allocation failed: %s", matinv_last_error());
    ensure(matinv_memcpy_2d(tab[0], pitch, h, bytes, bytes, 1, 1) == 0, "H2D failed: %s", matinv_last_error());
    return tab[0];
}

int main(int argc, char const *argv[])
{
    rng_state = argc > 1 ? strtoull(argv[1], NULL, 10) : 0x5EEDull;
    /* chunks of (n, count): several n per bin, a bin hit twice, chunk sizes 1 .. 40 */
    const int sizes[][2] = {{32, 40}, {7, 5}, {128, 6}, {100, 3}, {32, 17}, {300, 2}, {20, 9}, {64, 4}, {512, 1}, {16, 31}, {130, 2}};
    const int nchunks = (int)(sizeof sizes / sizeof sizes[0]);
    matinv_queue *q = NULL;
    ensure(matinv_queue_create(&q, QT_DTYPE, NULL, 0) == 0, "matinv_queue_create failed");
    size_t total = 0;
    for (int k = 0; k < nchunks; ++k) total += (size_t)sizes[k][1];
    Array wantM = (Array)malloc(sizeof(DataType) * total), wantV = (Array)malloc(sizeof(DataType) * total);
    void *dev[64][5];
    for (int k = 0; k < nchunks; ++k) {
        const int n = sizes[k][0], cnt = sizes[k][1];
        const size_t vec = (size_t)n * cnt, mat = vec * n;
        Array a = (Array)malloc(sizeof(DataType) * vec), B = (Array)malloc(sizeof(DataType) * mat);
        Array c = (Array)malloc(sizeof(DataType) * vec), d = (Array)malloc(sizeof(DataType) * vec), e = (Array)malloc(sizeof(DataType) * cnt);
        for (size_t i = 0; i < vec; ++i) a[i] = (DataType)urand(), c[i] = (DataType)urand(), d[i] = (DataType)urand();
        for (int i = 0; i < cnt; ++i) e[i] = (DataType)urand();
        for (int it = 0; it < cnt; ++it) /* B = R + R^T + n I (tests/generate_gaussian_matrices.m:15-28 of the reference) */
            for (int col = 0; col < n; ++col)
                for (int row = 0; row <= col; ++row) {
                    const DataType v = (DataType)(urand() + urand() + (row == col ? (double)n : 0.0));
                    B[(size_t)it * n * n + (size_t)col * n + row] = v;
                    B[(size_t)it * n * n + (size_t)row * n + col] = v;
                }
        size_t first = 0;
        dev[k][0] = to_device(a, sizeof(DataType) * vec), dev[k][1] = to_device(B, sizeof(DataType) * mat);
        dev[k][2] = to_device(c, sizeof(DataType) * vec), dev[k][3] = to_device(d, sizeof(DataType) * vec);
        dev[k][4] = to_device(e, sizeof(DataType) * cnt);
        ensure(matinv_queue_submit(q, n, dev[k][0], dev[k][1], dev[k][2], dev[k][3], dev[k][4], (size_t)cnt, &first) == 0,
               "submit failed: %s", matinv_queue_last_error(q));
        /* the CPU path destroys Bs and Cs (include/gauss_cpu.h): give it copies */
        Array B2 = (Array)malloc(sizeof(DataType) * mat), c2 = (Array)malloc(sizeof(DataType) * vec);
        memcpy(B2, B, sizeof(DataType) * mat), memcpy(c2, c, sizeof(DataType) * vec);
        calcluateMeanCPU(n, a, B2, c2, d, wantM + first, cnt);
        memcpy(B2, B, sizeof(DataType) * mat), memcpy(c2, c, sizeof(DataType) * vec);
        calcluateVarianceCPU(n, a, B2, c2, e, wantV + first, cnt);
        free(a), free(B), free(c), free(d), free(e), free(B2), free(c2);
    }
    size_t pending = 0;
    ensure(matinv_queue_pending(q, &pending, NULL) == 0 && pending == total, "pending count wrong");
    Array zeros = (Array)calloc(total, sizeof(DataType));
    void *dM = to_device(zeros, sizeof(DataType) * total), *dV = to_device(zeros, sizeof(DataType) * total);
    ensure(matinv_queue_flush(q, dM, dV, NULL) == 0, "flush failed: %s", matinv_queue_last_error(q));
    Array gotM = (Array)malloc(sizeof(DataType) * total), gotV = (Array)malloc(sizeof(DataType) * total);
    ensure(matinv_memcpy_2d(gotM, sizeof(DataType) * total, dM, sizeof(DataType) * total, sizeof(DataType) * total, 1, 0) == 0, "D2H failed");
    ensure(matinv_memcpy_2d(gotV, sizeof(DataType) * total, dV, sizeof(DataType) * total, sizeof(DataType) * total, 1, 0) == 0, "D2H failed");
    double em = 0, ev = 0;
    for (size_t i = 0; i < total; ++i) {
        const double dm = fabs((double)gotM[i] - (double)wantM[i]), dv = fabs((double)gotV[i] - (double)wantV[i]);
        if (!(dm <= em)) em = dm;
        if (!(dv <= ev)) ev = dv;
    }
    ensure(matinv_queue_pending(q, &pending, NULL) == 0 && pending == 0, "queue not empty after flush");
    /* the same items once more, submitted and flushed on the queue's OWN stream (matinv_queue_stream: the form a producer with two
     * flushes in flight uses): the same kernels on the same data, so the same bits */
    void *own = matinv_queue_stream(q);
    ensure(own != NULL, "matinv_queue_stream returned NULL");
    for (int k = 0; k < nchunks; ++k) {
        size_t first = 0;
        ensure(matinv_queue_submit(q, sizes[k][0], dev[k][0], dev[k][1], dev[k][2], dev[k][3], dev[k][4], (size_t)sizes[k][1], &first) == 0,
               "second submit failed: %s", matinv_queue_last_error(q));
    }
    void *dM2 = to_device(zeros, sizeof(DataType) * total), *dV2 = to_device(zeros, sizeof(DataType) * total);
    ensure(matinv_queue_flush(q, dM2, dV2, own) == 0, "flush on the queue's own stream failed: %s", matinv_queue_last_error(q));
    ensure(matinv_device_synchronize() == 0, "synchronize failed");
    Array gotM2 = (Array)malloc(sizeof(DataType) * total), gotV2 = (Array)malloc(sizeof(DataType) * total);
    ensure(matinv_memcpy_2d(gotM2, sizeof(DataType) * total, dM2, sizeof(DataType) * total, sizeof(DataType) * total, 1, 0) == 0, "D2H failed");
    ensure(matinv_memcpy_2d(gotV2, sizeof(DataType) * total, dV2, sizeof(DataType) * total, sizeof(DataType) * total, 1, 0) == 0, "D2H failed");
    ensure(memcmp(gotM, gotM2, sizeof(DataType) * total) == 0 && memcmp(gotV, gotV2, sizeof(DataType) * total) == 0,
           "the flush on the queue's own stream gave different bits");
    matinv_batched_free(&dM2), matinv_batched_free(&dV2);
    free(gotM2), free(gotV2);
    matinv_queue_destroy(q);
    for (int k = 0; k < nchunks; ++k)
        for (int j = 0; j < 5; ++j) matinv_batched_free(&dev[k][j]);
    matinv_batched_free(&dM), matinv_batched_free(&dV);
    printf("queue_test items=%zu max_err_mean=%.3e max_err_var=%.3e\n", total, em, ev);
    free(wantM), free(wantV), free(gotM), free(gotV), free(zeros);
    return (em < TOL && ev < TOL) ? 0 : 1;
}
