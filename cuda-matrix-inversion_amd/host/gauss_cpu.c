/*
 * gauss_cpu.c -- CPU mean / variance pipeline timed by gauss_bench next to the GPU path (product host code).
 * Same flow as /root/reference/src/gauss_cpu.c:23-77 (mean) and :156-209 (variance): add the diagonal, invert by
 * Cholesky in place, y = M^-1 x into C, dot with A. The BLAS calls of the reference (ssymv :54, sdot :66) are plain
 * loops here; the variance uses the documented sign E - A^T M^-1 A (see include/gauss_cpu.h).
 */
#include "../../include/gauss_cpu.h"
#include "../../include/helper_cpu.h"
#include "../../include/inverse_cpu.h"

/* y = M x for a full symmetric column-major M */
static void symv(int n, const DataType *M, const DataType *x, DataType *y)
{
    for (int i = 0; i < n; ++i) y[i] = 0;
    for (int j = 0; j < n; ++j) {
        const DataType xj = x[j];
        const DataType *col = M + (size_t)j * n;
        for (int i = 0; i < n; ++i) y[i] += col[i] * xj;
    }
}

static DataType dot(int n, const DataType *x, const DataType *y)
{
    DataType s = 0;
    for (int i = 0; i < n; ++i) s += x[i] * y[i];
    return s;
}

/* q_k = A_k^T (B_k + diag C_k)^-1 X_k ; B and C are destroyed */
static DataType quad_form(int n, const DataType *a, DataType *B, DataType *c, const DataType *x)
{
    for (int j = 0; j < n; ++j) B[j + (size_t)j * n] += c[j];
    inverse_chol_blas(B, n);
    symv(n, B, x, c);
    return dot(n, a, c);
}

void calcluateMeanCPU(int n, Array As, Array Bs, Array Cs, Array Ds, Array Means, int batchSize)
{
    int i;
#pragma omp parallel for schedule(dynamic, 8)
    for (i = 0; i < batchSize; ++i)
        Means[i] = quad_form(n, As + (size_t)i * n, Bs + (size_t)i * n * n, Cs + (size_t)i * n, Ds + (size_t)i * n);
}

void calcluateVarianceCPU(int n, Array As, Array Bs, Array Cs, Array Es, Array Variances, int batchSize)
{
    int i;
#pragma omp parallel for schedule(dynamic, 16)
    for (i = 0; i < batchSize; ++i)
        Variances[i] = Es[i] - quad_form(n, As + (size_t)i * n, Bs + (size_t)i * n * n, Cs + (size_t)i * n,
                                         As + (size_t)i * n);
}
