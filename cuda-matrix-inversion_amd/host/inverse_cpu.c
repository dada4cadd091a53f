/*
 * inverse_cpu.c -- the CPU inversion path the bench CLIs time next to the GPU (product host code, LAPACK-free).
 * Entry points mirror /root/reference/src/inverse.c:55-107; the arithmetic the reference delegates to LAPACK
 * (sgetrf_/sgetri_ :63-65, spotrf_/spotri_ :92-95) is written out here.
 */
#include <math.h>
#include <string.h>

#include "../../include/helper_cpu.h"
#include "../../include/inverse_cpu.h"

#define AT(p, r, c) (p)[(size_t)(c)*N + (r)]

/* LU with partial pivoting, then inv(U), then X*L = inv(U), then undo the row swaps as column swaps. */
void inverse_lu_blas(Array a, Array workspace, int N)
{
    int *piv = (int *)malloc(sizeof(int) * (size_t)(N + 1));
    ensure(piv != NULL, "Could not allocate pivot array for matrix inversion");

    for (int k = 0; k < N; ++k) {
        int p = k;
        DataType best = (DataType)fabs((double)AT(a, k, k));
        for (int i = k + 1; i < N; ++i) {
            DataType v = (DataType)fabs((double)AT(a, i, k));
            if (v > best) { best = v; p = i; }
        }
        ensure(best > 0, "Error code %d in LU-decomposition", k + 1);
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < N; ++j) { DataType t = AT(a, k, j); AT(a, k, j) = AT(a, p, j); AT(a, p, j) = t; }
        const DataType r = (DataType)1 / AT(a, k, k);
        for (int i = k + 1; i < N; ++i) AT(a, i, k) *= r;
        for (int j = k + 1; j < N; ++j) {
            const DataType u = AT(a, k, j);
            for (int i = k + 1; i < N; ++i) AT(a, i, j) -= AT(a, i, k) * u;
        }
    }
    for (int j = 0; j < N; ++j) { /* upper triangle <- inv(U) */
        AT(a, j, j) = (DataType)1 / AT(a, j, j);
        const DataType njj = -AT(a, j, j);
        for (int i = 0; i < j; ++i) workspace[i] = AT(a, i, j);
        for (int i = 0; i < j; ++i) {
            DataType s = 0;
            for (int k = i; k < j; ++k) s += AT(a, i, k) * workspace[k];
            AT(a, i, j) = s * njj;
        }
    }
    for (int j = N - 1; j >= 0; --j) { /* X * L = inv(U), last column first */
        for (int i = j + 1; i < N; ++i) { workspace[i] = AT(a, i, j); AT(a, i, j) = 0; }
        for (int k = j + 1; k < N; ++k) {
            const DataType w = workspace[k];
            for (int i = 0; i < N; ++i) AT(a, i, j) -= AT(a, i, k) * w;
        }
    }
    for (int k = N - 1; k >= 0; --k)
        if (piv[k] != k)
            for (int i = 0; i < N; ++i) { DataType t = AT(a, i, k); AT(a, i, k) = AT(a, i, piv[k]); AT(a, i, piv[k]) = t; }
    free(piv);
}

void inverse_lu_blas_omp(Array as, int N, int batchSize)
{
#pragma omp parallel shared(as)
    {
        Array workspace = (Array)malloc(sizeof(DataType) * (size_t)N * N);
        ensure(workspace != NULL, "Could not allocate workspace for matrix inversion");
        int i;
#pragma omp for schedule(dynamic, 8)
        for (i = 0; i < batchSize; ++i) inverse_lu_blas(as + (size_t)i * N * N, workspace, N);
        free(workspace);
    }
}

/* A = L L^T (lower), L <- L^-1 in place, A^-1 = L^-T L^-1 written to both triangles. */
void inverse_chol_blas(Array a, int N)
{
    for (int k = 0; k < N; ++k) {
        DataType d = AT(a, k, k);
        ensure(d > 0, "Error code %d in cholesky factorization", k + 1);
        d = (DataType)sqrt((double)d);
        AT(a, k, k) = d;
        const DataType r = (DataType)1 / d;
        for (int i = k + 1; i < N; ++i) AT(a, i, k) *= r;
        for (int j = k + 1; j < N; ++j) {
            const DataType ljk = AT(a, j, k);
            for (int i = j; i < N; ++i) AT(a, i, j) -= AT(a, i, k) * ljk;
        }
    }
    for (int j = N - 1; j >= 0; --j) { /* trtri, lower, last column first */
        const DataType ajj = (DataType)1 / AT(a, j, j);
        for (int i = N - 1; i > j; --i) {
            DataType s = 0;
            for (int k = j + 1; k <= i; ++k) s += AT(a, i, k) * AT(a, k, j);
            AT(a, i, j) = -s * ajj; /* rows below i still hold the old column: walk upwards */
        }
        AT(a, j, j) = ajj;
    }
    for (int j = 0; j < N; ++j) /* lauum: lower triangle <- L^-T L^-1, column by column, top row first */
        for (int i = j; i < N; ++i) {
            DataType s = 0;
            for (int k = i; k < N; ++k) s += AT(a, k, i) * AT(a, k, j);
            AT(a, i, j) = s;
        }
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < j; ++i) AT(a, i, j) = AT(a, j, i);
}

void inverse_chol_blas_omp(Array as, int N, int batchSize)
{
    int i;
#pragma omp parallel for schedule(dynamic, 8)
    for (i = 0; i < batchSize; ++i) inverse_chol_blas(as + (size_t)i * N * N, N);
}
