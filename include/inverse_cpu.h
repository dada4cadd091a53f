/*
 * inverse_cpu.h -- CPU inversion path of the bench CLIs (the reported CPU baseline of inverse_bench).
 *
 * Same four entry points as /root/reference/include/inverse_cpu.h:8-15, implemented in
 * cuda-matrix-inversion_amd/host/inverse_cpu.c WITHOUT LAPACK (the reference calls sgetrf_/sgetri_ and
 * spotrf_/spotri_, src/inverse.c:63-65,92-95; no LAPACK is guaranteed on an MI355X host):
 *   inverse_lu_blas(a, workspace, N)        in-place inverse of one column-major N x N matrix by LU with partial
 *                                           pivoting; workspace >= N*N scalars (reference: sgetri work array)
 *   inverse_lu_blas_omp(as, N, batchSize)   the same over a contiguous batch, OpenMP schedule(dynamic, 8)
 *   inverse_chol_blas(a, N)                 in-place inverse of one SPD matrix by Cholesky. The reference leaves only
 *                                           the UPPER triangle valid (spotri "U"); here the full symmetric inverse is
 *                                           written, a superset its consumers (ssymv Upper, src/gauss_cpu.c:54) accept
 *   inverse_chol_blas_omp(as, N, batchSize)
 * Errors (singular / not SPD) are fatal through ensure(), as in the reference (src/inverse.c:64,66,93,96).
 */
#ifndef HEADER_INVERSE_CPU_INCLUDED
#define HEADER_INVERSE_CPU_INCLUDED

#include "types.h"

#ifdef __cplusplus
extern "C" {
#endif

void inverse_lu_blas(Array a, Array workspace, int N);
void inverse_lu_blas_omp(Array as, int N, int batchSize);
void inverse_chol_blas(Array a, int N);
void inverse_chol_blas_omp(Array as, int N, int batchSize);

#ifdef __cplusplus
}
#endif

#endif
