/*
 * gauss_cpu.h -- CPU mean / variance pipeline of gauss_bench (same names as /root/reference/include/gauss_cpu.h:16-58,
 * including the reference's spelling "calcluate").
 *
 *   Means[k]     = A_k^T (B_k + diag(C_k))^-1 D_k          As, Cs, Ds: batchSize x n;  Bs: batchSize x n x n (SPD)
 *   Variances[k] = E_k - A_k^T (B_k + diag(C_k))^-1 A_k     Es, Variances: batchSize scalars
 *
 * As in the reference (gauss_cpu.h:42 there) Bs and Cs are DESTROYED: B gets the diagonal added and is then
 * overwritten by the inverse, C receives the intermediate vector (B+C)^-1 x (src/gauss_cpu.c:47-64).
 * The variance uses the DOCUMENTED sign (E - ...; gauss_cpu.h:34, report.tex:51, generate_gaussian_matrices.m:37);
 * the reference's code adds instead (src/gauss_cpu.c:198), which SURVEY.md fact 7 shows to be a bug against its own
 * goldens. The *SolveCPU variants (built only with solve=1 and wrong there) are not provided.
 */
#ifndef HEADER_GAUSS_CPU_INCLUDED
#define HEADER_GAUSS_CPU_INCLUDED

#include "types.h"

#ifdef __cplusplus
extern "C" {
#endif

void calcluateMeanCPU(int n, Array As, Array Bs, Array Cs, Array Ds, Array Means, int batchSize);
void calcluateVarianceCPU(int n, Array As, Array Bs, Array Cs, Array Es, Array Variances, int batchSize);

#ifdef __cplusplus
}
#endif

#endif
