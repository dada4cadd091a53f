/*
 * oracle.c -- CPU oracle for the batched small-matrix inversion hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The
 * product path (cuda-matrix-inversion_amd/) never links, imports or calls it.
 *
 * It restates, in plain C, the arithmetic of the reference's CPU/GPU
 * algorithms for this path (file:line citations are in oracle_impl.inc):
 *   - Gauss-Jordan as in src/gauss/batched_invert.cu:17-95 (reference
 *     zero-only pivoting) and with partial pivoting,
 *   - Cholesky inverse as in src/inverse_cholesky_cpu.c:17-85,
 *   - LU inverse as LAPACK getrf/getri (called at src/inverse.c:63-65),
 *   - mean / variance pipeline as in src/gauss_cpu.c:41-72,174-206.
 *
 * Parity pinning (see tests/test_oracle.py and DESIGN.md section 3):
 *   - tests/golden/ref/inverse_100_{8x8,16x16,32x32}: aInv.mats (4 digits),
 *   - tests/golden/ref/gaussian_100_{8,16,32,64}: means.mats, variances.mats,
 *   - tests/golden/ref/simpleMean: chol.mats / cholinv.mats (6 decimals),
 *   - oracle/_ref/inverse_cholesky_cpu: the reference's own scalar Cholesky
 *     (src/inverse_cholesky_cpu.c, N=4) compiled as it lies and run here.
 * The 64x64 / 128x128 inverse goldens are absent from the reference mount
 * (.MISSING_LARGE_BLOBS), so beyond n=32 the inverse is pinned only through
 * the n=64 pipeline scalars and through cross-agreement of the four
 * algorithms above.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define T double
#define SUF(x) x##_f64
#define ORACLE_IS_DOUBLE 1
#include "oracle_impl.inc"
#undef T
#undef SUF
#undef ORACLE_IS_DOUBLE

#define T float
#define SUF(x) x##_f32
#define ORACLE_IS_DOUBLE 0
#include "oracle_impl.inc"
#undef T
#undef SUF
#undef ORACLE_IS_DOUBLE

int oracle_abi_version(void) { return 1; }
