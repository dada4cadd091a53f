// chol_block.hpp -- blocked Cholesky factorisation of one matrix held in LDS by a 256-thread workgroup; shared by the LDS
// family (lds_kernels.hip) and the panel kernel of the blocked large-n path (blocked_gp_kernels.hip).
#pragma once
#include "common.hpp"

namespace matinv {

// sqrt(d) and 1/sqrt(d) from the hardware reciprocal-square-root estimate refined by Goldschmidt steps (coupled
// iteration g -> sqrt(d), h -> 1/(2 sqrt(d))) and one residual correction each: full working precision without the long
// IEEE sqrt / divide sequences, which matter here because every thread factors the diagonal block redundantly.
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double &sd, double &rs)
{
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y, r = fma(-h, g, 0.5);
    g = fma(g, r, g), h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g), h = fma(h, r, h);
    g = fma(fma(-g, g, d), h, g);
    rs = h + h;
    rs = fma(fma(-g, rs, 1.0), rs, rs);
    sd = g;
}
__device__ __forceinline__ void sqrt_and_rsqrt(float d, float &sd, float &rs)
{
    const float y = __builtin_amdgcn_rsqf(d);
    float g = d * y, h = 0.5f * y, r = fmaf(-h, g, 0.5f);
    g = fmaf(g, r, g), h = fmaf(h, r, h);
    g = fmaf(fmaf(-g, g, d), h, g);
    rs = h + h;
    rs = fmaf(fmaf(-g, rs, 1.0f), rs, rs);
    sd = g;
}

// phase 1: A = L L^T (choleskyDecomposition, inverse_cholesky_cpu.c:17-35; GPU kernels C4+C5,
// src/inverse_cholesky_gpu.cu:251-283), blocked right-looking with panels of CHOL_PB columns; two barriers per PANEL:
//   (1) every thread factors the CHOL_PB x CHOL_PB diagonal block in registers (redundantly -- broadcast LDS reads, no
//       communication) and solves ITS row of the panel against it (thread t <-> row k0 + t);
//   (2) rank-CHOL_PB update of the trailing lower triangle from registers: thread (ti, tj) of a 16 x 16 grid owns rows
//       ti + 16u and columns tj + 16v (cyclic, so the shrinking triangle stays balanced), keeps its rows of the panel in
//       registers, and touches each trailing element once per panel instead of once per column.
// Rows n .. nrows-1 are BORDER rows carried along (solved and updated, never pivots): the fused GP kernel stores its
// vectors there and reads (L^-1 a)^T, (L^-1 d)^T back. Only the lower triangle (row >= column) is read and written.
// Returns 0 or k+1 (block-uniform) when pivot k is not positive.
constexpr int CHOL_PB = 8;

template <class T>
__device__ __forceinline__ int chol_factor_lds(T *a, int ld, int n, int nrows)
{
    const int t = threadIdx.x, ti = t & 15, tj = t >> 4;
    for (int k0 = 0; k0 < n; k0 += CHOL_PB) {
        const int pb = (n - k0 < CHOL_PB) ? n - k0 : CHOL_PB;
        // (1) diagonal block (identity padded when the last panel is ragged)
        T l[CHOL_PB][CHOL_PB], inv[CHOL_PB];
#pragma unroll
        for (int c = 0; c < CHOL_PB; ++c)
#pragma unroll
            for (int r = c; r < CHOL_PB; ++r) l[r][c] = (r < pb) ? a[(k0 + c) * ld + k0 + r] : (T)(r == c);
        int bad = 0;
#pragma unroll
        for (int k = 0; k < CHOL_PB; ++k) {
            const T d = l[k][k];
            if ((!(d > 0) || d > max_finite<T>()) && bad == 0) bad = k0 + k + 1;  // non-positive, NaN or infinite pivot
            T sd, rs;
            sqrt_and_rsqrt(d, sd, rs);
            l[k][k] = sd, inv[k] = rs;
#pragma unroll
            for (int i = k + 1; i < CHOL_PB; ++i) l[i][k] *= rs;
#pragma unroll
            for (int j = k + 1; j < CHOL_PB; ++j)
#pragma unroll
                for (int i = j; i < CHOL_PB; ++i) l[i][j] = fma(-l[i][k], l[j][k], l[i][j]);
        }
        if (bad) return bad;  // every thread computed the same block
        // the rows INSIDE the diagonal block are stored after the barrier: other threads are still reading the block
        // (the update below never reads them)
        T x[CHOL_PB];
        const int r = k0 + t;
        if (r < nrows) {
#pragma unroll
            for (int c = 0; c < CHOL_PB; ++c) x[c] = (c < pb) ? a[(k0 + c) * ld + r] : (T)0;
#pragma unroll
            for (int c = 0; c < CHOL_PB; ++c) {
#pragma unroll
                for (int j = 0; j < c; ++j) x[c] = fma(-x[j], l[c][j], x[c]);
                x[c] *= inv[c];
            }
            if (t >= pb) {
#pragma unroll
                for (int c = 0; c < CHOL_PB; ++c)
                    if (c < pb) a[(k0 + c) * ld + r] = x[c];
            }
        }
        __syncthreads();
        if (t < pb) {
#pragma unroll
            for (int c = 0; c < CHOL_PB; ++c)
                if (c <= t) a[(k0 + c) * ld + r] = (t == c) ? l[c][c] : x[c];
        }
        // (2) trailing update
        const int j0 = k0 + pb;
        if (j0 < n) {
            for (int ub = 0; j0 + 16 * ub < nrows; ub += 8) {
                T li[8][CHOL_PB];
#pragma unroll
                for (int uu = 0; uu < 8; ++uu) {
                    const int row = j0 + ti + 16 * (ub + uu);
#pragma unroll
                    for (int c = 0; c < CHOL_PB; ++c) li[uu][c] = (row < nrows) ? a[(k0 + c) * ld + row] : (T)0;
                }
                const int last_row = j0 + ti + 16 * (ub + 7);
                for (int col = j0 + tj; col < n && col <= last_row; col += 16) {
                    T lj[CHOL_PB];
#pragma unroll
                    for (int c = 0; c < CHOL_PB; ++c) lj[c] = a[(k0 + c) * ld + col];
#pragma unroll
                    for (int uu = 0; uu < 8; ++uu) {
                        const int row = j0 + ti + 16 * (ub + uu);
                        if (row >= col && row < nrows) {
                            T x = a[col * ld + row];
#pragma unroll
                            for (int c = 0; c < CHOL_PB; ++c) x = fma(-li[uu][c], lj[c], x);
                            a[col * ld + row] = x;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    return 0;
}

}  // namespace matinv
