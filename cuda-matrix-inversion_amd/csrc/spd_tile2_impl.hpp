// spd_tile2_impl.hpp (instantiated by spd_tile2_kernels.hip and spd_tile2_wide*_kernels.hip, fp64) -- the symmetric blocked sweep of
// matinv_spd_tile_f64 (tile_kernels.inc: read its header first) on TWO wavefronts per matrix, LOWER tiles only: 112 < n <= 128
// (8 x 8 tiles, r03) and -- r04 -- 128 < n <= 176 (9 ... 11 tiles per dimension: 23 ... 33 lower tiles = up to 288 accumulator registers
// per wave, so these run ONE wave per SIMD on VGPRs + AGPRs, two matrices per CU; before, these sizes swept ALL n^2 tiles with one
// wavefront per tile column and one matrix per CU: 130^2 Cholesky 3.4e6 inv/s, a 2.9 x cliff behind 128^2).
//
// 8 x 8 tiles keep 36 lower tiles = 288 fp64 registers: 32 more than the AGPR file, so the one-wavefront form that serves n <= 112
// spills, and r01 / r02 ran these sizes on four wavefronts that sweep ALL 64 tiles (tile4_impl.hpp: the B operand there is a
// wave's own pivot-row registers, which needs the upper tiles to be current). Here the B operand comes from the LDS panel by
// symmetry, as in the one-wavefront kernel, so the upper tiles are never needed, and the tile columns are dealt to the two waves
// FOLDED -- wave 0 owns columns 0, 3, 4, 7 (8 + 5 + 4 + 1 lower tiles), wave 1 owns 1, 2, 5, 6 (7 + 6 + 3 + 2): 18 tiles each.
// Both waves run the same code on 20 register slots (columns of height 8 / 6 / 4 / 2; the two slots a wave does not own carry
// garbage that is never staged, stored or folded -- 2 of 20 MFMAs wasted, no wave-dependent control flow): 160 accumulator
// registers, two waves per SIMD, FOUR matrices per CU (the four-wave kernel: two), 40 MFMAs per block step and matrix instead of 64.
// Per block step: the owner of the pivot tile column stages its part of the panel, every wave adds the transposed pieces of tile
// row tK it owns (W[I, K] = W[K, I]^T), ONE workgroup barrier (the panel is double buffered), both waves solve the panel
// redundantly (PanelSolve, SPD mode: A operand and the symmetric B operand), prepare their operands and issue their MFMAs.
// (Measured and not kept: ONE wave -- taking turns -- solving the panel and publishing the A operand and the symmetric B operand
// through LDS, two barriers per step: Cholesky 128^2 9.8e6 -> 1.07e7 inv/s, pipeline 1.17e7 -> 1.04e7 items/s: a wash.)
// GP = the fused mean / variance on the same sweep (see SpdGp in tile_kernels.inc): diag c added while loading, a^T M^-1 d folded
// out of the accumulators of both waves, nothing stored.
//
// Replaces, for SPD input of these sizes, the Cholesky families of /root/reference/src/inverse_cholesky_gpu.cu:55-765 and
// calcluateMean / calcluateVariance (src/gauss_bench.cu:127-265,275-409).
#pragma once
#include <cstdio>

#include "tile_common.hpp"

namespace matinv {

template <class T>
struct Spd2Gp {
    const T *a, *c, *d, *e;  // d == nullptr: variance, out = e - a^T M^-1 a
    T *out;
};

// tile column of local column jl of wave w (folded: 0 1 | 1 0 | 0 1 | ... so that both waves hold the same number of lower tiles
// +- 1), and its inverse. Local column jl exists from tile row 2 jl on.
// (W = wavefronts per matrix: 2, or -- r04, 12 x 12 tiles -- 3: 0 1 2 | 2 1 0 | ..., 26 lower tiles each)
template <int W = 2>
constexpr int spd2_col_c(int w, int jl) { return (jl & 1) ? (2 * W * (jl >> 1) + 2 * W - 1 - w) : (2 * W * (jl >> 1) + w); }
template <int W = 2>
__device__ __forceinline__ int spd2_col(int w, int jl) { return (jl & 1) ? (2 * W * (jl >> 1) + 2 * W - 1 - w) : (2 * W * (jl >> 1) + w); }
template <int W = 2>
constexpr int spd2_owner(int tj) { return (tj % (2 * W)) < W ? (tj % (2 * W)) : 2 * W - 1 - (tj % (2 * W)); }
template <int W = 2>
constexpr int spd2_local(int tj) { return 2 * (tj / (2 * W)) + ((tj % (2 * W)) >= W ? 1 : 0); }

template <int V>
struct IntC2 {
    static constexpr int value = V;
};
// f(IntC2<K>()), ..., f(IntC2<END - 1>()): the block steps with their number as a compile-time constant
template <int K, int END>
struct Spd2Steps {
    template <class F>
    static __device__ __forceinline__ void run(F &f)
    {
        f(IntC2<K>());
        if constexpr (K + 1 < END) Spd2Steps<K + 1, END>::run(f);
    }
};

// SELF: no Cholesky kernel behind this one serves every such n (the LDS kernel stops at n = 137), so an item that is not positive
// definite is finished here: info = the column of the first non-positive pivot + 1 (the Cholesky contract), output NaN-filled.
template <int NT, bool GP, int W = 2>
__device__ __forceinline__ void spd_tile2_body(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt, unsigned batch,
                                               int *work_count, int *work_list, double *panel2, double *tbuf2, Spd2Gp<double> gp)
{
    typedef double T;
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    static_assert(NT >= 8 && NT <= 12 && (W == 2 || W == 3), "two or three wavefronts: 8 ... 12 tiles per dimension");
    constexpr int N = 16 * NT, NKB = 4 * NT, NL = (NT + W - 1) / W;
    constexpr bool SELF = NT > 8;
    constexpr int TSTRIDE = 17;  // padded row stride of the 16 x 16 transpose buffers (one per wave)
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;  // wave-uniform: 0 .. W - 1
    T *const tbuf = tbuf2 + w * (16 * TSTRIDE);
    int tjs[NL];
#pragma unroll
    for (int jl = 0; jl < NL; ++jl) tjs[jl] = spd2_col<W>(w, jl);

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int n = n_rt;
        asm volatile("" : "+s"(n));  // run-time n opaque once per matrix: see gj_tile_body
        int q = l >> 4, c = l & 15;
        asm volatile("" : "+v"(q), "+v"(c));

        // acc[jl][ti] = tile (ti, tjs[jl]) of W = A^T (symmetric); slots ti = W jl .. NT - 1; a slot with ti < tjs[jl] is not owned
        vec4 acc[NL][NT];
#pragma unroll
        for (int jl = 0; jl < NL; ++jl) {
            const int tj = tjs[jl];
#pragma unroll
            for (int ti = W * jl; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                    const bool in = row < n && col < n;
                    // only the LOWER triangle of A is read: inside the diagonal tile the upper elements come from their mirror
                    const int hi = row > col ? row : col, lo = row > col ? col : row;
                    T v = (ti >= tj) ? (in ? A[(unsigned)(lo * n + hi)] : ((row == col) ? (T)1 : (T)0)) : (T)0;
                    if (GP && ti == tj && row == col && in) v += gp.c[(size_t)mat * n + row];  // addDiagonal, gauss_bench.cu:38-43
                    acc[jl][ti][r] = v;
                }
        }
        unsigned long long bad = 0;
        int badinfo = 0;

        // panel of block kb into buffer kb & 1: rows >= 16 tK from the tiles of column tK (its owner), rows < 16 tK from the
        // pivot rows of tile row tK, each wave the columns it owns (W[16 ti + c][pivot q] = W[pivot q][16 ti + c])
        auto stage = [&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            constexpr int tK = kb >> 2, rK = kb & 3, jo = spd2_local<W>(tK);
            T *const buf = panel2 + (kb & 1) * (N * 4);
            if (w == spd2_owner<W>(tK) && G::blk(c) == rK) {
#pragma unroll
                for (int ti = tK; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[jo][ti][r];
            }
#pragma unroll
            for (int jl = 0; jl < NL; ++jl) {
                if (tK >= W * jl) {  // (folds after unrolling: the slot exists)
                    if (tjs[jl] < tK) buf[(16 * tjs[jl] + c) * 4 + q] = acc[jl][tK][rK];
                }
            }
        };

        stage(IntC2<0>());
        __syncthreads();
        const int last_blocks = G::real_blocks(n - 16 * (NT - 1));
        // (r04, measured and not kept: the look-ahead of the one-wavefront sweep -- the slots the next panel is read from updated first, the
        // next panel staged and solved between the remaining MFMAs. With the second operand set the 9 x 9 kernel needs 460 registers
        // instead of 368 and the 10 x 10 / 11 x 11 ones spill or crash hipcc's "Rewrite AGPR-Copy-MFMA" pass; where it compiles it is
        // SLOWER: Cholesky 130^2 6.0e6 -> 5.6e6 inv/s, 160^2 4.3e6 -> 3.2e6; at 8 x 8 tiles, two waves per SIMD, 1 413 registers spill.)
        auto step = [&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            constexpr int tK = kb >> 2, rK = kb & 3, jo = spd2_local<W>(tK);
            // ragged n: a block step over identity padding only changes nothing (see spd_tile_body); n is workgroup-uniform
            if (kb > 4 * (NT - 1) && kb - 4 * (NT - 1) >= last_blocks) return;
            const T *const buf = panel2 + (kb & 1) * (N * 4);
            T aop[NT], bsym[NT];
            PanelSolve<NT, true, T> ps;
            if (SELF) ps.binfo = &badinfo;
#pragma unroll
            for (int s = 0; s < PanelSolve<NT, true, T>::NSTAGE; ++s) ps.stage(s, buf, kb, q, c, aop, bsym, bad);
            // B operand of the wave's columns: the old panel by symmetry; -I_4 on the pivot columns (their owner)
            T bop[NL];
#pragma unroll
            for (int jl = 0; jl < NL; ++jl) {
                const int t0 = spd2_col_c<W>(0, jl), t1 = spd2_col_c<W>(1, jl), t2 = spd2_col_c<W>(W - 1, jl);  // fold to literals after unrolling
                const T b0 = t0 < NT ? bsym[t0 < NT ? t0 : 0] : (T)0, b1 = t1 < NT ? bsym[t1 < NT ? t1 : 0] : (T)0;
                if constexpr (W == 2) {
                    bop[jl] = w ? b1 : b0;
                } else {
                    const T b2 = t2 < NT ? bsym[t2 < NT ? t2 : 0] : (T)0;
                    bop[jl] = w == 0 ? b0 : (w == 1 ? b1 : b2);
                }
            }
            const bool panel_lane = (w == spd2_owner<W>(tK)) && G::blk(c) == rK;
            const bool diag_lane = panel_lane && (G::piv(c) == q);
            bop[jo] = panel_lane ? (diag_lane ? (T)-1 : (T)0) : bop[jo];
            // C operand: zero on the pivot columns (owner) and on the pivot rows (every owned tile of tile row tK)
#pragma unroll
            for (int ti = tK; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[jo][ti][r] = panel_lane ? (T)0 : acc[jo][ti][r];
#pragma unroll
            for (int jl = 0; jl < NL; ++jl) {
                if (tK >= W * jl) acc[jl][tK][rK] = (tjs[jl] <= tK) ? (T)0 : acc[jl][tK][rK];
            }
#pragma unroll
            for (int jl = 0; jl < NL; ++jl)
#pragma unroll
                for (int ti = W * jl; ti < NT; ++ti) acc[jl][ti] = G::mfma(aop[ti], bop[jl], acc[jl][ti]);
            if constexpr (kb + 1 < NKB) {
                if (!(kb + 1 > 4 * (NT - 1) && kb + 1 - 4 * (NT - 1) >= last_blocks)) stage(IntC2<kb + 1>());
            }
            __syncthreads();
        };
        Spd2Steps<0, NKB>::run(step);

        if (GP) {
            // s = sum_ij a_i W_ij d_j over the owned lower tiles (an off-diagonal tile also stands for its mirror); W = -M^-1
            const T *va = gp.a + (size_t)mat * n;
            const T *vd = gp.d ? gp.d + (size_t)mat * n : va;
            T *const sa = panel2, *const sd = panel2 + N, *const part = panel2 + 2 * N;  // both panel buffers are free now
            for (int i = threadIdx.x; i < N; i += 64 * W) {
                sa[i] = i < n ? va[i] : (T)0;
                sd[i] = i < n ? vd[i] : (T)0;
            }
            __syncthreads();
            T s = 0;
            if (bad == 0) {
#pragma unroll
                for (int jl = 0; jl < NL; ++jl) {
                    const int tj = tjs[jl];
                    const T ac = sa[16 * tj + c], dc = sd[16 * tj + c];
#pragma unroll
                    for (int ti = W * jl; ti < NT; ++ti) {
                        if (ti < tj) continue;  // the slot this wave does not own
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * ti + G::trow(r, q);
                            const T ar = sa[row], dr = sd[row];
                            const T wgt = (ti == tj) ? ar * dc : fma_t(ar, dc, dr * ac);
                            s = fma_t(acc[jl][ti][r], wgt, s);
                        }
                    }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
                if (l == 0) part[w] = s;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                if (bad == 0) {
                    const T sum = part[0] + part[1] + (W > 2 ? part[2] : (T)0);
                    gp.out[mat] = gp.d ? -sum : gp.e[mat] + sum;
                    if (info) info[mat] = 0;
                } else if (SELF) {
                    gp.out[mat] = nan_of<T>();
                    if (info) info[mat] = badinfo;
                } else {
                    const int slot = atomicAdd(work_count, 1);  // not SPD: the LDS pipeline kernel reports the column
                    work_list[slot] = (int)mat;
                }
            }
        } else if (bad == 0) {
            // W = -A^-1: owned lower tiles go out directly, the mirror of every off-diagonal one through the wave's transpose buffer
#pragma unroll
            for (int jl = 0; jl < NL; ++jl) {
                const int tj = tjs[jl];
#pragma unroll
                for (int ti = W * jl; ti < NT; ++ti) {
                    if (ti < tj) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                        if (row < n && col < n) X[(unsigned)(row * n + col)] = -acc[jl][ti][r];
                    }
                    if (ti > tj) {
                        wave_lds_sync();
#pragma unroll
                        for (int r = 0; r < 4; ++r) tbuf[G::trow(r, q) * TSTRIDE + c] = -acc[jl][ti][r];
                        wave_lds_sync();
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // element (row 16 tj + 4r + q, col 16 ti + c) of the result = tile (ti, tj)[c][4r + q]
                            const int row = 16 * tj + G::trow(r, q), col = 16 * ti + c;
                            const T v = tbuf[c * TSTRIDE + G::trow(r, q)];
                            if (row < n && col < n) X[(unsigned)(row * n + col)] = v;
                        }
                    }
                }
            }
            if (info && threadIdx.x == 0) info[mat] = 0;
        } else if (SELF) {
            for (unsigned e = threadIdx.x; e < (unsigned)(n * n); e += 64u * W) X[e] = nan_of<T>();  // (a plain strided fill: see gj_tile4_body)
            if (info && threadIdx.x == 0) info[mat] = badinfo;
        } else if (threadIdx.x == 0) {
            const int slot = atomicAdd(work_count, 1);  // not SPD: the LDS Cholesky kernel reports the column
            work_list[slot] = (int)mat;
        }
        __syncthreads();  // the next matrix stages its first panel into the same buffers
    }
}

template <bool GP>
__global__ __launch_bounds__(128, 2) void matinv_spd_tile2_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                              unsigned batch, int *work_count, int *work_list, Spd2Gp<double> gp)
{
    __shared__ __attribute__((aligned(16))) double panel2[2 * 128 * 4];  // double buffered [row][4 pivot columns]
    __shared__ __attribute__((aligned(16))) double tbuf2[2 * 16 * 17];   // one padded 16 x 16 transpose buffer per wave
    spd_tile2_body<8, GP>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel2, tbuf2, gp);
}

// 9 ... 12 tiles per dimension: one wave per SIMD (VGPRs + AGPRs), two matrices per CU
template <int NT, bool GP>
__global__ __launch_bounds__(128, 1) void matinv_spd_tile2w_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                               unsigned batch, int *work_count, int *work_list, Spd2Gp<double> gp)
{
    __shared__ __attribute__((aligned(16))) double panel2[2 * 16 * NT * 4];
    __shared__ __attribute__((aligned(16))) double tbuf2[2 * 16 * 17];
    spd_tile2_body<NT, GP>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel2, tbuf2, gp);
}

// 12 x 12 tiles on THREE wavefronts (r04: 176 < n <= 192): 26 lower tiles per wave in 30 slots = 240 accumulator registers, one wave per
// SIMD, one matrix per CU with a SIMD to spare -- against 12 waves sweeping all 144 tiles (tile4_impl.hpp) before. (Two waves would need
// 39 slots: 324 registers spilled with AGPR-form MFMAs, a hipcc crash with VGPR-form ones.)
// (Measured and not kept: three waves also at 9 x 9 / 10 x 10 tiles, two waves per SIMD (256 registers, 6 / 107 spilled): Cholesky 130^2
// 6.06e6 inv/s against 6.01e6 on two waves, 160^2 2.96e6 against 4.35e6, the fused pipeline 7 - 13 % slower.)
// (FOUR waves of the same body, 24 slots each: 2.07e6 inv/s at 192^2 against 1.88e6 -- and NaNs; not pursued. Two workgroups of four
// waves per CU: 4 225 registers spilled.)
template <int NT, bool GP>
__global__ __launch_bounds__(192, NT <= 10 ? 2 : 1) void matinv_spd_tile3w_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                               unsigned batch, int *work_count, int *work_list, Spd2Gp<double> gp)
{
    __shared__ __attribute__((aligned(16))) double panel2[2 * 16 * NT * 4];
    __shared__ __attribute__((aligned(16))) double tbuf2[3 * 16 * 17];
    spd_tile2_body<NT, GP, 3>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel2, tbuf2, gp);
}
template <int NT>
hipError_t enqueue_spd_tile3w(bool gp_mode, int n, BatchRef<const double> A, BatchRef<double> X, unsigned grid, unsigned batch, int *info,
                              int *ws, Spd2Gp<double> gp, hipStream_t stream);
template <>
hipError_t enqueue_spd_tile3w<12>(bool, int, BatchRef<const double>, BatchRef<double>, unsigned, unsigned, int *, int *, Spd2Gp<double>, hipStream_t);

// the launch of the NT x NT-tile kernel (NT = 9 ... 11), defined in spd_tile2w<NT>_kernels.hip
template <int NT>
hipError_t enqueue_spd_tile2w(bool gp_mode, int n, BatchRef<const double> A, BatchRef<double> X, unsigned grid, unsigned batch, int *info,
                              int *ws, Spd2Gp<double> gp, hipStream_t stream);
template <>
hipError_t enqueue_spd_tile2w<9>(bool, int, BatchRef<const double>, BatchRef<double>, unsigned, unsigned, int *, int *, Spd2Gp<double>, hipStream_t);
template <>
hipError_t enqueue_spd_tile2w<10>(bool, int, BatchRef<const double>, BatchRef<double>, unsigned, unsigned, int *, int *, Spd2Gp<double>, hipStream_t);
template <>
hipError_t enqueue_spd_tile2w<11>(bool, int, BatchRef<const double>, BatchRef<double>, unsigned, unsigned, int *, int *, Spd2Gp<double>, hipStream_t);

}  // namespace matinv
