// blocked_gj_kernels.hip -- Gauss-Jordan inversion with PARTIAL PIVOTING for LARGE general matrices (beyond what fits in
// LDS, n <= 1024 = the reference's own limit of one thread per row, /root/reference/src/gauss/batched_invert.cu:87-93).
//
// The blocked algorithm of the LDS family (lds_kernels.hip: panel eliminated in registers with thread = row, then a
// rank-PB update of every other column) spread over the chip: per panel of PB = 32 columns TWO launches over the whole batch,
//   matinv_bgj_panel   one workgroup per matrix, thread t <-> row t: per column a workgroup-wide arg-max (DPP inside the
//                      waves, LDS across them; lowest index on ties), the two rows change places through LDS together with
//                      the pivot-row broadcast, every thread eliminates its row in the panel columns. It then publishes
//                      the panel columns G[:, K] of the accumulated transform, the row map of this panel's swaps, and the
//                      old pivot-row entries b = x[K] of every other column;
//   matinv_bgj_update  one workgroup per 64 x 64 tile: x <- x (rows swapped, K entries zeroed) + G[:, K] b, the two
//                      operands staged through LDS, 4 x 4 outputs per thread,
// ping-ponging between two working copies (the swaps are applied as a row gather while reading, so no in-place hazard),
// and one last launch that undoes the row swaps as column swaps while copying into the caller's buffer. 2 n / 32 + 2
// launches replace the reference's 3 n (pivotRow / normalizeRow / transform_matrix, batched_invert.cu:84-95); a batch
// of few large matrices still fills the chip in the update, which is where the 2 n^3 flops are.
// (Measured and not used: 64-column panels for n <= 512, which halve the passes over the working copies. The panel kernel
// then needs 64 panel entries per thread -- 256 VGPRs and scratch in fp64 -- and runs 64 dependent column steps per launch:
// fp64 130^2 8.5e5 -> 6.1e5 inv/s, 256^2 2.4e5 -> 2.3e5, 512^2 3.4e4 -> 3.6e4; fp32 512^2 5.5e4 -> 7.1e4; and the file took
// 3m50s to compile.)
#include <stdlib.h>

#include "slab_mma.hpp"

namespace matinv {

constexpr int BGJ_PB = 32;    // sub-panel: columns eliminated by one launch of the panel kernel
constexpr int BGJ_NB = 128;   // block: columns between two rank-NB MFMA updates of the whole matrix
constexpr int BGJ_TILE = 64;

namespace {
constexpr int GDPP_QUAD_XOR1 = 0xB1, GDPP_QUAD_XOR2 = 0x4E, GDPP_ROW_MIRROR = 0x140, GDPP_ROW_HALF_MIRROR = 0x141;
template <int CTRL>
__device__ __forceinline__ unsigned gdppu(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned bgj_wave_max_u32(unsigned v)
{
    v = max(v, gdppu<GDPP_QUAD_XOR1>(v));
    v = max(v, gdppu<GDPP_QUAD_XOR2>(v));
    v = max(v, gdppu<GDPP_ROW_HALF_MIRROR>(v));
    v = max(v, gdppu<GDPP_ROW_MIRROR>(v));
    const unsigned m0 = __builtin_amdgcn_readlane(v, 0), m1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned m2 = __builtin_amdgcn_readlane(v, 32), m3 = __builtin_amdgcn_readlane(v, 48);
    const unsigned a = m0 > m1 ? m0 : m1, b = m2 > m3 ? m2 : m3;
    return a > b ? a : b;
}
// largest |v| among the active lanes and the lowest lane attaining it (exact: the bit pattern of a non-negative IEEE number
// orders like the number)
__device__ __forceinline__ int bgj_wave_argmax_abs(double v, bool active, double *best)
{
    const unsigned long long bits = active ? ((unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull) : 0ull;
    const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
    const unsigned mhi = bgj_wave_max_u32(hi);
    const unsigned mlo = bgj_wave_max_u32(hi == mhi ? lo : 0u);
    const unsigned long long vote = __ballot(active && hi == mhi && lo == mlo);
    *best = __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
    return vote ? (int)__builtin_ctzll(vote) : 0;
}
__device__ __forceinline__ int bgj_wave_argmax_abs(float v, bool active, float *best)
{
    const unsigned key = active ? (__float_as_uint(v) & 0x7fffffffu) : 0u;
    const unsigned mx = bgj_wave_max_u32(key);
    const unsigned long long vote = __ballot(active && key == mx);
    *best = __uint_as_float(mx);
    return vote ? (int)__builtin_ctzll(vote) : 0;
}
}  // namespace

template <class T>
__global__ __launch_bounds__(256) void matinv_bgj_init(BatchRef<const T> Ain, size_t first, T *W, int n, int *status)
{
    const size_t item = blockIdx.y;
    T *w = W + item * (size_t)n * n;
    const T *A = Ain.at(first + item);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)n * n; e += (size_t)gridDim.x * 256) w[e] = A[e];
    if (blockIdx.x == 0 && threadIdx.x == 0) status[item] = 0;
}

// ---- single-level scheme for n < 384 (r01): one 32-column panel, then the rank-32 update of every other column from VALU
// register tiles; the panel kernel publishes the pivot rows of ALL columns itself. With so few columns per matrix the extra
// launches of the two-level scheme below cost more than its traffic saves (n = 256: 2.4e5 inv/s against 1.9e5).
// blockDim = n rounded up to a multiple of 64 (<= 1024)
template <class T>
__global__ __launch_bounds__(1024) void matinv_bgj_panel1(const T *Win, T *Wout, T *Bbuf, int *rowsrc, int *pivots, int n, int k0,
                                                         int *status)
{
    __shared__ T s_val[16];
    __shared__ int s_idx[16];
    __shared__ T xch[2 * BGJ_PB];
    __shared__ int rs[1024];
    const size_t item = blockIdx.x;
    if (status[item] != 0) return;
    const T *win = Win + item * (size_t)n * n;
    T *wout = Wout + item * (size_t)n * n;
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6, nwaves = blockDim.x >> 6;
    const int pb = (n - k0 < BGJ_PB) ? n - k0 : BGJ_PB;

    T x[BGJ_PB];
#pragma unroll
    for (int c = 0; c < BGJ_PB; ++c) x[c] = (t < n && c < pb) ? win[(size_t)(k0 + c) * n + t] : (T)0;
    rs[t] = t;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BGJ_PB; ++j) {
        if (j < pb) {  // block-uniform
            const int k = k0 + j;
            T best;
            const int bi = (t & ~63) + bgj_wave_argmax_abs(x[j], t >= k && t < n, &best);
            if (tx == 0) { s_val[ty] = best; s_idx[ty] = bi; }
            __syncthreads();
            best = s_val[0];
            int p = s_idx[0];
            for (int w = 1; w < nwaves; ++w) {
                const T ob = s_val[w];
                const int oi = s_idx[w];
                if (ob > best) { best = ob; p = oi; }  // equal maxima: the lower wave (lower rows) keeps it
            }
            if (!(best > 0) || best > max_finite<T>()) {  // zero, NaN or infinite column: no usable pivot (block-uniform)
                if (t == 0) status[item] = k + 1;
                return;
            }
            if (t == 0) {
                pivots[item * (size_t)n + k] = p;
                const int u = rs[k];
                rs[k] = rs[p];
                rs[p] = u;
            }
            // [0] = the pivot row (row p; also when p == k) ALREADY scaled by 1 / pivot, with the reciprocal itself in column j
            // (r03: every thread used to scale it again, a third of the panel's multiplies); [1] = old row k
            if (t == p) {
                const T pvp = (T)1 / x[j];
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) xch[c] = (c == j) ? pvp : x[c] * pvp;
            }
            if (t == k && p != k) {
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) xch[BGJ_PB + c] = x[c];
            }
            __syncthreads();
            if (t == p && p != k) {
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) x[c] = xch[BGJ_PB + c];
            }
            if (t == k) {
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) x[c] = xch[c];
            } else if (t < n) {
                const T m = x[j];
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) x[c] = (c == j) ? -m * xch[j] : fma(-m, xch[c], x[c]);
            }
        }
    }
    __syncthreads();  // rs is final
    if (t < n) {
#pragma unroll
        for (int c = 0; c < BGJ_PB; ++c)
            if (c < pb) wout[(size_t)(k0 + c) * n + t] = x[c];
        rowsrc[item * (size_t)n + t] = rs[t];
        // old pivot-row entries (after this panel's swaps) of the other columns: thread t <-> column t
        if (t < k0 || t >= k0 + pb) {
            const T *col = win + (size_t)t * n;
            T *b = Bbuf + item * (size_t)BGJ_PB * n;
            for (int j = 0; j < pb; ++j) b[(size_t)j * n + t] = col[rs[k0 + j]];
        }
    }
}

// x_new[row][col] = (row in K ? 0 : x_old[rowsrc[row]][col]) + sum_k G[row][k] b[k][col]   for the columns outside K
template <class T>
__global__ __launch_bounds__(256) void matinv_bgj_update1(const T *Win, T *Wout, const T *Bbuf, const int *rowsrc, int n, int k0,
                                                         const int *status, unsigned g, unsigned nb)
{
    const XcdTile tile = xcd_tile_of(blockIdx.x, g, g, nb);  // XCD-aware tile order (slab_mma.hpp)
    if (!tile.valid) return;
    __shared__ T Gt[BGJ_PB][BGJ_TILE + 1], Bt[BGJ_PB][BGJ_TILE + 1];
    __shared__ int rsrc[BGJ_TILE];
    const size_t item = tile.z;
    if (status[item] != 0) return;
    const int pb = (n - k0 < BGJ_PB) ? n - k0 : BGJ_PB;
    const int j0 = tile.x * BGJ_TILE, i0 = tile.y * BGJ_TILE;
    const T *win = Win + item * (size_t)n * n;
    T *wout = Wout + item * (size_t)n * n;
    const T *b = Bbuf + item * (size_t)BGJ_PB * n;
    const int t = threadIdx.x;
    for (int e = t; e < BGJ_PB * BGJ_TILE; e += 256) {
        const int k = e / BGJ_TILE, r = e - k * BGJ_TILE;
        Gt[k][r] = (k < pb && i0 + r < n) ? wout[(size_t)(k0 + k) * n + i0 + r] : (T)0;
        Bt[k][r] = (k < pb && j0 + r < n) ? b[(size_t)k * n + j0 + r] : (T)0;
    }
    if (t < BGJ_TILE) rsrc[t] = (i0 + t < n) ? rowsrc[item * (size_t)n + i0 + t] : 0;
    __syncthreads();
    const int ti = (t & 15) * 4, tj = (t >> 4) * 4;
    T acc[4][4] = {};
    // (Tried and measured, not kept: this rank-32 product through slab_mma.hpp, the LDS-staged MFMA form that made the blocked
    // Cholesky update 1.5 x faster. Here every launch reads and writes the whole matrix for 32 multiply-adds per element, and the
    // MFMA accumulator layout stores 128-byte segments of four different columns per instruction where this thread mapping
    // stores 512 contiguous bytes: 200^2 x 5000 fp64 17.7 ms against 15.3, 256^2 x 3051 14.2 against 12.7.)
#pragma unroll 4  // (fully unrolled, hipcc hoists all 256 LDS reads and spills)
    for (int k = 0; k < BGJ_PB; ++k) {
        T a[4], bb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = Gt[k][ti + u]; bb[u] = Bt[k][tj + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], bb[v], acc[u][v]);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int col = j0 + tj + v;
        if (col >= n || (col >= k0 && col < k0 + pb)) continue;  // the panel columns were written by matinv_bgj_panel
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = i0 + ti + u;
            if (row < n) {
                const T old = (row >= k0 && row < k0 + pb) ? (T)0 : win[(size_t)col * n + rsrc[ti + u]];
                wout[(size_t)col * n + row] = old + acc[u][v];
            }
        }
    }
}

// ---- one block column of BGJ_NB = 128 columns -----------------------------------------------------------------------------
// The block lives in its own ping-pong buffers (ld = n, up to 128 columns per item); its BGJ_PB-wide sub-panels are
// eliminated one after the other, each followed by the rank-32 update of the OTHER COLUMNS OF THE BLOCK only. Everything
// outside the block is deferred to one rank-128 update per block (matinv_bgj_update_mfma): in exact arithmetic the 128
// elementary steps equal one block step x_new[:, J] = Z(x_perm[:, J]) + G x_perm[K, J] with G = the finished block columns,
// x_perm = the ORIGINAL columns with the block's row swaps applied and Z = zeroing of the rows K -- so the big update needs
// only the composite row map `comp` of the block and reads each outside element once per 128 columns instead of once per 32.
//
// blockDim = n rounded up to a multiple of 64 (<= 1024). Pin / Pout: block buffers of this item (column c of the block at
// + c * n); c0 = first column of the sub-panel inside the block, K0 = global index of the block's first column.
template <class T>
__global__ __launch_bounds__(1024) void matinv_bgj_panel(const T *Pin, size_t in_stride, T *Pout, size_t out_stride, T *Bin,
                                                         int *rowsrc, const int *comp_prev, int *comp_next, int *pivots, int n,
                                                         int bw, int c0, int K0, int *status)
{
    __shared__ T s_val[16];
    __shared__ int s_idx[16];
    __shared__ T xch[2 * BGJ_PB];
    __shared__ int rs[1024];
    const size_t item = blockIdx.x;
    if (status[item] != 0) return;
    const T *pin = Pin + item * in_stride;
    T *pout = Pout + item * out_stride;
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6, nwaves = blockDim.x >> 6;
    const int pb = (bw - c0 < BGJ_PB) ? bw - c0 : BGJ_PB;
    const int kg0 = K0 + c0;  // global index of the sub-panel's first pivot

    T x[BGJ_PB];
#pragma unroll
    for (int c = 0; c < BGJ_PB; ++c) x[c] = (t < n && c < pb) ? pin[(size_t)(c0 + c) * n + t] : (T)0;
    rs[t] = t;
    if (t < 16) { s_val[t] = (T)-1; s_idx[t] = 0; }  // slots of waves that do not exist never win the scan below
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BGJ_PB; ++j) {
        if (j < pb) {  // block-uniform
            const int k = kg0 + j;
            T best;
            const int bi = (t & ~63) + bgj_wave_argmax_abs(x[j], t >= k && t < n, &best);
            if (tx == 0) { s_val[ty] = best; s_idx[ty] = bi; }
            __syncthreads();
            // scan of the per-wave maxima, four slots per trip so that their LDS reads are in flight together (one slot per trip cost
            // 15 dependent LDS latencies per pivot at n = 1024)
            best = (T)-1;
            int p = 0;
            for (int w0 = 0; w0 < nwaves; w0 += 4) {
                T ob[4];
                int oi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ob[u] = s_val[w0 + u]; oi[u] = s_idx[w0 + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ob[u] > best) { best = ob[u]; p = oi[u]; }  // equal maxima: the lower wave (lower rows) keeps it
            }
            if (!(best > 0) || best > max_finite<T>()) {  // zero, NaN or infinite column: no usable pivot (block-uniform)
                if (t == 0) status[item] = k + 1;
                return;
            }
            if (t == 0) {
                pivots[item * (size_t)n + k] = p;
                const int u = rs[k];
                rs[k] = rs[p];
                rs[p] = u;
            }
            if (t == k || t == p) {
                T *dst = xch + (t == p ? 0 : BGJ_PB);  // [0] = the pivot row (row p; also when p == k), [1] = old row k
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) dst[c] = x[c];
            }
            __syncthreads();
            const T pv = (T)1 / xch[j];
            if (t == p && p != k) {
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) x[c] = xch[BGJ_PB + c];
            }
            if (t == k) {
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) x[c] = (c == j) ? pv : xch[c] * pv;
            } else if (t < n) {
                const T m = x[j];
#pragma unroll
                for (int c = 0; c < BGJ_PB; ++c) x[c] = (c == j) ? -m * pv : fma(-m, xch[c] * pv, x[c]);
            }
        }
    }
    __syncthreads();  // rs is final
    if (t < n) {
#pragma unroll
        for (int c = 0; c < BGJ_PB; ++c)
            if (c < pb) pout[(size_t)(c0 + c) * n + t] = x[c];
        rowsrc[item * (size_t)n + t] = rs[t];
        // composite row map of the block so far: row t now holds what sat in row comp[t] when the block started
        comp_next[item * (size_t)n + t] = comp_prev ? comp_prev[item * (size_t)n + rs[t]] : rs[t];
        // old pivot-row entries (after this sub-panel's swaps) of the other columns OF THE BLOCK: thread t <-> block column t
        if (t < bw && (t < c0 || t >= c0 + pb)) {
            const T *col = pin + (size_t)t * n;
            T *b = Bin + item * (size_t)BGJ_PB * BGJ_NB;
            T v[BGJ_PB];  // all loads in flight at once (a run-time loop issued them one memory latency after the other)
#pragma unroll
            for (int j = 0; j < BGJ_PB; ++j) v[j] = col[rs[kg0 + (j < pb ? j : 0)]];
#pragma unroll
            for (int j = 0; j < BGJ_PB; ++j)
                if (j < pb) b[(size_t)j * BGJ_NB + t] = v[j];
        }
    }
}

// B operand of the block-level update: Bfull[k][col] = x_old[col][comp[K0 + k]] (the block's pivot rows as they stood when
// the block started, in pivot order). 64 columns per workgroup through LDS: consecutive threads read consecutive k (comp is
// the identity except for the swapped rows: nearly contiguous) and write consecutive columns.
constexpr int BGJ_PRW = 32;  // columns per workgroup of matinv_bgj_pivot_rows: 33 KB of LDS in fp64 (64 columns = 66 KB left two
                             // workgroups per CU and the kernel at 2.5 TB/s)
template <class T>
__global__ __launch_bounds__(256) void matinv_bgj_pivot_rows(const T *Win, size_t in_stride, T *Bfull, const int *comp, int n, int bw, int K0,
                                                             const int *status)
{
    __shared__ T tile[BGJ_PRW][BGJ_NB + 1];
    __shared__ int ksrc[BGJ_NB];
    const size_t item = blockIdx.y;
    if (status[item] != 0) return;
    const int c0 = blockIdx.x * BGJ_PRW, t = threadIdx.x;
    const T *win = Win + item * in_stride;
    T *bf = Bfull + item * (size_t)BGJ_NB * n;
    if (t < BGJ_NB) ksrc[t] = (t < bw) ? comp[item * (size_t)n + K0 + t] : 0;
    __syncthreads();
    for (int e = t; e < BGJ_PRW * BGJ_NB; e += 256) {
        const int k = e % BGJ_NB, cc = e / BGJ_NB;
        if (k < bw && c0 + cc < n) tile[cc][k] = win[(size_t)(c0 + cc) * n + ksrc[k]];
    }
    __syncthreads();
    for (int e = t; e < BGJ_PRW * BGJ_NB; e += 256) {
        const int cc = e % BGJ_PRW, k = e / BGJ_PRW;
        if (k < bw && c0 + cc < n) bf[(size_t)k * n + c0 + cc] = tile[cc][k];
    }
}

// The rank-kw update on the matrix cores, used at both levels:
//   x_new[row][col] = (Z0 <= row < Z0 + kw ? 0 : x_old[rmap[row]][col]) + sum_{k < kw} G[row][k] x_old[rmap[Z0 + k]][col]
// * block level (INNER = false): x_old / x_new = the two working copies (all n columns), G = the finished block columns,
//   rmap = the block's composite row map, kw = bw <= 128, Z0 = K0; the block's own columns [K0, K0 + bw) are copied from G;
// * inside a block (INNER = true): x_old / x_new = the block buffers (ncols = bw columns), G = the sub-panel just eliminated
//   (columns [c0, c0 + pb) of x_new), rmap = that sub-panel's row map, kw = pb <= 32, Z0 = K0 + c0; the sub-panel's own
//   columns are left alone (matinv_bgj_panel wrote them).
// One workgroup per 64 x 64 tile, four wavefronts of 32 x 32, 16x16x4 MFMA tiles. The buffers are column-major, so the MFMA
// computes the TRANSPOSED tile (its rows <-> matrix columns, its columns <-> matrix rows): the 16 lanes of a C/D row group
// then own 16 consecutive matrix rows of one column -- 128-byte segments for the gather of x_old and for the store. G and
// the pivot rows (prepared k-major by matinv_bgj_panel / matinv_bgj_pivot_rows: gathering them here, per tile, from the
// column-major x_old doubled the time of this kernel) go through LDS in slabs of 32.
// MTC x MTR = 16 x 16 MFMA tiles per wavefront (columns x rows): 2 x 2 -> 64 x 64 per workgroup, the only one instantiated since
// r03 (4 x 4 = 128 x 128 and 2 x 4 = 64 columns x 128 rows halve the operand traffic per flop and were slower: see below). The slabs are double-buffered through
// registers: the next slab's global loads are in flight while the matrix cores work on the current one.
// Slab depth and occupancy (r03, measured on general fp64 input): the kernel is latency-bound -- each workgroup alternates between
// waiting for a slab and 4 MFMAs per k-step, and only OTHER workgroups on the CU fill its gaps -- so thin slabs that leave room for
// more workgroups win: 32-deep slabs / 3 waves per SIMD (130 VGPRs, 33 KB LDS) 2.60e5 inv/s at 256^2, 8.16e3 at 1024^2;
// 16 / 5 (94 VGPRs): 3.17e5, 9.95e3; 8 / 6 (76 VGPRs, 8 KB): 3.26e5, 1.03e4; 8 / 8 (64 VGPRs, 40 B scratch): 3.00e5, 8.95e3.
// A 64 x 128 workgroup tile (twice the flops per operand byte, 2 waves per SIMD) LOST 13 %. Two / three slabs in flight instead of one
// (rotating register slots, 80 / 96 VGPRs): 1024^2 1.08e4 -> 8.3e3 / 8.0e3, 256^2 3.5e5 -> 2.9e5 / 2.8e5 -- also lost. The fp32 128 x 128 tile: 16-deep slabs
// 7.6e4 / 1.40e4 inv/s at 512^2 / 1024^2, 8-deep at 2..3 waves per SIMD 8.3e4 / 1.54e4.
#ifndef MATINV_BGJ_KS
#define MATINV_BGJ_KS 8
#endif
#ifndef MATINV_BGJ_OCC
#define MATINV_BGJ_OCC 6
#endif
template <class T, bool INNER, int MTC, int MTR = MTC>
__global__ __launch_bounds__(256, MTC * MTR <= 4 ? MATINV_BGJ_OCC : 2) void matinv_bgj_update_mfma(const T *Xold, size_t old_stride, T *Xnew, size_t new_stride,
                                                              const T *Gsrc, size_t g_stride, const T *Bsrc, size_t b_stride, int ldb,
                                                              const int *rmap, int n, int ncols, int kw, int Z0, int skip0,
                                                              const int *status, unsigned gx, unsigned gy, unsigned nb,
                                                              const int *colmap = nullptr, T *const *xtable = nullptr)
{
    // colmap != nullptr: the LAST block-level update of an inversion writes the result itself -- column col of the working copy is
    // column colmap[col] of the inverse (matinv_bgj_colmap) -- into the caller's buffers (xtable: pointer table, else Xnew / new_stride)
    const XcdTile tile = xcd_tile_of(blockIdx.x, gx, gy, nb);  // tiles of one tile row share their G slab: one XCD (slab_mma.hpp)
    if (!tile.valid) return;
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    // wave tile: 16 MTC columns x 16 MTR rows; workgroup tile (2 x 2 waves): TSC columns x TSR rows
    constexpr int WTC = 16 * MTC, WTR = 16 * MTR, TSC = 2 * WTC, TSR = 2 * WTR;
    constexpr int KS = MATINV_BGJ_KS;  // slab depth
    constexpr int PERG = KS * TSR / 256, PERB = KS * TSC / 256;  // slab elements per thread: G (rows), pivot rows (columns)
    // fp32: row stride + 16, so that the four k-groups of an MFMA operand read (rows k .. k+3, 16 consecutive elements each)
    // land in disjoint LDS banks (1024^2 x 256: 19.6 -> 18.5 ms). fp64 keeps the plain stride: the padded slabs (41 KB) cost a
    // workgroup per CU, which outweighs the conflicts there (31.3 ms unpadded, 31.8 padded).
    constexpr int PAD = sizeof(T) == 8 ? 0 : 16;
    __shared__ T Gt[KS][TSR + PAD], Bt[KS][TSC + PAD];
    __shared__ int rsrc[TSR];
    const size_t item = tile.z;
    if (status[item] != 0) return;
    const int j0 = tile.x * TSC, i0 = tile.y * TSR;  // first column / row of the tile
    const T *xold = Xold + item * old_stride;
    T *xnew = (colmap && xtable) ? xtable[item] : Xnew + item * new_stride;
    const int *cmap = colmap ? colmap + item * (size_t)n : nullptr;
    const T *g = Gsrc + item * g_stride;
    const T *bsrc = Bsrc + item * b_stride;  // [k][column], ld = ldb: the pivot rows x_old[rmap[Z0 + k]][.] in pivot order
    const int *rm = rmap + item * (size_t)n;
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int q = l >> 4, c = l & 15;
    const int wc = (wv & 1) * WTC, wr = (wv >> 1) * WTR;  // this wave's columns x rows inside the tile
    if (t < TSR) rsrc[t] = (i0 + t < n) ? rm[i0 + t] : 0;
    // tiles whose columns all belong to the skipped range only copy (block level) or have nothing to do (inner)
    const bool all_skipped = j0 >= skip0 && j0 + TSC <= skip0 + kw;
    if (INNER && all_skipped) return;
    vec4 acc[MTC][MTR] = {};
    if (!all_skipped) {
        T gq[PERG], bq[PERB];
        // Slab element i of thread t is (k = t / TS + i * 256 / TS, r = t % TS): the row / column r is the same for every i, so the
        // addresses are one base pointer + multiples of a stride and the edge tests are hoisted (r03: the kernel issued 9.3 vector-ALU
        // instructions per MFMA, nearly all of them 64-bit address arithmetic and predicates, PMC on 1024^2). Only a ragged LAST slab
        // (kw not a multiple of KS) clamps k.
        constexpr int KRG = 256 / TSR, KRB = 256 / TSC;
        const int rG = t % TSR, kG = t / TSR, rB = t % TSC, kB = t / TSC;
        const bool g_in = i0 + rG < n, b_in = j0 + rB < ncols;
        const T *gcol = g + (g_in ? i0 + rG : n - 1);
        const T *bcol = bsrc + (b_in ? j0 + rB : ncols - 1);
        const size_t gstep = (size_t)KRG * n, bstep = (size_t)KRB * ldb;
        // The prefetch is RAW (addresses clamped into the operands, so the loads are unconditional); padding is zeroed when
        // the slab is staged. Written as `in ? load : 0` the select sits right behind the loads and the wave waits for them
        // before the MFMAs they are meant to hide behind (the same change made the blocked Cholesky update 1.3 x faster).
        auto fetch = [&](int ks) {
            if (ks + KS <= kw) {  // block-uniform
                const T *gp = gcol + (size_t)(ks + kG) * n;
#pragma unroll
                for (int i = 0; i < PERG; ++i, gp += gstep) gq[i] = *gp;
                const T *bp = bcol + (size_t)(ks + kB) * ldb;
#pragma unroll
                for (int i = 0; i < PERB; ++i, bp += bstep) bq[i] = *bp;
            } else {
#pragma unroll
                for (int i = 0; i < PERG; ++i) {
                    const int k = ks + kG + i * KRG;
                    gq[i] = gcol[(size_t)(k < kw ? k : kw - 1) * n];
                }
#pragma unroll
                for (int i = 0; i < PERB; ++i) {
                    const int k = ks + kB + i * KRB;
                    bq[i] = bcol[(size_t)(k < kw ? k : kw - 1) * ldb];
                }
            }
        };
        fetch(0);
        for (int ks = 0; ks < kw; ks += KS) {
            __syncthreads();  // the previous slab has been consumed (and rsrc is visible after the first one)
            const int kleft = kw - ks;  // >= KS except in a ragged last slab
#pragma unroll
            for (int i = 0; i < PERG; ++i) Gt[kG + i * KRG][rG] = (g_in && kG + i * KRG < kleft) ? gq[i] : (T)0;
#pragma unroll
            for (int i = 0; i < PERB; ++i) Bt[kB + i * KRB][rB] = (b_in && kB + i * KRB < kleft) ? bq[i] : (T)0;
            __syncthreads();
            if (ks + KS < kw) fetch(ks + KS);
#pragma unroll
            for (int kk = 0; kk < KS / 4; ++kk) {
                T a[MTC], b[MTR];
#pragma unroll
                for (int u = 0; u < MTC; ++u) a[u] = Bt[4 * kk + q][wc + 16 * u + c];  // MFMA rows  <-> matrix columns
#pragma unroll
                for (int v = 0; v < MTR; ++v) b[v] = Gt[4 * kk + q][wr + 16 * v + c];  // MFMA columns <-> matrix rows
#pragma unroll
                for (int u = 0; u < MTC; ++u)
#pragma unroll
                    for (int v = 0; v < MTR; ++v) acc[u][v] = G::mfma(a[u], b[v], acc[u][v]);
            }
        }
    } else {
        __syncthreads();
    }
    // epilogue: lane (q, c) owns rows i0 + wr + 16 v + c of columns j0 + wc + 16 u + trow(r, q); per column one base offset
    int rs_[MTR];
    bool zr[MTR];
#pragma unroll
    for (int v = 0; v < MTR; ++v) {
        const int lrow = wr + 16 * v + c, row = i0 + lrow;
        rs_[v] = rsrc[lrow];
        zr[v] = row >= Z0 && row < Z0 + kw;
    }
    const int row0 = i0 + wr + c;
#pragma unroll
    for (int u = 0; u < MTC; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = j0 + wc + 16 * u + G::trow(r, q);
            if (col >= ncols) continue;
            const bool skipped = col >= skip0 && col < skip0 + kw;
            if (INNER && skipped) continue;
            const size_t cb = (size_t)col * n;
            const T *xo = xold + cb;
            T *xn = xnew + (cmap ? (size_t)cmap[col] * n : cb) + row0;
            const T *gc = g + (size_t)(skipped ? col - skip0 : 0) * n + row0;
#pragma unroll
            for (int v = 0; v < MTR; ++v) {
                if (row0 + 16 * v >= n) continue;
                T out;
                if (skipped) out = gc[16 * v];  // the finished block columns
                else out = (zr[v] ? (T)0 : xo[rs_[v]]) + acc[u][v][r];
                xn[16 * v] = out;
            }
        }
}

// X[:, j] = W[:, src(j)], src(j) = the pivot swaps applied to the index j in forward order (the row swaps of P A become
// column swaps of (P A)^-1 in reverse order; following one index through them forwards is the same map)
template <class T>
__global__ __launch_bounds__(256) void matinv_bgj_finish(const T *W, BatchRef<T> Xout, size_t first, const int *pivots, int *info,
                                                         int n, const int *status)
{
    __shared__ int piv[1024];
    __shared__ int src[BGJ_TILE];
    const size_t item = blockIdx.y;
    const int j0 = blockIdx.x * BGJ_TILE, t = threadIdx.x;
    const T *w = W + item * (size_t)n * n;
    T *X = Xout.at(first + item);
    const int bad = status[item];
    if (!bad) {
        for (int i = t; i < n; i += 256) piv[i] = pivots[item * (size_t)n + i];
        __syncthreads();
        if (t < BGJ_TILE && j0 + t < n) {
            int idx = j0 + t;
            for (int k = 0; k < n; ++k) {
                const int p = piv[k];
                idx = (idx == k) ? p : ((idx == p) ? k : idx);
            }
            src[t] = idx;
        }
        __syncthreads();
    }
    for (int c = t >> 6; c < BGJ_TILE && j0 + c < n; c += 4) {
        const T *colsrc = bad ? nullptr : w + (size_t)src[c] * n;
        T *dst = X + (size_t)(j0 + c) * n;
        for (int r = t & 63; r < n; r += 64) dst[r] = bad ? nan_of<T>() : colsrc[r];
    }
    if (info && blockIdx.x == 0 && t == 0) info[first + item] = bad;
}

// The column permutation of matinv_bgj_finish folded into the last block-level update (r03: the separate copy was 7 % of a 256^2
// inversion): colmap[c] = the column of the inverse that column c of the working copy becomes = the pivot swaps applied to c in
// REVERSE order (the inverse of src above). Also takes over the finish kernel's other duties: info, and NaN for items without a
// usable pivot (the update kernels skip those). blockDim = n rounded up to a multiple of 64.
template <class T>
__global__ __launch_bounds__(1024) void matinv_bgj_colmap(const int *pivots, int *colmap, BatchRef<T> Xout, size_t first, int *info, int n,
                                                          const int *status)
{
    __shared__ int piv[1024];
    const size_t item = blockIdx.x;
    const int t = threadIdx.x, bad = status[item];
    if (info && t == 0) info[first + item] = bad;
    if (bad) {
        T *X = Xout.at(first + item);
        for (size_t e = t; e < (size_t)n * n; e += blockDim.x) X[e] = nan_of<T>();
        return;
    }
    if (t < n) piv[t] = pivots[item * (size_t)n + t];
    __syncthreads();
    if (t < n) {
        int idx = t;
        for (int k = n - 1; k >= 0; --k) {
            const int p = piv[k];
            idx = (idx == k) ? p : ((idx == p) ? k : idx);
        }
        colmap[item * (size_t)n + t] = idx;
    }
}

// workspace cap of the blocked paths (bytes); MATINV_BLOCKED_WS_MB overrides the 16 GiB default (tests use it to force chunking)
size_t blocked_workspace_cap()
{
    static const size_t cap = []() {
        const char *s = getenv("MATINV_BLOCKED_WS_MB");
        const long mb = s && *s ? atol(s) : 0;
        return mb > 0 ? (size_t)mb << 20 : (size_t)16 << 30;  // 288 GB of HBM: 16 GiB of scratch is small change
    }();
    return cap;
}

bool blocked_gj_supports(int n) { return n >= 1 && n <= 1024; }

// smallest n that takes the two-level scheme (MFMA updates); below it: one level, rank-32 vector-ALU updates. MATINV_BGJ_TWO_LEVEL_MIN
// overrides (A/B switch). r02: 384. r03: 224 at first (general input, f64: 256^2 2.25e5 -> 2.55e5 inv/s, 320^2 1.18e5 -> 1.49e5; f32
// 320^2 1.80e5 -> 2.00e5; 200^2 equal), then 160 once the MFMA update ran at six waves per SIMD: 193^2 / 200^2 / 216^2
// 3.25e5 / 3.15e5 / 2.88e5 -> 4.11e5 / 4.15e5 / 3.83e5
int blocked_gj_two_level_min()
{
    static const int v = []() {
        const char *s = getenv("MATINV_BGJ_TWO_LEVEL_MIN");
        const int e = s && *s ? atoi(s) : 0;
        return e > 0 ? e : 160;
    }();
    return v;
}

template <class T>
static hipError_t launch_gj_blocked_small(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    const size_t per_item = (2 * (size_t)n * n + (size_t)BGJ_PB * n) * sizeof(T);
    size_t chunk = blocked_workspace_cap() / per_item;  // bounded workspace, grid.y / grid.z limit
    if (chunk < 1) chunk = 1;
    if (chunk > 65535) chunk = 65535;
    if (chunk > batch) chunk = batch;
    T *ws = nullptr;
    int *iws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), chunk * per_item, stream);
    if (e != hipSuccess) return e;
    e = scratch_alloc(reinterpret_cast<void **>(&iws), chunk * (2 * (size_t)n + 1) * sizeof(int), stream);
    if (e != hipSuccess) { (void)scratch_free(ws, stream); return e; }
    T *W0 = ws, *W1 = ws + chunk * (size_t)n * n, *Bbuf = W1 + chunk * (size_t)n * n;
    int *rowsrc = iws, *pivots = iws + chunk * (size_t)n, *status = pivots + chunk * (size_t)n;
    const unsigned threads = (unsigned)((n + 63) / 64 * 64);
    const unsigned g = (unsigned)((n + BGJ_TILE - 1) / BGJ_TILE);
    for (size_t first = 0; first < batch; first += chunk) {
        const unsigned b = (unsigned)((batch - first < chunk) ? batch - first : chunk);
        hipLaunchKernelGGL(matinv_bgj_init<T>, dim3(64, b), dim3(256), 0, stream, A, first, W0, n, status);
        T *cur = W0, *nxt = W1;
        for (int k0 = 0; k0 < n; k0 += BGJ_PB) {
            hipLaunchKernelGGL(matinv_bgj_panel1<T>, dim3(b), dim3(threads), 0, stream, cur, nxt, Bbuf, rowsrc, pivots, n, k0, status);
            hipLaunchKernelGGL(matinv_bgj_update1<T>, dim3(xcd_tile_grid(g, g, b)), dim3(256), 0, stream, cur, nxt, Bbuf, rowsrc, n, k0, status, g, b);
            T *tmp = cur;
            cur = nxt;
            nxt = tmp;
        }
        hipLaunchKernelGGL(matinv_bgj_finish<T>, dim3(g, b), dim3(256), 0, stream, cur, X, first, pivots, info, n, status);
    }
    e = hipGetLastError();
    hipError_t e2 = scratch_free(ws, stream), e3 = scratch_free(iws, stream);
    return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
}

template <class T>
hipError_t launch_gj_blocked(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!blocked_gj_supports(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    if (n < blocked_gj_two_level_min()) return launch_gj_blocked_small<T>(n, A, X, batch, info, stream);
    // per item: two working copies, two block buffers (n x NB), the block's pivot rows (NB x n), a sub-panel's b strip
    const size_t per_item = (2 * (size_t)n * n + 3 * (size_t)BGJ_NB * n + (size_t)BGJ_PB * BGJ_NB) * sizeof(T);
    size_t chunk = blocked_workspace_cap() / per_item;  // bounded workspace, grid.y / grid.z limit
    if (chunk < 1) chunk = 1;
    if (chunk > 65535) chunk = 65535;
    if (chunk > batch) chunk = batch;
    T *ws = nullptr;
    int *iws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), chunk * per_item, stream);
    if (e != hipSuccess) return e;
    e = scratch_alloc(reinterpret_cast<void **>(&iws), chunk * (4 * (size_t)n + 1) * sizeof(int), stream);
    if (e != hipSuccess) { (void)scratch_free(ws, stream); return e; }
    const size_t nn = (size_t)n * n, blk = (size_t)BGJ_NB * n, strip = (size_t)BGJ_PB * BGJ_NB;
    T *W0 = ws, *W1 = W0 + chunk * nn, *P0 = W1 + chunk * nn, *P1 = P0 + chunk * blk, *Bfull = P1 + chunk * blk,
      *Bin = Bfull + chunk * blk;
    int *rowsrc = iws, *comp0 = rowsrc + chunk * (size_t)n, *comp1 = comp0 + chunk * (size_t)n, *pivots = comp1 + chunk * (size_t)n,
        *status = pivots + chunk * (size_t)n;
    const unsigned threads = (unsigned)((n + 63) / 64 * 64);
    const unsigned g = (unsigned)((n + BGJ_TILE - 1) / BGJ_TILE);
    // (128 x 128 workgroup tiles were used for fp32 until r03; with 8-deep slabs the 64 x 64 tile at six waves per SIMD is faster
    // there too: 1.65e4 against 1.54e4 inv/s at 1024^2, 8.9e4 against 8.3e4 at 512^2)
    // columns per block (a multiple of the sub-panel width, <= BGJ_NB): MATINV_BGJ_NB overrides (A/B switch)
    static const int nb_env = []() {
        const char *s = getenv("MATINV_BGJ_NB");
        const int v = s && *s ? atoi(s) : 0;
        return (v >= BGJ_PB && v <= BGJ_NB && v % BGJ_PB == 0) ? v : 0;
    }();
    const int nbw = nb_env ? nb_env : BGJ_NB;
    for (size_t first = 0; first < batch; first += chunk) {
        const unsigned b = (unsigned)((batch - first < chunk) ? batch - first : chunk);
        // A flat batch (one matrix after the other, the usual case) is read in place by the first block's kernels; only a pointer
        // table needs the gathering copy (r03: the copy was 6 % of the time at 256^2)
        const T *cur;
        size_t cur_stride = nn;
        T *nxt = W1;
        if (!A.table && A.stride >= nn) {
            cur = A.base + first * A.stride;
            cur_stride = A.stride;
            hipError_t em = hipMemsetAsync(status, 0, b * sizeof(int), stream);
            if (em != hipSuccess) { (void)scratch_free(ws, stream); (void)scratch_free(iws, stream); return em; }
        } else {
            hipLaunchKernelGGL(matinv_bgj_init<T>, dim3(64, b), dim3(256), 0, stream, A, first, W0, n, status);
            cur = W0;
        }
        bool fused_finish = false;
        for (int K0 = 0; K0 < n; K0 += nbw) {
            const int bw = n - K0 < nbw ? n - K0 : nbw;
            // the block's sub-panels, inside the block buffers: X0 = the block columns of cur, then P0, P1, P0, ...
            const T *pin = cur + (size_t)K0 * n;
            size_t in_stride = cur_stride;
            T *pout = P0;
            const int *cprev = nullptr;
            int *cnext = comp0;
            for (int c0 = 0; c0 < bw; c0 += BGJ_PB) {
                const int pb = bw - c0 < BGJ_PB ? bw - c0 : BGJ_PB;
                hipLaunchKernelGGL(matinv_bgj_panel<T>, dim3(b), dim3(threads), 0, stream, pin, in_stride, pout, blk, Bin, rowsrc, cprev,
                                   cnext, pivots, n, bw, c0, K0, status);
                if (bw > pb) {
                    hipLaunchKernelGGL((matinv_bgj_update_mfma<T, true, 2>), dim3(xcd_tile_grid((bw + 63) / 64, g, b)), dim3(256), 0, stream, pin,
                                       in_stride, pout, blk, pout + (size_t)c0 * n, blk, Bin, strip, BGJ_NB, rowsrc, n, bw, pb, K0 + c0,
                                       c0, status, (unsigned)((bw + 63) / 64), g, b);
                }
                pin = pout, in_stride = blk;
                pout = (pout == P0) ? P1 : P0;
                cprev = cnext;
                cnext = (cnext == comp0) ? comp1 : comp0;
            }
            // pin = the finished block columns G, cprev = the block's composite row map
            if (bw < n)
                hipLaunchKernelGGL(matinv_bgj_pivot_rows<T>, dim3((n + BGJ_PRW - 1) / BGJ_PRW, b), dim3(256), 0, stream, cur, cur_stride, Bfull,
                                   cprev, n, bw, K0, status);
            // the last of several block-level updates writes the inverse itself (a single block reads the caller's input in place:
            // no fusion there, input and output may be the same buffer)
            if (K0 + bw == n && K0 > 0) {
                fused_finish = true;
                hipLaunchKernelGGL(matinv_bgj_colmap<T>, dim3(b), dim3(threads), 0, stream, pivots, rowsrc, X, first, info, n, status);
                T *xbase = X.table ? nullptr : X.base + first * X.stride;
                hipLaunchKernelGGL((matinv_bgj_update_mfma<T, false, 2>), dim3(xcd_tile_grid(g, g, b)), dim3(256), 0, stream, cur, cur_stride, xbase,
                                   X.stride, pin, blk, Bfull, blk, n, cprev, n, n, bw, K0, K0, status, g, g, b, rowsrc,
                                   X.table ? X.table + first : nullptr);
                break;
            }
            hipLaunchKernelGGL((matinv_bgj_update_mfma<T, false, 2>), dim3(xcd_tile_grid(g, g, b)), dim3(256), 0, stream, cur, cur_stride, nxt, nn, pin,
                               blk, Bfull, blk, n, cprev, n, n, bw, K0, K0, status, g, g, b);
            cur = nxt;
            cur_stride = nn;
            nxt = (nxt == W1) ? W0 : W1;
        }
        if (!fused_finish) hipLaunchKernelGGL(matinv_bgj_finish<T>, dim3(g, b), dim3(256), 0, stream, cur, X, first, pivots, info, n, status);
    }
    e = hipGetLastError();
    hipError_t e2 = scratch_free(ws, stream), e3 = scratch_free(iws, stream);
    return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
}
template hipError_t launch_gj_blocked<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_gj_blocked<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);

}  // namespace matinv
