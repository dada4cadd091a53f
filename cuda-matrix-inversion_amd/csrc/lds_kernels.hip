// lds_kernels.hip -- kernel family "LDS": one 256-thread workgroup per matrix, the whole n x n matrix
// resident in the CU's 160 KiB LDS, read from HBM once and written once. It serves every n up to the LDS
// limit (n <= 137 for f64, n <= 197 for f32) and both algorithms, and is the fallback under the faster
// register-resident families.
//
//   matinv_gj_lds    in-place Gauss-Jordan with partial (row) pivoting. One launch replaces the 3n launches
//                    of pivotRow / normalizeRow / transform_matrix (/root/reference/src/gauss/
//                    batched_invert.cu:17-95) and never materialises the identity half.
//   matinv_chol_lds  Cholesky A = L L^T, in-place inverse of L, A^-1 = L^-T L^-1: the three phases of
//                    /root/reference/src/inverse_cholesky_cpu.c:17-85 and of GPU kernels C4-C7
//                    (src/inverse_cholesky_gpu.cu:251-312), one launch instead of 4N+1.
#include "common.hpp"
#include "chol_block.hpp"

namespace matinv {

constexpr int LDS_THREADS = 256;
constexpr int LDS_LIMIT_BYTES = 160 * 1024;

__host__ __device__ inline int lds_ld(int n) { return n | 1; }  // odd leading dimension: row walks spread over banks

template <class T>
__host__ __device__ inline size_t lds_bytes(int n)
{
    // matrix + 8 staged pivot rows (blocked Gauss-Jordan) + exchange / broadcast rows + int pivots
    return sizeof(T) * ((size_t)n * lds_ld(n) + 8 * (size_t)n + 32) + sizeof(int) * (size_t)n + 64;
}

template <class T>
bool lds_family_supports(int n)
{
    return n >= 1 && lds_bytes<T>(n) <= (size_t)LDS_LIMIT_BYTES;
}
template bool lds_family_supports<double>(int);
template bool lds_family_supports<float>(int);

template <class T>
__device__ __forceinline__ void fill_nan(T *X, int n)
{
    for (int e = threadIdx.x; e < n * n; e += LDS_THREADS) X[e] = nan_of<T>();
}

template <class T>
__device__ __forceinline__ T absval(T v) { return v < 0 ? -v : v; }

// ------------------------------------------------------------------------------------------------
// One matrix, whole workgroup: in-place Gauss-Jordan with PARTIAL PIVOTING, blocked over panels of GJ_PB columns.
//   phase 1  the n x GJ_PB panel lives in registers, thread t <-> row t: per column a workgroup-wide argmax (lowest index on
//            ties, the oracle's rule), the two rows change places between their threads, the pivot row is broadcast through
//            LDS and every thread eliminates its own row -- in the panel columns only. Afterwards the panel columns hold
//            G[:, K], the K columns of the accumulated transform G = G_8 P_8 ... G_1 P_1 (the in-place "inverse part").
//   phase 2  every other column takes the same row swaps, its old pivot entries b = x[K] are set aside, and
//            x <- x (K entries zeroed) + G[:, K] b  is a rank-GJ_PB update from register tiles (16 x 16 thread grid, rows
//            ti + 16u, columns tj + 16v): each matrix element is touched once per PANEL, not once per column.
// Same pivots as the column-by-column algorithm (the panel columns carry all earlier updates when they are searched).
// Block-uniform control flow; ends with every LDS access retired (callers that loop must __syncthreads() before the next
// matrix).
constexpr int GJ_PB = 8;

namespace {
constexpr int LDPP_QUAD_XOR1 = 0xB1, LDPP_QUAD_XOR2 = 0x4E, LDPP_ROW_MIRROR = 0x140, LDPP_ROW_HALF_MIRROR = 0x141;
template <int CTRL>
__device__ __forceinline__ unsigned ldppu(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
// maximum over the 64 lanes, wave-uniform: 4 DPP steps inside the rows of 16, then 4 v_readlane (no LDS traffic, unlike
// the ds_bpermute behind __shfl_down)
__device__ __forceinline__ unsigned lds_wave_max_u32(unsigned v)
{
    v = max(v, ldppu<LDPP_QUAD_XOR1>(v));
    v = max(v, ldppu<LDPP_QUAD_XOR2>(v));
    v = max(v, ldppu<LDPP_ROW_HALF_MIRROR>(v));
    v = max(v, ldppu<LDPP_ROW_MIRROR>(v));
    const unsigned m0 = __builtin_amdgcn_readlane(v, 0), m1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned m2 = __builtin_amdgcn_readlane(v, 32), m3 = __builtin_amdgcn_readlane(v, 48);
    const unsigned a = m0 > m1 ? m0 : m1, b = m2 > m3 ? m2 : m3;
    return a > b ? a : b;
}
// largest |v| among the lanes with `active` and the LOWEST lane attaining it (the bit pattern of a non-negative IEEE number
// orders like the number): exact, two 32-bit rounds for double. Returns the lane (0 when nothing is active), *best = the value.
__device__ __forceinline__ int wave_argmax_abs(double v, bool active, double *best)
{
    const unsigned long long bits = active ? ((unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull) : 0ull;
    const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
    const unsigned mhi = lds_wave_max_u32(hi);
    const unsigned mlo = lds_wave_max_u32(hi == mhi ? lo : 0u);
    const unsigned long long vote = __ballot(active && hi == mhi && lo == mlo);
    *best = __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
    return vote ? (int)__builtin_ctzll(vote) : 0;
}
__device__ __forceinline__ int wave_argmax_abs(float v, bool active, float *best)
{
    const unsigned key = active ? (__float_as_uint(v) & 0x7fffffffu) : 0u;
    const unsigned mx = lds_wave_max_u32(key);
    const unsigned long long vote = __ballot(active && key == mx);
    *best = __uint_as_float(mx);
    return vote ? (int)__builtin_ctzll(vote) : 0;
}
}  // namespace

template <class T>
__device__ __forceinline__ void gj_lds_one(const T *A, T *X, int *info_slot, int n, unsigned char *smem_raw,
                                           T *s_red_val, int *s_red_idx)
{
    const int ld = lds_ld(n);
    T *a = reinterpret_cast<T *>(smem_raw);  // a[c*ld + r]
    T *bbuf = a + (size_t)n * ld;            // [GJ_PB][n]: old pivot-row entries of the non-panel columns
    T *xch = bbuf + GJ_PB * n;               // [2][GJ_PB] row exchange, then [GJ_PB] pivot row broadcast
    T *prow = xch + 2 * GJ_PB;
    int *piv = reinterpret_cast<int *>(prow + GJ_PB);

    const int t = threadIdx.x;
    const int tx = t & 63, ty = t >> 6;

    for (int c = ty; c < n; c += LDS_THREADS / 64)
        for (int r = tx; r < n; r += 64) a[c * ld + r] = A[(size_t)c * n + r];
    __syncthreads();

    for (int k0 = 0; k0 < n; k0 += GJ_PB) {
        const int pb = (n - k0 < GJ_PB) ? n - k0 : GJ_PB;
        // ---- phase 1: the panel, one row per thread
        T x[GJ_PB];
#pragma unroll
        for (int c = 0; c < GJ_PB; ++c) x[c] = (t < n && c < pb) ? a[(k0 + c) * ld + t] : (T)0;
#pragma unroll
        for (int j = 0; j < GJ_PB; ++j) {
            if (j < pb) {  // block-uniform
                const int k = k0 + j;
                T best;
                int bi = (t & ~63) + wave_argmax_abs(x[j], t >= k && t < n, &best);
                if (tx == 0) { s_red_val[ty] = best; s_red_idx[ty] = bi; }
                __syncthreads();
                best = s_red_val[0];
                int p = s_red_idx[0];
#pragma unroll
                for (int w = 1; w < LDS_THREADS / 64; ++w) {
                    T ob = s_red_val[w];
                    int oi = s_red_idx[w];
                    if (ob > best) { best = ob; p = oi; }  // equal maxima: the lower wave (lower rows) keeps it
                }
                if (!(best > 0) || best > max_finite<T>()) {  // zero, NaN or infinite column: no usable pivot (block-uniform)
                    if (info_slot && t == 0) *info_slot = k + 1;
                    fill_nan(X, n);
                    return;
                }
                if (t == 0) piv[k] = p;
                // rows k and p change places through LDS, and row p (the pivot row) is what everybody needs: one barrier
                if (t == k || t == p) {
                    T *dst = xch + (t == p ? 0 : GJ_PB);  // [0] = the pivot row (row p; also when p == k), [1] = old row k
#pragma unroll
                    for (int c = 0; c < GJ_PB; ++c) dst[c] = x[c];
                }
                __syncthreads();
                const T pv = (T)1 / xch[j];
                if (t == p && p != k) {  // takes over what was in row k
#pragma unroll
                    for (int c = 0; c < GJ_PB; ++c) x[c] = xch[GJ_PB + c];
                }
                if (t == k) {  // the pivot row: scaled, its pivot entry 1/pivot
#pragma unroll
                    for (int c = 0; c < GJ_PB; ++c) x[c] = (c == j) ? pv : xch[c] * pv;
                } else if (t < n) {
                    const T m = x[j];
#pragma unroll
                    for (int c = 0; c < GJ_PB; ++c) x[c] = (c == j) ? -m * pv : fma(-m, xch[c] * pv, x[c]);
                }
            }
        }
        if (t < n) {
#pragma unroll
            for (int c = 0; c < GJ_PB; ++c)
                if (c < pb) a[(k0 + c) * ld + t] = x[c];
        }
        __syncthreads();
        // ---- phase 2: the other columns. Row swaps and the old pivot entries first (thread t <-> column t) ...
        if (t < n && (t < k0 || t >= k0 + pb)) {
            T *col = a + t * ld;
            for (int j = 0; j < pb; ++j) {
                const int k = k0 + j, p = piv[k];
                if (p != k) { const T u = col[k]; col[k] = col[p]; col[p] = u; }
            }
            for (int j = 0; j < pb; ++j) bbuf[j * n + t] = col[k0 + j];
        }
        __syncthreads();
        // ... then x <- x (K entries zeroed) + G[:, K] b from register tiles
        {
            const int ti = t & 15, tj = t >> 4;
            for (int ub = 0; 16 * ub < n; ub += 8) {
                T li[8][GJ_PB];
#pragma unroll
                for (int uu = 0; uu < 8; ++uu) {
                    const int row = ti + 16 * (ub + uu);
#pragma unroll
                    for (int c = 0; c < GJ_PB; ++c) li[uu][c] = (row < n && c < pb) ? a[(k0 + c) * ld + row] : (T)0;
                }
                for (int col = tj; col < n; col += 16) {
                    if (col >= k0 && col < k0 + pb) continue;
                    T bj[GJ_PB];
#pragma unroll
                    for (int c = 0; c < GJ_PB; ++c) bj[c] = (c < pb) ? bbuf[c * n + col] : (T)0;
#pragma unroll
                    for (int uu = 0; uu < 8; ++uu) {
                        const int row = ti + 16 * (ub + uu);
                        if (row < n) {
                            T v = (row >= k0 && row < k0 + pb) ? (T)0 : a[col * ld + row];
#pragma unroll
                            for (int c = 0; c < GJ_PB; ++c) v = fma(li[uu][c], bj[c], v);
                            a[col * ld + row] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // undo the row swaps as column swaps in reverse order: X[:, j] = a[:, src[j]]
    int *src = piv;  // the pivot list becomes the composite column source map
    if (t == 0) {
        int *s = reinterpret_cast<int *>(bbuf);  // free now
        for (int j = 0; j < n; ++j) s[j] = j;
        for (int k = n - 1; k >= 0; --k) {
            int p = piv[k];
            int tmp = s[k]; s[k] = s[p]; s[p] = tmp;
        }
        for (int j = 0; j < n; ++j) src[j] = s[j];
    }
    __syncthreads();
    for (int c = ty; c < n; c += LDS_THREADS / 64) {
        const int sc = src[c];
        for (int r = tx; r < n; r += 64) X[(size_t)c * n + r] = a[sc * ld + r];
    }
    if (info_slot && t == 0) *info_slot = 0;
}

template <class T>
__global__ __launch_bounds__(LDS_THREADS) void matinv_gj_lds(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ T s_red_val[LDS_THREADS / 64];
    __shared__ int s_red_idx[LDS_THREADS / 64];
    const size_t k_mat = blockIdx.x;
    gj_lds_one<T>(Ain.at(k_mat), Xout.at(k_mat), info ? info + k_mat : nullptr, n, smem_raw, s_red_val, s_red_idx);
}

// Same algorithm over a device-side work list (indices of matrices a register-resident fast path rejected):
// a fixed small grid strides over work_list[0 .. *work_count). An empty list costs one near-empty launch.
template <class T>
__global__ __launch_bounds__(LDS_THREADS) void matinv_gj_lds_worklist(BatchRef<const T> Ain, BatchRef<T> Xout, int *info,
                                                                      int n, const int *work_count, const int *work_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ T s_red_val[LDS_THREADS / 64];
    __shared__ int s_red_idx[LDS_THREADS / 64];
    const int count = *work_count;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const size_t k_mat = (size_t)work_list[i];
        gj_lds_one<T>(Ain.at(k_mat), Xout.at(k_mat), info ? info + k_mat : nullptr, n, smem_raw, s_red_val, s_red_idx);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T sqrt_of(T v);
template <>
__device__ __forceinline__ double sqrt_of<double>(double v) { return sqrt(v); }
template <>
__device__ __forceinline__ float sqrt_of<float>(float v) { return sqrtf(v); }

// Cholesky building blocks on an LDS-resident matrix a[c*ld + r] (lower triangle significant).
// phase 2: L <- L^-1 in place, last column first (inverseLower, inverse_cholesky_cpu.c:37-58; GPU C6 :286-301)
template <class T>
__device__ __forceinline__ void tri_inverse_lds(T *a, T *vec, int ld, int n)
{
    const int t = threadIdx.x;
    for (int j = n - 1; j >= 0; --j) {
        const T ajj = (T)1 / a[j * ld + j];
        for (int i = j + 1 + t; i < n; i += LDS_THREADS) vec[i] = a[j * ld + i];
        __syncthreads();
        for (int i = j + 1 + t; i < n; i += LDS_THREADS) {
            T s = 0;
            for (int k = j + 1; k <= i; ++k) s += a[k * ld + i] * vec[k];
            a[j * ld + i] = -s * ajj;
        }
        if (t == 0) a[j * ld + j] = ajj;
        __syncthreads();
    }
}

enum { CHOL_PHASE_FACTOR = 1, CHOL_PHASE_TRINV = 2, CHOL_PHASE_MULT = 4, CHOL_PHASE_ALL = 7 };

// `phases` selects which of the three phases run (the reference exports them separately,
// include/inverse_gpu.h:15-24 there). With CHOL_PHASE_MULT the full symmetric product is written;
// otherwise the lower triangle (L or L^-1) with the strict upper triangle zeroed, as the reference's
// decompose kernels do (src/inverse_cholesky_gpu.cu:268-270). Ain may equal Xout (in place).
template <class T>
__device__ __forceinline__ void chol_lds_one(const T *A, T *X, int *info_slot, int n, int phases, unsigned char *smem_raw)
{
    const int ld = lds_ld(n);
    T *a = reinterpret_cast<T *>(smem_raw);  // lower triangle: A, then L, then L^-1
    T *vec = a + (size_t)n * ld;
    const int t = threadIdx.x;
    const int tx = t & 63, ty = t >> 6;

    for (int c = ty; c < n; c += LDS_THREADS / 64)
        for (int r = tx; r < n; r += 64) a[c * ld + r] = A[(size_t)c * n + r];
    __syncthreads();

    if (phases & CHOL_PHASE_FACTOR) {
        const int bad = chol_factor_lds(a, ld, n, n);
        if (bad) {
            if (info_slot && t == 0) *info_slot = bad;
            fill_nan(X, n);
            return;
        }
    }
    if (phases & CHOL_PHASE_TRINV) tri_inverse_lds(a, vec, ld, n);
    if (phases & CHOL_PHASE_MULT) {
        // phase 3: X = L^-T L^-1, X[r][c] = sum_{k >= max(r,c)} Linv[k][r]*Linv[k][c] (inverse, :60-85; GPU C7 :303-312)
        for (int c = ty; c < n; c += LDS_THREADS / 64)
            for (int r = tx; r < n; r += 64) {
                T s = 0;
                for (int k = (r > c ? r : c); k < n; ++k) s += a[r * ld + k] * a[c * ld + k];
                X[(size_t)c * n + r] = s;
            }
    } else {
        for (int c = ty; c < n; c += LDS_THREADS / 64)
            for (int r = tx; r < n; r += 64) X[(size_t)c * n + r] = (r >= c) ? a[c * ld + r] : (T)0;
    }
    if (info_slot && t == 0) *info_slot = 0;
}

template <class T>
__global__ __launch_bounds__(LDS_THREADS) void matinv_chol_lds(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n,
                                                               int phases)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const size_t k_mat = blockIdx.x;
    chol_lds_one<T>(Ain.at(k_mat), Xout.at(k_mat), info ? info + k_mat : nullptr, n, phases, smem_raw);
}

// Full Cholesky inverse over a device-side work list (fallback of the register-resident SPD fast path).
template <class T>
__global__ __launch_bounds__(LDS_THREADS) void matinv_chol_lds_worklist(BatchRef<const T> Ain, BatchRef<T> Xout, int *info,
                                                                        int n, const int *work_count,
                                                                        const int *work_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int count = *work_count;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const size_t k_mat = (size_t)work_list[i];
        chol_lds_one<T>(Ain.at(k_mat), Xout.at(k_mat), info ? info + k_mat : nullptr, n, CHOL_PHASE_ALL, smem_raw);
        __syncthreads();
    }
}

// Fused Gaussian-process scalar: out = u^T (B + diag c)^-1 w  (mean: u=a, w=d)  or  e - a^T (B+diag c)^-1 a.
// Replaces addDiagonal + batched inverse + gemmBatched x2 of /root/reference/src/gauss_bench.cu:127-265,275-409
// and calcluateMeanCPU / calcluateVarianceCPU (src/gauss_cpu.c:41-72,174-206; documented sign, gauss_cpu.h:34).
// With M = L L^T:  u^T M^-1 w = (L^-1 u) . (L^-1 w): one factorisation and two forward substitutions;
// the inverse is never formed.
template <class T>
__device__ __forceinline__ void gp_lds_one(const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out,
                                           int *info, int n, size_t k_mat, unsigned char *smem_raw, T *s_part)
{
    // the vectors travel as border ROWS n (a) and n+1 (d) of the LDS matrix: after the factorisation they hold
    // (L^-1 a)^T and (L^-1 d)^T, so the answer is one dot product -- no separate substitution sweeps
    const bool variance = (Ds == nullptr);
    const int nrows = n + (variance ? 1 : 2);
    const int ld = (n + 2) | 1;  // n * ld fits the LDS budget of lds_bytes() (matrix + 3 vectors)
    T *a = reinterpret_cast<T *>(smem_raw);
    const T *B = Bs + k_mat * (size_t)n * n;
    const int t = threadIdx.x;
    const int tx = t & 63, ty = t >> 6;

    for (int c = ty; c < n; c += LDS_THREADS / 64) {
        const T cc = Cs[k_mat * n + c], uc = As[k_mat * n + c], wc = variance ? (T)0 : Ds[k_mat * n + c];
        for (int r = tx; r < n; r += 64) a[c * ld + r] = B[(size_t)c * n + r] + ((r == c) ? cc : (T)0);  // addDiagonal, gauss_bench.cu:38-43
        if (tx == 0) a[c * ld + n] = uc;
        if (tx == 1 && !variance) a[c * ld + n + 1] = wc;
    }
    __syncthreads();

    const int bad = chol_factor_lds(a, ld, n, nrows);
    if (bad) {
        if (info && t == 0) info[k_mat] = bad;
        if (t == 0) out[k_mat] = nan_of<T>();
        return;
    }
    T part = 0;
    for (int i = t; i < n; i += LDS_THREADS) part += a[i * ld + n] * a[i * ld + (variance ? n : n + 1)];
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off);
    if (tx == 0) s_part[ty] = part;
    __syncthreads();
    if (t == 0) {
        T q = 0;
#pragma unroll
        for (int i = 0; i < LDS_THREADS / 64; ++i) q += s_part[i];
        out[k_mat] = variance ? Es[k_mat] - q : q;
        if (info) info[k_mat] = 0;
    }
}

template <class T>
__global__ __launch_bounds__(LDS_THREADS) void matinv_gp_lds(const T *As, const T *Bs, const T *Cs, const T *Ds,
                                                             const T *Es, T *out, int *info, int n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ T s_part[LDS_THREADS / 64];
    gp_lds_one<T>(As, Bs, Cs, Ds, Es, out, info, n, blockIdx.x, smem_raw, s_part);
}

// the same over a device-side work list (fallback of matinv_gp_tile_f64)
template <class T>
__global__ __launch_bounds__(LDS_THREADS) void matinv_gp_lds_worklist(const T *As, const T *Bs, const T *Cs, const T *Ds,
                                                                      const T *Es, T *out, int *info, int n,
                                                                      const int *work_count, const int *work_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ T s_part[LDS_THREADS / 64];
    const int count = *work_count;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        gp_lds_one<T>(As, Bs, Cs, Ds, Es, out, info, n, (size_t)work_list[i], smem_raw, s_part);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
template <class K>
static hipError_t prepare_lds(K kernel, size_t bytes)
{
    // > 64 KiB of dynamic LDS needs the attribute; set it every time (cheap, and device-agnostic)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// the grid x-dimension limit (2^31-1 blocks) is far above any batch that fits in 288 GB
template <class T>
hipError_t launch_gj_lds(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!lds_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const size_t bytes = lds_bytes<T>(n);
    hipError_t e = prepare_lds(matinv_gj_lds<T>, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_gj_lds<T>, dim3((unsigned)batch), dim3(LDS_THREADS), bytes, stream, A, X, info, n);
    return hipGetLastError();
}
template <class T>
hipError_t launch_gj_lds_worklist(int n, BatchRef<const T> A, BatchRef<T> X, const int *work_count, const int *work_list,
                                  int *info, hipStream_t stream)
{
    if (!lds_family_supports<T>(n)) return hipErrorInvalidValue;
    const size_t bytes = lds_bytes<T>(n);
    hipError_t e = prepare_lds(matinv_gj_lds_worklist<T>, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_gj_lds_worklist<T>, dim3(1024), dim3(LDS_THREADS), bytes, stream, A, X, info, n, work_count,
                       work_list);
    return hipGetLastError();
}
template <class T>
hipError_t launch_chol_lds_worklist(int n, BatchRef<const T> A, BatchRef<T> X, const int *work_count,
                                    const int *work_list, int *info, hipStream_t stream)
{
    if (!lds_family_supports<T>(n)) return hipErrorInvalidValue;
    const size_t bytes = lds_bytes<T>(n);
    hipError_t e = prepare_lds(matinv_chol_lds_worklist<T>, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_chol_lds_worklist<T>, dim3(1024), dim3(LDS_THREADS), bytes, stream, A, X, info, n,
                       work_count, work_list);
    e = hipGetLastError();
    return e != hipSuccess ? e : debug_note_rejects(work_count, stream);
}
template <class T>
hipError_t launch_chol_lds(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                           int phases)
{
    if (!lds_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const size_t bytes = lds_bytes<T>(n);
    hipError_t e = prepare_lds(matinv_chol_lds<T>, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_chol_lds<T>, dim3((unsigned)batch), dim3(LDS_THREADS), bytes, stream, A, X, info, n,
                       phases);
    return hipGetLastError();
}
template <class T>
hipError_t launch_gp_lds(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                         int *info, hipStream_t stream)
{
    if (!lds_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const size_t bytes = lds_bytes<T>(n);
    hipError_t e = prepare_lds(matinv_gp_lds<T>, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_gp_lds<T>, dim3((unsigned)batch), dim3(LDS_THREADS), bytes, stream, As, Bs, Cs, Ds, Es,
                       out, info, n);
    return hipGetLastError();
}
template <class T>
hipError_t launch_gp_lds_worklist(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out,
                                  const int *work_count, const int *work_list, int *info, hipStream_t stream)
{
    if (!lds_family_supports<T>(n)) return hipErrorInvalidValue;
    const size_t bytes = lds_bytes<T>(n);
    hipError_t e = prepare_lds(matinv_gp_lds_worklist<T>, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_gp_lds_worklist<T>, dim3(1024), dim3(LDS_THREADS), bytes, stream, As, Bs, Cs, Ds, Es, out,
                       info, n, work_count, work_list);
    e = hipGetLastError();
    return e != hipSuccess ? e : debug_note_rejects(work_count, stream);
}
#define INST(T)                                                                                                        \
    template hipError_t launch_gj_lds<T>(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t);            \
    template hipError_t launch_gj_lds_worklist<T>(int, BatchRef<const T>, BatchRef<T>, const int *, const int *,      \
                                                  int *, hipStream_t);                                                \
    template hipError_t launch_chol_lds<T>(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t, int);     \
    template hipError_t launch_chol_lds_worklist<T>(int, BatchRef<const T>, BatchRef<T>, const int *, const int *,    \
                                                    int *, hipStream_t);                                              \
    template hipError_t launch_gp_lds<T>(int, const T *, const T *, const T *, const T *, const T *, T *, size_t,     \
                                         int *, hipStream_t);                                                         \
    template hipError_t launch_gp_lds_worklist<T>(int, const T *, const T *, const T *, const T *, const T *, T *,    \
                                                  const int *, const int *, int *, hipStream_t);
INST(double)
INST(float)
#undef INST

const char *name_gj_lds(bool f64) { return f64 ? "matinv_gj_lds<double>" : "matinv_gj_lds<float>"; }
const char *name_chol_lds(bool f64) { return f64 ? "matinv_chol_lds<double>" : "matinv_chol_lds<float>"; }
const char *name_gp_lds(bool f64) { return f64 ? "matinv_gp_lds<double>" : "matinv_gp_lds<float>"; }

}  // namespace matinv
