// lds_kernels.hip -- kernel family "LDS": one 256-thread workgroup per matrix, the whole n x n matrix
// resident in the CU's 160 KiB LDS, read from HBM once and written once. It serves every n up to the LDS
// limit (n <= 141 for f64, n <= 200 for f32) and both algorithms, and is the fallback under the faster
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
    // matrix + multiplier column + pivot row + staging vector + int pivots + reduction scratch
    return sizeof(T) * ((size_t)n * lds_ld(n) + 3 * (size_t)n) + sizeof(int) * (size_t)n + 64;
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
// One matrix, whole workgroup. Block-uniform control flow; ends with every LDS access retired
// (callers that loop must __syncthreads() before the next matrix).
template <class T>
__device__ __forceinline__ void gj_lds_one(const T *A, T *X, int *info_slot, int n, unsigned char *smem_raw,
                                           T *s_red_val, int *s_red_idx)
{
    const int ld = lds_ld(n);
    T *a = reinterpret_cast<T *>(smem_raw);  // a[c*ld + r]
    T *mcol = a + (size_t)n * ld;            // multipliers of the current step
    T *prow = mcol + n;                      // pivot row of the current step
    int *piv = reinterpret_cast<int *>(prow + 2 * n);  // (prow + n .. prow + 2n is spare)

    const int t = threadIdx.x;
    const int tx = t & 63, ty = t >> 6;

    for (int c = ty; c < n; c += LDS_THREADS / 64)
        for (int r = tx; r < n; r += 64) a[c * ld + r] = A[(size_t)c * n + r];
    __syncthreads();

    for (int k = 0; k < n; ++k) {
        // 1. pivot = largest |a[i][k]|, i >= k; lowest index wins ties (same rule as the oracle's scan)
        T best = (T)-1;
        int bi = k;
        for (int i = k + t; i < n; i += LDS_THREADS) {
            T v = absval(a[k * ld + i]);
            if (v > best) { best = v; bi = i; }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            T ob = __shfl_down(best, off);
            int oi = __shfl_down(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tx == 0) { s_red_val[ty] = best; s_red_idx[ty] = bi; }
        __syncthreads();
        best = s_red_val[0];
        int p = s_red_idx[0];
#pragma unroll
        for (int w = 1; w < LDS_THREADS / 64; ++w) {
            T ob = s_red_val[w];
            int oi = s_red_idx[w];
            if (ob > best || (ob == best && oi < p)) { best = ob; p = oi; }
        }
        if (!(best > 0)) {  // zero or NaN column: singular (block-uniform)
            if (info_slot && t == 0) *info_slot = k + 1;
            fill_nan(X, n);
            return;
        }
        // 2. swap rows k <-> p while lifting the pivot row out
        if (t < n) {
            T vp = a[t * ld + p];
            T vk = a[t * ld + k];
            a[t * ld + p] = vk;
            prow[t] = vp;
        }
        if (t == 0) piv[k] = p;
        __syncthreads();
        const T inv = (T)1 / prow[k];
        if (t < n) mcol[t] = (t == k) ? (T)0 : a[k * ld + t];
        __syncthreads();
        // 3. eliminate: row k <- scaled pivot row (its k-th entry 1/pivot); row r <- row r - m_r * row k
        for (int c = ty; c < n; c += LDS_THREADS / 64) {
            const T pr = (c == k) ? inv : prow[c] * inv;
            for (int r = tx; r < n; r += 64) {
                T v;
                if (r == k) v = pr;
                else if (c == k) v = -mcol[r] * inv;
                else v = a[c * ld + r] - mcol[r] * pr;
                a[c * ld + r] = v;
            }
        }
        __syncthreads();
    }
    // 4. undo the row swaps as column swaps in reverse order: X[:, j] = a[:, src[j]]
    int *src = piv;  // the pivot list becomes the composite column source map
    if (t == 0) {
        int *s = reinterpret_cast<int *>(mcol);  // mcol/prow are free now
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
    return hipGetLastError();
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
    return hipGetLastError();
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
