// global_kernels.hip -- kernel family "GLOBAL": any n up to 1024 (the reference's own limit: one thread per row,
// /root/reference/src/gauss/batched_invert.cu:87-93), for sizes whose matrix no longer fits on chip (n > 137 in f64,
// n > 200 in f32). One 1024-thread workgroup per matrix; the n x n working copy lives in the OUTPUT buffer (global memory,
// L2 / Infinity-Cache resident for the sizes in question) and only the pivot row and multiplier column of the current
// step are staged in LDS. Same arithmetic as the LDS family:
//   matinv_gj_global    in-place Gauss-Jordan with partial pivoting,
//   matinv_chol_global  Cholesky factor, in-place triangular inverse, L^-T L^-1 (SPD input, lower triangle read),
//   matinv_gp_global    fused mean / variance through the Cholesky factor and two forward substitutions.
// It is a functional path (2 n^3 * sizeof(T) bytes of cache traffic per matrix), not a tuned one: the blocked, MFMA-based
// large-n path is listed as next in DESIGN.md.
#include "common.hpp"

namespace matinv {

constexpr int GL_THREADS = 1024;

template <class T>
__device__ __forceinline__ T gl_abs(T v) { return v < 0 ? -v : v; }
template <class T>
__device__ __forceinline__ T gl_sqrt(T v);
template <>
__device__ __forceinline__ double gl_sqrt<double>(double v) { return sqrt(v); }
template <>
__device__ __forceinline__ float gl_sqrt<float>(float v) { return sqrtf(v); }

template <class T>
__device__ __forceinline__ void gl_fill_nan(T *X, int n)
{
    for (size_t e = threadIdx.x; e < (size_t)n * n; e += GL_THREADS) X[e] = nan_of<T>();
}

// block-wide arg-max of (val, idx); lowest index wins ties. Result broadcast to every thread.
template <class T>
__device__ __forceinline__ void gl_argmax(T &best, int &bi, T *s_val, int *s_idx)
{
    for (int off = 32; off >= 1; off >>= 1) {
        T ob = __shfl_down(best, off);
        int oi = __shfl_down(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_val[w] = best; s_idx[w] = bi; }
    __syncthreads();
    best = s_val[0];
    bi = s_idx[0];
    for (int k = 1; k < GL_THREADS / 64; ++k) {
        T ob = s_val[k];
        int oi = s_idx[k];
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __syncthreads();
}

template <class T>
__global__ __launch_bounds__(GL_THREADS) void matinv_gj_global(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n)
{
    __shared__ T prow[1024], mcol[1024];
    __shared__ T s_val[GL_THREADS / 64];
    __shared__ int s_idx[GL_THREADS / 64];
    __shared__ int piv[1024];
    const size_t k_mat = blockIdx.x;
    const T *A = Ain.at_uniform(k_mat);
    T *X = Xout.at_uniform(k_mat);
    const int t = threadIdx.x;
    const size_t nn = (size_t)n * n;

    if (A != X)
        for (size_t e = t; e < nn; e += GL_THREADS) X[e] = A[e];
    __syncthreads();

    for (int k = 0; k < n; ++k) {
        T best = (T)-1;
        int bi = k;
        for (int i = k + t; i < n; i += GL_THREADS) {
            const T v = gl_abs(X[(size_t)k * n + i]);
            if (v > best) { best = v; bi = i; }
        }
        gl_argmax(best, bi, s_val, s_idx);
        const int p = bi;
        if (!(best > 0)) {
            if (info && t == 0) info[k_mat] = k + 1;
            __syncthreads();
            gl_fill_nan(X, n);
            return;
        }
        for (int c = t; c < n; c += GL_THREADS) {  // swap rows k <-> p, lift the pivot row
            const T vp = X[(size_t)c * n + p], vk = X[(size_t)c * n + k];
            X[(size_t)c * n + p] = vk;
            prow[c] = vp;
        }
        if (t == 0) piv[k] = p;
        __syncthreads();
        const T inv = (T)1 / prow[k];
        for (int i = t; i < n; i += GL_THREADS) mcol[i] = (i == k) ? (T)0 : X[(size_t)k * n + i];
        __syncthreads();
        for (size_t e = t; e < nn; e += GL_THREADS) {
            const int c = (int)(e / n), r = (int)(e - (size_t)c * n);
            const T pr = (c == k) ? inv : prow[c] * inv;
            T v;
            if (r == k) v = pr;
            else if (c == k) v = -mcol[r] * inv;
            else v = X[e] - mcol[r] * pr;
            X[e] = v;
        }
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; --k) {  // undo the row swaps as column swaps, last first
        const int p = piv[k];
        if (p != k)
            for (int i = t; i < n; i += GL_THREADS) {
                const T a = X[(size_t)k * n + i];
                X[(size_t)k * n + i] = X[(size_t)p * n + i];
                X[(size_t)p * n + i] = a;
            }
        __syncthreads();
    }
    if (info && t == 0) info[k_mat] = 0;
}

// Cholesky pieces on a global-memory matrix W (lower triangle significant)
template <class T>
__device__ __forceinline__ int gl_chol_factor(T *W, int n, T *col)
{
    const int t = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        const T d = W[(size_t)k * n + k];
        if (!(d > 0)) return k + 1;  // block-uniform (read after the barrier below / the initial one)
        const T sd = gl_sqrt<T>(d), rs = (T)1 / sd;
        __syncthreads();
        for (int i = k + t; i < n; i += GL_THREADS) {
            const T v = (i == k) ? sd : W[(size_t)k * n + i] * rs;
            W[(size_t)k * n + i] = v;
            col[i] = v;
        }
        __syncthreads();
        const size_t m = (size_t)(n - k - 1);
        for (size_t e = t; e < m * m; e += GL_THREADS) {  // trailing lower triangle: (i, j), j >= k+1, i >= j
            const int j = k + 1 + (int)(e / m), i = k + 1 + (int)(e % m);
            if (i >= j) W[(size_t)j * n + i] -= col[i] * col[j];
        }
        __syncthreads();
    }
    return 0;
}

template <class T>
__global__ __launch_bounds__(GL_THREADS) void matinv_chol_global(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n,
                                                                 T *workspace)
{
    __shared__ T col[1024];
    const size_t k_mat = blockIdx.x;
    const T *A = Ain.at_uniform(k_mat);
    T *X = Xout.at_uniform(k_mat);
    T *W = workspace + k_mat * (size_t)n * n;  // factor / triangular inverse (X receives the product)
    const int t = threadIdx.x;
    const size_t nn = (size_t)n * n;
    for (size_t e = t; e < nn; e += GL_THREADS) W[e] = A[e];
    __syncthreads();
    const int bad = gl_chol_factor(W, n, col);
    if (bad) {
        if (info && t == 0) info[k_mat] = bad;
        gl_fill_nan(X, n);
        return;
    }
    for (int j = n - 1; j >= 0; --j) {  // L <- L^-1 in place, last column first
        const T ajj = (T)1 / W[(size_t)j * n + j];
        for (int i = j + 1 + t; i < n; i += GL_THREADS) col[i] = W[(size_t)j * n + i];
        __syncthreads();
        for (int i = j + 1 + t; i < n; i += GL_THREADS) {
            T s = 0;
            for (int k = j + 1; k <= i; ++k) s += W[(size_t)k * n + i] * col[k];
            W[(size_t)j * n + i] = -s * ajj;
        }
        if (t == 0) W[(size_t)j * n + j] = ajj;
        __syncthreads();
    }
    for (size_t e = t; e < nn; e += GL_THREADS) {  // X = L^-T L^-1
        const int c = (int)(e / n), r = (int)(e - (size_t)c * n);
        T s = 0;
        for (int k = (r > c ? r : c); k < n; ++k) s += W[(size_t)r * n + k] * W[(size_t)c * n + k];
        X[e] = s;
    }
    if (info && t == 0) info[k_mat] = 0;
}

template <class T>
__global__ __launch_bounds__(GL_THREADS) void matinv_gp_global(const T *As, const T *Bs, const T *Cs, const T *Ds,
                                                               const T *Es, T *out, int *info, int n, T *workspace)
{
    __shared__ T col[1024], u[1024], w[1024];
    __shared__ T s_part[GL_THREADS / 64];
    const size_t k_mat = blockIdx.x;
    const T *B = Bs + k_mat * (size_t)n * n;
    T *W = workspace + k_mat * (size_t)n * n;
    const int t = threadIdx.x;
    const bool variance = (Ds == nullptr);
    const size_t nn = (size_t)n * n;
    for (size_t e = t; e < nn; e += GL_THREADS) W[e] = B[e];
    for (int i = t; i < n; i += GL_THREADS) {
        u[i] = As[k_mat * n + i];
        w[i] = variance ? (T)0 : Ds[k_mat * n + i];
    }
    __syncthreads();
    for (int i = t; i < n; i += GL_THREADS) W[(size_t)i * n + i] += Cs[k_mat * n + i];
    __syncthreads();
    const int bad = gl_chol_factor(W, n, col);
    if (bad) {
        if (info && t == 0) info[k_mat] = bad;
        if (t == 0) out[k_mat] = nan_of<T>();
        return;
    }
    for (int k = 0; k < n; ++k) {  // forward substitution of both right-hand sides, column oriented
        const T rk = (T)1 / W[(size_t)k * n + k];
        const T uk = u[k] * rk, wk = w[k] * rk;
        __syncthreads();
        if (t == 0) { u[k] = uk; w[k] = wk; }
        for (int i = k + 1 + t; i < n; i += GL_THREADS) {
            const T l = W[(size_t)k * n + i];
            u[i] -= l * uk;
            w[i] -= l * wk;
        }
        __syncthreads();
    }
    T part = 0;
    for (int i = t; i < n; i += GL_THREADS) part += u[i] * (variance ? u[i] : w[i]);
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off);
    if ((t & 63) == 0) s_part[t >> 6] = part;
    __syncthreads();
    if (t == 0) {
        T q = 0;
        for (int i = 0; i < GL_THREADS / 64; ++i) q += s_part[i];
        out[k_mat] = variance ? Es[k_mat] - q : q;
        if (info) info[k_mat] = 0;
    }
}

template <class T>
bool global_family_supports(int n) { return n >= 1 && n <= 1024; }
template bool global_family_supports<double>(int);
template bool global_family_supports<float>(int);

template <class T>
hipError_t launch_gj_global(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!global_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(matinv_gj_global<T>, dim3((unsigned)batch), dim3(GL_THREADS), 0, stream, A, X, info, n);
    return hipGetLastError();
}

template <class T>
hipError_t launch_chol_global(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!global_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    T *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), batch * (size_t)n * n * sizeof(T), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_chol_global<T>, dim3((unsigned)batch), dim3(GL_THREADS), 0, stream, A, X, info, n, ws);
    e = hipGetLastError();
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

template <class T>
hipError_t launch_gp_global(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                            int *info, hipStream_t stream)
{
    if (!global_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    T *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), batch * (size_t)n * n * sizeof(T), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(matinv_gp_global<T>, dim3((unsigned)batch), dim3(GL_THREADS), 0, stream, As, Bs, Cs, Ds, Es, out, info,
                       n, ws);
    e = hipGetLastError();
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

#define INST(T)                                                                                                        \
    template hipError_t launch_gj_global<T>(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t);         \
    template hipError_t launch_chol_global<T>(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t);       \
    template hipError_t launch_gp_global<T>(int, const T *, const T *, const T *, const T *, const T *, T *, size_t,  \
                                            int *, hipStream_t);
INST(double)
INST(float)
#undef INST

const char *name_gj_global(bool f64) { return f64 ? "matinv_gj_global<double>" : "matinv_gj_global<float>"; }
const char *name_chol_global(bool f64) { return f64 ? "matinv_chol_global<double>" : "matinv_chol_global<float>"; }

}  // namespace matinv
