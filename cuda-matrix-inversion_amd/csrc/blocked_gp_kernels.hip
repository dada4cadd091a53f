// blocked_gp_kernels.hip -- fused Gaussian-process scalars for LARGE n (beyond what fits in LDS; bins 512 / 1024 of
// BASELINE configs[4]):   mean = a^T (B + diag c)^-1 d,   var = e - a^T (B + diag c)^-1 a.
//
// Blocked right-looking Cholesky on a global-memory working copy, with the two vectors carried as two extra ROWS of the
// matrix (bordered trick: row n = a^T, row n+1 = d^T; after the factorisation they hold (L^-1 a)^T and (L^-1 d)^T, and the
// answer is their dot product -- no inverse, no separate triangular solves). Per panel of PB = 32 columns, TWO launches
// over the whole batch:
//   matinv_bgp_panel   one workgroup per matrix: factor the PB x PB diagonal block in LDS, then solve the column panel
//                      below it (all remaining rows incl. the two border rows) against it;
//   matinv_bgp_update  one workgroup per 64 x 64 tile of the trailing lower triangle (and of the border rows):
//                      W[I,J] -= L[I,K] L[J,K]^T, both panels staged in LDS, 4 x 4 outputs per thread.
// so a batch of few large matrices still fills the chip (the GLOBAL family runs one workgroup per matrix). The launch
// count is 2 n / PB per batch -- this replaces the reference's 4N+1 launches per batch (src/inverse_cholesky_gpu.cu:
// 323-354) plus its two gemmBatched calls (src/gauss_bench.cu:210,232) for the sizes its design note calls "512, 1024"
// (README.md:41-44).
#include "common.hpp"

namespace matinv {

constexpr int BGP_PB = 32;     // panel width
constexpr int BGP_TILE = 64;   // update tile edge
constexpr int BGP_THREADS = 256;

template <class T>
__device__ __forceinline__ T bgp_sqrt(T v);
template <>
__device__ __forceinline__ double bgp_sqrt<double>(double v) { return sqrt(v); }
template <>
__device__ __forceinline__ float bgp_sqrt<float>(float v) { return sqrtf(v); }

// working copy layout per item: (n + 2) rows x n columns, COLUMN-major with leading dimension ld = n + 2:
// element (r, c) at c*ld + r; rows n and n+1 are the border rows a^T and d^T (d = a for the variance).
template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_init(const T *As, const T *Bs, const T *Cs, const T *Ds, T *W,
                                                               int n, int *status)
{
    const size_t item = blockIdx.y;
    const int ld = n + 2;
    T *w = W + item * (size_t)ld * n;
    const T *B = Bs + item * (size_t)n * n;
    for (size_t e = (size_t)blockIdx.x * BGP_THREADS + threadIdx.x; e < (size_t)ld * n; e += (size_t)gridDim.x * BGP_THREADS) {
        const int c = (int)(e / ld), r = (int)(e - (size_t)c * ld);
        T v;
        if (r < n) v = (r >= c) ? B[(size_t)c * n + r] + ((r == c) ? Cs[item * n + c] : (T)0) : (T)0;  // lower triangle only
        else if (r == n) v = As[item * n + c];
        else v = Ds ? Ds[item * n + c] : As[item * n + c];
        w[e] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) status[item] = 0;
}

template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_panel(T *W, int n, int k0, int *status)
{
    __shared__ T L11[BGP_PB][BGP_PB + 1];
    const size_t item = blockIdx.x;
    const int ld = n + 2, pb = (n - k0 < BGP_PB) ? n - k0 : BGP_PB, t = threadIdx.x;
    T *w = W + item * (size_t)ld * n;
    if (status[item] != 0) return;  // an earlier panel found a non-positive pivot
    for (int e = t; e < BGP_PB * BGP_PB; e += BGP_THREADS) {
        const int c = e / BGP_PB, r = e - c * BGP_PB;
        L11[r][c] = (r < pb && c < pb) ? w[(size_t)(k0 + c) * ld + k0 + r] : (T)(r == c);  // ragged last panel: identity padding
    }
    __syncthreads();
    int bad = 0;
    for (int k = 0; k < pb; ++k) {  // unblocked Cholesky of the diagonal block, in LDS
        const T d = L11[k][k];
        if (!(d > 0)) { bad = k0 + k + 1; break; }  // block-uniform
        const T sd = bgp_sqrt<T>(d), rs = (T)1 / sd;
        __syncthreads();
        for (int i = k + t; i < pb; i += BGP_THREADS) L11[i][k] = (i == k) ? sd : L11[i][k] * rs;
        __syncthreads();
        for (int e = t; e < (pb - k - 1) * (pb - k - 1); e += BGP_THREADS) {
            const int j = k + 1 + e / (pb - k - 1), i = k + 1 + e % (pb - k - 1);
            if (i >= j) L11[i][j] -= L11[i][k] * L11[j][k];
        }
        __syncthreads();
    }
    if (bad) {
        if (t == 0) status[item] = bad;
        return;
    }
    for (int e = t; e < pb * pb; e += BGP_THREADS) {
        const int c = e / pb, r = e - c * pb;
        if (r >= c) w[(size_t)(k0 + c) * ld + k0 + r] = L11[r][c];
    }
    // rows below the block (incl. the two border rows): x L11^T = row. Right-looking substitution: once x[c] is final it is
    // eliminated from all later entries, so the dependent chain is PB long and the inner updates are independent FMAs.
    __syncthreads();
    if (t < pb) L11[t][t] = (T)1 / L11[t][t];  // reciprocal diagonal (the factor itself is already written back)
    __syncthreads();
    for (int r = k0 + pb + t; r < ld; r += BGP_THREADS) {
        T x[BGP_PB];
#pragma unroll
        for (int c = 0; c < BGP_PB; ++c) x[c] = (c < pb) ? w[(size_t)(k0 + c) * ld + r] : (T)0;
#pragma unroll
        for (int c = 0; c < BGP_PB; ++c) {
            x[c] *= L11[c][c];
#pragma unroll
            for (int j = c + 1; j < BGP_PB; ++j) x[j] -= x[c] * L11[j][c];
        }
#pragma unroll
        for (int c = 0; c < BGP_PB; ++c)
            if (c < pb) w[(size_t)(k0 + c) * ld + r] = x[c];
    }
}

// trailing update: W[I, J] -= L[I, K] L[J, K]^T for 64 x 64 tiles with J >= k0 + pb, I >= J (rows up to n + 1)
template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_update(T *W, int n, int k0, const int *status)
{
    __shared__ T Li[BGP_PB][BGP_TILE + 1], Lj[BGP_PB][BGP_TILE + 1];
    const size_t item = blockIdx.z;
    if (status[item] != 0) return;
    const int ld = n + 2, pb = (n - k0 < BGP_PB) ? n - k0 : BGP_PB;
    const int j0 = k0 + pb + blockIdx.x * BGP_TILE, i0 = k0 + pb + blockIdx.y * BGP_TILE;
    if (j0 >= n || i0 >= ld || i0 + BGP_TILE <= j0) return;  // outside, or strictly above the diagonal
    T *w = W + item * (size_t)ld * n;
    const int t = threadIdx.x;
    for (int e = t; e < pb * BGP_TILE; e += BGP_THREADS) {
        const int k = e / BGP_TILE, r = e - k * BGP_TILE;
        Li[k][r] = (i0 + r < ld) ? w[(size_t)(k0 + k) * ld + i0 + r] : (T)0;
        Lj[k][r] = (j0 + r < n) ? w[(size_t)(k0 + k) * ld + j0 + r] : (T)0;
    }
    __syncthreads();
    const int ti = (t & 15) * 4, tj = (t >> 4) * 4;  // 4 x 4 outputs per thread: rows i0+ti.., columns j0+tj..
    T acc[4][4] = {};
    for (int k = 0; k < pb; ++k) {
        T a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = Li[k][ti + u]; b[u] = Lj[k][tj + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = i0 + ti + u, c = j0 + tj + v;
            if (r < ld && c < n && r >= c) w[(size_t)c * ld + r] -= acc[u][v];
        }
}

template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_finish(const T *W, const T *Es, T *out, int *info, int n,
                                                                 const int *status)
{
    __shared__ T part[BGP_THREADS / 64];
    const size_t item = blockIdx.x;
    const int ld = n + 2, t = threadIdx.x;
    const T *w = W + item * (size_t)ld * n;
    const int bad = status[item];
    T s = 0;
    if (!bad)
        for (int c = t; c < n; c += BGP_THREADS) s += w[(size_t)c * ld + n] * w[(size_t)c * ld + n + 1];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off);
    if ((t & 63) == 0) part[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
        T q = 0;
        for (int i = 0; i < BGP_THREADS / 64; ++i) q += part[i];
        out[item] = bad ? nan_of<T>() : (Es ? Es[item] - q : q);
        if (info) info[item] = bad;
    }
}

template <class T>
hipError_t launch_gp_blocked(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                             int *info, hipStream_t stream)
{
    if (n < 1 || n > 4096) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;  // grid.y / grid.z limit; callers split larger batches
    const int ld = n + 2;
    T *W = nullptr;
    int *status = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&W), batch * (size_t)ld * n * sizeof(T), stream);
    if (e != hipSuccess) return e;
    e = hipMallocAsync(reinterpret_cast<void **>(&status), batch * sizeof(int), stream);
    if (e != hipSuccess) { (void)hipFreeAsync(W, stream); return e; }
    hipLaunchKernelGGL(matinv_bgp_init<T>, dim3(64, (unsigned)batch), dim3(BGP_THREADS), 0, stream, As, Bs, Cs, Ds, W, n, status);
    for (int k0 = 0; k0 < n; k0 += BGP_PB) {
        hipLaunchKernelGGL(matinv_bgp_panel<T>, dim3((unsigned)batch), dim3(BGP_THREADS), 0, stream, W, n, k0, status);
        const int rem_cols = n - (k0 + BGP_PB), rem_rows = ld - (k0 + BGP_PB);
        if (rem_cols > 0) {
            const unsigned gx = (rem_cols + BGP_TILE - 1) / BGP_TILE, gy = (rem_rows + BGP_TILE - 1) / BGP_TILE;
            hipLaunchKernelGGL(matinv_bgp_update<T>, dim3(gx, gy, (unsigned)batch), dim3(BGP_THREADS), 0, stream, W, n, k0, status);
        }
    }
    hipLaunchKernelGGL(matinv_bgp_finish<T>, dim3((unsigned)batch), dim3(BGP_THREADS), 0, stream, W, Ds ? nullptr : Es, out, info,
                       n, status);
    e = hipGetLastError();
    hipError_t e2 = hipFreeAsync(W, stream), e3 = hipFreeAsync(status, stream);
    return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
}
template hipError_t launch_gp_blocked<double>(int, const double *, const double *, const double *, const double *,
                                              const double *, double *, size_t, int *, hipStream_t);
template hipError_t launch_gp_blocked<float>(int, const float *, const float *, const float *, const float *, const float *,
                                             float *, size_t, int *, hipStream_t);

}  // namespace matinv
