// tile_kernels.hip -- kernel family "TILE": one matrix per wavefront, register-resident in 16x16 fp64 MFMA
// accumulator tiles (v_mfma_f64_16x16x4_f64), for 16 < n <= 64 (n is padded to NT*16 with an identity block).
//
// Why MFMA for an inversion: Gauss-Jordan is a sequence of rank-1 updates whose cost on the VALU is dominated by
// BROADCASTING the multiplier column / pivot row across the wavefront (2 v_readlane per fp64 value per step).
// Blocking 4 elimination steps turns the trailing update into a rank-4 update  W += Aop(n x 4) * Bop(4 x n),
// which is exactly one 16x16x4 MFMA per tile: the matrix core performs the broadcast for free and runs at the
// fp64 vector FMA rate (MI355X: fp64 matrix peak = fp64 vector peak). The VALU is left with the 4-wide panel.
//
// Data layout in registers (C/D layout of v_mfma_f64_16x16x4_f64, guide section 3 "Fragment layout"):
//   lane l = 16*q + c, tile (ti, tj), register r  <->  W[16*ti + 4*r + q][16*tj + c]
// W is the TRANSPOSE of the caller's column-major matrix (W[i][j] = mem[i*n + j]) so the 16 lanes of a row
// group read 128 contiguous bytes; inv(A^T) = inv(A)^T, so storing the result the same way yields inv(A)
// column-major. Consequences that make the blocked step cheap:
//   * Bop for block kb (pivot rows 4kb..4kb+3) is the register acc[kb/4][tj][kb%4] AS IT STANDS (lane group q
//     already holds pivot row 4kb+q) -- no data movement;
//   * the 4 pivot columns live in 16 lanes (c in [4(kb%4), +4)) of tile column kb/4; they are staged through a
//     2 KB LDS buffer to be re-read in the A-operand layout (row per lane).
//
// Blocked in-place Gauss-Jordan step (D = W[K,K], K = 4 pivot indices):
//   Aop[i,:] = -W[i,K] D^-1 (i not in K),   Aop[K,:] = D^-1,
//   W[i,J] <- W[i,J] + Aop[i,:] W[K,J]  (i not in K),   W[K,J] <- Aop[K,:] W[K,J]   for the columns J not in K,
//   W[:,K] <- Aop.
// All of it is ONE MFMA per tile: C = W with the K rows and K columns zeroed, B = W[K,:] with I_4 on the K columns.
//
// Pivoting: this fast path eliminates in natural order and VERIFIES instead of searching: every multiplier it
// forms (the 6 LU multipliers of each 4x4 pivot block and every entry of Aop outside the pivot rows) must be
// <= TAU in magnitude (threshold pivoting acceptance; scale invariant; NaN/Inf fail it). A matrix that fails is
// appended to a device work list and redone, in the same stream, by the partially pivoted LDS kernel
// (lds_kernels.hip) -- no host round trip. Diagonally dominant / SPD batches (the reference's fixtures,
// tests/generate_inverse_matrices.m:12-18) never take the fallback.
//
// Replaces the 3n launches of /root/reference/src/gauss/batched_invert.cu:84-95.
#include "common.hpp"

namespace matinv {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr double TILE_TAU = 4.0;

__device__ __forceinline__ double fast_rcp(double x)
{
    // v_rcp_f64 + two Newton steps: full fp64 accuracy for normal x (no denormal/overflow fix-up needed here:
    // a pivot that small or large fails the TAU acceptance test and the matrix goes to the pivoted fallback)
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

__device__ __forceinline__ bool le_tau(double v) { return __builtin_fabs(v) <= TILE_TAU; }

// FULL: n == 16*NT known at compile time (constant address offsets, no bounds checks).
template <int NT, bool FULL>
__global__ __launch_bounds__(64, (FULL || NT < 3) ? 2 : 1) void matinv_gj_tile_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info,
                                                           int n_rt, unsigned batch, int *work_count, int *work_list)
{
    constexpr int N = 16 * NT;
    const int n = FULL ? N : n_rt;
    __shared__ __attribute__((aligned(16))) double panel[N * 4];  // [row][4 pivot columns]
    const int l = threadIdx.x;
    const int q = l >> 4, c = l & 15;

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const double *A = Ain.at(mat);
        double *X = Xout.at(mat);
        // one per-lane element offset + wave-uniform (compile-time when FULL) tile offsets: keeps the 16*NT*NT
        // addresses out of VGPRs
        const unsigned lane_off = (unsigned)(q * n + c);
        v4d acc[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + q, col = 16 * tj + c;
                    // identity padding beyond n: blockdiag(A, I)^-1 = blockdiag(A^-1, I)
                    const unsigned uoff = (unsigned)((16 * ti + 4 * r) * n + 16 * tj);
                    acc[ti][tj][r] = (FULL || (row < n && col < n)) ? A[uoff + lane_off] : ((row == col) ? 1.0 : 0.0);
                }
        bool ok = true;

#pragma unroll
        for (int kb = 0; kb < 4 * NT; ++kb) {
            const int tK = kb >> 2, rK = kb & 3, c0 = 4 * (kb & 3), K0 = 4 * kb;
            const bool panel_lane = (c >= c0) && (c < c0 + 4);
            const bool diag_lane = panel_lane && (c - c0 == q);  // lane holding I_4's ones in the B operand
            // 1. the 4 pivot columns -> LDS, [row][4]
            if (panel_lane) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) panel[(16 * ti + 4 * r + q) * 4 + (c - c0)] = acc[ti][tK][r];
            }
            __syncthreads();
            // 2. pivot block D (4x4), same for every lane
            double d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = panel[(K0 + i) * 4 + j];
            // 3. LU of D without pivoting (multipliers checked), then column q of D^-1: x = D^-1 e_q
            const double r0 = fast_rcp(d[0][0]);
            const double l10 = d[1][0] * r0, l20 = d[2][0] * r0, l30 = d[3][0] * r0;
            const double u11 = __builtin_fma(-l10, d[0][1], d[1][1]), u12 = __builtin_fma(-l10, d[0][2], d[1][2]),
                         u13 = __builtin_fma(-l10, d[0][3], d[1][3]);
            const double a21 = __builtin_fma(-l20, d[0][1], d[2][1]), a22 = __builtin_fma(-l20, d[0][2], d[2][2]),
                         a23 = __builtin_fma(-l20, d[0][3], d[2][3]);
            const double a31 = __builtin_fma(-l30, d[0][1], d[3][1]), a32 = __builtin_fma(-l30, d[0][2], d[3][2]),
                         a33 = __builtin_fma(-l30, d[0][3], d[3][3]);
            const double r1 = fast_rcp(u11);
            const double l21 = a21 * r1, l31 = a31 * r1;
            const double u22 = __builtin_fma(-l21, u12, a22), u23 = __builtin_fma(-l21, u13, a23);
            const double b32 = __builtin_fma(-l31, u12, a32), b33 = __builtin_fma(-l31, u13, a33);
            const double r2 = fast_rcp(u22);
            const double l32 = b32 * r2;
            const double u33 = __builtin_fma(-l32, u23, b33);
            const double r3 = fast_rcp(u33);
            ok = ok && le_tau(l10) && le_tau(l20) && le_tau(l30) && le_tau(l21) && le_tau(l31) && le_tau(l32) &&
                 (__builtin_fabs(r3) < 1.7e308);
            // L y = e_q
            const double y0 = (q == 0) ? 1.0 : 0.0;
            const double y1 = __builtin_fma(-l10, y0, (q == 1) ? 1.0 : 0.0);
            const double y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, (q == 2) ? 1.0 : 0.0));
            const double y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, y1, __builtin_fma(-l30, y0, (q == 3) ? 1.0 : 0.0)));
            // U x = y
            const double x3 = y3 * r3;
            const double x2 = __builtin_fma(-u23, x3, y2) * r2;
            const double x1 = __builtin_fma(-u13, x3, __builtin_fma(-u12, x2, y1)) * r1;
            const double x0 = __builtin_fma(-d[0][3], x3, __builtin_fma(-d[0][2], x2, __builtin_fma(-d[0][1], x1, y0))) * r0;
            // 4. A operand, lane (q, c) of tile row ti: Aop[16ti + c][q]
            double aop[NT];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const double *w = &panel[(16 * ti + c) * 4];
                const double w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
                double v = -__builtin_fma(w3, x3, __builtin_fma(w2, x2, __builtin_fma(w1, x1, w0 * x0)));
                if (ti == tK) {
                    // pivot rows: D^-1 itself (their C operand is zeroed in step 6), exempt from the multiplier
                    // test (they are not multipliers)
                    const int m = c - c0;
                    const double xm = (m == 0) ? x0 : (m == 1) ? x1 : (m == 2) ? x2 : x3;
                    ok = ok && (panel_lane || le_tau(v));
                    v = panel_lane ? xm : v;
                } else {
                    ok = ok && le_tau(v);
                }
                aop[ti] = v;
            }
            __syncthreads();  // panel is rewritten by the next block step
            // 5. B operand: pivot rows as they stand; identity on the pivot columns
            double bop[NT];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) bop[tj] = acc[tK][tj][rK];
            bop[tK] = panel_lane ? (diag_lane ? 1.0 : 0.0) : bop[tK];
            // 6. C operand: zero on the pivot columns (the MFMA then leaves Aop * I_4 = the new K columns there)
            //    and zero on the pivot rows (they become D^-1 * W[K,:], a pure product: no cancellation, and the
            //    whole step stays exactly equivariant under power-of-two scaling of the input)
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ti][tK][r] = panel_lane ? 0.0 : acc[ti][tK][r];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) acc[tK][tj][rK] = 0.0;
            // 7. rank-4 update of every tile on the matrix cores
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc[ti][tj], 0, 0, 0);
        }

        const bool all_ok = __all(ok);
        if (all_ok) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + 4 * r + q, col = 16 * tj + c;
                        const unsigned uoff = (unsigned)((16 * ti + 4 * r) * n + 16 * tj);
                        if (FULL || (row < n && col < n)) X[uoff + lane_off] = acc[ti][tj][r];
                    }
            if (info && l == 0) info[mat] = 0;
        } else if (l == 0) {
            const int slot = atomicAdd(work_count, 1);
            work_list[slot] = (int)mat;
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <class T>
bool tile_family_supports(int n);
template <>
bool tile_family_supports<double>(int n) { return n >= 1 && n <= 64; }
template <>
bool tile_family_supports<float>(int) { return false; }

template <class T>
hipError_t launch_gj_tile(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);

template <>
hipError_t launch_gj_tile<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t)
{
    return hipErrorInvalidValue;
}

template <>
hipError_t launch_gj_tile<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info,
                                  hipStream_t stream)
{
    if (!tile_family_supports<double>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    // work list for matrices that fail the acceptance test: [0] = count, [1..batch] = indices (stream-ordered pool)
    int *ws = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int nt = (n + 15) / 16;
    // grid-stride over the batch: enough waves to fill 256 CUs several times over, few enough to amortise setup
    const unsigned grid = (unsigned)(batch < 256u * 8u * 4u ? batch : 256u * 8u * 4u);
    const unsigned b = (unsigned)batch;
#define TILE_LAUNCH(NT_)                                                                                              \
    if (n == 16 * NT_)                                                                                                \
        hipLaunchKernelGGL((matinv_gj_tile_f64<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
    else                                                                                                              \
        hipLaunchKernelGGL((matinv_gj_tile_f64<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1)
    switch (nt) {
    case 1: TILE_LAUNCH(1); break;
    case 2: TILE_LAUNCH(2); break;
    case 3: TILE_LAUNCH(3); break;
    default: TILE_LAUNCH(4); break;
    }
#undef TILE_LAUNCH
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_gj_lds_worklist<double>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

const char *name_gj_tile(bool f64, int n)
{
    if (!f64) return "";
    const bool full = (n % 16) == 0;
    switch ((n + 15) / 16) {
    case 1: return full ? "matinv_gj_tile_f64<1, true>" : "matinv_gj_tile_f64<1, false>";
    case 2: return full ? "matinv_gj_tile_f64<2, true>" : "matinv_gj_tile_f64<2, false>";
    case 3: return full ? "matinv_gj_tile_f64<3, true>" : "matinv_gj_tile_f64<3, false>";
    default: return full ? "matinv_gj_tile_f64<4, true>" : "matinv_gj_tile_f64<4, false>";
    }
}

}  // namespace matinv
