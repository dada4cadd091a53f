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
#include <stdlib.h>

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

// Acceptance test, wave-wide, evaluated on the spot and accumulated in an SGPR pair: bad |= ballot(!(|v| <= TAU)) (NaN
// fails). It is inline asm on purpose: written in C++ (`ok = ok && ...`, `bad |= __ballot(...)`, or a lane-local running
// max) hipcc sinks every comparison to the end of the kernel and keeps all multipliers of all 4*NT block steps alive
// (hundreds of VGPRs, or dozens of SGPR pairs spilled through v_writelane -- measured: 2x slower). TAU = 4.0 is an
// inline constant of the ISA.
__device__ __forceinline__ void note_fail(unsigned long long &bad, double v)
{
    static_assert(TILE_TAU == 4.0, "the asm below hard-codes the inline constant 4.0");
    asm volatile("v_cmp_nle_f64_e64 vcc, |%1|, 4.0\n\ts_or_b64 %0, %0, vcc" : "+s"(bad) : "v"(v) : "vcc");
}

// ---- pieces of one block step ---------------------------------------------------------------------------------

// 1. the 4 pivot columns of block kb -> LDS, [row][4]. They live in the 16 lanes c in [c0, c0+4) of tile column tK.
template <int NT>
__device__ __forceinline__ void panel_to_lds(double *panel, const v4d (&acc)[NT][NT], int kb, int q, int c)
{
    const int tK = kb >> 2, c0 = 4 * (kb & 3);
    if (c >= c0 && c < c0 + 4) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) panel[(16 * ti + 4 * r + q) * 4 + (c - c0)] = acc[ti][tK][r];
    }
}

// 2.-4. read the pivot block D and this lane's panel rows from LDS, invert D (column q), form the A operand
//       aop[ti] = Aop[16ti + c][q] and update the acceptance flag. Split into NSTAGE pieces of roughly equal
//       VALU/LDS work so the look-ahead loop can issue one MFMA of the CURRENT block step between two pieces of the
//       NEXT step's panel (hardware issues in order: MFMA, ~64 cycles of VALU, MFMA, ... keeps both pipes busy).
template <int NT>
struct PanelSolve {
    static constexpr int NSTAGE = 6 + NT;
    double d[4][4];
    double r0, r1, r2, r3, l10, l20, l30, l21, l31, l32, u11, u12, u13, u22, u23, u33;
    double a21, a22, a23, a31, a32, a33, b32, b33, y0, y1, y2, y3, x0, x1, x2, x3;

    __device__ __forceinline__ void stage(int s, const double *panel, int kb, int q, int c, double (&aop)[NT],
                                          unsigned long long &bad)
    {
        const int tK = kb >> 2, c0 = 4 * (kb & 3), K0 = 4 * kb;
        const bool panel_lane = (c >= c0) && (c < c0 + 4);
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = panel[(K0 + i) * 4 + j];
            // LU of D without pivoting (multipliers checked below)
            r0 = fast_rcp(d[0][0]);
            l10 = d[1][0] * r0, l20 = d[2][0] * r0, l30 = d[3][0] * r0;
        } else if (s == 1) {
            u11 = __builtin_fma(-l10, d[0][1], d[1][1]), u12 = __builtin_fma(-l10, d[0][2], d[1][2]);
            u13 = __builtin_fma(-l10, d[0][3], d[1][3]);
            a21 = __builtin_fma(-l20, d[0][1], d[2][1]), a22 = __builtin_fma(-l20, d[0][2], d[2][2]);
            a23 = __builtin_fma(-l20, d[0][3], d[2][3]);
            a31 = __builtin_fma(-l30, d[0][1], d[3][1]), a32 = __builtin_fma(-l30, d[0][2], d[3][2]);
            a33 = __builtin_fma(-l30, d[0][3], d[3][3]);
            r1 = fast_rcp(u11);
        } else if (s == 2) {
            l21 = a21 * r1, l31 = a31 * r1;
            u22 = __builtin_fma(-l21, u12, a22), u23 = __builtin_fma(-l21, u13, a23);
            b32 = __builtin_fma(-l31, u12, a32), b33 = __builtin_fma(-l31, u13, a33);
            r2 = fast_rcp(u22);
            l32 = b32 * r2;
            u33 = __builtin_fma(-l32, u23, b33);
        } else if (s == 3) {
            r3 = fast_rcp(u33);
            note_fail(bad, l10), note_fail(bad, l20), note_fail(bad, l30);
            note_fail(bad, l21), note_fail(bad, l31), note_fail(bad, l32);
            // a zero / non-finite last pivot needs no test of its own: r3 = inf/NaN makes x, hence every Aop entry
            // outside the pivot rows (0 * inf = NaN included), fail the test in the last stages
        } else if (s == 4) {
            // L y = e_q
            y0 = (q == 0) ? 1.0 : 0.0;
            y1 = __builtin_fma(-l10, y0, (q == 1) ? 1.0 : 0.0);
            y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, (q == 2) ? 1.0 : 0.0));
            y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, y1, __builtin_fma(-l30, y0, (q == 3) ? 1.0 : 0.0)));
        } else if (s == 5) {
            // U x = y : x = column q of D^-1
            x3 = y3 * r3;
            x2 = __builtin_fma(-u23, x3, y2) * r2;
            x1 = __builtin_fma(-u13, x3, __builtin_fma(-u12, x2, y1)) * r1;
            x0 = __builtin_fma(-d[0][3], x3, __builtin_fma(-d[0][2], x2, __builtin_fma(-d[0][1], x1, y0))) * r0;
        } else {
            const int ti = s - 6;
            const double *w = &panel[(16 * ti + c) * 4];
            const double w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            double v = -__builtin_fma(w3, x3, __builtin_fma(w2, x2, __builtin_fma(w1, x1, w0 * x0)));
            if (ti == tK) {
                // pivot rows: D^-1 itself (their C operand is zeroed), exempt from the multiplier test
                const int m = c - c0;
                const double x01 = (m & 1) ? x1 : x0, x23 = (m & 1) ? x3 : x2;
                const double xm = (m & 2) ? x23 : x01;
                note_fail(bad, panel_lane ? 0.0 : v);
                v = panel_lane ? xm : v;
            } else {
                note_fail(bad, v);
            }
            aop[ti] = v;
        }
    }
};

template <int NT>
__device__ __forceinline__ void panel_solve(const double *panel, int kb, int q, int c, double (&aop)[NT],
                                            unsigned long long &bad)
{
    PanelSolve<NT> ps;
#pragma unroll
    for (int s = 0; s < PanelSolve<NT>::NSTAGE; ++s) ps.stage(s, panel, kb, q, c, aop, bad);
}

// 5.+6. B operand (pivot rows as they stand, I_4 on the pivot columns) and C operand (zero on the pivot columns: the
// MFMA then leaves Aop * I_4 = the new K columns there; zero on the pivot rows: they become D^-1 * W[K,:], a pure
// product -- no cancellation, and the step stays exactly equivariant under power-of-two scaling of the input).
template <int NT>
__device__ __forceinline__ void prep_operands(v4d (&acc)[NT][NT], double (&bop)[NT], int kb, int q, int c)
{
    const int tK = kb >> 2, rK = kb & 3, c0 = 4 * (kb & 3);
    const bool panel_lane = (c >= c0) && (c < c0 + 4);
    const bool diag_lane = panel_lane && (c - c0 == q);
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) bop[tj] = acc[tK][tj][rK];
    bop[tK] = panel_lane ? (diag_lane ? 1.0 : 0.0) : bop[tK];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ti][tK][r] = panel_lane ? 0.0 : acc[ti][tK][r];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[tK][tj][rK] = 0.0;
}

// FULL: n == 16*NT known at compile time (constant address offsets, no bounds checks).
// LOOKAHEAD: software pipelining across block steps -- the tile column that holds the NEXT pivot columns is updated
// first, the next panel is extracted and solved while the remaining MFMAs of the current step are in flight.
template <int NT, bool FULL, bool LOOKAHEAD>
__global__ __launch_bounds__(64, (FULL || NT < 3) ? 2 : 1) void matinv_gj_tile_f64(BatchRef<const double> Ain,
                                                                                 BatchRef<double> Xout, int *info,
                                                                                 int n_rt, unsigned batch,
                                                                                 int *work_count, int *work_list)
{
    constexpr int N = 16 * NT;
    constexpr int NKB = 4 * NT;
    const int n = FULL ? N : n_rt;
    __shared__ __attribute__((aligned(16))) double panel[N * 4];  // [row][4 pivot columns]
    const int l = threadIdx.x;

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const double *A = Ain.at(mat);
        double *X = Xout.at(mat);
        // Launder the lane coordinates once per matrix: otherwise LICM hoists the ~60 per-lane constants of the 4*NT
        // unrolled block steps (I_4 lanes, e_q entries, lane masks) out of this loop and the allocator spills them.
        int q = l >> 4, c = l & 15;
        // (addresses keep using the un-laundered lane id so they stay in saddr + 32-bit voffset + immediate form)
        const unsigned lane_off = (unsigned)((l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c));
        // one per-lane element offset + wave-uniform (compile-time when FULL) tile offsets keep the 16*NT*NT
        // addresses out of VGPRs
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
        unsigned long long bad = 0;  // wave-uniform: lanes that saw a multiplier above TAU (or NaN)
        double aop[NT], bop[NT];

        if (LOOKAHEAD) {
            panel_to_lds<NT>(panel, acc, 0, q, c);
            __syncthreads();
            panel_solve<NT>(panel, 0, q, c, aop, bad);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                prep_operands<NT>(acc, bop, kb, q, c);
                if (kb + 1 < NKB) {
                    const int tn = (kb + 1) >> 2;
                    // (a) the tile column holding the next pivot columns first ...
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti)
                        acc[ti][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tn], acc[ti][tn], 0, 0, 0);
                    // (b) the other NT*(NT-1) tiles, pinned in program order between the pieces of the next
                    //     panel: 2 MFMAs cover the latency of (a) before the panel columns are read back, then one
                    //     MFMA after every stage. sched_barrier(0) keeps hipcc from re-clustering them.
                    constexpr int NB = NT * (NT - 1);
                    int pend = 0;  // folds to a literal: everything here is fully unrolled
                    auto issue_b = [&](int count) {
#pragma unroll
                        for (int z = 0; z < count; ++z) {
                            if (pend < NB) {
                                const int tjx = pend / NT, ti = pend % NT;
                                const int tj = tjx + (tjx >= tn ? 1 : 0);
                                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc[ti][tj], 0, 0, 0);
                                ++pend;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    issue_b(2);
                    __syncthreads();  // panel(kb) has been consumed (aop is in registers)
                    panel_to_lds<NT>(panel, acc, kb + 1, q, c);
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
                    double aop_next[NT];
                    PanelSolve<NT> ps;
                    constexpr int NS = PanelSolve<NT>::NSTAGE;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        // spread the remaining MFMAs evenly over the stages
                        issue_b(((NB - 2) * (s + 1)) / NS - ((NB - 2) * s) / NS);
                        ps.stage(s, panel, kb + 1, q, c, aop_next, bad);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    issue_b(NB);  // whatever is left (NT < 3)
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti) aop[ti] = aop_next[ti];
                } else {
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj)
                            acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc[ti][tj], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                panel_to_lds<NT>(panel, acc, kb, q, c);
                __syncthreads();
                panel_solve<NT>(panel, kb, q, c, aop, bad);
                __syncthreads();  // panel is rewritten by the next block step
                prep_operands<NT>(acc, bop, kb, q, c);
                // 7. rank-4 update of every tile on the matrix cores
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc[ti][tj], 0, 0, 0);
            }
        }

        if (bad == 0) {
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
        if (LOOKAHEAD) __syncthreads();  // the next matrix's first panel write must not pass this one's last reads
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
    static const bool lookahead = []() {
        const char *s = getenv("MATINV_TILE_LOOKAHEAD");  // A/B switch for profiling; default on
        return !(s && *s == '0');
    }();
#define TILE_LAUNCH(NT_)                                                                                              \
    if (n == 16 * NT_ && lookahead)                                                                                   \
        hipLaunchKernelGGL((matinv_gj_tile_f64<NT_, true, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
    else if (n == 16 * NT_)                                                                                           \
        hipLaunchKernelGGL((matinv_gj_tile_f64<NT_, true, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
    else                                                                                                              \
        hipLaunchKernelGGL((matinv_gj_tile_f64<NT_, false, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1)
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
    case 1: return full ? "matinv_gj_tile_f64<1, true, true>" : "matinv_gj_tile_f64<1, false, false>";
    case 2: return full ? "matinv_gj_tile_f64<2, true, true>" : "matinv_gj_tile_f64<2, false, false>";
    case 3: return full ? "matinv_gj_tile_f64<3, true, true>" : "matinv_gj_tile_f64<3, false, false>";
    default: return full ? "matinv_gj_tile_f64<4, true, true>" : "matinv_gj_tile_f64<4, false, false>";
    }
}

}  // namespace matinv
