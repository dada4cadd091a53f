// rowlane2_kernels.hip -- the natural-order pass of the Gauss-Jordan entry point for 16 < n <= 32.
//
// Same design as rowlane_kernels.hip (rows in registers, pivot row folded into the FMA as a DPP row broadcast) with TWO rows
// per lane and, like the natural-order tile kernels it stands in for, VERIFIED pivots instead of a search: a matrix in which
// some multiplier exceeds TAU (or is NaN) is not finished here but appended to the work list of the pivoting tile kernel
// (tilep_impl.hpp), so general input costs this pass once and then goes straight there (tile_policy_*).
// Layout: a wavefront holds 4 matrices, lane
// (g, i) owns rows i and i + 16 of matrix g, register c of each half holds column c (NC = 24 or 32 columns, identity
// padded). A DPP row is still one matrix, so the pivot row of step k -- lane k % 16, lower or upper register half, which
// half is known at compile time -- reaches every lane of its matrix inside the v_fmac_*_dpp that consumes it:
// 2 NC multiply-adds per lane and step, no LDS, no MFMA. For these sizes that is 2 n^3 useful flops on the vector ALUs at
// (n / NC) lane utilisation, against 16 x 16 x 4 matrix-core tiles that are mostly padding at n = 20 or 24 and a panel
// factorisation that keeps 32 of 64 lanes busy: the job is HBM-bound either way (16 n^2 bytes per matrix), and this form
// needs a quarter of the instructions per matrix.
//
// r03: the same kernel as the fused mean / variance for these sizes (MODE RL2_GP): diag(c) is added while loading, the natural
// pivots are accepted when they are all POSITIVE (they are the squares of the Cholesky diagonal; no multiplier test: the sweep is
// stable on SPD input), rejected items go to the LDS pipeline kernel through the same work list, and a^T M^-1 d is folded out of
// the register rows (24 or 32 DPP-broadcast FMAs per row, a 16-lane reduction, ONE scalar written per item): f64 7.1e8 / 5.0e8 /
// 4.0e8 items/s at 17^2 / 20^2 / 24^2 against 3.0e8 on the 2 x 2-tile MFMA kernel. (MODE RL2_SPD, the Cholesky ENTRY POINT on this
// kernel with lower-triangle-only loads -- an upper element from its mirror address --, was measured and is not dispatched: the
// mirrored loads are uncoalesced, 3.7e8 / 2.9e8 inv/s at 20^2 / 24^2 f64 against 4.4e8 / 3.3e8 on the MFMA tile kernel.)
//
// Replaces, for these n, the 3n launches of /root/reference/src/gauss/batched_invert.cu:84-95 (pivotRow :17-45,
// normalizeRow :47-57, transform_matrix :59-82) like the other families; MODE 2 calcluateMean / calcluateVariance
// (src/gauss_bench.cu:127-265,275-409).
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"
#include "tile_common.hpp"
#include "wave_util.hpp"

namespace matinv {

constexpr int RL2_THREADS = 256;
constexpr double RL2_TAU = 4.0;  // as ROWLANE_TAU: the diagonal pivot is kept while every multiplier is <= TAU

constexpr int RL2_DPP_ROW_NEWBCAST = 0x150;  // + lane: broadcast that lane of each row of 16 (LLVM DppCtrl encoding)

template <int K, class V>
__device__ __forceinline__ V rl2_bcast(V v)  // lane K of this lane's row of 16 (= its matrix)
{
    return __builtin_amdgcn_update_dpp(V(0), v, RL2_DPP_ROW_NEWBCAST + K, 0xf, 0xf, false);
}

// Eight columns of one elimination step as ONE asm block (fixed instruction order; hipcc does not fold the DPP broadcast
// into the FMA by itself): d[c] += s[c](lane LK of the matrix) * m. `s_nop 1` first: the ISA wants two wait states between a
// VALU write of a VGPR and a DPP read of it and hipcc pads nothing in or before an asm block (the instruction before the
// block may be the select that rewrote the pivot column of the previous step).
//   SAME: d and s are the same registers -- the pivot row's own lane has m = 0, so the value every lane reads is unchanged
//   by the instruction that reads it.
#define RL2_F(T_, j) "v_fmac_" T_ "_dpp %[d" #j "], %[s" #j "], %[m] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define RL2_F8(T_) RL2_F(T_, 0) RL2_F(T_, 1) RL2_F(T_, 2) RL2_F(T_, 3) RL2_F(T_, 4) RL2_F(T_, 5) RL2_F(T_, 6) RL2_F(T_, 7)
#define RL2_S(T_, j) "v_fmac_" T_ "_dpp %[d" #j "], %[d" #j "], %[m] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define RL2_S8(T_) RL2_S(T_, 0) RL2_S(T_, 1) RL2_S(T_, 2) RL2_S(T_, 3) RL2_S(T_, 4) RL2_S(T_, 5) RL2_S(T_, 6) RL2_S(T_, 7)
#define RL2_DST(d, B) [d0] "+v"(d[B]), [d1] "+v"(d[B + 1]), [d2] "+v"(d[B + 2]), [d3] "+v"(d[B + 3]), [d4] "+v"(d[B + 4]), \
                      [d5] "+v"(d[B + 5]), [d6] "+v"(d[B + 6]), [d7] "+v"(d[B + 7])
#define RL2_SRC(s, B) [s0] "v"(s[B]), [s1] "v"(s[B + 1]), [s2] "v"(s[B + 2]), [s3] "v"(s[B + 3]), [s4] "v"(s[B + 4]), \
                      [s5] "v"(s[B + 5]), [s6] "v"(s[B + 6]), [s7] "v"(s[B + 7])

template <int LK, int B, int NC>
__device__ __forceinline__ void rl2_elim8_same(double (&d)[NC], double m)
{
    asm volatile("s_nop 1\n\t" RL2_S8("f64") : RL2_DST(d, B) : [m] "v"(m), [k] "n"(LK));
}
template <int LK, int B, int NC>
__device__ __forceinline__ void rl2_elim8_same(float (&d)[NC], float m)
{
    asm volatile("s_nop 1\n\t" RL2_S8("f32") : RL2_DST(d, B) : [m] "v"(m), [k] "n"(LK));
}
template <int LK, int B, int NC>
__device__ __forceinline__ void rl2_elim8_cross(double (&d)[NC], const double (&s)[NC], double m)
{
    asm volatile("s_nop 1\n\t" RL2_F8("f64") : RL2_DST(d, B) : RL2_SRC(s, B), [m] "v"(m), [k] "n"(LK));
}
template <int LK, int B, int NC>
__device__ __forceinline__ void rl2_elim8_cross(float (&d)[NC], const float (&s)[NC], float m)
{
    asm volatile("s_nop 1\n\t" RL2_F8("f32") : RL2_DST(d, B) : RL2_SRC(s, B), [m] "v"(m), [k] "n"(LK));
}

template <int LK, int NC, class T>
__device__ __forceinline__ void rl2_elim_same(T (&d)[NC], T m)
{
    rl2_elim8_same<LK, 0>(d, m);
    rl2_elim8_same<LK, 8>(d, m);
    rl2_elim8_same<LK, 16>(d, m);
    if constexpr (NC > 24) rl2_elim8_same<LK, 24>(d, m);
}
template <int LK, int NC, class T>
__device__ __forceinline__ void rl2_elim_cross(T (&d)[NC], const T (&s)[NC], T m)
{
    rl2_elim8_cross<LK, 0>(d, s, m);
    rl2_elim8_cross<LK, 8>(d, s, m);
    rl2_elim8_cross<LK, 16>(d, s, m);
    if constexpr (NC > 24) rl2_elim8_cross<LK, 24>(d, s, m);
}

__device__ __forceinline__ double rl2_abs(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float rl2_abs(float v) { return __builtin_fabsf(v); }

// One step, K a literal: which register half holds the pivot row is a compile-time fact. Natural order, VERIFIED: a
// multiplier above TAU (or NaN: zero / non-finite pivot) marks the matrix as rejected, nothing is exchanged here.
enum { RL2_GJ = 0, RL2_SPD = 1, RL2_GP = 2 };

template <int K, int NC, class T, int MODE = RL2_GJ>
struct Rl2Step {
    static constexpr bool HK = K >= 16;
    static constexpr int LK = K & 15;

    static __device__ __forceinline__ void run(T (&lo)[NC], T (&hi)[NC], int i, bool &rej, T &rs_lo, T &rs_hi)
    {
        const bool me_lo = !HK && i == LK, me_hi = HK && i == LK;  // this lane's lower / upper row is row K
        const T piv = rl2_bcast<LK>(HK ? hi[K] : lo[K]);
        const T inv = rcp_full(piv);
        const T nm_lo = me_lo ? (T)0 : -(lo[K] * inv);
        const T nm_hi = me_hi ? (T)0 : -(hi[K] * inv);
        if constexpr (MODE == RL2_GJ) rej = rej || !(rl2_abs(nm_lo) <= (T)RL2_TAU) || !(rl2_abs(nm_hi) <= (T)RL2_TAU);
        else rej = rej || !(piv > (T)0);  // SPD: leading principal minors positive (NaN fails)
        // eliminate: row r -= (a[r][K] / pivot) * row K for every other row; the pivot row keeps its values (unscaled until
        // the end). Column K itself is rewritten below.
        if constexpr (HK) {
            rl2_elim_cross<LK>(lo, hi, nm_lo);
            rl2_elim_same<LK>(hi, nm_hi);
        } else {
            rl2_elim_same<LK>(lo, nm_lo);
            rl2_elim_cross<LK>(hi, lo, nm_hi);
        }
        lo[K] = me_lo ? (T)1 : nm_lo;
        hi[K] = me_hi ? (T)1 : nm_hi;
        if constexpr (HK) rs_hi = me_hi ? inv : rs_hi;
        else rs_lo = me_lo ? inv : rs_lo;
        // pin the select HERE: hipcc otherwise sinks all of them to the end of the elimination and keeps the 32 reciprocals
        // alive until then (64 VGPRs in fp64)
        asm volatile("" : "+v"(rs_lo), "+v"(rs_hi));
    }
};

// waves per SIMD the allocator is asked to fit -- the largest that does not spill
constexpr int rl2_occupancy(size_t elem, int nc, bool full) { return elem == 8 ? 3 : (nc > 24 ? (full ? 5 : 4) : (full ? 6 : 5)); }

// NC serves n = NC - 7 .. NC: only the last seven columns, rows and steps can be padding -- everything before them carries no
// run-time predicate.
// acc += v[c] * (lane c % 16 of the matrix's row of 16 lanes)(dlo or dhi): the dot product of a register row with a vector that
// lies one element per lane, the broadcast folded into the FMA like the elimination's
template <int C, class T>
__device__ __forceinline__ void rl2_dot_step(T &acc, T v, T dlo, T dhi)
{
    if constexpr (sizeof(T) == 8)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %[acc], %[d], %[v] row_newbcast:%[k] row_mask:0xf bank_mask:0xf"
                     : [acc] "+v"(acc) : [d] "v"(C < 16 ? dlo : dhi), [v] "v"(v), [k] "n"(C & 15));
    else
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %[acc], %[d], %[v] row_newbcast:%[k] row_mask:0xf bank_mask:0xf"
                     : [acc] "+v"(acc) : [d] "v"(C < 16 ? dlo : dhi), [v] "v"(v), [k] "n"(C & 15));
}
template <int C, int NC, class T>
struct Rl2Dot {
    static __device__ __forceinline__ void run(T (&acc)[4], const T (&row)[NC], T dlo, T dhi)
    {
        rl2_dot_step<C>(acc[C & 3], row[C], dlo, dhi);
        if constexpr (C + 1 < NC) Rl2Dot<C + 1, NC, T>::run(acc, row, dlo, dhi);
    }
};

template <class T>
struct Rl2Gp {
    const T *a, *c, *d, *e;  // d == nullptr: variance, out = e - a^T M^-1 a
    T *out;
};

template <class T, int NC, bool FULL, int MODE = RL2_GJ>
__global__ __launch_bounds__(RL2_THREADS, rl2_occupancy(sizeof(T), NC, FULL)) void matinv_gj_rowlane2(BatchRef<const T> Ain, BatchRef<T> Xout,
                                                                                              int *info, int n_rt, unsigned batch,
                                                                                              int *work_count, int *work_list, Rl2Gp<T> gp)
{
    constexpr int GPW = 4;       // matrices per wavefront
    constexpr int CMIN = NC - 7;  // smallest n this instantiation is launched for
    int n = FULL ? NC : n_rt;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, i = lane & 15;
    const unsigned waves_per_block = RL2_THREADS / 64;
    const unsigned wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const unsigned wave_stride = gridDim.x * waves_per_block;
    const unsigned n_waves = (batch + GPW - 1) / GPW;

    for (unsigned w = wave0; w < n_waves; w += wave_stride) {
        // run-time n opaque once per iteration: otherwise every `c < n` is hoisted out of this loop and kept in SGPR pairs
        if (!FULL) asm volatile("" : "+s"(n));
        const unsigned mat = w * GPW + g;
        const bool valid = mat < batch;
        const T *A = Ain.at(valid ? mat : batch - 1);
        T *X = Xout.at(valid ? mat : batch - 1);
        const bool hi_in = i + 16 < n;  // the lower row always exists (n > 16)
        // Loads are UNCONDITIONAL (written as `in ? A[..] : pad` hipcc branches around every single load and they no longer
        // overlap): the address is clamped into the matrix instead and padding is selected afterwards. The lanes of a
        // matrix slot beyond the batch work on a copy of the last matrix and store nothing.
        T lo[NC], hi[NC];
        const T *Alo = A + i, *Ahi = A + (hi_in ? 16 + i : i);
        const int rhi = hi_in ? 16 + i : i;  // the upper row this lane really reads
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool col_in = FULL || c < CMIN || c < n;  // wave-uniform; literal for c < CMIN
            const int cc = col_in ? c : 0;
            T vl, vh;
            if constexpr (MODE != RL2_SPD) {
                vl = Alo[cc * n], vh = Ahi[cc * n];
            } else {
                // only the LOWER triangle is read (the Cholesky contract): element (row, col) with col > row comes from its mirror
                vl = A[(cc <= i) ? cc * n + i : i * n + cc];
                vh = A[(cc <= rhi) ? cc * n + rhi : rhi * n + cc];
            }
            lo[c] = col_in ? vl : (T)0;  // i < 16 <= c: never the diagonal
            hi[c] = (col_in && hi_in) ? vh : ((i + 16 == c) ? (T)1 : (T)0);
        }
        if constexpr (MODE == RL2_GP) {
            // addDiagonal (gauss_bench.cu:38-43): this lane's rows i and i + 16
            const T *vc = gp.c + (size_t)(valid ? mat : batch - 1) * n;
            const T clo = vc[i], chi = vc[hi_in ? 16 + i : i];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c < 16) lo[c] += (c == i) ? clo : (T)0;
                else hi[c] += (c == i + 16 && hi_in) ? chi : (T)0;
            }
        }
        bool rej = false;              // some multiplier of this lane's rows exceeded TAU (or was NaN)
        T rs_lo = (T)1, rs_hi = (T)1;  // 1 / pivot of the step in which the row was the pivot row

#define RL2_RUN(K) \
    if (FULL || K < CMIN || K < n) Rl2Step<K, NC, T, MODE>::run(lo, hi, i, rej, rs_lo, rs_hi);
        RL2_RUN(0) RL2_RUN(1) RL2_RUN(2) RL2_RUN(3) RL2_RUN(4) RL2_RUN(5) RL2_RUN(6) RL2_RUN(7)
        RL2_RUN(8) RL2_RUN(9) RL2_RUN(10) RL2_RUN(11) RL2_RUN(12) RL2_RUN(13) RL2_RUN(14) RL2_RUN(15)
        RL2_RUN(16) RL2_RUN(17) RL2_RUN(18) RL2_RUN(19) RL2_RUN(20) RL2_RUN(21) RL2_RUN(22) RL2_RUN(23)
        if constexpr (NC > 24) {
            RL2_RUN(24) RL2_RUN(25) RL2_RUN(26) RL2_RUN(27) RL2_RUN(28) RL2_RUN(29) RL2_RUN(30) RL2_RUN(31)
        }
#undef RL2_RUN
        // rejected by any row of the matrix (a DPP row = one matrix)
        const unsigned long long votes = __ballot(rej);
        const bool bad = ((votes >> (16 * g)) & 0xffffull) != 0;
        if (MODE == RL2_GP && !bad) {
            // s = sum_r a_r / pivot_r * (sum_c X'[r][c] d_c), X' = the rows before the deferred normalisation; padding rows and
            // columns carry zeros in a and d
            const size_t item = valid ? mat : batch - 1;
            const T *va = gp.a + item * n, *vd = gp.d ? gp.d + item * n : va;
            const T alo = va[i], ahi = hi_in ? va[16 + i] : (T)0;
            const T dlo = vd[i], dhi = hi_in ? vd[16 + i] : (T)0;
            T tl[4] = {0, 0, 0, 0}, th[4] = {0, 0, 0, 0};
            Rl2Dot<0, NC, T>::run(tl, lo, dlo, dhi);
            Rl2Dot<0, NC, T>::run(th, hi, dlo, dhi);
            T s_ = alo * rs_lo * ((tl[0] + tl[1]) + (tl[2] + tl[3])) + ahi * rs_hi * ((th[0] + th[1]) + (th[2] + th[3]));
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) s_ += __shfl_xor(s_, off);  // the 16 lanes of the matrix
            if (valid && i == 0) {
                gp.out[mat] = gp.d ? s_ : gp.e[mat] - s_;
                if (info) info[mat] = 0;
            }
        } else if (!bad) {
            // the deferred normalisation: every row was the pivot row of exactly one step
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                lo[c] *= rs_lo;
                hi[c] *= rs_hi;
            }
            if (valid) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (FULL || c < CMIN || c < n) X[c * n + i] = lo[c];
            }
            if (valid && hi_in) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (FULL || c < CMIN || c < n) X[c * n + 16 + i] = hi[c];
            }
            if (info && valid && i == 0) info[mat] = 0;
        } else if (valid && i == 0) {
            const int slot = atomicAdd(work_count, 1);
            work_list[slot] = (int)mat;
        }
    }
}

bool rowlane2_supports(int n) { return n > 16 && n <= 32; }

// Where this pass replaces the natural-order tile kernel. MATINV_ROWLANE2=0: nowhere, =2: everywhere it can run (A/B
// measurements); default: where it measured faster.
bool rowlane2_natural_use(bool f64, int n)
{
    static const int mode = [] { const char *s = getenv("MATINV_ROWLANE2"); return s ? atoi(s) : 1; }();
    if (!rowlane2_supports(n) || mode == 0) return false;
    if (mode == 2) return true;
    // measured (100 k matrices, natural-order pass only, this kernel against matinv_gj_tile_*<2, ..>): fp64 n = 17 / 20 / 24 / 25
    // 1.60 / 1.50 / 1.33 / 1.04 x, n = 28 / 32 0.85 / 0.98 x; fp32 1.52 / 1.65 / 1.52 / 1.24 x and 0.97 / 0.87 x
    (void)f64;
    return n <= 25;
}

template <class T, int NC, bool FULL, int MODE = RL2_GJ>
static hipError_t launch_rl2(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, int *work_count,
                             int *work_list, Rl2Gp<T> gp = Rl2Gp<T>())
{
    const size_t waves = (batch + 3) / 4;
    const size_t blocks = (waves + RL2_THREADS / 64 - 1) / (RL2_THREADS / 64);
    // persistent-style grid: what stays resident, every wave strides over the batch
    const unsigned resident = 256u * (unsigned)rl2_occupancy(sizeof(T), NC, FULL);
    const unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
    hipLaunchKernelGGL((matinv_gj_rowlane2<T, NC, FULL, MODE>), dim3(grid), dim3(RL2_THREADS), 0, stream, A, X, info, n, (unsigned)batch,
                       work_count, work_list, gp);
    return hipGetLastError();
}

template <class T, int MODE>
static hipError_t enqueue_rl2_mode(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, int *work_count,
                                   int *work_list, Rl2Gp<T> gp)
{
    if (!rowlane2_supports(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    // run-time n only (the compile-time-n instantiations of these modes spill under the occupancy the plain kernel is built for)
    if (n <= 24) return launch_rl2<T, 24, false, MODE>(n, A, X, batch, info, stream, work_count, work_list, gp);
    return launch_rl2<T, 32, false, MODE>(n, A, X, batch, info, stream, work_count, work_list, gp);
}

// the fused mean / variance on this kernel (Ds == nullptr: variance); rejects -> work list (-> LDS pipeline kernel)
template <class T>
hipError_t enqueue_gp_rowlane2(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch, int *info,
                               hipStream_t stream, int *work_count, int *work_list)
{
    BatchRef<const T> A{Bs, (size_t)n * n, nullptr};
    BatchRef<T> X{nullptr, 0, nullptr};
    return enqueue_rl2_mode<T, RL2_GP>(n, A, X, batch, info, stream, work_count, work_list, Rl2Gp<T>{As, Cs, Ds, Es, out});
}
template hipError_t enqueue_gp_rowlane2<double>(int, const double *, const double *, const double *, const double *, const double *, double *,
                                                size_t, int *, hipStream_t, int *, int *);
template hipError_t enqueue_gp_rowlane2<float>(int, const float *, const float *, const float *, const float *, const float *, float *, size_t,
                                               int *, hipStream_t, int *, int *);

// natural-order pass over the whole batch; the matrices it rejects are appended to work_list[0 .. *work_count)
template <class T>
hipError_t enqueue_gj_rowlane2(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, int *work_count,
                               int *work_list)
{
    if (!rowlane2_supports(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    if (n == 32) return launch_rl2<T, 32, true>(n, A, X, batch, info, stream, work_count, work_list);
    if (n == 24) return launch_rl2<T, 24, true>(n, A, X, batch, info, stream, work_count, work_list);
    if (n < 24) return launch_rl2<T, 24, false>(n, A, X, batch, info, stream, work_count, work_list);
    return launch_rl2<T, 32, false>(n, A, X, batch, info, stream, work_count, work_list);
}
template hipError_t enqueue_gj_rowlane2<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t, int *, int *);
template hipError_t enqueue_gj_rowlane2<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t, int *, int *);

// the pipeline entry point with its work list: items this kernel rejects (not SPD) are finished by the LDS pipeline kernel,
// which reports the failing column
template <class T>
hipError_t launch_gp_rowlane2(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch, int *info,
                              hipStream_t stream)
{
    if (!rowlane2_supports(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e == hipSuccess) e = enqueue_gp_rowlane2<T>(n, As, Bs, Cs, Ds, Es, out, batch, info, stream, ws, ws + 1);
    if (e == hipSuccess) e = launch_gp_lds_worklist<T>(n, As, Bs, Cs, Ds, Es, out, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}
template hipError_t launch_gp_rowlane2<double>(int, const double *, const double *, const double *, const double *, const double *, double *,
                                               size_t, int *, hipStream_t);
template hipError_t launch_gp_rowlane2<float>(int, const float *, const float *, const float *, const float *, const float *, float *, size_t,
                                              int *, hipStream_t);

// MATINV_ROWLANE2_GP=0: the pipeline of these sizes stays on the MFMA tile kernel (A/B switch)
bool rowlane2_gp_use(bool f64, int n)
{
    static const bool on = [] { const char *s = getenv("MATINV_ROWLANE2_GP"); return !(s && *s == '0'); }();
    return on && rowlane2_natural_use(f64, n);
}

const char *name_gp_rowlane2(bool f64, int n)
{
    static thread_local char buf[80];
    snprintf(buf, sizeof buf, "matinv_gj_rowlane2<%s, %d, false, 2>", f64 ? "double" : "float", n <= 24 ? 24 : 32);
    return buf;
}

const char *name_gj_rowlane2(bool f64, int n)
{
    static thread_local char buf[80];
    snprintf(buf, sizeof buf, "matinv_gj_rowlane2<%s, %d, %s, 0>", f64 ? "double" : "float", n <= 24 ? 24 : 32,
             (n == 24 || n == 32) ? "true" : "false");
    return buf;
}

}  // namespace matinv
