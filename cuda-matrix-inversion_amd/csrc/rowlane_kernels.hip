// rowlane_kernels.hip -- kernel family "ROWLANE": n <= 16, several matrices per wavefront, register resident.
//
// A wavefront holds 64/NP matrices (NP = 8 or 16 = n rounded up, identity padded): lane (g, i) owns row i of
// matrix g, register c holds column c, so the load `a[c] = A[c*n + i]` is a coalesced 8*NP-byte segment per matrix
// and consecutive matrices of the batch are adjacent in the wave. In-place Gauss-Jordan with partial (row) pivoting:
//   * pivot choice   = threshold partial pivoting: the diagonal entry is kept while no multiplier exceeds 4; otherwise the
//                      column maximum is found by a DPP max-reduction of |a[k]| over the NP lanes of the matrix (quad_perm /
//                      row_half_mirror / row_mirror, no LDS) + ballot + ctz and brought to lane k with a ds_bpermute row swap
//                      (wave-uniform branch; never taken on diagonally dominant input);
//   * pivot row      = broadcast with DPP row_newbcast:k straight into the FMA operand;
//   * elimination    = NP-1 fused multiply-adds per lane per step, multiplier lane-local; the pivot rows stay unscaled
//                      until one final multiply per element.
// Row swaps are undone as a column permutation folded into the store addresses.
//
// Replaces, for small n, the 3n launches of /root/reference/src/gauss/batched_invert.cu:84-95 (pivotRow :17-45,
// normalizeRow :47-57, transform_matrix :59-82) with one launch that touches HBM once per element.
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"

namespace matinv {

constexpr int ROWLANE_THREADS = 256;
constexpr double ROWLANE_TAU = 4.0;  // a diagonal pivot is accepted while every multiplier is <= TAU (threshold pivoting)

// DPP controls (LLVM AMDGPU DppCtrl encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15-i within a row of 16
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i within each half row
constexpr int DPP_ROW_NEWBCAST = 0x150;    // + lane: broadcast that lane of each row of 16

template <int NP, int K, class V>
__device__ __forceinline__ V bcast_const(V v)
{
    // value of lane K of the matrix this lane belongs to
    if (NP == 16) {
        // every lane is written, so `old` is dead: pass a constant to avoid a tied copy
        return __builtin_amdgcn_update_dpp(V(0), v, DPP_ROW_NEWBCAST + K, 0xf, 0xf, false);
    } else {
        // two matrices per DPP row: lanes 0-7 (banks 0,1) take lane K, lanes 8-15 (banks 2,3) take lane K+8
        V t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_NEWBCAST + K, 0xf, 0x3, false);
        return __builtin_amdgcn_update_dpp(t, v, DPP_ROW_NEWBCAST + K + 8, 0xf, 0xc, false);
    }
}

// k comes from a fully unrolled loop: the switch folds to the single case
template <int NP, class V>
__device__ __forceinline__ V bcast_lane(V v, int k)
{
    switch (k) {
    case 0: return bcast_const<NP, 0>(v);
    case 1: return bcast_const<NP, 1>(v);
    case 2: return bcast_const<NP, 2>(v);
    case 3: return bcast_const<NP, 3>(v);
    case 4: return bcast_const<NP, 4>(v);
    case 5: return bcast_const<NP, 5>(v);
    case 6: return bcast_const<NP, 6>(v);
    case 7: return bcast_const<NP, 7>(v);
    case 8: return bcast_const<16, 8>(v);
    case 9: return bcast_const<16, 9>(v);
    case 10: return bcast_const<16, 10>(v);
    case 11: return bcast_const<16, 11>(v);
    case 12: return bcast_const<16, 12>(v);
    case 13: return bcast_const<16, 13>(v);
    case 14: return bcast_const<16, 14>(v);
    default: return bcast_const<16, 15>(v);
    }
}

// One elimination step on the register-resident rows, as ONE asm block so the instruction order is fixed:
//   for every column c != K:  a[c] += a_pivotrow[c] * negm      (v_fmac_*_dpp: the DPP row broadcast of lane K is
//                                                                folded into the FMA operand -- hipcc does not form
//                                                                this from __builtin_amdgcn_update_dpp)
// The pivot row has negm = 0 and is left UNSCALED for the whole elimination (its 1/pivot is applied once, at the end:
// later steps treat it like any other row and row scaling commutes with row operations), so there is no per-step
// normalisation pass. Hazards: the ISA wants 2 wait states between a VALU write of a VGPR and a DPP read of it, and hipcc
// pads nothing inside or before an asm block: the leading `s_nop 1` covers whatever VALU instruction the compiler
// scheduled last; inside the block every register is DPP-read by the one instruction that also rewrites it.
#define RL_FMAC16(T_, c) \
    ".if %[k] != " #c "\n\tv_fmac_" T_ "_dpp %[a" #c "], %[a" #c "], %[m] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t.endif\n\t"
// NP = 8: two matrices share a DPP row of 16 lanes. 64-bit DPP ignores bank_mask (measured on gfx950: both halves were
// written), so the half-row selection is done with EXEC instead: lanes 0-7 of every row take lane K, then lanes 8-15
// take lane K+8 (the source lane is always inside the enabled half).
#define RL_FMAC8(T_, c, kk) \
    ".if %[k] != " #c "\n\tv_fmac_" T_ "_dpp %[a" #c "], %[a" #c "], %[m] row_newbcast:%[" kk "] row_mask:0xf bank_mask:0xf\n\t.endif\n\t"
#define RL_FMAC8_LO(T_, c) RL_FMAC8(T_, c, "k")
#define RL_FMAC8_HI(T_, c) RL_FMAC8(T_, c, "k8")
#define RL_STEP8(T_)                                                                                                  \
    "s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, 0x00ff00ff\n\ts_mov_b32 exec_hi, 0x00ff00ff\n\ts_nop 1\n\t"          \
    RL_8(RL_FMAC8_LO, T_)                                                                                             \
    "s_mov_b32 exec_lo, 0xff00ff00\n\ts_mov_b32 exec_hi, 0xff00ff00\n\ts_nop 1\n\t"                                  \
    RL_8(RL_FMAC8_HI, T_)                                                                                             \
    "s_mov_b64 exec, %[sv]\n\t"
#define RL_8(M, T_) M(T_, 0) M(T_, 1) M(T_, 2) M(T_, 3) M(T_, 4) M(T_, 5) M(T_, 6) M(T_, 7)
#define RL_16(M, T_) RL_8(M, T_) M(T_, 8) M(T_, 9) M(T_, 10) M(T_, 11) M(T_, 12) M(T_, 13) M(T_, 14) M(T_, 15)
#define RL_OPS8 [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), \
                [a6] "+v"(a[6]), [a7] "+v"(a[7])
#define RL_OPS16 RL_OPS8, [a8] "+v"(a[8]), [a9] "+v"(a[9]), [a10] "+v"(a[10]), [a11] "+v"(a[11]), [a12] "+v"(a[12]),  \
                 [a13] "+v"(a[13]), [a14] "+v"(a[14]), [a15] "+v"(a[15])

template <int K>
__device__ __forceinline__ void elim_step_const(double (&a)[16], double negm)
{
    asm volatile("s_nop 1\n\t" RL_16(RL_FMAC16, "f64")
                 : RL_OPS16 : [m] "v"(negm), [k] "n"(K));
}
template <int K>
__device__ __forceinline__ void elim_step_const(float (&a)[16], float negm)
{
    asm volatile("s_nop 1\n\t" RL_16(RL_FMAC16, "f32")
                 : RL_OPS16 : [m] "v"(negm), [k] "n"(K));
}
template <int K>
__device__ __forceinline__ void elim_step_const(double (&a)[8], double negm)
{
    unsigned long long sv;
    asm volatile(RL_STEP8("f64") : RL_OPS8, [sv] "=&s"(sv) : [m] "v"(negm), [k] "n"(K), [k8] "n"(K + 8));
}
template <int K>
__device__ __forceinline__ void elim_step_const(float (&a)[8], float negm)
{
    unsigned long long sv;
    asm volatile(RL_STEP8("f32") : RL_OPS8, [sv] "=&s"(sv) : [m] "v"(negm), [k] "n"(K), [k8] "n"(K + 8));
}

template <class T>
__device__ __forceinline__ void elim_step(T (&a)[8], T negm, int k)
{
    switch (k) {
    case 0: elim_step_const<0>(a, negm); break;
    case 1: elim_step_const<1>(a, negm); break;
    case 2: elim_step_const<2>(a, negm); break;
    case 3: elim_step_const<3>(a, negm); break;
    case 4: elim_step_const<4>(a, negm); break;
    case 5: elim_step_const<5>(a, negm); break;
    case 6: elim_step_const<6>(a, negm); break;
    default: elim_step_const<7>(a, negm); break;
    }
}
template <class T>
__device__ __forceinline__ void elim_step(T (&a)[16], T negm, int k)
{
    switch (k) {
    case 0: elim_step_const<0>(a, negm); break;
    case 1: elim_step_const<1>(a, negm); break;
    case 2: elim_step_const<2>(a, negm); break;
    case 3: elim_step_const<3>(a, negm); break;
    case 4: elim_step_const<4>(a, negm); break;
    case 5: elim_step_const<5>(a, negm); break;
    case 6: elim_step_const<6>(a, negm); break;
    case 7: elim_step_const<7>(a, negm); break;
    case 8: elim_step_const<8>(a, negm); break;
    case 9: elim_step_const<9>(a, negm); break;
    case 10: elim_step_const<10>(a, negm); break;
    case 11: elim_step_const<11>(a, negm); break;
    case 12: elim_step_const<12>(a, negm); break;
    case 13: elim_step_const<13>(a, negm); break;
    case 14: elim_step_const<14>(a, negm); break;
    default: elim_step_const<15>(a, negm); break;
    }
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

template <int NP>
__device__ __forceinline__ unsigned group_max(unsigned v)
{
    v = max(v, dpp_u32<DPP_QUAD_XOR1>(v));
    v = max(v, dpp_u32<DPP_QUAD_XOR2>(v));
    v = max(v, dpp_u32<DPP_ROW_HALF_MIRROR>(v));
    if (NP == 16) v = max(v, dpp_u32<DPP_ROW_MIRROR>(v));
    return v;
}

// magnitude key: monotone in |v| (top 32 bits of the IEEE pattern; for fp64 ties within 2^-20 relative pick the lowest row)
__device__ __forceinline__ unsigned mag_key(double v) { return (unsigned)(__double_as_longlong(v) >> 32) & 0x7fffffffu; }
__device__ __forceinline__ unsigned mag_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ bool key_not_finite(double, unsigned k) { return k >= 0x7ff00000u; }
__device__ __forceinline__ bool key_not_finite(float, unsigned k) { return k >= 0x7f800000u; }

__device__ __forceinline__ double absval(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float absval(float v) { return __builtin_fabsf(v); }

__device__ __forceinline__ double recip(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float recip(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}

// waves per SIMD the register allocator is asked to fit: two register sets (current + prefetched matrices) must stay in
// VGPRs -- a spilled accumulator costs far more than the lost occupancy (measured at n = 16, f64: 4.7 -> 5.7 TB/s)
constexpr int rowlane_occupancy(size_t elem, int np, bool full)
{
    return elem == 8 ? (np == 16 ? (full ? 3 : 2) : (full ? 5 : 4)) : (np == 16 ? (full ? 5 : 4) : 6);
}

// SPD = the Cholesky entry point for n <= 16 (the reference benchmarks its Cholesky kernels at 8x8 and 16x16): the same
// elimination in natural order -- no search, no multiplier test: on SPD input every pivot is positive and the sweep is
// stable --, only the LOWER triangle is USED (the load stays the coalesced full-matrix one; every lane then replaces its
// upper entries by the mirrored lower ones through a padded LDS tile of its wave: 2 LDS accesses per element against an
// uncoalesced second address pattern that cost 40 % at n = 16) and a pivot that is not positive marks the matrix as not
// positive definite (info = that column, as the Cholesky factorisation would).
// GP = the fused Gaussian-process scalars for n <= 16 (SPD mode on M = B + diag c): lane i ends up with row i of M^-1, so
// a^T M^-1 d = sum_i a_i (sum_c X[i][c] d_c) is 16 DPP-broadcast FMAs and one sum over the lanes of the matrix; nothing
// but ONE scalar per item is written.
template <class T>
struct RowlaneGp {
    const T *a, *c, *d, *e;  // d == nullptr: variance, out = e - a^T M^-1 a
    T *out;
};

template <class T, int NP, bool FULL, bool SPD = false, bool GP = false>
__global__ __launch_bounds__(ROWLANE_THREADS, rowlane_occupancy(sizeof(T), NP, FULL)) void matinv_gj_rowlane(BatchRef<const T> Ain, BatchRef<T> Xout, int *info,
                                                                     int n_rt, unsigned batch, RowlaneGp<T> gp)
{
    static_assert(!GP || SPD, "the fused pipeline runs the SPD elimination");
    constexpr int GPW = 64 / NP;  // matrices per wavefront
    const int n = FULL ? NP : n_rt;
    const int lane = threadIdx.x & 63;
    const int g = lane / NP, i = lane % NP;
    const unsigned waves_per_block = ROWLANE_THREADS / 64;
    const unsigned wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const unsigned wave_stride = gridDim.x * waves_per_block;
    const unsigned n_waves = (batch + GPW - 1) / GPW;

    __shared__ T sym_tile[SPD ? (ROWLANE_THREADS / 64) * 64 * (NP + 1) : 1];
    T *tile = sym_tile + (SPD ? (threadIdx.x >> 6) * 64 * (NP + 1) : 0);
    const bool row_in = FULL || i < n;
    // Software prefetch: the matrices of the wave's NEXT grid-stride iteration are loaded into a second register set
    // before the current ones are eliminated, so their HBM latency hides behind ~600 VALU instructions instead of being
    // exposed once per iteration (the kernel was measured 79 % waiting on memory without it).
    T nxt[NP];
    T nxt_a = 0, nxt_c = 0, nxt_d = 0;
    auto load = [&](unsigned w) {
        const unsigned mat = w * GPW + g;
        const bool valid = mat < batch;
        const T *A = Ain.at(valid ? mat : batch - 1);
        if (GP) {
            const bool in = valid && row_in;
            const size_t off = (size_t)(valid ? mat : 0) * n + (row_in ? i : 0);
            nxt_a = in ? gp.a[off] : (T)0;
            nxt_c = in ? gp.c[off] : (T)0;
            nxt_d = in ? (gp.d ? gp.d[off] : nxt_a) : (T)0;
        }
#pragma unroll
        for (int c = 0; c < NP; ++c)
            nxt[c] = (valid && row_in && (FULL || c < n)) ? A[c * n + i] : ((i == c) ? (T)1 : (T)0);
    };
    if (wave0 < n_waves) load(wave0);

    for (unsigned w = wave0; w < n_waves; w += wave_stride) {
        const unsigned mat = w * GPW + g;
        const bool valid = mat < batch;
        T *X = GP ? nullptr : Xout.at(valid ? mat : batch - 1);

        T a[NP];
#pragma unroll
        for (int c = 0; c < NP; ++c) a[c] = nxt[c];
        const T va = nxt_a, vd = nxt_d;
        if (GP) {  // addDiagonal (gauss_bench.cu:38-43)
#pragma unroll
            for (int c = 0; c < NP; ++c) a[c] += (i == c) ? nxt_c : (T)0;
        }
        if (w + wave_stride < n_waves) load(w + wave_stride);
        if (SPD) {  // upper triangle <- mirror of the lower one (whatever the caller left there is never used)
#pragma unroll
            for (int c = 0; c < NP; ++c)
                if (c <= i) tile[lane * (NP + 1) + c] = a[c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < NP; ++c)
                if (c > i) a[c] = tile[(g * NP + c) * (NP + 1) + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        int src = i;            // lane j: source column of output column j (composite of the undone row swaps)
        int bad = 0;            // k+1 of the first step without a usable pivot
        bool any_swap = false;  // wave-uniform
        T rowscale = (T)1;      // 1/pivot of the step in which this lane's row was the pivot row

#ifdef ROWLANE_DBG_NO_COMPUTE
        for (int k = 0; k < 0; ++k) {
#else
#pragma unroll
        for (int k = 0; k < NP; ++k) {
#endif
            // Threshold partial pivoting (same policy as the tile kernels): the diagonal entry is accepted as pivot when
            // no multiplier of the step exceeds ROWLANE_TAU in magnitude -- always the case for diagonally dominant input,
            // at the price of one compare. Otherwise (also for a zero / NaN pivot: the multipliers are then Inf / NaN) the
            // column maximum is searched and brought to row k, i.e. classical partial pivoting for this step.
            T piv = bcast_lane<NP>(a[k], k);
            T inv = recip(piv);
            T negm = (i == k) ? (T)0 : -(a[k] * inv);
            if (SPD && !(piv > 0) && bad == 0) bad = k + 1;  // not positive definite (NaN included)
            const bool viol = !SPD && !(absval(negm) <= (T)ROWLANE_TAU);
            if (!SPD && __any(viol)) {
                // pivot search in column k over rows >= k of this lane's matrix
                const unsigned key = (i >= k) ? mag_key(a[k]) : 0u;
                const unsigned mx = group_max<NP>(key);
                const bool is_max = (key == mx) && (i >= k);
                const unsigned long long vote = __ballot(is_max);
                const unsigned gbits = (unsigned)(vote >> (g * NP)) & ((1u << NP) - 1u);
                const int p = __builtin_ctz(gbits | 0x80000000u);  // lowest row attaining the maximum
                const bool singular = (mx == 0u) || key_not_finite(T(0), mx);
                if (singular && bad == 0) bad = k + 1;
                const bool need_swap = (p != k) && !singular;
                if (__any(need_swap)) {
                    any_swap = true;
                    const int partner = need_swap ? ((i == k) ? p : (i == p) ? k : i) : i;
                    const int from = g * NP + partner;
#pragma unroll
                    for (int c = 0; c < NP; ++c) a[c] = __shfl(a[c], from);
                    // rows >= k have not been pivot rows yet: their rowscale is still 1, nothing else to move
                    if (need_swap) src = (src == k) ? p : (src == p) ? k : src;
                }
                piv = bcast_lane<NP>(a[k], k);
                inv = recip(piv);
                negm = (i == k) ? (T)0 : -(a[k] * inv);
            }
            // eliminate: a[i][c] -= (a[i][k] / pivot) * a[k][c] for every other row; the pivot row keeps its values
            elim_step(a, negm, k);
            a[k] = (i == k) ? (T)1 : negm;
            rowscale = (i == k) ? inv : rowscale;
        }
        // the deferred normalisation: row i was the pivot row of exactly one step
#pragma unroll
        for (int c = 0; c < NP; ++c) a[c] *= rowscale;

        // 3. store; undo the row swaps as a column permutation of the addresses
        const bool fail = bad != 0;
        if (GP) {
            T t = 0;
#pragma unroll
            for (int c = 0; c < NP; ++c) t = fma(a[c], bcast_lane<NP>(vd, c), t);  // (M^-1 d)_i
            T sres = va * t;
#pragma unroll
            for (int off = 1; off < NP; off <<= 1) sres += __shfl_xor(sres, off);  // sum over the lanes of the matrix
            if (valid && i == 0) gp.out[mat] = fail ? nan_of<T>() : (gp.d ? sres : gp.e[mat] - sres);
        } else if (!any_swap) {
#pragma unroll
            for (int c = 0; c < NP; ++c)
                if (valid && row_in && (FULL || c < n)) X[c * n + i] = fail ? nan_of<T>() : a[c];
        } else {
            // lane j pushes j to lane src(j): afterwards lane c holds the output column of register c
            const int dst = __builtin_amdgcn_ds_permute((g * NP + src) << 2, i);
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                const int dcol = bcast_lane<NP>(dst, c);
                if (valid && row_in && (FULL || c < n)) X[dcol * n + i] = fail ? nan_of<T>() : a[c];
            }
        }
        if (info && valid && i == 0) info[mat] = bad;
    }
}

// ------------------------------------------------------------------------------------------------
template <class T>
bool rowlane_family_supports(int n) { return n >= 1 && n <= 16; }
template bool rowlane_family_supports<double>(int);
template bool rowlane_family_supports<float>(int);

template <class T, int NP, bool FULL, bool SPD, bool GP = false>
static hipError_t launch_one(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                             RowlaneGp<T> gp = RowlaneGp<T>())
{
    const unsigned gpw = 64 / NP;
    const size_t waves = (batch + gpw - 1) / gpw;
    const size_t blocks = (waves + ROWLANE_THREADS / 64 - 1) / (ROWLANE_THREADS / 64);
    // persistent-style grid: as many 4-wave blocks as stay resident (one wave per SIMD per block), each wave strides
    // over the batch; every wave then gets within one iteration of the same work and always has a next matrix to prefetch
    static const unsigned per_cu = []() {
        const char *s = getenv("MATINV_ROWLANE_BLOCKS_PER_CU");  // tuning knob for profiling
        return (unsigned)(s && atoi(s) > 0 ? atoi(s) : 0);
    }();
    const unsigned resident = 256u * (per_cu ? per_cu : (unsigned)rowlane_occupancy(sizeof(T), NP, FULL));
    const unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
    hipLaunchKernelGGL((matinv_gj_rowlane<T, NP, FULL, SPD, GP>), dim3(grid), dim3(ROWLANE_THREADS), 0, stream, A, X, info, n,
                       (unsigned)batch, gp);
    return hipGetLastError();
}

template <class T, bool SPD>
static hipError_t launch_rowlane(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!rowlane_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    if (n == 16) return launch_one<T, 16, true, SPD>(n, A, X, batch, info, stream);
    if (n == 8) return launch_one<T, 8, true, SPD>(n, A, X, batch, info, stream);
    if (n < 8) return launch_one<T, 8, false, SPD>(n, A, X, batch, info, stream);
    return launch_one<T, 16, false, SPD>(n, A, X, batch, info, stream);
}
template <class T>
hipError_t launch_gj_rowlane(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_rowlane<T, false>(n, A, X, batch, info, stream);
}
template <class T>
hipError_t launch_spd_rowlane(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_rowlane<T, true>(n, A, X, batch, info, stream);
}
template hipError_t launch_gj_rowlane<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_gj_rowlane<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);
template hipError_t launch_spd_rowlane<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_spd_rowlane<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);

template <class T>
hipError_t launch_gp_rowlane(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                             int *info, hipStream_t stream)
{
    if (!rowlane_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    BatchRef<const T> A{Bs, (size_t)n * n, nullptr};
    BatchRef<T> X{nullptr, 0, nullptr};
    const RowlaneGp<T> gp{As, Cs, Ds, Es, out};
    if (n == 16) return launch_one<T, 16, true, true, true>(n, A, X, batch, info, stream, gp);
    if (n == 8) return launch_one<T, 8, true, true, true>(n, A, X, batch, info, stream, gp);
    if (n < 8) return launch_one<T, 8, false, true, true>(n, A, X, batch, info, stream, gp);
    return launch_one<T, 16, false, true, true>(n, A, X, batch, info, stream, gp);
}
template hipError_t launch_gp_rowlane<double>(int, const double *, const double *, const double *, const double *,
                                              const double *, double *, size_t, int *, hipStream_t);
template hipError_t launch_gp_rowlane<float>(int, const float *, const float *, const float *, const float *, const float *,
                                             float *, size_t, int *, hipStream_t);

// as rocprofv3 prints the instantiations (default template arguments spelled out)
static const char *rowlane_name(bool f64, int n, bool spd)
{
    static thread_local char buf[80];
    snprintf(buf, sizeof buf, "matinv_gj_rowlane<%s, %d, %s, %s, false>", f64 ? "double" : "float", n <= 8 ? 8 : 16,
             (n == 8 || n == 16) ? "true" : "false", spd ? "true" : "false");
    return buf;
}
const char *name_spd_rowlane(bool f64, int n) { return rowlane_name(f64, n, true); }
const char *name_gj_rowlane(bool f64, int n) { return rowlane_name(f64, n, false); }

}  // namespace matinv
