// tile_common.hpp -- pieces shared by the MFMA-tile kernel families (tile_kernels.hip: one wavefront per matrix,
// n <= 64; tile4_kernels.hip: four wavefronts per matrix, 64 < n <= 128).
#pragma once
#include "common.hpp"

namespace matinv {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr double TILE_TAU = 4.0;

// Scalar-type traits of the tile kernels. The two 16x16x4 MFMAs differ in their C/D lane map (checked on hardware with
// tools/mfma_layout_check.hip):   f64: tile row = 4*reg + (lane>>4)      f32: tile row = 4*(lane>>4) + reg
// (column = lane&15 for both). A pivot block must be the 4 rows that ONE accumulator register holds across the four lane
// groups q = 0..3 (that is what makes the B operand free), and its 4 columns must carry the same tile-local indices:
//   f64: block b = rows/cols 4b .. 4b+3          (lane c belongs to block c>>2, is pivot number c&3)
//   f32: block b = rows/cols b, b+4, b+8, b+12   (lane c belongs to block c&3,  is pivot number c>>2)
// i.e. the f32 kernels eliminate in a permuted order -- a symmetric relabelling, invisible in the result.
template <class T>
struct TileGeo;
template <>
struct TileGeo<double> {
    typedef v4d vec4;
    typedef v2d vec2;
    static __device__ __forceinline__ int trow(int r, int q) { return 4 * r + q; }
    static __device__ __forceinline__ int blk(int c) { return c >> 2; }
    static __device__ __forceinline__ int piv(int c) { return c & 3; }
    // tile-local column of pivot t in block rK; register / lane group of the tile-local row s (inverse of trow)
    static __device__ __forceinline__ int pcol(int rK, int t) { return 4 * rK + t; }
    // ragged n: blocks of the LAST tile column that hold at least one real column when rem = n - 16 (NT - 1) of its columns
    // are real (blocks from this number on are identity padding only and are not run)
    static __device__ __forceinline__ int real_blocks(int rem) { return (rem + 3) >> 2; }
    static __device__ __forceinline__ int slot_r(int s) { return (s >> 2) & 3; }
    static __device__ __forceinline__ int slot_q(int s) { return s & 3; }
    static __device__ __forceinline__ vec4 mfma(double a, double b, vec4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <>
struct TileGeo<float> {
    typedef v4f vec4;
    typedef v2f vec2;
    static __device__ __forceinline__ int trow(int r, int q) { return 4 * q + r; }
    static __device__ __forceinline__ int blk(int c) { return c & 3; }
    static __device__ __forceinline__ int piv(int c) { return c >> 2; }
    static __device__ __forceinline__ int pcol(int rK, int t) { return 4 * t + rK; }
    static __device__ __forceinline__ int real_blocks(int rem) { return rem < 4 ? rem : 4; }  // block rK = columns rK, rK + 4, ...
    static __device__ __forceinline__ int slot_r(int s) { return s & 3; }
    static __device__ __forceinline__ int slot_q(int s) { return (s >> 2) & 3; }
    static __device__ __forceinline__ vec4 mfma(float a, float b, vec4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// LDS ordering inside a ONE-wavefront workgroup: the LDS executes a wave's DS instructions in issue order, so a ds_read
// after a ds_write needs no counter wait -- only the compiler must keep the order. __syncthreads() would also work but
// is a workgroup-scope fence: it drains vmcnt, i.e. it would wait for the asynchronous LDS-DMA prefetch of the next
// matrix (and for the previous matrix's stores) in the middle of the elimination.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

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
__device__ __forceinline__ void note_fail(unsigned long long &bad, float v)
{
    asm volatile("v_cmp_nle_f32_e64 vcc, |%1|, 4.0\n\ts_or_b64 %0, %0, vcc" : "+s"(bad) : "v"(v) : "vcc");
}
__device__ __forceinline__ float fast_rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}

// 2.-4. read the pivot block D and this lane's panel rows from LDS, invert D (column q), form the A operand
//       aop[ti] = Aop[16ti + c][q] and update the acceptance flag. Split into NSTAGE pieces of roughly equal
//       VALU/LDS work so the look-ahead loop can issue one MFMA of the CURRENT block step between two pieces of the
//       NEXT step's panel (hardware issues in order: MFMA, ~64 cycles of VALU, MFMA, ... keeps both pipes busy).
// !(v > 0) accumulated like note_fail: SPD mode rejects a non-positive (or NaN) pivot
__device__ __forceinline__ void note_nonpositive(unsigned long long &bad, double v)
{
    asm volatile("v_cmp_ngt_f64_e64 vcc, %1, 0\n\ts_or_b64 %0, %0, vcc" : "+s"(bad) : "v"(v) : "vcc");
}
__device__ __forceinline__ void note_nonpositive(unsigned long long &bad, float v)
{
    asm volatile("v_cmp_ngt_f32_e64 vcc, %1, 0\n\ts_or_b64 %0, %0, vcc" : "+s"(bad) : "v"(v) : "vcc");
}

// the same, and binfo = val at the FIRST pivot that fails (binfo stays 0 while every pivot was positive). All in scalar
// instructions inside one asm block: written as a C++ select on `bad` hipcc spills hundreds of SGPRs in the 12 x 12 tile kernel.
__device__ __forceinline__ void note_nonpositive_first(unsigned long long &bad, int &binfo, double v, int val)
{
    int tmp;
    asm volatile("v_cmp_ngt_f64_e64 vcc, %[v], 0\n\t"
                 "s_cmp_eq_u64 %[bad], 0\n\t"
                 "s_cselect_b32 %[tmp], %[val], 0\n\t"
                 "s_cmp_lg_u64 vcc, 0\n\t"
                 "s_cselect_b32 %[tmp], %[tmp], 0\n\t"
                 "s_or_b32 %[binfo], %[binfo], %[tmp]\n\t"
                 "s_or_b64 %[bad], %[bad], vcc"
                 : [bad] "+s"(bad), [binfo] "+s"(binfo), [tmp] "=&s"(tmp)
                 : [v] "v"(v), [val] "s"(val)
                 : "vcc", "scc");
}
__device__ __forceinline__ void note_nonpositive_first(unsigned long long &bad, int &binfo, float v, int val)
{
    int tmp;
    asm volatile("v_cmp_ngt_f32_e64 vcc, %[v], 0\n\t"
                 "s_cmp_eq_u64 %[bad], 0\n\t"
                 "s_cselect_b32 %[tmp], %[val], 0\n\t"
                 "s_cmp_lg_u64 vcc, 0\n\t"
                 "s_cselect_b32 %[tmp], %[tmp], 0\n\t"
                 "s_or_b32 %[binfo], %[binfo], %[tmp]\n\t"
                 "s_or_b64 %[bad], %[bad], vcc"
                 : [bad] "+s"(bad), [binfo] "+s"(binfo), [tmp] "=&s"(tmp)
                 : [v] "v"(v), [val] "s"(val)
                 : "vcc", "scc");
}

// SPD = true: symmetric blocked sweep for SPD input (see matinv_spd_tile_f64). Same arithmetic for D^-1 and Aop; the
// acceptance test becomes "all four pivots of D positive" (they are the squares of the Cholesky diagonal), and the
// stage of tile row ti also returns bsym[ti] = P[16ti + c][q], the B operand by symmetry (W[K, J] = W[J, K]^T).
template <int NT, bool SPD = false, class T = double>
struct PanelSolve {
    typedef TileGeo<T> G;
    static constexpr int NSTAGE = 6 + NT;
    int *binfo = nullptr;  // SPD: when set, *binfo becomes (column of the FIRST non-positive pivot) + 1 (scalar selects only)
    T d[4][4];
    T r0, r1, r2, r3, l10, l20, l30, l21, l31, l32, u11, u12, u13, u22, u23, u33;
    T a21, a22, a23, a31, a32, a33, b32, b33, y0, y1, y2, y3, x0, x1, x2, x3;

    __device__ __forceinline__ void stage(int s, const T *panel, int kb, int q, int c, T (&aop)[NT],
                                          unsigned long long &bad)
    {
        T unused[NT];
        stage(s, panel, kb, q, c, aop, unused, bad);
    }
    __device__ __forceinline__ void stage(int s, const T *panel, int kb, int q, int c, T (&aop)[NT],
                                          T (&bsym)[NT], unsigned long long &bad)
    {
        const int tK = kb >> 2, rK = kb & 3;
        const bool panel_lane = G::blk(c) == rK;
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = panel[(16 * tK + G::trow(rK, i)) * 4 + j];
            // LU of D without pivoting (multipliers checked below)
            r0 = fast_rcp(d[0][0]);
            l10 = d[1][0] * r0, l20 = d[2][0] * r0, l30 = d[3][0] * r0;
        } else if (s == 1) {
            u11 = fma_t(-l10, d[0][1], d[1][1]), u12 = fma_t(-l10, d[0][2], d[1][2]);
            u13 = fma_t(-l10, d[0][3], d[1][3]);
            a21 = fma_t(-l20, d[0][1], d[2][1]), a22 = fma_t(-l20, d[0][2], d[2][2]);
            a23 = fma_t(-l20, d[0][3], d[2][3]);
            a31 = fma_t(-l30, d[0][1], d[3][1]), a32 = fma_t(-l30, d[0][2], d[3][2]);
            a33 = fma_t(-l30, d[0][3], d[3][3]);
            r1 = fast_rcp(u11);
        } else if (s == 2) {
            l21 = a21 * r1, l31 = a31 * r1;
            u22 = fma_t(-l21, u12, a22), u23 = fma_t(-l21, u13, a23);
            b32 = fma_t(-l31, u12, a32), b33 = fma_t(-l31, u13, a33);
            r2 = fast_rcp(u22);
            l32 = b32 * r2;
            u33 = fma_t(-l32, u23, b33);
        } else if (s == 3) {
            r3 = fast_rcp(u33);
            if (SPD) {
                if (binfo) {
                    note_nonpositive_first(bad, *binfo, d[0][0], 16 * tK + G::pcol(rK, 0) + 1);
                    note_nonpositive_first(bad, *binfo, u11, 16 * tK + G::pcol(rK, 1) + 1);
                    note_nonpositive_first(bad, *binfo, u22, 16 * tK + G::pcol(rK, 2) + 1);
                    note_nonpositive_first(bad, *binfo, u33, 16 * tK + G::pcol(rK, 3) + 1);
                } else {
                    note_nonpositive(bad, d[0][0]), note_nonpositive(bad, u11);
                    note_nonpositive(bad, u22), note_nonpositive(bad, u33);
                }
            } else {
                note_fail(bad, l10), note_fail(bad, l20), note_fail(bad, l30);
                note_fail(bad, l21), note_fail(bad, l31), note_fail(bad, l32);
            }
            // a zero / non-finite last pivot needs no test of its own: r3 = inf/NaN makes x, hence every Aop entry
            // outside the pivot rows (0 * inf = NaN included), fail the test in the last stages
        } else if (s == 4) {
            // L y = e_q
            y0 = (q == 0) ? (T)1 : (T)0;
            y1 = fma_t(-l10, y0, (q == 1) ? (T)1 : (T)0);
            y2 = fma_t(-l21, y1, fma_t(-l20, y0, (q == 2) ? (T)1 : (T)0));
            y3 = fma_t(-l32, y2, fma_t(-l31, y1, fma_t(-l30, y0, (q == 3) ? (T)1 : (T)0)));
        } else if (s == 5) {
            // U x = y : x = column q of D^-1
            x3 = y3 * r3;
            x2 = fma_t(-u23, x3, y2) * r2;
            x1 = fma_t(-u13, x3, fma_t(-u12, x2, y1)) * r1;
            x0 = fma_t(-d[0][3], x3, fma_t(-d[0][2], x2, fma_t(-d[0][1], x1, y0))) * r0;
        } else {
            const int ti = s - 6;
            const T *w = &panel[(16 * ti + c) * 4];
            const T w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            T v = -fma_t(w3, x3, fma_t(w2, x2, fma_t(w1, x1, w0 * x0)));
            if (ti == tK) {
                // pivot rows: D^-1 itself (their C operand is zeroed), exempt from the multiplier test
                const int m = G::piv(c);
                const T x01 = (m & 1) ? x1 : x0, x23 = (m & 1) ? x3 : x2;
                const T xm = (m & 2) ? x23 : x01;
                if (!SPD) note_fail(bad, panel_lane ? (T)0 : v);
                v = panel_lane ? xm : v;
            } else {
                if (!SPD) note_fail(bad, v);
            }
            aop[ti] = v;
            if (SPD) {
                const T w01 = (q & 1) ? w1 : w0, w23 = (q & 1) ? w3 : w2;
                bsym[ti] = (q & 2) ? w23 : w01;
            }
        }
    }
};

template <int NT, class T>
__device__ __forceinline__ void panel_solve(const T *panel, int kb, int q, int c, T (&aop)[NT],
                                            unsigned long long &bad)
{
    PanelSolve<NT, false, T> ps;
#pragma unroll
    for (int s = 0; s < PanelSolve<NT, false, T>::NSTAGE; ++s) ps.stage(s, panel, kb, q, c, aop, bad);
}

// ---- symmetric (lower-triangular tile storage) helpers, shared by the SPD inverse and the fused GP kernel ----------
template <int NT, class T>
__device__ __forceinline__ void spd_panel_to_lds(T *panel, const typename TileGeo<T>::vec4 (&acc)[NT][NT], int kb, int q, int c)
{
    typedef TileGeo<T> G;
    const int tK = kb >> 2, rK = kb & 3;
    if (G::blk(c) == rK) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            if (ti < tK) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) panel[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][tK][r];
        }
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        if (ti >= tK) continue;
        panel[(16 * ti + c) * 4 + q] = acc[tK][ti][rK];  // W[16ti + c][pivot q] = W[pivot q][16ti + c]
    }
}

template <int NT, class T>
__device__ __forceinline__ void spd_prep_operands(typename TileGeo<T>::vec4 (&acc)[NT][NT], T (&bop)[NT], int kb, int q, int c)
{
    typedef TileGeo<T> G;
    const int tK = kb >> 2, rK = kb & 3;
    const bool panel_lane = G::blk(c) == rK;
    const bool diag_lane = panel_lane && (G::piv(c) == q);
    bop[tK] = panel_lane ? (diag_lane ? (T)-1 : (T)0) : bop[tK];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        if (ti < tK) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ti][tK][r] = panel_lane ? (T)0 : acc[ti][tK][r];
    }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        if (tj > tK) continue;
        acc[tK][tj][rK] = (T)0;
    }
}

}  // namespace matinv
