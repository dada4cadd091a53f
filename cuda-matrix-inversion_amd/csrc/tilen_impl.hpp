// tilen_impl.hpp (instantiated by tilen_kernels.hip for f64 and tilen_f32_kernels.hip for f32) -- kernel family "TILE",
// natural-order Gauss-Jordan, second generation (r02): the accumulator-tile sweep of tile_kernels.inc (read that header
// first: layout, the one-MFMA-per-tile block step, verified natural pivots, work list for rejected matrices) with the PANEL
// solved one row per lane, as the pivoting kernel tilep_impl.hpp does it, instead of redundantly in every lane:
//   * the four pivot columns go to LDS and come back with lane i holding row i (n <= 64 rows, 64 lanes);
//   * four in-place Gauss-Jordan steps of the n x 4 panel: the pivot row's four values reach all lanes as scalars
//     (v_readlane with a COMPILE-TIME lane: the pivots are the natural ones), each lane eliminates its own row: 1 multiply +
//     3 FMAs per step and row, where tile_kernels.inc inverts the 4 x 4 pivot block in every lane (LU, two triangular
//     solves) and then forms a 4-term dot product per row;
//   * afterwards lane i holds Aop[i, 0:4] and the A operand is a 4 x 4 transpose across the four lane groups: 8
//     v_permlane32/16_swap, no second LDS trip;
//   * the acceptance test (threshold pivoting, TAU = 4, scale invariant; NaN / Inf fail it) looks at those final panel
//     entries of the rows outside the pivot block: 4 compares per block step under an EXEC mask instead of 10.
// ~2 100 VALU instructions per 64 x 64 matrix instead of 2 585, and 168 VGPRs instead of 202: three waves per SIMD.
// Everything else -- B operand = the accumulator register as it stands, C zeroed on the pivot rows and columns, look-ahead,
// 16-byte global accesses through the symmetric relabelling, rejected matrices to the pivoting kernel -- is unchanged.
//
// Replaces the 3n launches of /root/reference/src/gauss/batched_invert.cu:84-95.
#pragma once
#include "tilep_impl.hpp"

namespace matinv {

// bad |= lanes (of `lanes`) whose |v| is not <= TAU. EXEC is narrowed inside, so lanes outside contribute nothing.
__device__ __forceinline__ void note_fail_masked(unsigned long long &bad, unsigned long long lanes, double v0, double v1, double v2,
                                                 double v3)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %[save], %[lanes]\n\t"
                 "v_cmp_nle_f64_e64 vcc, |%[v0]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "v_cmp_nle_f64_e64 vcc, |%[v1]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "v_cmp_nle_f64_e64 vcc, |%[v2]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "v_cmp_nle_f64_e64 vcc, |%[v3]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [bad] "+s"(bad), [save] "=&s"(save)
                 : [lanes] "s"(lanes), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3)
                 : "vcc", "scc");
}
__device__ __forceinline__ void note_fail_masked(unsigned long long &bad, unsigned long long lanes, float v0, float v1, float v2, float v3)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %[save], %[lanes]\n\t"
                 "v_cmp_nle_f32_e64 vcc, |%[v0]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "v_cmp_nle_f32_e64 vcc, |%[v1]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "v_cmp_nle_f32_e64 vcc, |%[v2]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "v_cmp_nle_f32_e64 vcc, |%[v3]|, 4.0\n\t"
                 "s_or_b64 %[bad], %[bad], vcc\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [bad] "+s"(bad), [save] "=&s"(save)
                 : [lanes] "s"(lanes), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3)
                 : "vcc", "scc");
}

// B operand = the pivot-row register of each tile of tile row tK as it stands, then zeroed; pivot columns zeroed under a
// narrowed EXEC; I_4 on the pivot columns of B (see prep_operands in tile_kernels.inc)
template <int NT, class T>
__device__ __forceinline__ void tilen_prep(typename TileGeo<T>::vec4 (&acc)[NT][NT], T (&bop)[NT], int kb, int q, int c)
{
    typedef TileGeo<T> G;
    const int tK = kb >> 2, rK = kb & 3;
    const bool panel_lane = G::blk(c) == rK;
    const bool diag_lane = panel_lane && (G::piv(c) == q);
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        if constexpr (sizeof(T) == 8)
            asm volatile("v_mov_b64_e32 %0, %1\n\tv_mov_b64_e32 %1, 0\n\ts_nop 1" : "=&v"(bop[tj]), "+v"(acc[tK][tj][rK]));
        else
            asm volatile("v_mov_b32_e32 %0, %1\n\tv_mov_b32_e32 %1, 0\n\ts_nop 1" : "=&v"(bop[tj]), "+v"(acc[tK][tj][rK]));
    }
    bop[tK] = panel_lane ? (diag_lane ? (T)1 : (T)0) : bop[tK];
    const unsigned long long zmask = __ballot(panel_lane);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        unsigned long long save;
        if constexpr (sizeof(T) == 8)
            asm volatile("s_and_saveexec_b64 %[save], %[mask]\n\t"
                         "v_mov_b64_e32 %0, 0\n\t"
                         "v_mov_b64_e32 %1, 0\n\t"
                         "v_mov_b64_e32 %2, 0\n\t"
                         "v_mov_b64_e32 %3, 0\n\t"
                         "s_nop 1\n\t"
                         "s_mov_b64 exec, %[save]"
                         : "+v"(acc[ti][tK][0]), "+v"(acc[ti][tK][1]), "+v"(acc[ti][tK][2]), "+v"(acc[ti][tK][3]), [save] "=&s"(save)
                         : [mask] "s"(zmask)
                         : "scc");
        else
            asm volatile("s_and_saveexec_b64 %[save], %[mask]\n\t"
                         "v_mov_b32_e32 %0, 0\n\t"
                         "v_mov_b32_e32 %1, 0\n\t"
                         "v_mov_b32_e32 %2, 0\n\t"
                         "v_mov_b32_e32 %3, 0\n\t"
                         "s_nop 1\n\t"
                         "s_mov_b64 exec, %[save]"
                         : "+v"(acc[ti][tK][0]), "+v"(acc[ti][tK][1]), "+v"(acc[ti][tK][2]), "+v"(acc[ti][tK][3]), [save] "=&s"(save)
                         : [mask] "s"(zmask)
                         : "scc");
    }
}

template <class T, int NT, bool FULL>
__device__ __forceinline__ void gj_tilen_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,
                                              int *work_count, int *work_list, T *panel)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    typedef typename G::vec2 vec2;
    constexpr int N = 16 * NT;
    constexpr int NKB = 4 * NT;
    constexpr bool PAIRED = FULL && (NT % 2 == 0);  // 16-byte global accesses: see gj_tile_body
    const int l = threadIdx.x;
    // rows that exist (lanes beyond N hold no row: NT < 4)
    const unsigned long long row_lanes = N >= 64 ? ~0ull : ((1ull << (N & 63)) - 1ull);

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));
        int q = l >> 4, c = l & 15, lr = l;
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c), "+v"(lr));
        vec4 acc[NT][NT];
        if (PAIRED) {
            const unsigned lane_off2 = (unsigned)(2 * G::trow(0, l >> 4) * N + 2 * (l & 15));
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int u = 0; u < NT / 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned uoff = (unsigned)((32 * (ti >> 1) + 2 * G::trow(r, 0) + (ti & 1)) * N + 32 * u);
                        const vec2 v = __builtin_nontemporal_load(reinterpret_cast<const vec2 *>(A + uoff + lane_off2));
                        acc[ti][2 * u][r] = v[0];
                        acc[ti][2 * u + 1][r] = v[1];
                    }
        } else {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                        const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n + 16 * tj);
                        const bool edge = !FULL && (ti == NT - 1 || tj == NT - 1);
                        acc[ti][tj][r] = (!edge || (row < n && col < n)) ? A[uoff + lane_off] : ((row == col) ? (T)1 : (T)0);
                    }
        }
        unsigned long long bad = 0;
        T aop[NT], bop[NT];

        // the panel of block kb: stage, one row per lane back, 4 Gauss-Jordan steps with the natural pivots, transpose.
        // `between(s)` is called between its 8 stages (the caller issues MFMAs of the previous block there).
        auto solve_panel = [&](int kb, T (&aout)[NT], auto &&between) {
            const int tK = kb >> 2, rK = kb & 3;
            if (G::blk(c) == rK) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) panel[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][tK][r];
            }
            wave_lds_sync();
            between(0);
            const vec4 wv = *reinterpret_cast<const vec4 *>(&panel[lr * 4]);
            T w[4] = {wv[0], wv[1], wv[2], wv[3]};
            unsigned long long pivots = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int p = 16 * tK + G::trow(rK, t);  // the natural pivot row: a compile-time lane
                pivots |= 1ull << p;
                T u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = lane_value(w[j], p);
                const T rp = fast_rcp(u[t]);
                between(1 + 2 * t);
                const T f = -(w[t] * rp);
                const bool me = lr == p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j == t) continue;
                    const T piv_j = u[j] * rp;  // wave-uniform
                    w[j] = me ? piv_j : fma_t(f, u[j], w[j]);
                }
                w[t] = me ? rp : f;
                if (t < 3) between(2 + 2 * t);
            }
            // acceptance: every entry of the finished panel outside the pivot rows is a multiplier of this block step
            note_fail_masked(bad, row_lanes & ~pivots, w[0], w[1], w[2], w[3]);
            lane_rows_swap<true>(w[0], w[2]);
            lane_rows_swap<true>(w[1], w[3]);
            lane_rows_swap<false>(w[0], w[1]);
            lane_rows_swap<false>(w[2], w[3]);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) aout[ti] = w[ti];
            wave_lds_sync();  // the panel may be rewritten
        };

        solve_panel(0, aop, [](int) {});
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            tilen_prep<NT, T>(acc, bop, kb, q, c);
            if (kb + 1 < NKB) {
                const int tn = (kb + 1) >> 2;
                // (a) the tile column holding the next pivot columns first ...
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][tn] = G::mfma(aop[ti], bop[tn], acc[ti][tn]);
                // (b) ... the other NT (NT - 1) tiles pinned between the stages of the next panel
                constexpr int NB = NT * (NT - 1);
                constexpr int NSLOT = 8;
                int pend = 0;  // folds to a literal: everything here is fully unrolled
                auto issue_b = [&](int count) {
#pragma unroll
                    for (int z = 0; z < count; ++z) {
                        if (pend < NB) {
                            const int tjx = pend / NT, ti = pend % NT;
                            const int tj = tjx + (tjx >= tn ? 1 : 0);
                            acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
                            ++pend;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                __builtin_amdgcn_sched_barrier(0);
                issue_b(2);  // cover the latency of (a) before its results are staged
                T aop_next[NT];
                solve_panel(kb + 1, aop_next, [&](int s) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_b(((NB - 2) * (s + 1)) / NSLOT - ((NB - 2) * s) / NSLOT);
                });
                __builtin_amdgcn_sched_barrier(0);
                issue_b(NB);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) aop[ti] = aop_next[ti];
            } else {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
            }
        }

        if (bad == 0) {
            if (PAIRED) {
                const unsigned lane_off2 = (unsigned)(2 * G::trow(0, l >> 4) * N + 2 * (l & 15));
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int u = 0; u < NT / 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned uoff = (unsigned)((32 * (ti >> 1) + 2 * G::trow(r, 0) + (ti & 1)) * N + 32 * u);
                            vec2 v;
                            v[0] = acc[ti][2 * u][r];
                            v[1] = acc[ti][2 * u + 1][r];
                            __builtin_nontemporal_store(v, reinterpret_cast<vec2 *>(X + uoff + lane_off2));
                        }
            } else {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                            const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n + 16 * tj);
                            const bool edge = !FULL && (ti == NT - 1 || tj == NT - 1);
                            if (!edge || (row < n && col < n)) X[uoff + lane_off] = acc[ti][tj][r];
                        }
            }
            if (info && l == 0) info[mat] = 0;
        } else if (l == 0) {
            const int slot = atomicAdd(work_count, 1);
            work_list[slot] = (int)mat;
        }
    }
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, 3) void matinv_gj_tilen_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                            unsigned batch, int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) double panel[64 * 4];
    gj_tilen_body<double, NT, FULL>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel);
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, 4) void matinv_gj_tilen_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info, int n_rt,
                                                            unsigned batch, int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) float panel[64 * 4];
    gj_tilen_body<float, NT, FULL>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel);
}

// enqueue only: the caller owns the work list (count at work_count, indices at work_list) and what follows it
template <class T>
static hipError_t enqueue_tilen(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                                int *work_count, int *work_list)
{
    const int nt = (n + 15) / 16;
    const unsigned cap = 256u * 12u * tile_grid_rounds();
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    const unsigned b = (unsigned)batch;
#define TN_LAUNCH(NT_)                                                                                                 \
    if constexpr (sizeof(T) == 8) {                                                                                    \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilen_f64<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, work_count, work_list); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilen_f64<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, work_count, work_list); \
    } else {                                                                                                           \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilen_f32<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, work_count, work_list); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilen_f32<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, work_count, work_list); \
    }
    switch (nt) {
    case 1: TN_LAUNCH(1) break;
    case 2: TN_LAUNCH(2) break;
    case 3: TN_LAUNCH(3) break;
    default: TN_LAUNCH(4) break;
    }
#undef TN_LAUNCH
    return hipGetLastError();
}

}  // namespace matinv
