// tile4_impl.hpp (instantiated by tile4_kernels.hip for f64 and tile4_f32_kernels.hip for f32 -- two translation units so
// that the two halves compile in parallel) -- kernel family "TILE" for 64 < n <= 128: SEVERAL wavefronts per matrix (1, 2 or 4: t4_waves() below; the
// description is for four).
//
// Same blocked Gauss-Jordan on 16x16 fp64 MFMA accumulator tiles as tile_kernels.hip (read that header first), but a
// 128 x 128 matrix is 64 tiles = 512 VGPRs per lane -- more than one wavefront may hold. A 256-thread workgroup owns a
// matrix; wavefront w holds the tile COLUMNS w, w+4 (all NT tile rows of them, <= 16 tiles = 128 VGPRs). That split
// keeps the two operands of the rank-4 update cheap:
//   * B operand = pivot rows of the wave's own columns = its own accumulator registers, no exchange at all;
//   * A operand = -W[:,K] D^-1 is needed by every wave: the wave that owns tile column kb/4 stages the 4 pivot columns
//     in LDS (4 KB), then all four waves solve the 4x4 pivot block redundantly and form the same Aop (no second
//     exchange, and the acceptance flag comes out identical in every wave).
// Look-ahead: every wave first updates the local tile column that (for the next owner) holds the next pivot columns, the
// next owner stages them into the OTHER half of a double-buffered LDS panel, one workgroup barrier, then the remaining
// MFMAs run pinned between the stages of the next panel's solve. One barrier per block step.
// Rejected matrices (a multiplier above TAU) go to the same device work list and are redone by the pivoted LDS kernel,
// which handles every n this family serves.
//
// Replaces, for 64 < n <= 128, the 3n launches of /root/reference/src/gauss/batched_invert.cu:84-95.
#pragma once
#include <cstdio>

#include "tile_common.hpp"
#include "tile_screen.hpp"

namespace matinv {


// NT = tiles per dimension, T4_WAVES = wavefronts per matrix, NC = tile columns per wave = ceil(NT / T4_WAVES).
// (Measured and not used: NT = 4 with 2 waves per 64 x 64 matrix, 142 VGPRs, 3 waves per SIMD: 4.7e7 inv/s against 6.1e7
// for the one-wave kernel of tile_kernels.hip -- the redundant panel solve and the barriers cost more than the occupancy buys.)
// SPD = the Cholesky entry point for 64 < n <= 128: the same sweep, but only the LOWER triangle of A is read (the upper
// tiles are mirrored while loading, as the Cholesky contract demands -- include/matinv.h), the natural pivots are accepted
// when they are all POSITIVE (leading principal minors of a symmetric matrix: positive definite; no multiplier test, the
// sweep is stable on SPD input), and rejected matrices go to the LDS Cholesky, which reports the failing column.
// (r01 - r03 also ran the fused mean / variance on this kernel -- a GP mode that added diag c while loading and folded a^T M^-1 d out
// of the accumulators. Since r04 every size it served has a lower-tile kernel: gp_tile / gp_spd_tile up to n = 112 / 128,
// spd_tile2_impl.hpp up to 176 in fp64, gp_spd_wide_f32 up to 160 in fp32, the blocked path beyond; the mode and its switch are gone.)
template <class T, int NT, bool FULL, int T4_WAVES, bool SPD>
__device__ __forceinline__ void gj_tile4_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,
                                              int *work_count, int *work_list, T *panel, const int *in_count = nullptr,
                                              const int *in_list = nullptr)
{
    // NT > 8, SPD sweep: no Cholesky kernel behind this one serves every such n, so a matrix that is not positive definite is
    // finished here: info = the column of the first non-positive pivot + 1 (the Cholesky contract), output NaN-filled.
    // (Gauss-Jordan, NT > 8: rejected = needs row exchanges -> work list -> the pivoting kernel of that size, tileq_impl.hpp)
    constexpr bool SELF = NT > 8 && SPD;
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int NKB = 4 * NT;
    constexpr int NC = (NT + T4_WAVES - 1) / T4_WAVES;
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;  // wave-uniform

    // accept-list form (behind the screening kernel, tile_screen.hpp): in_list[0 .. *in_count)
    const unsigned todo = in_count ? (unsigned)*in_count : batch;
    for (unsigned item = blockIdx.x; item < todo; item += gridDim.x) {
        const unsigned mat = in_list ? (unsigned)in_list[item] : item;
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        // run-time n made opaque once per matrix: keeps LICM from hoisting the tile offsets and bounds predicates of the
        // load and store loops out of the batch loop (370-510 VGPRs otherwise)
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));
        int q = l >> 4, c = l & 15;
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c));  // keep LICM from hoisting ~100 per-lane constants (see tile_kernels.hip)

        // acc[ti][jl] = tile (ti, w + 4*jl); W = A^T as in the single-wave kernel
        vec4 acc[NT][NC];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) {
                const int tj = w + T4_WAVES * jl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                    const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n + 16 * tj);
                    // only the last tile row / a wave's last tile column can reach beyond n: interior tiles skip the test
                    const bool edge = !FULL && (ti == NT - 1 || jl == NC - 1);
                    const bool in = (tj < NT) && (!edge || (row < n && col < n));
                    // W = A^T: W[row][col] = A[col][row] at col*... the batch is column-major, so uoff + lane_off addresses
                    // A(col, row); its mirror A(row, col) sits at col * n + row
                    const bool mirror = SPD && (col < row);  // A(col,row) with col < row is an UPPER element: read A(row,col)
                    acc[ti][jl][r] = in ? (mirror ? A[(unsigned)(col * n + row)] : A[uoff + lane_off]) : ((row == col) ? (T)1 : (T)0);
                }
            }
        unsigned long long bad = 0;
        int badinfo = 0;
        T aop[NT], bop[NC];
        // Look-ahead pipeline with ONE workgroup barrier per block step (the panel is double buffered in LDS):
        //   every wave updates its local column jo_n first (for the next owner that is the column holding the next
        //   pivot columns); the next owner stages them; barrier; the other local column is updated while every wave
        //   solves the next panel (MFMAs pinned between the solve stages).
        auto stage_panel = [&](int kb) {
            const int tK = kb >> 2, rK = kb & 3, jo = tK / T4_WAVES;
            T *buf = panel + (kb & 1) * (N * 4);
            if (w == tK % T4_WAVES && G::blk(c) == rK) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][jo][r];
            }
        };
        stage_panel(0);
        __syncthreads();
        {
            PanelSolve<NT, SPD, T> ps0;
            if (SELF) ps0.binfo = &badinfo;
#pragma unroll
            for (int s0 = 0; s0 < PanelSolve<NT, SPD, T>::NSTAGE; ++s0) ps0.stage(s0, panel, 0, q, c, aop, bad);
        }

#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            // ragged n: block steps over identity padding only are no-ops (see gj_tile_body); n is workgroup-uniform, so
            // every wave skips the same (trailing) steps and their barriers
            if (!FULL && kb > 4 * (NT - 1) && kb - 4 * (NT - 1) >= G::real_blocks(n - 16 * (NT - 1))) continue;
            const int tK = kb >> 2, rK = kb & 3;
            const int owner = tK % T4_WAVES, jo = tK / T4_WAVES;  // wave and local column holding the pivot columns
            const bool panel_lane = G::blk(c) == rK;
            const bool diag_lane = panel_lane && (G::piv(c) == q);
            // B operand: pivot rows of the wave's own columns; I_4 on the pivot columns (owner only)
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) bop[jl] = acc[tK][jl][rK];
            if (w == owner) bop[jo] = panel_lane ? (diag_lane ? (T)1 : (T)0) : bop[jo];
            // C operand: zero on the pivot columns (owner) and on the pivot rows (everyone)
            // One asm block per tile row, EXEC narrowed to the owner's pivot-column lanes and the block skipped when that is
            // empty (every other wave): written as a C++ select hipcc emits 64 v_cndmask per step in EVERY wave (146 of
            // ~250 VALU instructions per step; fp64 VALU and MFMA do not overlap on gfx950, so they cost), and with a
            // scalar branch it merges the two paths with 32 register copies and twice the registers.
            {
                const unsigned long long zmask = __ballot((w == owner) && panel_lane);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    unsigned long long save;
                    if constexpr (sizeof(T) == 8)
                        asm volatile("s_and_saveexec_b64 %[save], %[mask]\n\t"
                                     "s_cbranch_execz 1f\n\t"
                                     "v_mov_b64_e32 %0, 0\n\t"
                                     "v_mov_b64_e32 %1, 0\n\t"
                                     "v_mov_b64_e32 %2, 0\n\t"
                                     "v_mov_b64_e32 %3, 0\n\t"
                                     "s_nop 1\n"  // wait states for the MFMA that reads these registers: hipcc pads nothing inside asm
                                     "1:\n\t"
                                     "s_mov_b64 exec, %[save]"
                                     : "+v"(acc[ti][jo][0]), "+v"(acc[ti][jo][1]), "+v"(acc[ti][jo][2]), "+v"(acc[ti][jo][3]),
                                       [save] "=&s"(save)
                                     : [mask] "s"(zmask)
                                     : "scc");
                    else
                        asm volatile("s_and_saveexec_b64 %[save], %[mask]\n\t"
                                     "s_cbranch_execz 1f\n\t"
                                     "v_mov_b32_e32 %0, 0\n\t"
                                     "v_mov_b32_e32 %1, 0\n\t"
                                     "v_mov_b32_e32 %2, 0\n\t"
                                     "v_mov_b32_e32 %3, 0\n\t"
                                     "s_nop 1\n"
                                     "1:\n\t"
                                     "s_mov_b64 exec, %[save]"
                                     : "+v"(acc[ti][jo][0]), "+v"(acc[ti][jo][1]), "+v"(acc[ti][jo][2]), "+v"(acc[ti][jo][3]),
                                       [save] "=&s"(save)
                                     : [mask] "s"(zmask)
                                     : "scc");
                }
            }
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) acc[tK][jl][rK] = (T)0;

            if (kb + 1 < NKB) {
                const int jn = ((kb + 1) >> 2) / T4_WAVES;  // local column updated first
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
                    acc[ti][jn] = G::mfma(aop[ti], bop[jn], acc[ti][jn]);
                stage_panel(kb + 1);
                __syncthreads();
                const T *pnext = panel + ((kb + 1) & 1) * (N * 4);
                constexpr int NS = PanelSolve<NT, SPD, T>::NSTAGE;
                constexpr int NB = NT * (NC - 1);
                T aop_next[NT];
                PanelSolve<NT, SPD, T> ps;
                if (SELF) ps.binfo = &badinfo;
                int count = 0, ev = 0;
                auto run_events = [&](bool flush) {
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        const int thr = (NB * e) / NS;
                        if (e == ev && (flush || thr <= count)) {
                            __builtin_amdgcn_sched_barrier(0);
                            ps.stage(e, pnext, kb + 1, q, c, aop_next, bad);
                            __builtin_amdgcn_sched_barrier(0);
                            ++ev;
                        }
                    }
                };
                run_events(false);
#pragma unroll
                for (int jl = 0; jl < NC; ++jl) {
                    if (jl == jn) continue;
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti) {
                        acc[ti][jl] = G::mfma(aop[ti], bop[jl], acc[ti][jl]);
                        ++count;
                        run_events(false);
                    }
                }
                run_events(true);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) aop[ti] = aop_next[ti];
            } else {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int jl = 0; jl < NC; ++jl)
                        acc[ti][jl] = G::mfma(aop[ti], bop[jl], acc[ti][jl]);
            }
        }
        __syncthreads();  // both panel buffers are free again before the next matrix stages its first panel

        if (bad == 0) {  // identical in all four waves (they evaluate the same D and the same Aop)
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int jl = 0; jl < NC; ++jl) {
                    const int tj = w + T4_WAVES * jl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                        const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n + 16 * tj);
                        const bool edge = !FULL && (ti == NT - 1 || jl == NC - 1);
                        if ((tj < NT) && (!edge || (row < n && col < n))) X[uoff + lane_off] = acc[ti][jl][r];
                    }
                }
            if (info && threadIdx.x == 0) info[mat] = 0;
        } else if (SELF) {
            // (a plain strided fill: written over the tile structure, the 16 NT address offsets of this rare path are
            // hoisted out of the batch loop by LICM and the whole kernel spills -- 4 700 VGPR spills at 12 x 12 tiles)
            for (unsigned e = threadIdx.x; e < (unsigned)(n * n); e += 64u * T4_WAVES) X[e] = nan_of<T>();
            if (info && threadIdx.x == 0) info[mat] = badinfo;
        } else if (threadIdx.x == 0) {
            const int slot = atomicAdd(work_count, 1);
            work_list[slot] = (int)mat;
        }
    }
}

// SPD = the Cholesky entry point for 64 < n <= 128 (see gj_tile4_body).
template <int NT, bool FULL, int T4_WAVES = 4, bool SPD = false>
__global__ __launch_bounds__(64 * T4_WAVES, T4_WAVES > 4 ? 1 : ((NT <= 4) ? 3 : 2)) void matinv_gj_tile4_f64(BatchRef<const double> Ain, BatchRef<double> Xout,
                                                                       int *info, int n_rt, unsigned batch,
                                                                       int *work_count, int *work_list, const int *in_count,
                                                                       const int *in_list)
{
    __shared__ __attribute__((aligned(16))) double panel[2 * 16 * NT * 4];  // double buffered [row][4 pivot columns]
    gj_tile4_body<double, NT, FULL, T4_WAVES, SPD>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel, in_count, in_list);
}
// fp32 (the reference's DataType; its benchmark sweep goes up to n = 128): 16 tiles x 4 VGPRs per wave
template <int NT, bool FULL, int T4_WAVES = 4, bool SPD = false>
__global__ __launch_bounds__(64 * T4_WAVES, T4_WAVES > 4 ? 1 : (T4_WAVES <= 2 ? 2 : 3)) void matinv_gj_tile4_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info,
                                                                       int n_rt, unsigned batch, int *work_count, int *work_list,
                                                                       const int *in_count, const int *in_list)
{
    __shared__ __attribute__((aligned(16))) float panel[2 * 16 * NT * 4];
    gj_tile4_body<float, NT, FULL, T4_WAVES, SPD>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel, in_count, in_list);
}

// Wavefronts per matrix, by measurement (inv/s, FULL Gauss-Jordan, 1 / 2 / 3 / 4 waves; "-" = does not fit or spills badly):
//   f64  n=80: - / 2.20e7 / 2.03e7 / 1.73e7    n=96: - / 1.66e7 / 1.23e7 / 1.16e7    n=112: - / - / 7.9e6 / 9.7e6    n=128: 4
//   f32  n=80: 5.09e7 / 4.17e7 / 3.01e7 / 2.96e7    n=96: 2.99e7 / 2.86e7 / 2.22e7 / 2.22e7
//        n=112: - / 1.90e7 / 1.75e7 / 1.73e7        n=128: - / 1.40e7 / 1.11e7 / 1.45e7
// Every wave repeats the panel solve, so the fewest waves whose tile columns still fit the register file win (f32 n=96 on one
// wave spills in the SPD / pipeline variants -- 2.6e7 and 1.9e7 against 2.7e7 and 2.2e7 on four -- so it takes two).
// Beyond 8 tiles per dimension (r02: the Cholesky entry point and the fused pipeline up to n = 192 in f64, 256 in f32) every
// wavefront holds ONE tile column: NT wavefronts per matrix, one workgroup per CU. An f64 matrix of 12 x 12 tiles is 1 152 of
// the CU's 2 048 VGPRs per lane; 13 x 13 no longer leaves room for the working registers of 13 waves.
// (r03, measured and not kept: fp32 9 ... 12 tiles per dimension with TWO tile columns per wave and (NT + 1) / 2 waves, three waves per
// SIMD so that two matrices fit a CU: 130^2 / 160^2 4.85e6 / 3.76e6 inv/s against 4.95e6 / 3.83e6, and 176^2 / 192^2 1.9e6 / 1.65e6
// against 3.0e6 / 2.66e6 -- 200+ B of scratch there, and every wave's redundant panel solve now serves two columns' worth of waiting)
// (r04, measured and not kept: fp64 9 ... 11 tiles per dimension on FOUR waves of three tile columns, one wave per SIMD on VGPRs + AGPRs
// (498 registers, none spilled at 9 x 9; 15 / 137 spilled at 10 x 10 / 12 x 12): 130^2 2.9e6 inv/s against 3.3e6 with one wave per tile
// column, 160^2 2.1e6 against 2.5e6, 176^2 1.2e6 against 2.0e6 -- fewer redundant panel solves, but nothing left to overlap them with.)
constexpr int t4_waves(bool f64, int nt) { return nt > 8 ? nt : (f64 ? (nt <= 6 ? 2 : 4) : (nt <= 5 ? 1 : (nt <= 7 ? 2 : 4))); }
constexpr int t4_wide_limit(bool f64) { return f64 ? 192 : 256; }

template <class T, bool SPD>
static hipError_t launch_tile4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!(tile4_supports(n) || (n > 128 && n <= t4_wide_limit(sizeof(T) == 8)))) return hipErrorInvalidValue;
    // SPD sweep: only the sizes no lower-tile kernel serves (fp64 176 < n <= 192, fp32 160 < n <= 256) are instantiated since r04
    if (SPD && (n + 15) / 16 < (sizeof(T) == 8 ? 12 : 11)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    // Gauss-Jordan: general batches go straight to the PIVOTING kernel of this size once a natural-order launch of this size
    // has seen most of its matrices rejected (tile_kernels.inc "natural order or pivot search?")
    if (!SPD && tile_policy_use_pivot(sizeof(T) == 8, (n + 15) / 16))
        return n > 128 ? launch_gj_tileq<T>(n, A, X, batch, info, stream, nullptr, nullptr, nullptr, nullptr, nullptr)
                       : launch_gj_tilep4<T>(n, A, X, batch, info, stream);
    // [0], [1] = counts; [2 .. batch+2) = rejected matrices; [batch+2 ..) = (Gauss-Jordan) the singular ones among them; behind the
    // screening pass (r04, tile_screen.hpp: general batches under the default policy) [2 batch + 4] = count of the accepted matrices,
    // then their list
    const int nt = (n + 15) / 16;
    const bool screen = !SPD && tile_policy_use_screen(sizeof(T) == 8, nt);
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), ((screen ? 3 : 2) * batch + 6) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, 2 * sizeof(int), stream);
    if (e == hipSuccess && screen) e = hipMemsetAsync(ws + 2 * batch + 4, 0, sizeof(int), stream);
    if (e != hipSuccess) {
        (void)scratch_free(ws, stream);
        return e;
    }
    int *const acc_count = ws + 2 * batch + 4, *const acc_list = ws + 2 * batch + 5;
    const int *const in_count = screen ? acc_count : nullptr, *const in_list = screen ? acc_list : nullptr;
    const unsigned resident = nt > 8 ? 256u : 256u * 3u;  // NT wavefronts per matrix: one workgroup per CU
    const unsigned sgrid = (unsigned)(batch < 256u * 16u ? batch : 256u * 16u);  // screening: every resident wave takes many matrices
    const unsigned grid = (unsigned)(batch < resident * tile_grid_rounds() ? batch : resident * tile_grid_rounds());
    const unsigned b = (unsigned)batch;
// more than 8 x 8 tiles: run-time n only, f64 up to 12 x 12
#define T4_WIDE(NT_)                                                                                                  \
    if constexpr ((sizeof(T) == 4 || NT_ <= 12) && (!SPD || NT_ >= (sizeof(T) == 8 ? 12 : 11))) {                                                             \
        if constexpr (sizeof(T) == 8)                                                                                 \
            hipLaunchKernelGGL((matinv_gj_tile4_f64<NT_, false, NT_, SPD>), dim3(grid), dim3(64 * NT_), 0, stream, A, X, info, n, b, ws, ws + 2, in_count, in_list); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gj_tile4_f32<NT_, false, NT_, SPD>), dim3(grid), dim3(64 * NT_), 0, stream, A, X, info, n, b, ws, ws + 2, in_count, in_list); \
    }
#define T4_LAUNCH(NT_)                                                                                                \
    if constexpr (SPD) {                                                                                              \
        /* not instantiated: the lower-tile kernels serve every SPD n <= 128 (refused at the top of this function) */     \
    } else if constexpr (sizeof(T) == 8) {                                                                                   \
        if (n == 16 * NT_)                                                                                            \
            hipLaunchKernelGGL((matinv_gj_tile4_f64<NT_, true, t4_waves(true, NT_), SPD>), dim3(grid), dim3(64 * t4_waves(true, NT_)), 0, stream, A, X, info, n, b, ws, ws + 2, in_count, in_list); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gj_tile4_f64<NT_, false, t4_waves(true, NT_), SPD>), dim3(grid), dim3(64 * t4_waves(true, NT_)), 0, stream, A, X, info, n, b, ws, ws + 2, in_count, in_list); \
    } else {                                                                                                          \
        if (n == 16 * NT_)                                                                                            \
            hipLaunchKernelGGL((matinv_gj_tile4_f32<NT_, true, t4_waves(false, NT_), SPD>), dim3(grid), dim3(64 * t4_waves(false, NT_)), 0, stream, A, X, info, n, b, ws, ws + 2, in_count, in_list); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gj_tile4_f32<NT_, false, t4_waves(false, NT_), SPD>), dim3(grid), dim3(64 * t4_waves(false, NT_)), 0, stream, A, X, info, n, b, ws, ws + 2, in_count, in_list); \
    }
#define T4_SCREEN(NT_)                                                                                                \
    if constexpr (!SPD && (sizeof(T) == 4 || NT_ <= 12)) {                                                            \
        if constexpr (sizeof(T) == 8)                                                                                 \
            hipLaunchKernelGGL((matinv_gj_tile4_screen_f64<NT_>), dim3(sgrid), dim3(64), 0, stream, A, n, b, ws, ws + 2, acc_count, acc_list); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gj_tile4_screen_f32<NT_>), dim3(sgrid), dim3(64), 0, stream, A, n, b, ws, ws + 2, acc_count, acc_list); \
    }
    if (screen) {
        switch (nt) {
        case 5: T4_SCREEN(5) break;
        case 6: T4_SCREEN(6) break;
        case 7: T4_SCREEN(7) break;
        case 8: T4_SCREEN(8) break;
        case 9: T4_SCREEN(9) break;
        case 10: T4_SCREEN(10) break;
        case 11: T4_SCREEN(11) break;
        case 12: T4_SCREEN(12) break;
        case 13: T4_SCREEN(13) break;
        case 14: T4_SCREEN(14) break;
        case 15: T4_SCREEN(15) break;
        default: T4_SCREEN(16) break;
        }
    }
#undef T4_SCREEN
    switch (nt) {
    case 5: T4_LAUNCH(5) break;
    case 6: T4_LAUNCH(6) break;
    case 7: T4_LAUNCH(7) break;
    case 8: T4_LAUNCH(8) break;
    case 9: T4_WIDE(9) break;
    case 10: T4_WIDE(10) break;
    case 11: T4_WIDE(11) break;
    case 12: T4_WIDE(12) break;
    case 13: T4_WIDE(13) break;
    case 14: T4_WIDE(14) break;
    case 15: T4_WIDE(15) break;
    default: T4_WIDE(16) break;
    }
#undef T4_LAUNCH
#undef T4_WIDE
    e = hipGetLastError();
    if (e == hipSuccess) {
        if (SPD) {  // beyond 8 x 8 tiles the SPD kernel finishes its rejects itself
            if (nt <= 8) e = launch_chol_lds_worklist<T>(n, A, X, ws, ws + 2, info, stream);
        } else if (nt <= 8) {  // rejected = needs row exchanges: the PIVOTING kernel of this size, in the same stream
            e = launch_gj_tilep4_worklist<T>(n, A, X, batch, ws, ws + 2, ws + 1, ws + 2 + batch, info, stream,
                                             tile_policy_record(sizeof(T) == 8, nt, batch), screen);
        } else {
            e = launch_gj_tileq<T>(n, A, X, batch, info, stream, ws, ws + 2, tile_policy_record(sizeof(T) == 8, nt, batch), nullptr, nullptr);
        }
    }
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

template <class T>
hipError_t launch_gj_tile4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tile4<T, false>(n, A, X, batch, info, stream);
}
template <class T>
hipError_t launch_spd_tile4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tile4<T, true>(n, A, X, batch, info, stream);
}
}  // namespace matinv
