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
// appended to a device work list and redone, in the same stream, by the partially pivoted ROW kernel
// (row_kernels.hip; the LDS kernel for n > 64) -- no host round trip. Diagonally dominant / SPD batches (the reference's fixtures,
// tests/generate_inverse_matrices.m:12-18) never take the fallback.
//
// Replaces the 3n launches of /root/reference/src/gauss/batched_invert.cu:84-95.
#include <stdio.h>
#include <stdlib.h>

#include "tile_common.hpp"

namespace matinv {

// ---- pieces of one block step ---------------------------------------------------------------------------------

// 1. the 4 pivot columns of block kb -> LDS, [row][4]. They live in the 16 lanes c in [c0, c0+4) of tile column tK.
template <int NT, class T>
__device__ __forceinline__ void panel_to_lds(T *panel, const typename TileGeo<T>::vec4 (&acc)[NT][NT], int kb, int q, int c)
{
    typedef TileGeo<T> G;
    const int tK = kb >> 2, rK = kb & 3;
    if (G::blk(c) == rK) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) panel[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][tK][r];
    }
}

// 5.+6. B operand (pivot rows as they stand, I_4 on the pivot columns) and C operand (zero on the pivot columns: the
// MFMA then leaves Aop * I_4 = the new K columns there; zero on the pivot rows: they become D^-1 * W[K,:], a pure
// product -- no cancellation, and the step stays exactly equivariant under power-of-two scaling of the input).
template <int NT, class T>
__device__ __forceinline__ void prep_operands(typename TileGeo<T>::vec4 (&acc)[NT][NT], T (&bop)[NT], int kb, int q, int c)
{
    typedef TileGeo<T> G;
    const int tK = kb >> 2, rK = kb & 3;
    const bool panel_lane = G::blk(c) == rK;
    const bool diag_lane = panel_lane && (G::piv(c) == q);
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) bop[tj] = acc[tK][tj][rK];
    bop[tK] = panel_lane ? (diag_lane ? (T)1 : (T)0) : bop[tK];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ti][tK][r] = panel_lane ? (T)0 : acc[ti][tK][r];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[tK][tj][rK] = (T)0;
}

// One matrix per wavefront; see the file header. T = double or float.
// (Tried, not kept: letting a matrix that has already failed the acceptance test skip the remaining block steps through
// nested scalar branches after every fourth step -- no loop exit, accumulators dead on the rejected path. hipcc answers the
// control flow with 256 VGPRs + 344 B of scratch in the headline kernel instead of 212 and none.)
// (Tried and measured, not kept: pinning "B operand = copy of the pivot-row register, then zero it" as two asm moves per tile
// column -- hipcc copies the whole 4-register tile instead, 768 v_mov per 64x64 matrix. The asm version issues 173 fewer
// VALU instructions per matrix (2 653 -> 2 480, 200 VGPRs instead of 212) and, A/B on one box, is 1 % faster at 64x64
// (1.556 vs 1.575 ms per 100 k), 2-6 % at 48x48. But hipcc inserts no hazard wait states after an asm block: a move inside it
// followed directly by the MFMA that reads the register is a VALU-write -> MFMA-read hazard, and the same change in the
// four-wave kernel did produce wrong results. Not worth 1 %.)
// (Tried and measured, not kept: streaming half of the wave's NEXT matrix into LDS with global_load_lds_dwordx4 during
// the elimination. The exposed time per matrix is load LATENCY, not bytes: 1.651 ms with, 1.645 ms without at 100 k x 64^2.)
template <class T, int NT, bool FULL, bool LOOKAHEAD>
__device__ __forceinline__ void gj_tile_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,
                                             int *work_count, int *work_list, T *panel)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    typedef typename G::vec2 vec2;
    constexpr int N = 16 * NT;
    constexpr int NKB = 4 * NT;
    // PAIRED: 16-byte global accesses. The labels (tile, register, lane) -> (matrix row, matrix column) are ours to
    // choose as long as pivot block kb uses the same index set for its rows (tile row kb/4, register kb%4, q = 0..3)
    // and columns (tile column kb/4, lanes c = 4(kb%4)..+3). With
    //     row(ti, r, q) = 32(ti>>1) + 8r + 2q + (ti&1),    col(tj, c) = 32(tj>>1) + 2c + (tj&1)
    // lane c of the tile-column pair (2u, 2u+1) owns the ADJACENT columns 32u+2c, 32u+2c+1 of its row, i.e. one
    // 16-byte access feeds two tiles and a 16-lane group covers 256 contiguous bytes. Nothing else in the kernel
    // depends on the relabelling (a symmetric permutation of the matrix: inv(P A P^T) = P inv(A) P^T).
    constexpr bool PAIRED = FULL && (NT % 2 == 0);
    const int l = threadIdx.x;

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        // run-time n: made opaque once per matrix, otherwise LICM hoists the 16 NT^2 tile offsets (products with n) and
        // the bounds predicates of both the load and the store loop out of this loop (370-510 VGPRs, one wave per SIMD)
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));
        // Launder the lane coordinates once per matrix: otherwise LICM hoists the ~60 per-lane constants of the 4*NT
        // unrolled block steps (I_4 lanes, e_q entries, lane masks) out of this loop and the allocator spills them.
        int q = l >> 4, c = l & 15;
        // (addresses keep using the un-laundered lane id so they stay in saddr + 32-bit voffset + immediate form)
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c));
        // one per-lane element offset + wave-uniform (compile-time when FULL) tile offsets keep the 16*NT*NT
        // addresses out of VGPRs
        vec4 acc[NT][NT];
        if (PAIRED) {
            // 16-byte accesses: see the index relabelling above (rows/cols of tile pairs interleaved by parity)
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
                        // identity padding beyond n: blockdiag(A, I)^-1 = blockdiag(A^-1, I). Only the last tile row and
                        // column can reach beyond n (n > 16 (NT - 1)): the interior tiles load without a predicate.
                        const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n + 16 * tj);
                        const bool edge = !FULL && (ti == NT - 1 || tj == NT - 1);
                        acc[ti][tj][r] = (!edge || (row < n && col < n)) ? A[uoff + lane_off] : ((row == col) ? (T)1 : (T)0);
                    }
        }
        unsigned long long bad = 0;  // wave-uniform: lanes that saw a multiplier above TAU (or NaN)
        T aop[NT], bop[NT];

#ifdef TILE_DBG_REPEAT
        for (int rep_ = 0; rep_ < TILE_DBG_REPEAT; ++rep_)
#endif
#ifdef TILE_DBG_NO_COMPUTE
        if (false) {
#else
        if (LOOKAHEAD) {
#endif
            panel_to_lds<NT, T>(panel, acc, 0, q, c);
            wave_lds_sync();
            panel_solve<NT>(panel, 0, q, c, aop, bad);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                prep_operands<NT, T>(acc, bop, kb, q, c);
                if (kb + 1 < NKB) {
                    const int tn = (kb + 1) >> 2;
                    // (a) the tile column holding the next pivot columns first ...
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti)
                        acc[ti][tn] = G::mfma(aop[ti], bop[tn], acc[ti][tn]);
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
#ifndef TILE_DBG_NO_MFMA
                                acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
#else
                                acc[ti][tj][0] += aop[ti] * bop[tj];
#endif
                                ++pend;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    issue_b(2);
                    wave_lds_sync();  // panel(kb) has been consumed (aop is in registers)
                    panel_to_lds<NT, T>(panel, acc, kb + 1, q, c);
                    wave_lds_sync();
                    __builtin_amdgcn_sched_barrier(0);
                    T aop_next[NT];
                    PanelSolve<NT, false, T> ps;
                    constexpr int NS = PanelSolve<NT, false, T>::NSTAGE;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        // spread the remaining MFMAs evenly over the stages
                        issue_b(((NB - 2) * (s + 1)) / NS - ((NB - 2) * s) / NS);
#ifndef TILE_DBG_NO_PANEL
                        ps.stage(s, panel, kb + 1, q, c, aop_next, bad);
#else
                        if (s >= 6) aop_next[s - 6] = aop[s - 6] * 0.5;
#endif
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
                            acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
                }
            }
        } else {
#ifndef TILE_DBG_NO_COMPUTE
#pragma unroll
#endif
            for (int kb = 0; kb < (
#ifdef TILE_DBG_NO_COMPUTE
                0
#else
                NKB
#endif
                ); ++kb) {
                panel_to_lds<NT, T>(panel, acc, kb, q, c);
                wave_lds_sync();
                panel_solve<NT>(panel, kb, q, c, aop, bad);
                wave_lds_sync();  // panel is rewritten by the next block step
                prep_operands<NT, T>(acc, bop, kb, q, c);
                // 7. rank-4 update of every tile on the matrix cores
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
            }
        }

#ifdef TILE_DBG_REPEAT
        bad = 0;
#endif
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
        if (LOOKAHEAD) wave_lds_sync();  // the next matrix's first panel write must not pass this one's last reads
    }
}


// FULL: n == 16*NT known at compile time (constant address offsets, no bounds checks).
// LOOKAHEAD: software pipelining across block steps -- the tile column that holds the NEXT pivot columns is updated
// first, the next panel is extracted and solved while the remaining MFMAs of the current step are in flight.
template <int NT, bool FULL, bool LOOKAHEAD>
__global__ __launch_bounds__(64, 2) void matinv_gj_tile_f64(BatchRef<const double> Ain,
                                                                                 BatchRef<double> Xout, int *info,
                                                                                 int n_rt, unsigned batch,
                                                                                 int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) double panel[16 * NT * 4];  // [row][4 pivot columns]
    gj_tile_body<double, NT, FULL, LOOKAHEAD>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel);
}

// fp32 (the reference's DataType): v_mfma_f32_16x16x4_f32, 4 VGPRs per tile (64 at n = 64), same algorithm; the pivot
// blocks follow the f32 accumulator layout (TileGeo<float>).
template <int NT, bool FULL, bool LOOKAHEAD>
__global__ __launch_bounds__(64, FULL ? 4 : 3) void matinv_gj_tile_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info,
                                                           int n_rt, unsigned batch, int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) float panel[16 * NT * 4];
    gj_tile_body<float, NT, FULL, LOOKAHEAD>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel);
}

// ================================================================================================================
// SPD inputs: symmetric blocked sweep on LOWER-TRIANGULAR tile storage (the square-root-free member of the Cholesky
// family: the same Schur complements as A = L L^T, pivots = squares of the Cholesky diagonal, no pivot search needed
// and none wanted). Serves MATINV_ALGO_CHOLESKY for n <= 64 in place of the reference's four Cholesky kernel families
// (/root/reference/src/inverse_cholesky_gpu.cu:55-765); the literal L L^T / L^-1 / L^-T L^-1 phases stay available in
// the LDS family (and behind the reference's sub-phase entry points).
//
// Sweeping the pivot block K (D = W[K,K], P = W[:,K], Q = P D^-1) maps the symmetric W to the symmetric
//     W[I,J] <- W[I,J] - Q[I] P[J]^T,    W[I,K] <- Q[I],    W[K,J] <- Q[J]^T,    W[K,K] <- -D^-1       (I, J not in K)
// and after all blocks W = -A^-1. Only tiles (ti >= tj) are kept: NT(NT+1)/2 tiles = 80 VGPRs at n = 64 instead of 128
// (3 waves per SIMD instead of 2), 10 MFMAs per block step instead of 16, and only the lower triangle is read from HBM.
// Per tile it is the same single MFMA as the Gauss-Jordan kernel: A operand = -Q (rows K: +D^-1, C zeroed), B operand
// = P^T -- by symmetry the OLD panel itself, read back from LDS in the layout it was staged in -- with -I_4 on the K
// columns. Panel rows above the pivot block are not stored as a column: they are the pivot ROWS of tile row tK
// (W[I,K] = W[K,I]^T), which already sit in A-operand lane order.
// The upper triangle of the result is produced at the end by transposing each off-diagonal tile through LDS.
// Rejected (some pivot <= 0: not SPD, or NaN): work list -> LDS Cholesky kernel, which also reports info exactly.
template <class T, int NT, bool FULL>
__device__ __forceinline__ void spd_tile_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,
                                              int *work_count, int *work_list, T *panel)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int NKB = 4 * NT;
    constexpr int TSTRIDE = 17;  // padded row stride of the 16x16 transpose buffer (conflict-free reads)
    const int l = threadIdx.x;

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int n = FULL ? N : n_rt;  // run-time n opaque once per matrix, predicates on the edge tiles only: see gj_tile_body
        if (!FULL) asm volatile("" : "+s"(n));
        int q = l >> 4, c = l & 15;
        const unsigned lane_off = (unsigned)((l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c));  // see matinv_gj_tile_f64

        // W = A^T tile layout as in the Gauss-Jordan kernel; lower tiles only. In the diagonal tiles the strictly
        // upper elements are fetched from their mirror position, so ONLY the lower triangle of A is ever read.
        vec4 acc[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                if (tj > ti) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                    T v;
                    if (ti == tj) {
                        const int hi = row > col ? row : col, lo = row > col ? col : row;
                        // memory element (r_mem, c_mem) of column-major A sits at c_mem*n + r_mem; W[row][col] =
                        // mem[row*n + col] = A[col][row]; its mirror mem[col*n + row]. Lower triangle of A
                        // (r_mem >= c_mem) <=> mem index (small*n + big).
                        v = (FULL || ti < NT - 1 || (row < n && col < n)) ? A[(unsigned)(lo * n + hi)] : ((row == col) ? (T)1 : (T)0);
                    } else {
                        // ti > tj: row > col: W[row][col] = mem[row*n + col] = A[col][row] is in A's UPPER triangle;
                        // take its mirror A[row][col] = mem[col*n + row] instead
                        v = (FULL || ti < NT - 1 || (row < n && col < n)) ? A[(unsigned)(col * n + row)] : (T)0;  // tj < ti <= NT-1
                    }
                    acc[ti][tj][r] = v;
                }
            }
        (void)lane_off;
        unsigned long long bad = 0;
        T aop[NT], bop[NT];

        spd_panel_to_lds<NT, T>(panel, acc, 0, q, c);
        wave_lds_sync();
        {
            PanelSolve<NT, true, T> ps0;
#pragma unroll
            for (int s = 0; s < PanelSolve<NT, true, T>::NSTAGE; ++s) ps0.stage(s, panel, 0, q, c, aop, bop, bad);
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int tK = kb >> 2;
            spd_prep_operands<NT, T>(acc, bop, kb, q, c);
            if (kb + 1 < NKB) {
                const int tn = (kb + 1) >> 2;
                // (a) the tiles the next panel is read from: column tn (ti >= tn) and row tn (tj < tn)
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    if (ti < tn) continue;
                    acc[ti][tn] = G::mfma(aop[ti], bop[tn], acc[ti][tn]);
                }
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    if (tj >= tn) continue;
                    acc[tn][tj] = G::mfma(aop[tn], bop[tj], acc[tn][tj]);
                }
                // (b) the other lower tiles, pinned between the pieces of the next panel: 2 MFMAs cover the latency of
                //     (a), then the panel is staged, then the remaining MFMAs are spread evenly over the solve stages.
                //     Everything below is fully unrolled: the counters fold to literals.
                constexpr int NB = NT * (NT + 1) / 2 - NT;
                constexpr int NS = PanelSolve<NT, true, T>::NSTAGE;
                T aop_next[NT], bop_next[NT];
                PanelSolve<NT, true, T> ps;
                int count = 0, ev = 0;  // MFMAs of (b) issued so far; next event (0 = stage the panel, 1 + s = stage s)
                auto run_events = [&](bool flush) {
#pragma unroll
                    for (int e = 0; e < NS + 1; ++e) {
                        const int lead = NB < 2 ? NB : 2;
                        const int thr = (e == 0) ? lead : lead + ((NB - lead) * e) / NS;
                        if (e == ev && (flush || thr <= count)) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (e == 0) {
                                wave_lds_sync();
                                spd_panel_to_lds<NT, T>(panel, acc, kb + 1, q, c);
                                wave_lds_sync();
                            } else {
                                ps.stage(e - 1, panel, kb + 1, q, c, aop_next, bop_next, bad);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            ++ev;
                        }
                    }
                };
                run_events(false);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        if (tj > ti || ti == tn || tj == tn) continue;
                        acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
                        ++count;
                        run_events(false);
                    }
                run_events(true);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) { aop[ti] = aop_next[ti]; bop[ti] = bop_next[ti]; }
            } else {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        if (tj > ti) continue;
                        acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
                    }
            }
            (void)tK;
        }

        if (bad == 0) {
            // W = -A^-1 (lower tiles). Lower tiles + diagonal tiles go out directly; the mirror of every off-diagonal
            // tile is transposed through LDS so that it, too, is written as 128-byte row segments.
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    if (tj > ti) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                        if (FULL || ti < NT - 1 || (row < n && col < n)) X[(unsigned)(row * n + col)] = -acc[ti][tj][r];
                    }
                    if (tj < ti) {
                        wave_lds_sync();
#pragma unroll
                        for (int r = 0; r < 4; ++r) panel[G::trow(r, q) * TSTRIDE + c] = -acc[ti][tj][r];
                        wave_lds_sync();
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // element (row 16tj + 4r + q, col 16ti + c) of the result = tile(ti,tj)[c][4r + q]
                            const int row = 16 * tj + G::trow(r, q), col = 16 * ti + c;
                            const T v = panel[c * TSTRIDE + G::trow(r, q)];
                            if (FULL || ti < NT - 1 || (row < n && col < n)) X[(unsigned)(row * n + col)] = v;
                        }
                    }
                }
            if (info && l == 0) info[mat] = 0;
        } else if (l == 0) {
            const int slot = atomicAdd(work_count, 1);
            work_list[slot] = (int)mat;
        }
        wave_lds_sync();
    }
}


template <int NT, bool FULL>
__global__ __launch_bounds__(64, NT >= 5 ? 2 : (NT >= 4 ? 3 : 4)) void matinv_spd_tile_f64(BatchRef<const double> Ain, BatchRef<double> Xout,
                                                                          int *info, int n_rt, unsigned batch,
                                                                          int *work_count, int *work_list)
{
    // the LDS buffer serves both as the [row][4] panel and as the padded 16x16 transpose buffer
    __shared__ __attribute__((aligned(16))) double panel[(16 * NT * 4 > 16 * 17) ? 16 * NT * 4 : 16 * 17];
    spd_tile_body<double, NT, FULL>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel);
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, NT >= 7 ? 2 : (NT >= 5 ? 3 : 4)) void matinv_spd_tile_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info,
                                                            int n_rt, unsigned batch, int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) float panel[(16 * NT * 4 > 16 * 17) ? 16 * NT * 4 : 16 * 17];
    spd_tile_body<float, NT, FULL>(Ain, Xout, info, n_rt, batch, work_count, work_list, panel);
}

// ------------------------------------------------------------------------------------------------
// Measured at 100 k x 64^2 f64: 1 / 4 / 16 / 64 rounds -> 1.631 / 1.600 / 1.579 / 1.570 ms per launch: the hardware
// dispatcher balances better than a static stride does, so the grids are (nearly) one workgroup per matrix and the stride
// loop only matters for batches beyond 64 rounds.
unsigned tile_grid_rounds()
{
    static const unsigned rounds = []() {
        const char *s = getenv("MATINV_TILE_GRID_MULT");
        const int v = s && *s ? atoi(s) : 64;
        return (unsigned)(v < 1 ? 1 : v);
    }();
    return rounds;
}

template <class T>
bool tile_family_supports(int n);
template <>
bool tile_family_supports<double>(int n) { return n >= 1 && n <= 128; }
template <>
bool tile_family_supports<float>(int n) { return n >= 1 && n <= 128; }

template <class T>
hipError_t launch_gj_tile(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);

template <>
hipError_t launch_gj_tile<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info,
                                 hipStream_t stream)
{
    if (!tile_family_supports<float>(n)) return hipErrorInvalidValue;
    if (n > 64) return launch_gj_tile4<float>(n, A, X, batch, info, stream);
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int nt = (n + 15) / 16;
    const unsigned grid = (unsigned)(batch < 256u * 16u * tile_grid_rounds() ? batch : 256u * 16u * tile_grid_rounds());
    const unsigned b = (unsigned)batch;
#define TILE_LAUNCH_F32(NT_)                                                                                          \
    if (n == 16 * NT_)                                                                                                \
        hipLaunchKernelGGL((matinv_gj_tile_f32<NT_, true, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
    else                                                                                                              \
        hipLaunchKernelGGL((matinv_gj_tile_f32<NT_, false, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1)
    switch (nt) {
    case 1: TILE_LAUNCH_F32(1); break;
    case 2: TILE_LAUNCH_F32(2); break;
    case 3: TILE_LAUNCH_F32(3); break;
    default: TILE_LAUNCH_F32(4); break;
    }
#undef TILE_LAUNCH_F32
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_gj_row_worklist<float>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

template <>
hipError_t launch_gj_tile<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info,
                                  hipStream_t stream)
{
    if (!tile_family_supports<double>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    if (n > 64) return launch_gj_tile4<double>(n, A, X, batch, info, stream);
    // work list for matrices that fail the acceptance test: [0] = count, [1..batch] = indices (stream-ordered pool)
    int *ws = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int nt = (n + 15) / 16;
    // grid-stride over the batch: enough waves to fill 256 CUs several times over, few enough to amortise setup
    const unsigned grid_mult = tile_grid_rounds();
    const unsigned grid = (unsigned)(batch < 256u * 8u * grid_mult ? batch : 256u * 8u * grid_mult);
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
    else if (lookahead)                                                                                               \
        hipLaunchKernelGGL((matinv_gj_tile_f64<NT_, false, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
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
    if (e == hipSuccess) e = launch_gj_row_worklist<double>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

template <class T>
bool spd_tile_supports(int n);
template <>
bool spd_tile_supports<double>(int n) { return n >= 1 && n <= 128; }  // 64 < n <= 128: tile4_kernels.hip
template <>
bool spd_tile_supports<float>(int n) { return n >= 1 && n <= 128; }

template <class T>
hipError_t launch_spd_tile(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <>
hipError_t launch_spd_tile<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info,
                                  hipStream_t stream)
{
    if (!spd_tile_supports<float>(n)) return hipErrorInvalidValue;
    if (n > 96) return launch_spd_tile4<float>(n, A, X, batch, info, stream);  // one wavefront holds the lower triangle up to 6 x 6 tiles
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int nt = (n + 15) / 16;
    const unsigned grid = (unsigned)(batch < 256u * 16u * tile_grid_rounds() ? batch : 256u * 16u * tile_grid_rounds());
    const unsigned b = (unsigned)batch;
#define SPD_LAUNCH_F32(NT_)                                                                                           \
    if (n == 16 * NT_)                                                                                                \
        hipLaunchKernelGGL((matinv_spd_tile_f32<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
    else                                                                                                              \
        hipLaunchKernelGGL((matinv_spd_tile_f32<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1)
    switch (nt) {
    case 1: SPD_LAUNCH_F32(1); break;
    case 2: SPD_LAUNCH_F32(2); break;
    case 3: SPD_LAUNCH_F32(3); break;
    case 4: SPD_LAUNCH_F32(4); break;
    case 5: SPD_LAUNCH_F32(5); break;
    default: SPD_LAUNCH_F32(6); break;
    }
#undef SPD_LAUNCH_F32
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_chol_lds_worklist<float>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

template <>
hipError_t launch_spd_tile<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info,
                                   hipStream_t stream)
{
    if (!spd_tile_supports<double>(n)) return hipErrorInvalidValue;
    if (n > 96) return launch_spd_tile4<double>(n, A, X, batch, info, stream);  // one wavefront holds the lower triangle up to 6 x 6 tiles
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int nt = (n + 15) / 16;
    const unsigned grid = (unsigned)(batch < 256u * 12u * tile_grid_rounds() ? batch : 256u * 12u * tile_grid_rounds());
    const unsigned b = (unsigned)batch;
#define SPD_LAUNCH(NT_)                                                                                               \
    if (n == 16 * NT_)                                                                                                \
        hipLaunchKernelGGL((matinv_spd_tile_f64<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1); \
    else                                                                                                              \
        hipLaunchKernelGGL((matinv_spd_tile_f64<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, ws, ws + 1)
    switch (nt) {
    case 1: SPD_LAUNCH(1); break;
    case 2: SPD_LAUNCH(2); break;
    case 3: SPD_LAUNCH(3); break;
    case 4: SPD_LAUNCH(4); break;
    case 5: SPD_LAUNCH(5); break;
    default: SPD_LAUNCH(6); break;
    }
#undef SPD_LAUNCH
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_chol_lds_worklist<double>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

const char *name_spd_tile(bool f64, int n)
{
    if (n > 96) return name_tile4(f64, true, n);
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_spd_tile_%s<%d, %s>", f64 ? "f64" : "f32", (n + 15) / 16, (n % 16) == 0 ? "true" : "false");
    return buf;
}

const char *name_gj_tile(bool f64, int n)
{
    if (n > 64) return name_tile4(f64, false, n);
    if (!f64) {
        const bool fullf = (n % 16) == 0;
        switch ((n + 15) / 16) {
        case 1: return fullf ? "matinv_gj_tile_f32<1, true, true>" : "matinv_gj_tile_f32<1, false, true>";
        case 2: return fullf ? "matinv_gj_tile_f32<2, true, true>" : "matinv_gj_tile_f32<2, false, true>";
        case 3: return fullf ? "matinv_gj_tile_f32<3, true, true>" : "matinv_gj_tile_f32<3, false, true>";
        default: return fullf ? "matinv_gj_tile_f32<4, true, true>" : "matinv_gj_tile_f32<4, false, true>";
        }
    }
    const bool full = (n % 16) == 0;
    switch ((n + 15) / 16) {
    case 1: return full ? "matinv_gj_tile_f64<1, true, true>" : "matinv_gj_tile_f64<1, false, true>";
    case 2: return full ? "matinv_gj_tile_f64<2, true, true>" : "matinv_gj_tile_f64<2, false, true>";
    case 3: return full ? "matinv_gj_tile_f64<3, true, true>" : "matinv_gj_tile_f64<3, false, true>";
    default: return full ? "matinv_gj_tile_f64<4, true, true>" : "matinv_gj_tile_f64<4, false, true>";
    }
}

}  // namespace matinv
