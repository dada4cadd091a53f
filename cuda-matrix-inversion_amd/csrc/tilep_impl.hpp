// tilep_impl.hpp (instantiated by tilep_kernels.hip for f64 and tilep_f32_kernels.hip for f32) -- kernel family "TILEP":
// the MFMA accumulator-tile Gauss-Jordan of tile_kernels.inc (read that header first) with TRUE PARTIAL PIVOTING inside
// the kernel, for GENERAL matrices, n <= 64, one matrix per wavefront.
//
// What pivoting changes. tile_kernels.inc eliminates in natural order: the four pivot rows of block kb are the rows that
// ONE accumulator register (tile row kb/4, register kb%4) holds across the four lane groups, which makes the B operand
// of the rank-4 update free. With a pivot search the four pivot rows of a block are wherever the search finds them:
//   * rows never move (IMPLICIT pivoting): every row keeps its register slot for the whole elimination and the row /
//     column permutation that this leaves behind is folded into the store addresses (two byte tables in LDS);
//   * the search runs on the LDS-staged n x 4 panel with ONE ROW PER LANE (n <= 64 rows, 64 lanes): per pivot column a
//     wave-wide DPP max over the top 32 bits of |x| of the rows not used yet, ballot + s_ff1 for the lowest row attaining
//     it (the oracle's tie rule up to 2^-20 relative), the pivot row's four panel values and its earlier multipliers come
//     back as SCALARS (8 v_readlane per pivot -- against 128 per step in row_kernels.hip), and each lane eliminates its
//     own row: after four steps lane i holds the LU multipliers L_i of its row, and
//           -W[i,K] D^-1 = -L_i L_D^-1        (D = L_D U, the four pivot rows in pivot order)
//     so the A operand is a 4-term dot product per row exactly as in the unpivoted kernel; the pivot lanes publish
//     -U^-1[t,:] instead of their L so that the SAME dot product yields their row of D^-1 -- no per-row select;
//   * the four pivot rows reach the B operand through a 2 KB LDS buffer: a wave-uniform switch on the pivot's register
//     slot (tile row, register) lets the one lane group that holds it store the row and zero it in C. That switch is the
//     only place where a run-time register index is needed, and it is why the block-step loop is unrolled over the tile
//     column only (the register index of the pivot COLUMNS) and rolled over the four blocks inside it.
// Cost per block step at n = 64: ~45 more VALU instructions than the verified-natural-order kernel and two LDS round
// trips instead of one; no acceptance test; a work list only for singular input (no finite non-zero pivot candidate in
// some column), which the ROW kernel redoes for the oracle's info code (this kernel eliminates A^T).
//
// Replaces pivotRow / normalizeRow / transform_matrix of /root/reference/src/gauss/batched_invert.cu:17-82 for inputs that
// need row exchanges (the reference swaps only on an exactly zero diagonal, :19-35; tests/square_5_*.mats are such inputs).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tile_common.hpp"
#include "wave_util.hpp"
#include "gather_tree.inc"

namespace matinv {

template <class T>
struct Vec4Of;
template <>
struct Vec4Of<double> {
    typedef v4d type;
};
template <>
struct Vec4Of<float> {
    typedef v4f type;
};

// B-operand gather of ONE pivot row (slot s = 16 ti + tile-local row) into the LDS row at byte address `bb_addr` (+ lane
// offset): only the lane group that holds the row is active; its registers are zeroed afterwards (C operand of the pivot
// rows: they become D^-1 W[P,:], a pure product). The register that holds the row is known at RUN time only. Written in
// C++ (a switch over the 4 NT slots whose cases modify acc) hipcc merges the cases with whole-tile copies: 2 565 v_mov_b64
// in the 64 x 64 kernel, 256 VGPRs and scratch. So: one asm block per tile row, unconditional for the compiler, that
// branches over itself unless the pivot lives in this tile row, narrows EXEC to the lane group and picks the register with
// scalar compares -- a dozen SALU instructions per pivot around NT ds_write + NT v_mov.
template <class T, int NT, int TI>
__device__ __forceinline__ void gather_zero_tile_row(typename TileGeo<T>::vec4 (&acc)[NT][NT], unsigned addr, int pos,
                                                    unsigned long long mask)
{
    unsigned long long save;
    unsigned tmp;
    // dummy tiles keep the operand list of the asm the same for every NT (columns beyond NT are never touched at run time:
    // their code is there, but the tile they name is a throw-away)
    typename TileGeo<T>::vec4 dummy = {};
#define TP_A(TJ, R) "+v"(((TJ) < NT ? acc[TI][(TJ) < NT ? (TJ) : 0] : dummy)[R])
#define TP_W64(J, R, OFF) "ds_write_b64 %[addr], %[a" #J #R "] offset:" #OFF "\n\t"
#define TP_W32(J, R, OFF) "ds_write_b32 %[addr], %[a" #J #R "] offset:" #OFF "\n\t"
#define TP_Z64(J, R) "v_mov_b64_e32 %[a" #J #R "], 0\n\t"
#define TP_Z32(J, R) "v_mov_b32_e32 %[a" #J #R "], 0\n\t"
#define TP_WZ64_1(R) TP_W64(0, R, 0) TP_Z64(0, R)
#define TP_WZ64_2(R) TP_W64(0, R, 0) TP_W64(1, R, 128) TP_Z64(0, R) TP_Z64(1, R)
#define TP_WZ64_3(R) TP_W64(0, R, 0) TP_W64(1, R, 128) TP_W64(2, R, 256) TP_Z64(0, R) TP_Z64(1, R) TP_Z64(2, R)
#define TP_WZ64_4(R) TP_W64(0, R, 0) TP_W64(1, R, 128) TP_W64(2, R, 256) TP_W64(3, R, 384) TP_Z64(0, R) TP_Z64(1, R) TP_Z64(2, R) TP_Z64(3, R)
#define TP_WZ32_1(R) TP_W32(0, R, 0) TP_Z32(0, R)
#define TP_WZ32_2(R) TP_W32(0, R, 0) TP_W32(1, R, 64) TP_Z32(0, R) TP_Z32(1, R)
#define TP_WZ32_3(R) TP_W32(0, R, 0) TP_W32(1, R, 64) TP_W32(2, R, 128) TP_Z32(0, R) TP_Z32(1, R) TP_Z32(2, R)
#define TP_WZ32_4(R) TP_W32(0, R, 0) TP_W32(1, R, 64) TP_W32(2, R, 128) TP_W32(3, R, 192) TP_Z32(0, R) TP_Z32(1, R) TP_Z32(2, R) TP_Z32(3, R)
#define TP_BODY(WZ)                                                                                                    \
    "s_lshr_b32 %[tmp], %[pos], 2\n\t"                                                                                 \
    "s_cmp_lg_u32 %[tmp], %[ti]\n\t"                                                                                   \
    "s_cbranch_scc1 9f\n\t"                                                                                            \
    "s_and_saveexec_b64 %[save], %[mask]\n\t"                                                                          \
    "s_and_b32 %[tmp], %[pos], 3\n\t"                                                                                  \
    "s_cmp_lg_u32 %[tmp], 0\n\t"                                                                                       \
    "s_cbranch_scc1 1f\n\t" WZ(0) "s_branch 8f\n"                                                                      \
    "1:\n\t"                                                                                                           \
    "s_cmp_lg_u32 %[tmp], 1\n\t"                                                                                       \
    "s_cbranch_scc1 2f\n\t" WZ(1) "s_branch 8f\n"                                                                      \
    "2:\n\t"                                                                                                           \
    "s_cmp_lg_u32 %[tmp], 2\n\t"                                                                                       \
    "s_cbranch_scc1 3f\n\t" WZ(2) "s_branch 8f\n"                                                                      \
    "3:\n\t" WZ(3) "8:\n\t"                                                                                            \
    "s_nop 1\n\t"                                                                                                      \
    "s_mov_b64 exec, %[save]\n"                                                                                        \
    "9:"
#define TP_OPERANDS                                                                                                    \
    [a00] TP_A(0, 0), [a01] TP_A(0, 1), [a02] TP_A(0, 2), [a03] TP_A(0, 3), [a10] TP_A(1, 0), [a11] TP_A(1, 1),        \
        [a12] TP_A(1, 2), [a13] TP_A(1, 3), [a20] TP_A(2, 0), [a21] TP_A(2, 1), [a22] TP_A(2, 2), [a23] TP_A(2, 3),    \
        [a30] TP_A(3, 0), [a31] TP_A(3, 1), [a32] TP_A(3, 2), [a33] TP_A(3, 3), [save] "=&s"(save), [tmp] "=&s"(tmp)
#define TP_EMIT(WZ) asm volatile(TP_BODY(WZ) : TP_OPERANDS : [addr] "v"(addr), [pos] "s"(pos), [mask] "s"(mask), [ti] "n"(TI) : "scc", "memory")
    static_assert(NT >= 1 && NT <= 4, "one wavefront holds at most 4 x 4 tiles");
    if constexpr (sizeof(T) == 8) {
        if constexpr (NT == 1) TP_EMIT(TP_WZ64_1);
        else if constexpr (NT == 2) TP_EMIT(TP_WZ64_2);
        else if constexpr (NT == 3) TP_EMIT(TP_WZ64_3);
        else TP_EMIT(TP_WZ64_4);
    } else {
        if constexpr (NT == 1) TP_EMIT(TP_WZ32_1);
        else if constexpr (NT == 2) TP_EMIT(TP_WZ32_2);
        else if constexpr (NT == 3) TP_EMIT(TP_WZ32_3);
        else TP_EMIT(TP_WZ32_4);
    }
#undef TP_EMIT
#undef TP_OPERANDS
#undef TP_BODY
#undef TP_A
}

// v_permlane32_swap / v_permlane16_swap (gfx950) on one scalar of the tile type: rows of 32 (16) lanes; the odd rows of `a`
// change places with the even rows of `b`.
template <bool WIDE>
__device__ __forceinline__ void lane_rows_swap(unsigned &a, unsigned &b)
{
    if constexpr (WIDE) {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0], b = r[1];
    } else {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0], b = r[1];
    }
}
template <bool WIDE>
__device__ __forceinline__ void lane_rows_swap(float &a, float &b)
{
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    lane_rows_swap<WIDE>(ua, ub);
    a = __uint_as_float(ua), b = __uint_as_float(ub);
}
template <bool WIDE>
__device__ __forceinline__ void lane_rows_swap(double &a, double &b)
{
    unsigned long long xa = (unsigned long long)__double_as_longlong(a), xb = (unsigned long long)__double_as_longlong(b);
    unsigned alo = (unsigned)xa, ahi = (unsigned)(xa >> 32), blo = (unsigned)xb, bhi = (unsigned)(xb >> 32);
    lane_rows_swap<WIDE>(alo, blo);
    lane_rows_swap<WIDE>(ahi, bhi);
    a = __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo));
    b = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
}

// one tile row of the gather, NT tile columns wide, as ONE asm statement with a branch tree over the register (gather_tree.inc)
template <class T, int NT, int TI>
__device__ __forceinline__ void gather_row_tree(typename TileGeo<T>::vec4 (&acc)[NT][NT], unsigned addr, int pos, unsigned long long mask)
{
    if constexpr (NT == 1) gather_tree_1x1<4 * TI>(acc[TI][0], addr, pos, mask);
    else if constexpr (NT == 2) gather_tree_2x1<4 * TI>(acc[TI][0], acc[TI][1], addr, pos, mask);
    else if constexpr (NT == 3) gather_tree_3x1<4 * TI>(acc[TI][0], acc[TI][1], acc[TI][2], addr, pos, mask);
    else gather_tree_4x1<4 * TI>(acc[TI][0], acc[TI][1], acc[TI][2], acc[TI][3], addr, pos, mask);
}

template <int V>
struct IntC {
    static constexpr int value = V;
};

template <class T, int NT, bool FULL>
__device__ __forceinline__ void gj_tilep_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,
                                              T *lds, unsigned char *tab, const int *in_count, const int *in_list, hint_t *hint_out,
                                              int *bad_count, int *bad_list)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    T *const panel = lds;        // [64][4]  the four pivot columns, one row per lane
    T *const bbuf = lds + 256;   // [4][N]   the four pivot rows in B-operand order
    unsigned char *const rowaddr = tab, *const coladdr = tab + 64;
    const int l = threadIdx.x;
    // LDS byte address of this lane's element of pivot row 0 in bbuf (ds_write in gather_zero_tile_row)
    typedef __attribute__((address_space(3))) T *lds_ptr;
    const unsigned bb_lane = (unsigned)(size_t)(lds_ptr)(bbuf + (l & 15));

    // work-list form (the matrices the natural-order kernel rejected): in_list[0 .. *in_count)
    const unsigned todo = in_count ? (unsigned)*in_count : batch;
    // work-list form: the length of the list is what the launcher's natural-order / pivot guess feeds on; it goes back to
    // the host through a store into pinned memory (no copy command in the stream, nobody waits for it)
    if (hint_out && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hint_out, ((hint_t)batch << 32) | todo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (unsigned item = blockIdx.x; item < todo; item += gridDim.x) {
        const unsigned mat = in_list ? (unsigned)in_list[item] : item;
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));  // see gj_tile_body: keeps LICM away from the tile offsets
        int q = l >> 4, c = l & 15, lr = l;
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c), "+v"(lr));

        vec4 acc[NT][NT];
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

        bool used = lr >= N;  // rows that may still become pivots: false. Lanes beyond N hold no row.
        int bad = 0;
        T aop[NT], bop[NT];  // operands of the block step whose MFMAs are still owed

        // One pipeline turn = everything block nb = (tile column tKn [compile time], block rKn inside it [run time]) needs
        // before its MFMAs, interleaved with the MFMAs of the PREVIOUS block (look-ahead, as in tile_kernels.inc):
        //   1. previous block's update of tile column tKn (it holds the pivot columns of block nb)
        //   2. the four pivot columns -> LDS; one row per lane back
        //   3. pivot search + in-place Gauss-Jordan of the n x 4 panel, 12 stages, the other NT (NT-1) MFMAs of the
        //      previous block pinned between them: while a 64-cycle MFMA runs, the scalar half of a stage (reductions,
        //      v_readlane results, ff1, waits) proceeds
        //   4. A operand through LDS (row i of the finished panel IS Aop[i,:]), B operand gather, C zeroing
        auto turn = [&](auto tKc, int rKn, auto firstc) {
            constexpr int tKn = decltype(tKc)::value;
            constexpr bool first = decltype(firstc)::value != 0;
            constexpr int NB = first ? 0 : NT * (NT - 1);
            const bool panel_lane = G::blk(c) == rKn;
            if (!first) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][tKn] = G::mfma(aop[ti], bop[tKn], acc[ti][tKn]);
            }
            int pend = 0;  // folds to a literal: everything here is fully unrolled
            auto issue_b = [&](int count) {
#pragma unroll
                for (int z = 0; z < count; ++z) {
                    if (pend < NB) {
                        const int tjx = pend / NT, ti = pend % NT;
                        const int tj = tjx + (tjx >= tKn ? 1 : 0);
                        acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
                        ++pend;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            issue_b(2);  // cover the latency of (1) before its results are staged
            wave_lds_sync();
            if (panel_lane) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) panel[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][tKn][r];
            }
            wave_lds_sync();
            __builtin_amdgcn_sched_barrier(0);
            issue_b(NT >= 4 ? 2 : 0);  // the LDS turn-around of the panel runs under these
            const vec4 wv = *reinterpret_cast<const vec4 *>(&panel[lr * 4]);
            T w[4] = {wv[0], wv[1], wv[2], wv[3]};
            T u[4] = {};
            T rp = (T)0;
            int p = 0, pv = 0;  // current pivot row slot; lane t of pv holds the slot of pivot t
            constexpr int NS = 12;
            constexpr int NLEAD = NT >= 4 ? 4 : 2;  // MFMAs already issued around the panel staging
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int t = st / 3;
                issue_b(((NB - NLEAD) * (st + 1)) / NS - ((NB - NLEAD) * st) / NS);
                if (st % 3 == 0) {
                    // largest |w_t| over the rows not used yet; lowest row slot on ties
                    const unsigned key = used ? 0u : magkey(w[t]);
                    const unsigned mx = wave_max_u32(key);
                    if (key_bad(T(0), mx)) bad = 1;  // no usable pivot in this column: singular (or NaN / Inf in the input)
                    // key == mx != 0 implies a candidate row (the others carry key 0; mx == 0 is the singular case)
                    const unsigned long long vote = __builtin_amdgcn_uicmp(key, mx, 32 /* ICMP_EQ */);
                    p = vote ? (int)__builtin_ctzll(vote) : 0;
                    used = used || (lr == p);
                    pv = (lr == t) ? p : pv;
                } else if (st % 3 == 1) {
                    // the pivot row's four panel values as scalars, and the reciprocal of the pivot
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = lane_value(w[j], p);
                    rp = rcp_full(u[t]);
                } else {
                    // in-place Gauss-Jordan step of the panel: column t becomes the inverse column
                    const T f = -(w[t] * rp);
                    const bool me = lr == p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j == t) continue;
                        const T piv_j = u[j] * rp;  // wave-uniform
                        w[j] = me ? piv_j : fma_t(f, u[j], w[j]);
                    }
                    w[t] = me ? rp : f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_b(NB);  // whatever is left (NT < 3)
            // Row i of the finished panel is Aop[i, 0..3] (pivot rows: their row of D^-1) and sits in lane i; lane (q, c) needs
            // Aop[16 ti + c][q] = component q of lane group ti: a 4 x 4 transpose of (w0..w3) across the four lane groups,
            // two rounds of v_permlane32_swap / v_permlane16_swap -- no LDS round trip.
            lane_rows_swap<true>(w[0], w[2]);
            lane_rows_swap<true>(w[1], w[3]);
            lane_rows_swap<false>(w[0], w[1]);
            lane_rows_swap<false>(w[2], w[3]);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) aop[ti] = w[ti];
            // permutation tables: column slot j_t was eliminated with row slot p_t
            if (lr < 4) {
                const int j = 16 * tKn + G::pcol(rKn, lr);
                coladdr[j] = (unsigned char)pv;
                rowaddr[pv] = (unsigned char)j;
            }
            // B operand: the four pivot rows through LDS (and zero them in C)
#pragma nounroll
            for (int t = 0; t < 4; ++t) {
                const int s = __builtin_amdgcn_readlane(pv, t);
                const int loc = s & 15;
                const int pos = 4 * (s >> 4) + G::slot_r(loc);  // wave-uniform: 4 * tile row + register
                const unsigned long long mask = 0xffffull << (16 * G::slot_q(loc));
                const unsigned addr = bb_lane + (unsigned)(t * N * (int)sizeof(T));
#ifdef MATINV_GATHER_LINEAR
                gather_zero_tile_row<T, NT, 0>(acc, addr, pos, mask);
                if constexpr (NT > 1) gather_zero_tile_row<T, NT, 1>(acc, addr, pos, mask);
                if constexpr (NT > 2) gather_zero_tile_row<T, NT, 2>(acc, addr, pos, mask);
                if constexpr (NT > 3) gather_zero_tile_row<T, NT, 3>(acc, addr, pos, mask);
#else
                // r03: inside a tile row the register is found with the branch tree of gather_tree.inc instead of a compare chain.
                // (Finding the tile ROW with one wave-uniform C++ branch over the halves instead of NT skipped blocks was tried: the
                // two paths meet with the accumulators in different registers and hipcc spills -- 272 B of scratch at 4 x 4 tiles.)
                gather_row_tree<T, NT, 0>(acc, addr, pos, mask);
                if constexpr (NT > 1) gather_row_tree<T, NT, 1>(acc, addr, pos, mask);
                if constexpr (NT > 2) gather_row_tree<T, NT, 2>(acc, addr, pos, mask);
                if constexpr (NT > 3) gather_row_tree<T, NT, 3>(acc, addr, pos, mask);
#endif
            }
            // pivot columns: zero in C (the MFMA then leaves Aop there) -- placed here, it runs under the gather's LDS turn-around
            {
                // EXEC narrowed to the pivot-column lanes (written as a C++ select hipcc branches and copies the
                // tile column: see tile4_impl.hpp)
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
                                     : "+v"(acc[ti][tKn][0]), "+v"(acc[ti][tKn][1]), "+v"(acc[ti][tKn][2]), "+v"(acc[ti][tKn][3]),
                                       [save] "=&s"(save)
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
                                     : "+v"(acc[ti][tKn][0]), "+v"(acc[ti][tKn][1]), "+v"(acc[ti][tKn][2]), "+v"(acc[ti][tKn][3]),
                                       [save] "=&s"(save)
                                     : [mask] "s"(zmask)
                                     : "scc");
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) bop[tj] = bbuf[q * N + 16 * tj + c];
            // pivot columns: I_4 in B
            bop[tKn] = panel_lane ? ((G::piv(c) == q) ? (T)1 : (T)0) : bop[tKn];
        };

        // ragged n: blocks of the last tile column that hold identity padding only are not run. A padding row is zero in
        // every real column and stays so (it is never a pivot row of one, and elimination adds nothing to it), a padding
        // column is zero in every real row: such a step changes nothing. Their table entries keep the 0xff written below.
        int last_blocks = 4;
        if (!FULL) {
            last_blocks = G::real_blocks(n - 16 * (NT - 1));
            if (lr < N) rowaddr[lr] = coladdr[lr] = (unsigned char)0xff;  // >= n: never stored (ordered by the turn's LDS syncs)
        }
        turn(IntC<0>(), 0, IntC<1>());
#pragma nounroll
        for (int rK = 1; rK < (NT == 1 ? last_blocks : 4); ++rK) turn(IntC<0>(), rK, IntC<0>());
        if constexpr (NT > 1) {
#pragma nounroll
            for (int rK = 0; rK < (NT == 2 ? last_blocks : 4); ++rK) turn(IntC<1>(), rK, IntC<0>());
        }
        if constexpr (NT > 2) {
#pragma nounroll
            for (int rK = 0; rK < (NT == 3 ? last_blocks : 4); ++rK) turn(IntC<2>(), rK, IntC<0>());
        }
        if constexpr (NT > 3) {
#pragma nounroll
            for (int rK = 0; rK < (NT == 4 ? last_blocks : 4); ++rK) turn(IntC<3>(), rK, IntC<0>());
        }
        // the last block's update
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);

        if (bad == 0) {
            // F[i][j] = inverse[rowaddr[i]][coladdr[j]] (see the header): W = A^T is stored as W[a][b] at a*n + b
            unsigned ca[NT];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) ca[tj] = coladdr[16 * tj + c];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned ra = rowaddr[16 * ti + G::trow(r, q)];
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        if (FULL || (ra < (unsigned)n && ca[tj] < (unsigned)n)) X[ra * (unsigned)n + ca[tj]] = acc[ti][tj][r];
                    }
                }
            if (info && l == 0) info[mat] = 0;
        } else if (l == 0) {
            // singular: this kernel eliminates W = A^T, so the column at which IT runs out of pivots is not the oracle's
            // (row pivoting on A). The ROW kernel redoes the matrix for the exact info code and the NaN fill.
            const int slot = atomicAdd(bad_count, 1);
            bad_list[slot] = (int)mat;
        }
        wave_lds_sync();
    }
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, 3) void matinv_gj_tilep_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                            unsigned batch, const int *in_count, const int *in_list, hint_t *hint_out,
                                                            int *bad_count, int *bad_list)
{
    __shared__ __attribute__((aligned(16))) double lds[256 + 4 * 16 * NT];
    __shared__ unsigned char tab[128];
    gj_tilep_body<double, NT, FULL>(Ain, Xout, info, n_rt, batch, lds, tab, in_count, in_list, hint_out, bad_count, bad_list);
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, 3) void matinv_gj_tilep_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info, int n_rt,
                                                            unsigned batch, const int *in_count, const int *in_list, hint_t *hint_out,
                                                            int *bad_count, int *bad_list)
{
    __shared__ __attribute__((aligned(16))) float lds[256 + 4 * 16 * NT];
    __shared__ unsigned char tab[128];
    gj_tilep_body<float, NT, FULL>(Ain, Xout, info, n_rt, batch, lds, tab, in_count, in_list, hint_out, bad_count, bad_list);
}

// in_count / in_list != nullptr: work-list form (one round of resident workgroups strides over the list; usually empty).
// Singular matrices are appended to (bad_count, bad_list) for the ROW kernel.
template <class T>
static hipError_t enqueue_tilep(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                                const int *in_count, const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list, bool expect_many = false)
{
    const int nt = (n + 15) / 16;
    unsigned cap = 256u * 12u * tile_grid_rounds();
    // work-list form: the list is usually empty -- one round of resident workgroups that stride over it. Behind the screening pass
    // (expect_many) most of the batch is on it: one workgroup per matrix as in the direct form (a workgroup beyond the list's
    // length reads the count and leaves), otherwise 30 matrices per wave, each behind a dependent load of its index
    if (in_list && !expect_many) cap = 256u * 12u;
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    const unsigned b = (unsigned)batch;
#define TP_LAUNCH(NT_)                                                                                                 \
    if constexpr (sizeof(T) == 8) {                                                                                    \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilep_f64<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, in_count, in_list, hint_out, bad_count, bad_list); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilep_f64<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, in_count, in_list, hint_out, bad_count, bad_list); \
    } else {                                                                                                           \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilep_f32<NT_, true>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, in_count, in_list, hint_out, bad_count, bad_list); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilep_f32<NT_, false>), dim3(grid), dim3(64), 0, stream, A, X, info, n, b, in_count, in_list, hint_out, bad_count, bad_list); \
    }
    switch (nt) {
    case 1: TP_LAUNCH(1) break;
    case 2: TP_LAUNCH(2) break;
    case 3: TP_LAUNCH(3) break;
    default: TP_LAUNCH(4) break;
    }
#undef TP_LAUNCH
    return hipGetLastError();
}

template <class T>
static hipError_t launch_tilep(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (n < 1 || n > 64) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) {
        (void)scratch_free(ws, stream);
        return e;
    }
    e = enqueue_tilep<T>(n, A, X, batch, info, stream, nullptr, nullptr, nullptr, ws, ws + 1);
    if (e == hipSuccess) e = launch_gj_row_worklist<T>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

// the matrices the natural-order kernel rejected: (in_count, in_list) in device memory; the singular ones among them go on
// to the ROW kernel through (bad_count, bad_list), which the caller has zeroed
template <class T>
static hipError_t launch_tilep_worklist(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, const int *in_count,
                                        const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                        hint_t *hint_out, bool expect_many = false)
{
    hipError_t e = enqueue_tilep<T>(n, A, X, batch, info, stream, in_count, in_list, hint_out, bad_count, bad_list, expect_many);
    if (e == hipSuccess) e = launch_gj_row_worklist<T>(n, A, X, bad_count, bad_list, info, stream);
    if (e == hipSuccess) e = debug_note_rejects(in_count, stream);
    return e;
}

}  // namespace matinv
