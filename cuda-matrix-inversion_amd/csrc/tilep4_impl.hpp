// tilep4_impl.hpp (instantiated by tilep4_kernels.hip for f64 and tilep4_f32_kernels.hip for f32) -- the pivoting MFMA tile
// Gauss-Jordan of tilep_impl.hpp (read that header first) for 64 < n <= 128: FOUR wavefronts per matrix.
//
// As in tile4_impl.hpp a 256-thread workgroup owns a matrix and wavefront w holds the tile columns w and w + 4 (all NT tile
// rows of them, <= 16 tiles). What the pivot search adds to that split:
//   * the wave that owns the pivot columns stages them into a double-buffered LDS panel, ONE workgroup barrier per block
//     step; then EVERY wave runs the search and the in-place Gauss-Jordan of the n x 4 panel redundantly, two rows per
//     lane (rows l and l + 64): no second exchange, and pivots, A operand and permutation come out identical in all four;
//   * the B operand (the four pivot rows) is local: each wave gathers the part of the pivot rows that lies in ITS tile
//     columns from its own registers through a private 1 KB LDS strip (run-time register index: the asm blocks of
//     gather_zero_tile_row, two tile columns wide here) and zeroes it in C;
//   * the A operand comes out of the search registers by the same lane-group transposes (permlane swaps), once for the
//     rows below 64 and once for the rows above.
// (Measured and not kept here: ONE searching wave per block -- the owner -- publishing the finished panel through LDS while the
// other three only run their MFMAs until the barrier. It removes 3/4 of the search instructions and loses 10 %: 3.8e6 inv/s
// at 128 x 128 f64 against 4.2e6, 6.1e6 against 6.8e6 at 96 x 96 -- with four waves the search of ONE wave is the critical
// path either way, and the redundant version needs no second LDS trip.)
// Look-ahead as in the one-wave kernel: the local tile column that (for the next owner) holds the next pivot columns is
// updated first, the next panel is staged, and the other column's MFMAs run pinned between the stages of the next search.
//
// Replaces, for general 64 < n <= 128 input, pivotRow / normalizeRow / transform_matrix of
// /root/reference/src/gauss/batched_invert.cu:17-82 (the reference's sweep goes to n = 128, Makefile:202-220).
#pragma once
#include "tilep_impl.hpp"

namespace matinv {

// gather_zero_tile_row for a wave that holds TWO tile columns (acc[ti][0], acc[ti][1]); see tilep_impl.hpp
template <class T, int NT, int TI>
__device__ __forceinline__ void gather_zero_tile_row2(typename TileGeo<T>::vec4 (&acc)[NT][2], unsigned addr, int pos,
                                                     unsigned long long mask)
{
    unsigned long long save;
    unsigned tmp;
#define TP4_W64(J, R, OFF) "ds_write_b64 %[addr], %[a" #J #R "] offset:" #OFF "\n\t"
#define TP4_W32(J, R, OFF) "ds_write_b32 %[addr], %[a" #J #R "] offset:" #OFF "\n\t"
#define TP4_Z64(J, R) "v_mov_b64_e32 %[a" #J #R "], 0\n\t"
#define TP4_Z32(J, R) "v_mov_b32_e32 %[a" #J #R "], 0\n\t"
#define TP4_WZ64(R) TP4_W64(0, R, 0) TP4_W64(1, R, 128) TP4_Z64(0, R) TP4_Z64(1, R)
#define TP4_WZ32(R) TP4_W32(0, R, 0) TP4_W32(1, R, 64) TP4_Z32(0, R) TP4_Z32(1, R)
#define TP4_BODY(WZ)                                                                                                   \
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
#define TP4_OPERANDS                                                                                                   \
    [a00] "+v"(acc[TI][0][0]), [a01] "+v"(acc[TI][0][1]), [a02] "+v"(acc[TI][0][2]), [a03] "+v"(acc[TI][0][3]),        \
        [a10] "+v"(acc[TI][1][0]), [a11] "+v"(acc[TI][1][1]), [a12] "+v"(acc[TI][1][2]), [a13] "+v"(acc[TI][1][3]),    \
        [save] "=&s"(save), [tmp] "=&s"(tmp)
    if constexpr (sizeof(T) == 8)
        asm volatile(TP4_BODY(TP4_WZ64) : TP4_OPERANDS : [addr] "v"(addr), [pos] "s"(pos), [mask] "s"(mask), [ti] "n"(TI) : "scc", "memory");
    else
        asm volatile(TP4_BODY(TP4_WZ32) : TP4_OPERANDS : [addr] "v"(addr), [pos] "s"(pos), [mask] "s"(mask), [ti] "n"(TI) : "scc", "memory");
#undef TP4_OPERANDS
#undef TP4_BODY
#undef TP4_WZ32
#undef TP4_WZ64
#undef TP4_Z32
#undef TP4_Z64
#undef TP4_W32
#undef TP4_W64
}

template <class T, int NT, bool FULL, int W = 4>
__device__ __forceinline__ void gj_tilep4_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,
                                               T *panel2, T *bball, unsigned char *tab, int *bad_count, int *bad_list,
                                               const int *in_count, const int *in_list, hint_t *hint_out)
{
    static_assert(NT >= 5 && NT <= 2 * W && W <= 4, "W wavefronts of two tile columns each serve 64 < n <= 32 W");
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int NC = 2;
    unsigned char *const rowaddr = tab, *const coladdr = tab + 128;
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;  // wave-uniform
    T *const bbuf = bball + w * (4 * 16 * NC);  // [4 pivots][2 tile columns x 16], private to the wave
    typedef __attribute__((address_space(3))) T *lds_ptr;
    const unsigned bb_lane = (unsigned)(size_t)(lds_ptr)(bbuf + (l & 15));

    // work-list form (the matrices the natural-order four-wave kernel rejected): in_list[0 .. *in_count); its length goes
    // back to the launcher's natural / pivot guess through pinned host memory (see tilep_impl.hpp)
    const unsigned todo = in_count ? (unsigned)*in_count : batch;
    if (hint_out && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hint_out, ((hint_t)batch << 32) | todo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (unsigned item = blockIdx.x; item < todo; item += gridDim.x) {
        const unsigned mat = in_list ? (unsigned)in_list[item] : item;
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));
        int q = l >> 4, c = l & 15, lr = l;
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c), "+v"(lr));

        // acc[ti][jl] = tile (ti, w + 4 jl); W = A^T as in the one-wave kernel. Tile columns beyond NT hold zeros.
        vec4 acc[NT][NC];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) {
                const int tj = w + W * jl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                    const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n + 16 * tj);
                    const bool edge = !FULL && (ti == NT - 1 || jl == NC - 1);
                    const bool in = (tj < NT) && (!edge || (row < n && col < n));
                    acc[ti][jl][r] = in ? A[uoff + lane_off] : ((row == col) ? (T)1 : (T)0);
                }
            }

        bool used_lo = false, used_hi = lr + 64 >= N;  // rows l and l + 64
        int bad = 0;
        T aop[NT], bop[NC];

        auto turn = [&](auto tKc, int rKn, auto firstc) {
            constexpr int tKn = decltype(tKc)::value;
            constexpr bool first = decltype(firstc)::value != 0;
            constexpr int owner = tKn % W, jo = tKn / W;  // wave and local column holding the pivot columns of this block
            constexpr int NB = first ? 0 : NT * (NC - 1);
            const bool panel_lane = G::blk(c) == rKn;
            T *const pbuf = panel2 + ((4 * tKn + rKn) & 1) * (N * 4);
            // tile columns beyond NT (second local column of the last waves when NT < 8) hold zeros and are left alone
            const bool have_jo = w + W * jo < NT, have_other = w + W * (1 - jo) < NT;  // wave-uniform
            if (!first && have_jo) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][jo] = G::mfma(aop[ti], bop[jo], acc[ti][jo]);
            }
            int pend = 0;
            auto issue_b = [&](int count) {
#pragma unroll
                for (int z = 0; z < count; ++z) {
                    if (pend < NB) {
                        if (have_other) acc[pend][1 - jo] = G::mfma(aop[pend], bop[1 - jo], acc[pend][1 - jo]);
                        ++pend;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            issue_b(2);
            if (w == owner && panel_lane) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pbuf[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][jo][r];
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            const vec4 wlo = *reinterpret_cast<const vec4 *>(&pbuf[lr * 4]);
            vec4 whi = {};
            if (lr + 64 < N) whi = *reinterpret_cast<const vec4 *>(&pbuf[(lr + 64) * 4]);
            T a0[4] = {wlo[0], wlo[1], wlo[2], wlo[3]}, a1[4] = {whi[0], whi[1], whi[2], whi[3]};
            T u[4] = {};
            T rp = (T)0;
            int p = 0, pv = 0;
            constexpr int NS = 12;
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int t = st / 3;
                issue_b(((NB - 2) * (st + 1)) / NS - ((NB - 2) * st) / NS);
                if (st % 3 == 0) {
                    const unsigned k0 = used_lo ? 0u : magkey(a0[t]), k1 = used_hi ? 0u : magkey(a1[t]);
                    const unsigned mx = wave_max_u32(k0 > k1 ? k0 : k1);
                    if (key_bad(T(0), mx)) bad = 1;
                    const unsigned long long v0 = __builtin_amdgcn_uicmp(k0, mx, 32), v1 = __builtin_amdgcn_uicmp(k1, mx, 32);
                    p = v0 ? (int)__builtin_ctzll(v0) : (v1 ? 64 + (int)__builtin_ctzll(v1) : 0);  // lowest row first
                    used_lo = used_lo || (lr == p);
                    used_hi = used_hi || (lr + 64 == p);
                    pv = (lr == t) ? p : pv;
                } else if (st % 3 == 1) {
                    const bool hi = p >= 64;  // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = lane_value(hi ? a1[j] : a0[j], p & 63);
                    rp = rcp_full(u[t]);
                } else {
                    const T f0 = -(a0[t] * rp), f1 = -(a1[t] * rp);
                    const bool me0 = lr == p, me1 = lr + 64 == p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j == t) continue;
                        const T piv_j = u[j] * rp;
                        a0[j] = me0 ? piv_j : fma_t(f0, u[j], a0[j]);
                        a1[j] = me1 ? piv_j : fma_t(f1, u[j], a1[j]);
                    }
                    a0[t] = me0 ? rp : f0;
                    a1[t] = me1 ? rp : f1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_b(NB);
            // A operand: lane (q, c) needs Aop[16 ti + c][q]; rows below 64 sit in a0 (lane group ti), rows above in a1
            lane_rows_swap<true>(a0[0], a0[2]);
            lane_rows_swap<true>(a0[1], a0[3]);
            lane_rows_swap<false>(a0[0], a0[1]);
            lane_rows_swap<false>(a0[2], a0[3]);
            lane_rows_swap<true>(a1[0], a1[2]);
            lane_rows_swap<true>(a1[1], a1[3]);
            lane_rows_swap<false>(a1[0], a1[1]);
            lane_rows_swap<false>(a1[2], a1[3]);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) aop[ti] = ti < 4 ? a0[ti] : a1[ti - 4];
            // permutation tables (every wave writes the same values)
            if (lr < 4) {
                const int j = 16 * tKn + G::pcol(rKn, lr);
                coladdr[j] = (unsigned char)pv;
                rowaddr[pv] = (unsigned char)j;
            }
            // B operand: this wave's part of the four pivot rows through its LDS strip (and zero it in C)
#pragma nounroll
            for (int t = 0; t < 4; ++t) {
                const int s = __builtin_amdgcn_readlane(pv, t);
                const int loc = s & 15;
                const int pos = 4 * (s >> 4) + G::slot_r(loc);
                const unsigned long long mask = 0xffffull << (16 * G::slot_q(loc));
                const unsigned addr = bb_lane + (unsigned)(t * 16 * NC * (int)sizeof(T));
#ifdef MATINV_GATHER_LINEAR
                gather_zero_tile_row2<T, NT, 0>(acc, addr, pos, mask);
                gather_zero_tile_row2<T, NT, 1>(acc, addr, pos, mask);
                gather_zero_tile_row2<T, NT, 2>(acc, addr, pos, mask);
                gather_zero_tile_row2<T, NT, 3>(acc, addr, pos, mask);
                gather_zero_tile_row2<T, NT, 4>(acc, addr, pos, mask);
                if constexpr (NT > 5) gather_zero_tile_row2<T, NT, 5>(acc, addr, pos, mask);
                if constexpr (NT > 6) gather_zero_tile_row2<T, NT, 6>(acc, addr, pos, mask);
                if constexpr (NT > 7) gather_zero_tile_row2<T, NT, 7>(acc, addr, pos, mask);
#else
                // r03: blocks of three tile rows, each one asm statement with a binary branch tree over the slot index
                // (gather_tree.inc): a skipped per-tile-row block of the r02 form cost ~75 cycles, ~600 per pivot row here
                gather_tree_2x3<0>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], addr, pos, mask);
                if constexpr (NT == 5) gather_tree_2x2<12>(acc[3][0], acc[3][1], acc[4][0], acc[4][1], addr, pos, mask);
                if constexpr (NT >= 6) gather_tree_2x3<12>(acc[3][0], acc[3][1], acc[4][0], acc[4][1], acc[5][0], acc[5][1], addr, pos, mask);
                if constexpr (NT == 7) gather_tree_2x1<24>(acc[6][0], acc[6][1], addr, pos, mask);
                if constexpr (NT == 8) gather_tree_2x2<24>(acc[6][0], acc[6][1], acc[7][0], acc[7][1], addr, pos, mask);
#endif
            }
            // pivot columns (owner only): zero in C, I_4 in B
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
                                     "s_nop 1\n"
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
            wave_lds_sync();
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) bop[jl] = bbuf[q * (16 * NC) + 16 * jl + c];
            if (w == owner) bop[jo] = panel_lane ? ((G::piv(c) == q) ? (T)1 : (T)0) : bop[jo];
        };

        // ragged n: the all-padding blocks of the last tile column are not run (see tilep_impl.hpp); their table entries keep
        // the 0xff every wave writes here (before the first turn's barrier, the turns' own entries after it)
        int last_blocks = 4;
        if (!FULL) {
            last_blocks = G::real_blocks(n - 16 * (NT - 1));
            rowaddr[lr] = coladdr[lr] = (unsigned char)0xff;
            rowaddr[lr + 64] = coladdr[lr + 64] = (unsigned char)0xff;
        }
        auto column = [&](auto tKc, int from) {
            const int to = (decltype(tKc)::value == NT - 1) ? last_blocks : 4;
#pragma nounroll
            for (int rK = from; rK < to; ++rK) turn(tKc, rK, IntC<0>());
        };
        turn(IntC<0>(), 0, IntC<1>());
        column(IntC<0>(), 1);
        column(IntC<1>(), 0);
        column(IntC<2>(), 0);
        column(IntC<3>(), 0);
        column(IntC<4>(), 0);
        if constexpr (NT > 5) column(IntC<5>(), 0);
        if constexpr (NT > 6) column(IntC<6>(), 0);
        if constexpr (NT > 7) column(IntC<7>(), 0);
#pragma unroll
        for (int jl = 0; jl < NC; ++jl) {
            if (w + W * jl < NT) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][jl] = G::mfma(aop[ti], bop[jl], acc[ti][jl]);
            }
        }
        __syncthreads();  // the tables are complete; both panel buffers are free for the next matrix

        if (bad == 0) {
            unsigned ca[NC];
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) ca[jl] = (w + W * jl < NT) ? (unsigned)coladdr[16 * (w + W * jl) + c] : 0xffffu;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned ra = rowaddr[16 * ti + G::trow(r, q)];
#pragma unroll
                    for (int jl = 0; jl < NC; ++jl) {
                        if (ca[jl] != 0xffffu && (FULL || (ra < (unsigned)n && ca[jl] < (unsigned)n)))
                            X[ra * (unsigned)n + ca[jl]] = acc[ti][jl][r];
                    }
                }
            if (info && threadIdx.x == 0) info[mat] = 0;
        } else if (threadIdx.x == 0) {
            // singular: the pivoted LDS kernel redoes the matrix for the oracle's info code and the NaN fill
            const int slot = atomicAdd(bad_count, 1);
            bad_list[slot] = (int)mat;
        }
        __syncthreads();  // the next matrix rewrites the tables
    }
}

// fp64, NT <= 7: three workgroups per CU (168 VGPRs; 6 / 32 / 58 registers of the full 80^2 / 96^2 / 112^2 instantiations spill)
// -- the kernel waits on its per-step critical path most of the time, so the extra workgroup is worth more than the spills
// cost: general 80^2 9.1e6 -> 1.15e7 inv/s, 96^2 6.6e6 -> 8.0e6, 100^2 5.9e6 -> 6.3e6. NT = 8 would spill 95 of its 254
// registers and loses (128^2: 4.1e6 -> 3.6e6): two per CU.
template <int NT, bool FULL>
__global__ __launch_bounds__(256, NT <= 7 ? 3 : 2) void matinv_gj_tilep4_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                              unsigned batch, int *bad_count, int *bad_list, const int *in_count,
                                                              const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) double panel2[2 * 16 * NT * 4];
    __shared__ __attribute__((aligned(16))) double bball[4 * 4 * 32];
    __shared__ unsigned char tab[256];
    gj_tilep4_body<double, NT, FULL>(Ain, Xout, info, n_rt, batch, panel2, bball, tab, bad_count, bad_list, in_count, in_list, hint_out);
}

// THREE wavefronts per matrix for 5 x 5 / 6 x 6 tiles (r03): with four, the tile columns fall 2 + 1 + 1 + 1 / 2 + 2 + 1 + 1 on the waves and
// the block step lasts as long as the wave with two; with three (2 + 2 + 1 / 2 + 2 + 2) a CU holds four matrices instead of three
template <int NT, bool FULL>
__global__ __launch_bounds__(192, 3) void matinv_gj_tilep3_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n_rt,
                                                              unsigned batch, int *bad_count, int *bad_list, const int *in_count,
                                                              const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) double panel2[2 * 16 * NT * 4];
    __shared__ __attribute__((aligned(16))) double bball[3 * 4 * 32];
    __shared__ unsigned char tab[256];
    gj_tilep4_body<double, NT, FULL, 3>(Ain, Xout, info, n_rt, batch, panel2, bball, tab, bad_count, bad_list, in_count, in_list, hint_out);
}
template <int NT, bool FULL>
__global__ __launch_bounds__(192, 3) void matinv_gj_tilep3_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info, int n_rt,
                                                              unsigned batch, int *bad_count, int *bad_list, const int *in_count,
                                                              const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) float panel2[2 * 16 * NT * 4];
    __shared__ __attribute__((aligned(16))) float bball[3 * 4 * 32];
    __shared__ unsigned char tab[256];
    gj_tilep4_body<float, NT, FULL, 3>(Ain, Xout, info, n_rt, batch, panel2, bball, tab, bad_count, bad_list, in_count, in_list, hint_out);
}

template <int NT, bool FULL>
__global__ __launch_bounds__(256, 3) void matinv_gj_tilep4_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info, int n_rt,
                                                              unsigned batch, int *bad_count, int *bad_list, const int *in_count,
                                                              const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) float panel2[2 * 16 * NT * 4];
    __shared__ __attribute__((aligned(16))) float bball[4 * 4 * 32];
    __shared__ unsigned char tab[256];
    gj_tilep4_body<float, NT, FULL>(Ain, Xout, info, n_rt, batch, panel2, bball, tab, bad_count, bad_list, in_count, in_list, hint_out);
}

template <class T>
static hipError_t enqueue_tilep4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, int *bad_count,
                                 int *bad_list, const int *in_count, const int *in_list, hint_t *hint_out, bool expect_many = false)
{
    const int nt = (n + 15) / 16;
    unsigned cap = 256u * 4u * tile_grid_rounds();
    if (in_list && !expect_many) cap = 256u * 4u;  // usually empty: one round of resident workgroups (see enqueue_tilep)
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    const unsigned b = (unsigned)batch;
#define TP4_LAUNCH(NT_)                                                                                                \
    if constexpr (sizeof(T) == 8) {                                                                                    \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilep4_f64<NT_, true>), dim3(grid), dim3(256), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilep4_f64<NT_, false>), dim3(grid), dim3(256), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
    } else {                                                                                                           \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilep4_f32<NT_, true>), dim3(grid), dim3(256), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilep4_f32<NT_, false>), dim3(grid), dim3(256), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
    }
#define TP3_LAUNCH(NT_)                                                                                                \
    if constexpr (sizeof(T) == 8) {                                                                                    \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilep3_f64<NT_, true>), dim3(grid), dim3(192), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilep3_f64<NT_, false>), dim3(grid), dim3(192), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
    } else {                                                                                                           \
        if (n == 16 * NT_)                                                                                             \
            hipLaunchKernelGGL((matinv_gj_tilep3_f32<NT_, true>), dim3(grid), dim3(192), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
        else                                                                                                           \
            hipLaunchKernelGGL((matinv_gj_tilep3_f32<NT_, false>), dim3(grid), dim3(192), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out); \
    }
    switch (nt) {
    case 5: TP3_LAUNCH(5) break;
    case 6: TP3_LAUNCH(6) break;
    case 7: TP4_LAUNCH(7) break;
    default: TP4_LAUNCH(8) break;
    }
#undef TP4_LAUNCH
#undef TP3_LAUNCH
    return hipGetLastError();
}

template <class T>
static hipError_t launch_tilep4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (n <= 64 || n > 128) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) {
        (void)scratch_free(ws, stream);
        return e;
    }
    e = enqueue_tilep4<T>(n, A, X, batch, info, stream, ws, ws + 1, nullptr, nullptr, nullptr);
    // singular input only: the pivoted LDS kernel reports the exact step and NaN-fills the output
    if (e == hipSuccess) e = launch_gj_lds_worklist<T>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

// the matrices the natural-order four-wave kernel rejected: (in_count, in_list); the singular ones among them go on to the LDS
// kernel through (bad_count, bad_list), zeroed by the caller
template <class T>
static hipError_t launch_tilep4_worklist(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, const int *in_count,
                                         const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                         hint_t *hint_out, bool expect_many = false)
{
    hipError_t e = enqueue_tilep4<T>(n, A, X, batch, info, stream, bad_count, bad_list, in_count, in_list, hint_out, expect_many);
    if (e == hipSuccess) e = launch_gj_lds_worklist<T>(n, A, X, bad_count, bad_list, info, stream);
    if (e == hipSuccess) e = debug_note_rejects(in_count, stream);
    return e;
}

}  // namespace matinv
