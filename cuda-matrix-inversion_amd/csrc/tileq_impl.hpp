// tileq_impl.hpp (instantiated by tileq_kernels.hip for f64 and tileq_f32_kernels.hip for f32) -- kernel family "TILEQ":
// the MFMA accumulator-tile Gauss-Jordan with TRUE PARTIAL PIVOTING for GENERAL matrices, 128 < n <= 192 (f64) / 256 (f32),
// one wavefront per tile column, WITHOUT any run-time register index.
//
// The r02 / r03 pivoting kernels (tilep4_impl.hpp; tilepw_impl.hpp until r03) search the pivot along the pivot COLUMNS of the
// register-resident W (static registers: the panel) and therefore find their four pivot ROWS wherever the search puts
// them -- (tile row, accumulator register, lane group) known at run time only. Feeding those rows to the B operand of the
// rank-4 update needed a register index the ISA only offers through branching (gather_tree.inc: ~2.6 k of the ~7 k cycles
// of a block step at 128 x 128) and, with the search repeated by every wave, 48 k vector instructions per 128 x 128 matrix.
//
// This family turns the step around. The four pivot ROWS of block kb are fixed: the rows that accumulator register kb % 4
// of tile row kb / 4 holds across the four lane groups -- the rows the natural-order kernel uses (tile_kernels.inc), so
// staging them is ONE LDS store per tile column and perfectly balanced over the waves. The search runs ALONG those rows,
// over the columns not used yet: with W = A^T this is exactly the oracle's row pivoting on A (column k of A, rows not
// used yet; /root/reference/src/gauss/inverse_gpu.cu:24-33 = cublasSgetrfBatched, the LAPACK rule). What is found at run
// time is the pivot COLUMN: a run-time LANE (and wave), but static registers -- its owner stores it to LDS under a
// four-lane EXEC mask with plain code. Per block step (TWO workgroup barriers):
//   1. (look-ahead) the previous block's MFMAs on tile row tK, then the four pivot rows -> LDS row panel [4 x N]; barrier
//   2. ONE wave (the searcher, rotating with tK) runs four steps of in-place Gauss-Jordan on the 4 x N panel, R = ceil(N / 64)
//      columns per lane: wave-wide DPP max of |.| over the unused columns, lowest column on ties, the pivot column's four
//      entries as scalars, one row operation. It leaves B' = D^-1 W[P, :] with D^-1 in the pivot columns -- the B operand as
//      it stands. Each column index goes out through a tagged LDS word THE MOMENT IT IS FOUND; the other waves issue the
//      rest of the previous block's MFMAs, then spin on those words, and the owner of a pivot column writes it (N x 1,
//      pre-update values) into the LDS column panel and zeroes it in C while the search of the next pivot is still running
//      (r04 stamps, tools/tileq_stamps.hip: with the gather behind a barrier after the search it cost 2.4 k of 6.3 k cycles)
//   3. barrier; B operand = one LDS load per tile column; A operand = the NEGATED column panel with I_4 in the pivot rows:
//      no arithmetic.     W_new = C_masked + A' B'   (checked against numpy in tools/tileq_emulate.py)
// The permutation this leaves behind (row slot i was eliminated with column j_i) is folded into the store addresses:
// inverse[j_i][b] = F[i][j_b], two byte tables in LDS, as in tilep_impl.hpp.
#pragma once
#include "tilep_impl.hpp"

namespace matinv {

// the register of a tile that run-time rK names (wave-uniform): three selects on scalar conditions
template <class T, class V>
__device__ __forceinline__ T pick_reg(const V &t, int rK)
{
    T v = t[0];
    v = rK == 1 ? t[1] : v;
    v = rK == 2 ? t[2] : v;
    v = rK == 3 ? t[3] : v;
    return v;
}

// key of a column that is already used = 0: v_cndmask on the scalar mask of used columns (bit = lane)
__device__ __forceinline__ unsigned mask_key(unsigned key, unsigned long long used)
{
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(r) : "v"(key), "s"(used));
    return r;
}

// -DMATINV_TILEQ_STAMPS (tools/tileq_stamps.hip only): s_memtime at the phase boundaries of a block step, summed per wave
#ifdef MATINV_TILEQ_STAMPS
__device__ unsigned long long matinv_tileq_stamps[16 * 16];  // [wave][phase]
#define TQ_STAMP(PH)                                                                                                   \
    do {                                                                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter();                                                  \
        stamp_acc[PH] += now_ - stamp_last;                                                                            \
        stamp_last = now_;                                                                                             \
    } while (0)
#else
#define TQ_STAMP(PH) ((void)0)
#endif

// the searcher's column index of pivot k reaches the other waves through an LDS word tagged with the block step: they spin on
// it (all waves of a workgroup are resident, the searcher never waits for them) and gather their column WHILE the search of the
// next pivots goes on
__device__ __forceinline__ unsigned wait_pivot(int *slot, unsigned tag)
{
    unsigned v;
    for (;;) {
        v = (unsigned)__builtin_amdgcn_readfirstlane(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((v >> 8) == tag) break;
        __builtin_amdgcn_s_sleep(1);
    }
    return v & 255u;
}

// Wave-wide maximum of the keys AND, in the wait states its six DPP steps need anyway, the full-accuracy reciprocal of each lane's own
// two candidates (the instruction sequence of rcp_full): once the pivot lane is known its reciprocal is ONE readlane pair away instead
// of a dependent v_rcp + four fused multiply-adds (~100 cycles per pivot on the searcher's critical path).
__device__ __forceinline__ unsigned wave_max_rcp2(unsigned v, double x0, double x1, double &r0, double &r1)
{
    unsigned m;
    double e0, e1;
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_rcp_f64_e32 %[r0], %[x0]\n\t"
                 "v_rcp_f64_e32 %[r1], %[x1]\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fma_f64 %[e0], -%[x0], %[r0], 1.0\n\t"
                 "v_fma_f64 %[e1], -%[x1], %[r1], 1.0\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_fma_f64 %[r0], %[r0], %[e0], %[r0]\n\t"
                 "v_fma_f64 %[r1], %[r1], %[e1], %[r1]\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_fma_f64 %[e0], -%[x0], %[r0], 1.0\n\t"
                 "v_fma_f64 %[e1], -%[x1], %[r1], 1.0\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_fma_f64 %[r0], %[r0], %[e0], %[r0]\n\t"
                 "v_fma_f64 %[r1], %[r1], %[e1], %[r1]\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_readlane_b32 %[m], %[v], 63"
                 : [v] "+v"(v), [m] "=s"(m), [r0] "=&v"(r0), [r1] "=&v"(r1), [e0] "=&v"(e0), [e1] "=&v"(e1)
                 : [x0] "v"(x0), [x1] "v"(x1));
    return m;
}
__device__ __forceinline__ unsigned wave_max_rcp2(unsigned v, float x0, float x1, float &r0, float &r1)
{
    unsigned m;
    float e0, e1;
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_rcp_f32_e32 %[r0], %[x0]\n\t"
                 "v_rcp_f32_e32 %[r1], %[x1]\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fma_f32 %[e0], -%[x0], %[r0], 1.0\n\t"
                 "v_fma_f32 %[e1], -%[x1], %[r1], 1.0\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_fma_f32 %[r0], %[r0], %[e0], %[r0]\n\t"
                 "v_fma_f32 %[r1], %[r1], %[e1], %[r1]\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_readlane_b32 %[m], %[v], 63"
                 : [v] "+v"(v), [m] "=s"(m), [r0] "=&v"(r0), [r1] "=&v"(r1), [e0] "=&v"(e0), [e1] "=&v"(e1)
                 : [x0] "v"(x0), [x1] "v"(x1));
    return m;
}

template <class T, int NT, int W, int NC, bool FULL>
__device__ __forceinline__ void gj_tileq_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch, T *rowpanel,
                                              T *bprime, T *colpanel, unsigned char *tab, int *meta, int *bad_count, int *bad_list,
                                              const int *in_count, const int *in_list, hint_t *hint_out)
{
    static_assert(NT >= 5 && NT <= W * NC && NT <= 16, "W wavefronts of NC tile columns each");
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int R = (N + 63) / 64;  // columns per lane in the search
    unsigned char *const rowaddr = tab, *const coladdr = tab + 256;
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    const unsigned todo = in_count ? (unsigned)*in_count : batch;
    if (hint_out && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hint_out, ((hint_t)batch << 32) | todo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x < 8) meta[threadIdx.x] = 0;  // no stale tag: ordered before the first wait by the first step's barrier
#ifdef MATINV_TILEQ_STAMPS
    unsigned long long stamp_acc[16] = {}, stamp_last = 0;
#endif
    for (unsigned item = blockIdx.x; item < todo; item += gridDim.x) {
        const unsigned mat = in_list ? (unsigned)in_list[item] : item;
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));
        int q = l >> 4, c = l & 15, lr = l;
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c), "+v"(lr));
#ifdef MATINV_TILEQ_STAMPS
        stamp_last = __builtin_readcyclecounter();
#endif

        // acc[ti][jl] = tile (ti, w + W jl) of W = A^T: element (a, b) at a * n + b. Tile columns beyond NT hold zeros.
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
                    const bool edge = !FULL && (ti == NT - 1 || tj == NT - 1);
                    const bool in = (tj < NT) && (!edge || (row < n && col < n));
                    acc[ti][jl][r] = in ? A[uoff + lane_off] : ((row == col) ? (T)1 : (T)0);
                }
            }

        // columns already used as pivots: one 64-bit scalar mask per column set (bit l = column 64 s + l); columns >= N never qualify
        unsigned long long used[R];
#pragma unroll
        for (int s = 0; s < R; ++s) used[s] = (N - 64 * s >= 64) ? 0ull : (~0ull << (N - 64 * s));
        int bad = 0;
        T aop[NT], bop[NC];

        auto turn = [&](auto tKc, int rK, auto firstc) {
            constexpr int tK = decltype(tKc)::value;
            constexpr bool first = decltype(firstc)::value != 0;
            constexpr int NB = first ? 0 : (NT - 1) * NC;  // MFMAs of the previous block still owed after the look-ahead row
            const bool searcher = w == tK % W;             // wave-uniform; rotates with the tile row
            const unsigned tag = (unsigned)(4 * tK + rK + 1);
            // (the run-time test `w + W jl < NT` around every MFMA stays even where it is always true, NT = W NC: each MFMA in its own
            // basic block keeps hipcc's scheduler from stretching live ranges -- without the branches 256 VGPRs and 90 spilled at 8 x 8 tiles)
            TQ_STAMP(0);  // (first step: the loads; otherwise nothing)
            // 1. look-ahead: tile row tK first, then its register rK (the four pivot rows) -> LDS, column-major [N][4]
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) {
                if (w + W * jl < NT) {
                    if (!first) acc[tK][jl] = G::mfma(aop[tK], bop[jl], acc[tK][jl]);
                }
            }
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) {
                if (w + W * jl < NT) rowpanel[(16 * (w + W * jl) + c) * 4 + q] = pick_reg<T>(acc[tK][jl], rK);
            }
            int pend = 0;  // folds to a literal: everything here is fully unrolled
            auto issue_b = [&](int count) {
#pragma unroll
                for (int z = 0; z < count; ++z) {
                    if (pend < NB) {
                        const int tix = pend / NC, jl = pend % NC;
                        const int ti = tix + (tix >= tK ? 1 : 0);
                        if (w + W * jl < NT) acc[ti][jl] = G::mfma(aop[ti], bop[jl], acc[ti][jl]);
                        ++pend;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            TQ_STAMP(1);  // look-ahead MFMAs + pivot rows -> LDS
            __syncthreads();
            TQ_STAMP(2);  // barrier 1
            __builtin_amdgcn_sched_barrier(0);
            // 2. the search: four steps of in-place Gauss-Jordan on the 4 x N row panel, R columns per lane; the MFMAs the wave still owes
            //    the previous block between its stages. (The MFMAs stay OUTSIDE the wave-uniform branch: issued on both sides of one,
            //    hipcc merges the accumulators with copies and spills 276 registers.)
            T a[R][4];
            if (searcher) {
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    vec4 v = {};
                    if (64 * s + 64 <= N || lr + 64 * s < N) v = *reinterpret_cast<const vec4 *>(&rowpanel[(lr + 64 * s) * 4]);
                    a[s][0] = v[0], a[s][1] = v[1], a[s][2] = v[2], a[s][3] = v[3];
                }
            }
            unsigned ppack = 0;
            int p = 0;
            T u[4] = {};
            T rp = (T)0;
            T rl[2] = {};  // R == 2: the reciprocals of the lane's own candidates (wave_max_rcp2)
            constexpr int NS = 12;
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int k = st / 3;
                issue_b((NB * (st + 1)) / NS - (NB * st) / NS);
#ifdef MATINV_TILEQ_STAMPS_FINE
                if (searcher) TQ_STAMP(14);  // the searcher's MFMAs between its stages
#endif
                if (searcher) {
                    if (st % 3 == 0) {
                        unsigned key[R], kmax = 0;
#pragma unroll
                        for (int s = 0; s < R; ++s) {
                            key[s] = mask_key(magkey(a[s][k]), used[s]);
                            kmax = key[s] > kmax ? key[s] : kmax;
                        }
                        unsigned mx;
#ifndef MATINV_TILEQ_NO_SPECRCP
                        if constexpr (R == 2) mx = wave_max_rcp2(kmax, a[0][k], a[1][k], rl[0], rl[1]);
                        else
#endif
                            mx = wave_max_u32(kmax);
                        if (key_bad(T(0), mx) && bad == 0) bad = 16 * tK + G::trow(rK, k) + 1;  // row of W = column of A without a usable pivot
                        p = 0;
                        bool found = false;
#pragma unroll
                        for (int s = 0; s < R; ++s) {
                            const unsigned long long v = __builtin_amdgcn_uicmp(key[s], mx, 32 /* ICMP_EQ */);
                            if (!found && v) {
                                p = 64 * s + (int)__builtin_ctzll(v);
                                found = true;
                            }
                        }
                        // out at once: the column's owner gathers it while the search goes on
                        if (lr == 0) __hip_atomic_store(&meta[k], (int)((tag << 8) | (unsigned)p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                        for (int s = 0; s < R; ++s) used[s] |= ((p >> 6) == s) ? (1ull << (p & 63)) : 0ull;
                        ppack |= (unsigned)p << (8 * k);
                    } else if (st % 3 == 1) {
                        // the pivot column's four entries as scalars, the reciprocal of the pivot
                        const int pset = p >> 6;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            T src = a[0][i];
#pragma unroll
                            for (int s = 1; s < R; ++s) src = (pset == s) ? a[s][i] : src;
                            u[i] = lane_value(src, p & 63);
                        }
#ifndef MATINV_TILEQ_NO_SPECRCP
                        if constexpr (R == 2) rp = lane_value(pset ? rl[1] : rl[0], p & 63);
                        else
#endif
                            rp = rcp_full(u[k]);
                    } else {
                        // row k /= pivot; the other rows lose their entry of the pivot column; the pivot column becomes column k of the inverse
#pragma unroll
                        for (int s = 0; s < R; ++s) {
                            const bool me = lr + 64 * s == p;
                            const T nk = a[s][k] * rp;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (i == k) continue;
                                a[s][i] = me ? -(u[i] * rp) : fma_t(-u[i], nk, a[s][i]);
                            }
                            a[s][k] = me ? rp : nk;
                        }
                    }
#ifdef MATINV_TILEQ_STAMPS_FINE
                    TQ_STAMP(11 + st % 3);  // 11: keys, maximum, vote; 12: pivot column as scalars, reciprocal; 13: row operation
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_b(NB);
            if (searcher) {
                TQ_STAMP(3);  // search, the searcher's other MFMAs between its stages
                // B' = the B operand as it stands; the permutation tables; the singular flag
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    vec4 v;
                    v[0] = a[s][0], v[1] = a[s][1], v[2] = a[s][2], v[3] = a[s][3];
                    if (64 * s + 64 <= N || lr + 64 * s < N) *reinterpret_cast<vec4 *>(&bprime[(lr + 64 * s) * 4]) = v;
                }
                if (lr < 4) {
                    const int pk = (int)((ppack >> (8 * lr)) & 255u);
                    const int i = 16 * tK + G::trow(rK, lr);
                    rowaddr[i] = (unsigned char)pk;
                    coladdr[pk] = (unsigned char)i;
                }
                if (lr == 0) meta[4] = bad;
            } else {
                TQ_STAMP(10);  // the other MFMAs
            }
            // 3. the pivot columns -> LDS column panel (their owners; the searcher's own after its search), zeroed in C
#pragma nounroll
            for (int k = 0; k < 4; ++k) {
                int pk;
                if (searcher) {
                    pk = (int)((ppack >> (8 * k)) & 255u);
                } else {
                    pk = (int)wait_pivot(&meta[k], tag);
#pragma unroll
                    for (int s = 0; s < R; ++s) used[s] |= ((pk >> 6) == s) ? (1ull << (pk & 63)) : 0ull;
                }
#ifdef MATINV_TILEQ_STAMPS_FINE2
                TQ_STAMP(11);  // waiting for pivot k
#endif
                const int tj = pk >> 4, cc = pk & 15;
                if (tj % W != w) continue;  // wave-uniform
                const int jlk = tj / W;
                const bool mine = c == cc;
#pragma unroll
                for (int jl = 0; jl < NC; ++jl) {
                    if (jlk != jl) continue;  // wave-uniform
                    if (mine) {
#pragma unroll
                        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                colpanel[(16 * ti + G::trow(r, q)) * 4 + k] = acc[ti][jl][r];
                                acc[ti][jl][r] = (T)0;
                            }
                    }
                }
#ifdef MATINV_TILEQ_STAMPS_FINE2
                TQ_STAMP(12);  // one column gathered and zeroed
                stamp_acc[13] += 1;  // (count of columns this wave gathered)
#endif
            }
            // the pivot rows: zero in C
#pragma unroll
            for (int jl = 0; jl < NC; ++jl)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tK][jl][r] = (r == rK) ? (T)0 : acc[tK][jl][r];
            TQ_STAMP(5);  // waiting for the pivots, column gather, zeroing
            __syncthreads();
            TQ_STAMP(6);  // barrier 2
            // 4. B operand = B'; A operand = -(column panel), I_4 in the pivot rows
            if (bad == 0) bad = __builtin_amdgcn_readfirstlane(meta[4]);
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) bop[jl] = (w + W * jl < NT) ? bprime[(16 * (w + W * jl) + c) * 4 + q] : (T)0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) aop[ti] = -colpanel[(16 * ti + c) * 4 + q];
            aop[tK] = (G::blk(c) == rK) ? ((G::piv(c) == q) ? (T)1 : (T)0) : aop[tK];
            TQ_STAMP(7);  // operands
        };

        // ragged n: the all-padding blocks of the last tile row are not run (a padding row is e_i^T and stays so, a padding
        // column is zero in every real row: such a step changes nothing); their table entries keep the 0xff written here
        int last_blocks = 4;
        if (!FULL) {
            last_blocks = G::real_blocks(n - 16 * (NT - 1));
            if (w == 0) {
#pragma unroll
                for (int s = 0; s < R; ++s)
                    if (lr + 64 * s < N) rowaddr[lr + 64 * s] = coladdr[lr + 64 * s] = (unsigned char)0xff;
            }
        }
        auto tile_row = [&](auto tKc, int from) {
            const int to = (decltype(tKc)::value == NT - 1) ? last_blocks : 4;
#pragma nounroll
            for (int rK = from; rK < to; ++rK) turn(tKc, rK, IntC<0>());
        };
        turn(IntC<0>(), 0, IntC<1>());
        tile_row(IntC<0>(), 1);
        tile_row(IntC<1>(), 0);
        tile_row(IntC<2>(), 0);
        tile_row(IntC<3>(), 0);
        tile_row(IntC<4>(), 0);
        if constexpr (NT > 5) tile_row(IntC<5>(), 0);
        if constexpr (NT > 6) tile_row(IntC<6>(), 0);
        if constexpr (NT > 7) tile_row(IntC<7>(), 0);
        if constexpr (NT > 8) tile_row(IntC<8>(), 0);
        if constexpr (NT > 9) tile_row(IntC<9>(), 0);
        if constexpr (NT > 10) tile_row(IntC<10>(), 0);
        if constexpr (NT > 11) tile_row(IntC<11>(), 0);
        if constexpr (NT > 12) tile_row(IntC<12>(), 0);
        if constexpr (NT > 13) tile_row(IntC<13>(), 0);
        if constexpr (NT > 14) tile_row(IntC<14>(), 0);
        if constexpr (NT > 15) tile_row(IntC<15>(), 0);
#pragma unroll
        for (int jl = 0; jl < NC; ++jl) {
            if (w + W * jl < NT) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][jl] = G::mfma(aop[ti], bop[jl], acc[ti][jl]);
            }
        }
        TQ_STAMP(8);  // last block's MFMAs

        if (bad == 0) {
            // F[i][j] = inverse(rowaddr[i], coladdr[j]); W = A^T: element (a, b) of its inverse at a * n + b. (The tables are complete
            // since the last step's second barrier.)
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
                        if (ca[jl] != 0xffffu && (FULL || (ra < (unsigned)n && ca[jl] < (unsigned)n))) X[ra * (unsigned)n + ca[jl]] = acc[ti][jl][r];
                    }
                }
            if (info && threadIdx.x == 0) info[mat] = 0;
        } else if (bad_list) {
            // singular: the pivoted LDS kernel redoes the matrix for the oracle's info code and the NaN fill
            if (threadIdx.x == 0) {
                const int slot = atomicAdd(bad_count, 1);
                bad_list[slot] = (int)mat;
            }
        } else {
            // no kernel behind this one at this size: the first row of W = column of A without a usable pivot is the oracle's info
            for (unsigned e = threadIdx.x; e < (unsigned)(n * n); e += 64u * W) X[e] = nan_of<T>();
            if (info && threadIdx.x == 0) info[mat] = bad;
        }
        __syncthreads();  // the next matrix rewrites the tables and the panels
        TQ_STAMP(9);  // stores
    }
#ifdef MATINV_TILEQ_STAMPS
    if (l == 0) {
        for (int ph = 0; ph < 16; ++ph) atomicAdd(&matinv_tileq_stamps[w * 16 + ph], stamp_acc[ph]);
    }
#endif
}

// LDS: row panel, published B', column panel, tables, pivots
#define MATINV_TILEQ_KERNEL(NAME, T, NTHREADS, OCC, WV, NCV)                                                           \
    template <int NT, bool FULL>                                                                                       \
    __global__ __launch_bounds__(NTHREADS, OCC) void NAME(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n_rt, unsigned batch,           \
                                                          int *bad_count, int *bad_list, const int *in_count, const int *in_list,              \
                                                          hint_t *hint_out)                                            \
    {                                                                                                                  \
        __shared__ __attribute__((aligned(16))) T rowpanel[16 * NT * 4];                                               \
        __shared__ __attribute__((aligned(16))) T bprime[16 * NT * 4];                       \
        __shared__ __attribute__((aligned(16))) T colpanel[16 * NT * 4];                                               \
        __shared__ unsigned char tab[512];                                                                             \
        __shared__ int meta[8];                                                                                        \
        gj_tileq_body<T, NT, WV, NCV, FULL>(Ain, Xout, info, n_rt, batch, rowpanel, bprime, colpanel, tab, meta, bad_count,   \
                                                        bad_list, in_count, in_list, hint_out);                        \
    }

// One wavefront per tile column. (The body also serves W < NT wavefronts of two tile columns each -- three or four per 64 < n <= 128
// matrix: measured with tools/tileq_stamps.hip, within 2 - 7 % of tilep4_impl.hpp in fp64 and 1.7 x slower in fp32, not instantiated here.)
MATINV_TILEQ_KERNEL(matinv_gj_tileqw_f64, double, 64 * NT, 1, NT, 1)
MATINV_TILEQ_KERNEL(matinv_gj_tileqw_f32, float, 64 * NT, 1, NT, 1)

constexpr int tileq_limit(bool f64) { return f64 ? 192 : 256; }

template <class T>
static hipError_t enqueue_tileq(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, int *bad_count,
                                int *bad_list, const int *in_count, const int *in_list, hint_t *hint_out)
{
    const int nt = (n + 15) / 16;
    const unsigned per_cu = 1u;  // the matrix fills most of the CU's register file
    unsigned cap = 256u * per_cu * tile_grid_rounds();
    if (in_list) cap = 256u * per_cu;  // usually empty: one round of resident workgroups
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    const unsigned b = (unsigned)batch;
// (run-time n only: at these sizes the kernel is far from memory-bound, the compile-time-n twins bought nothing measurable and doubled
// the build time of this family)
#define TQ_LAUNCH(KERN, NT_, THREADS)                                                                                  \
    hipLaunchKernelGGL((KERN<NT_, false>), dim3(grid), dim3(THREADS), 0, stream, A, X, info, n, b, bad_count, bad_list, in_count, in_list, hint_out);
#define TQ_CASEW(NT_)                                                                                                  \
    case NT_:                                                                                                          \
        if constexpr (sizeof(T) == 8) {                                                                                \
            if constexpr (NT_ <= 12) { TQ_LAUNCH(matinv_gj_tileqw_f64, NT_, 64 * NT_) }                                \
        } else { TQ_LAUNCH(matinv_gj_tileqw_f32, NT_, 64 * NT_) }                                                      \
        break;
    switch (nt) {
        TQ_CASEW(9)
        TQ_CASEW(10)
        TQ_CASEW(11)
        TQ_CASEW(12)
        TQ_CASEW(13)
        TQ_CASEW(14)
        TQ_CASEW(15)
        TQ_CASEW(16)
    default: return hipErrorInvalidValue;
    }
#undef TQ_CASEW
#undef TQ_LAUNCH
    return hipGetLastError();
}

// Direct form (in_list == nullptr): the whole batch. Work-list form: the matrices the natural-order kernel of this size rejected,
// (in_count, in_list) in device memory. No kernel behind this one serves every 128 < n <= 256, so a singular matrix is finished
// here: info = first column of A without a usable pivot + 1 (the oracle's code), NaN-filled output. (bad_count / bad_list: the hand-over
// to the pivoted LDS kernel that the n <= 128 instantiations of tools/tileq_stamps.hip can use; nullptr here.)
template <class T>
static hipError_t launch_tileq(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, const int *in_count,
                               const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list)
{
    if (n <= 128 || n > tileq_limit(sizeof(T) == 8)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    return enqueue_tileq<T>(n, A, X, batch, info, stream, bad_count, bad_list, in_count, in_list, hint_out);
}

}  // namespace matinv
