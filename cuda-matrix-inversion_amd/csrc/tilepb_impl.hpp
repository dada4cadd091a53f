// tilepb_impl.hpp (instantiated by tilepb_kernels.hip for f64 and tilepb_f32_kernels.hip for f32) -- the pivoting MFMA tile
// Gauss-Jordan for 64 < n <= 192 (f64) / 256 (f32), r03: ONE WAVEFRONT PER TILE COLUMN as tilepw_impl.hpp (read that header and
// tilep_impl.hpp first), but the matrix advances a whole TILE COLUMN (16 pivots) per workgroup barrier instead of 4:
//
//   * the wave that owns tile column K factors it ALONE: four block steps of the one-wavefront algorithm (stage 4 pivot columns
//     in its private LDS panel, search with R = ceil(n / 64) rows per lane, 4 x 4 transposes, gather of its own pivot rows,
//     rank-4 MFMA update) applied to its own NT x 1 tiles only -- no barrier, nobody else repeats the search. In the in-place
//     Gauss-Jordan form the 16 columns then hold T[:, P], the 16 non-trivial columns of the accumulated transformation
//     T = M4 M3 M2 M1 of the four steps (P = their 16 pivot rows);
//   * it publishes T[:, P] (n x 16, laid out as four A operands), the 16 pivot slots and the singular flag in LDS: ONE barrier;
//   * every other wave then applies T to its own tile column in one rank-16 update: x <- x (rows P zeroed) + T[:, P] x[P]:
//     it gathers ITS part of the 16 pivot rows from its own registers through its private LDS strip (the run-time-register asm
//     blocks of tilep_impl.hpp) and issues 4 NT MFMAs whose A operands come from the published block.
// Per 16 columns the four-wave kernel (tilep4_impl.hpp) pays 4 barriers and 4 x 4 redundant searches, tilepw 4 barriers; here it
// is one barrier and one search per pivot. What is left on the critical path is the owner's factorization; the other waves'
// MFMAs of block K run while the owner of block K + 1 (which updates its column first) factors.
// LDS (f64, n = 128): panel 4 KB + A-operand scratch 4 KB + 2 x 16 KB published block (double buffered: the owner of K + 2 writes
// while nobody can still be reading K) + 2 KB gather strip per wave.
//
// Works on W = A itself like tilepw: the row search IS the oracle's partial pivoting, a singular matrix is finished here (info =
// first column without a usable pivot + 1, NaN-filled output). Replaces, for general 64 < n <= 192 / 256 input, pivotRow /
// normalizeRow / transform_matrix of /root/reference/src/gauss/batched_invert.cu:17-82.
#pragma once
#include "tilepw_impl.hpp"

namespace matinv {

template <class T, int NT>
__device__ __forceinline__ void gj_tilepb_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n, unsigned batch, T *panel2,
                                               T *aopl, T *S2, T *bball, int *pvl, unsigned char *tab, const int *in_count,
                                               const int *in_list, hint_t *hint_out)
{
    static_assert(NT >= 5 && NT <= 16, "one wavefront per tile column: 64 < n <= 256");
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int R = (N + 63) / 64;  // rows per lane in the search
    constexpr int PVS = 20;           // ints per published pivot record: 16 slots + the singular flag
    unsigned char *const rowaddr = tab, *const coladdr = tab + 256;
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;           // wave-uniform: this wave's tile column
    T *const bbuf = bball + w * (16 * 16);    // [16 pivots][16 columns], private to the wave
    typedef __attribute__((address_space(3))) T *lds_ptr;
    const unsigned bb_lane = (unsigned)(size_t)(lds_ptr)(bbuf + (l & 15));

    const unsigned todo = in_count ? (unsigned)*in_count : batch;
    if (hint_out && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hint_out, ((hint_t)batch << 32) | todo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (unsigned item = blockIdx.x; item < todo; item += gridDim.x) {
        const unsigned mat = in_list ? (unsigned)in_list[item] : item;
        const T *A = Ain.at_uniform(mat);
        T *X = Xout.at_uniform(mat);
        int nn = n;
        asm volatile("" : "+s"(nn));  // keeps LICM away from the tile offsets (see gj_tile_body)
        int q = l >> 4, c = l & 15, lr = l;
        asm volatile("" : "+v"(q), "+v"(c), "+v"(lr));

        // acc[ti][0] = tile (ti, w) of W = A: element (row, col) at col * n + row
        vec4 acc[NT][1];
        const int col = 16 * w + c;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + G::trow(r, q);
                acc[ti][0][r] = (row < nn && col < nn) ? A[(unsigned)(col * nn + row)] : ((row == col) ? (T)1 : (T)0);
            }

        bool used[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) used[rr] = lr + 64 * rr >= N;
        int bad = 0;

        // ragged n: the all-padding blocks of the last tile column are not run (see tilep_impl.hpp); their table entries keep
        // the 0xff written here by wave 0 (before the first barrier; every block's owner writes its entries before its barrier)
        const int last_blocks = G::real_blocks(nn - 16 * (NT - 1));
        if (w == 0) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) rowaddr[lr + 64 * rr] = coladdr[lr + 64 * rr] = (unsigned char)0xff;
        }

#ifdef MATINV_TILEPB_STAMPS
        unsigned long long st_search = 0, st_aop = 0, st_gather = 0, st_mfma = 0, st_update = 0, st_wait = 0, st_gat16 = 0, st_total = __builtin_amdgcn_s_memtime();
#define TPB_STAMP(var) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - last_; last_ = now_; }
#else
#define TPB_STAMP(var)
#endif
        // ---- one block step of the owner's factorization of its own tile column (tile column tK, block rK inside it)
        auto panel_step = [&](int tK, int rK, int *pv16) {
#ifdef MATINV_TILEPB_STAMPS
            unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
            const bool panel_lane = G::blk(c) == rK;
            if (panel_lane) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) panel2[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][0][r];
            }
            wave_lds_sync();
            T a[R][4];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                vec4 v = {};
                if (lr + 64 * rr < N) v = *reinterpret_cast<const vec4 *>(&panel2[(lr + 64 * rr) * 4]);
                a[rr][0] = v[0], a[rr][1] = v[1], a[rr][2] = v[2], a[rr][3] = v[3];
            }
            int pv = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // largest |.| over the unused rows; lowest row on ties (rows of one lane set first, then the next set)
                unsigned key[R], kmax = 0;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    key[rr] = used[rr] ? 0u : magkey(a[rr][t]);
                    kmax = key[rr] > kmax ? key[rr] : kmax;
                }
                const unsigned mx = wave_max_u32(kmax);
                if (key_bad(T(0), mx) && bad == 0) bad = 16 * tK + G::pcol(rK, t) + 1;  // no usable pivot in this column
                int p = 0;
                bool found = false;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const unsigned long long v = __builtin_amdgcn_uicmp(key[rr], mx, 32 /* ICMP_EQ */);
                    if (!found && v) {
                        p = 64 * rr + (int)__builtin_ctzll(v);
                        found = true;
                    }
                }
                const int pset = p >> 6;  // wave-uniform
#pragma unroll
                for (int rr = 0; rr < R; ++rr) used[rr] = used[rr] || (lr + 64 * rr == p);
                pv = (lr == t) ? p : pv;
                T u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    T src = a[0][j];
#pragma unroll
                    for (int rr = 1; rr < R; ++rr) src = (pset == rr) ? a[rr][j] : src;
                    u[j] = lane_value(src, p & 63);
                }
                const T rp = rcp_full(u[t]);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const T f = -(a[rr][t] * rp);
                    const bool me = lr + 64 * rr == p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j == t) continue;
                        a[rr][j] = me ? u[j] * rp : fma_t(f, u[j], a[rr][j]);
                    }
                    a[rr][t] = me ? rp : f;
                }
            }
            TPB_STAMP(st_search)
            // the finished panel (row i = Aop[i, 0:4]) through the wave's private LDS scratch into the A-operand layout
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                vec4 v;
                v[0] = a[rr][0], v[1] = a[rr][1], v[2] = a[rr][2], v[3] = a[rr][3];
                if (lr + 64 * rr < N) *reinterpret_cast<vec4 *>(&aopl[(lr + 64 * rr) * 4]) = v;
            }
            if (lr < 4) {
                pv16[4 * rK + lr] = pv;
                const int j = 16 * tK + G::pcol(rK, lr);
                coladdr[j] = (unsigned char)pv;
                rowaddr[pv] = (unsigned char)j;
            }
            wave_lds_sync();
            T aop[NT];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) aop[ti] = aopl[(16 * ti + c) * 4 + q];
            TPB_STAMP(st_aop)
            // B operand: the four pivot rows of this tile column through the LDS strip (and zero them in C)
#pragma nounroll
            for (int t = 0; t < 4; ++t) {
                const int s = __builtin_amdgcn_readlane(pv, t);
                const int loc = s & 15;
                const int pos = 4 * (s >> 4) + G::slot_r(loc);
                const unsigned long long mask = 0xffffull << (16 * G::slot_q(loc));
                GatherAllRows<T, NT, 0>::run(acc, bb_lane + (unsigned)(t * 16 * (int)sizeof(T)), pos, mask);
            }
            // pivot columns: zero in C, I_4 in B
            {
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
                                     : "+v"(acc[ti][0][0]), "+v"(acc[ti][0][1]), "+v"(acc[ti][0][2]), "+v"(acc[ti][0][3]), [save] "=&s"(save)
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
                                     : "+v"(acc[ti][0][0]), "+v"(acc[ti][0][1]), "+v"(acc[ti][0][2]), "+v"(acc[ti][0][3]), [save] "=&s"(save)
                                     : [mask] "s"(zmask)
                                     : "scc");
                }
            }
            wave_lds_sync();
            T bop = bbuf[q * 16 + c];
            bop = panel_lane ? ((G::piv(c) == q) ? (T)1 : (T)0) : bop;
            TPB_STAMP(st_gather)
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) acc[ti][0] = G::mfma(aop[ti], bop, acc[ti][0]);
#ifdef MATINV_TILEPB_STAMPS
            asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[NT - 1][0][0]));
#endif
            TPB_STAMP(st_mfma)
        };

        // ---- rank-16 update of this wave's tile column with a published block (nblk of its four 4-column blocks are real)
        auto update = [&](const T *S, const int *pv16, int nblk) {
#ifdef MATINV_TILEPB_STAMPS
            unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
            if (bad == 0) bad = pv16[16];
#pragma nounroll
            for (int t = 0; t < 4 * nblk; ++t) {
                const int s = __builtin_amdgcn_readfirstlane(pv16[t]);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) used[rr] = used[rr] || (lr + 64 * rr == s);
                const int loc = s & 15;
                const int pos = 4 * (s >> 4) + G::slot_r(loc);
                const unsigned long long mask = 0xffffull << (16 * G::slot_q(loc));
                GatherAllRows<T, NT, 0>::run(acc, bb_lane + (unsigned)(t * 16 * (int)sizeof(T)), pos, mask);
            }
            wave_lds_sync();
            TPB_STAMP(st_gat16)
#pragma nounroll
            for (int kk = 0; kk < nblk; ++kk) {
                const T bop = bbuf[(4 * kk + q) * 16 + c];
                const T *Sk = S + kk * (N * 4);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][0] = G::mfma(Sk[(16 * ti + c) * 4 + q], bop, acc[ti][0]);
            }
#ifdef MATINV_TILEPB_STAMPS
            asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[NT - 1][0][0]));
#endif
            TPB_STAMP(st_update)
        };

#pragma nounroll
        for (int tK = 0; tK < NT; ++tK) {
            const int par = tK & 1;
            const int nblk = (tK == NT - 1) ? last_blocks : 4;
            if (tK > 0 && w != tK - 1) update(S2 + (1 - par) * (16 * N), pvl + (1 - par) * PVS, 4);
            if (w == tK) {
                int *const pv16 = pvl + par * PVS;
#pragma nounroll
                for (int rK = 0; rK < nblk; ++rK) panel_step(tK, rK, pv16);
                // publish T[:, P] as four A operands: S[blk][row][piv]
                T *const S = S2 + par * (16 * N);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[(G::blk(c) * N + 16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][0][r];
                if (lr == 0) pv16[16] = bad;
            }
#ifdef MATINV_TILEPB_STAMPS
            {
                unsigned long long last_ = __builtin_amdgcn_s_memtime();
                __syncthreads();
                TPB_STAMP(st_wait)
            }
#else
            __syncthreads();
#endif
        }
        if (w != NT - 1) update(S2 + ((NT - 1) & 1) * (16 * N), pvl + ((NT - 1) & 1) * PVS, last_blocks);  // (also picks up its singular flag)

        if (bad == 0) {
            // F[i][j] = inverse(rowaddr[i], coladdr[j]); W = A: element (a, b) of the inverse at b * n + a
            const unsigned ca = coladdr[16 * w + c];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned ra = rowaddr[16 * ti + G::trow(r, q)];
                    if (ra < (unsigned)nn && ca < (unsigned)nn) X[ca * (unsigned)nn + ra] = acc[ti][0][r];
                }
            if (info && threadIdx.x == 0) info[mat] = 0;
        } else {
            for (unsigned e = threadIdx.x; e < (unsigned)(nn * nn); e += 64u * NT) X[e] = nan_of<T>();
            if (info && threadIdx.x == 0) info[mat] = bad;
        }
        __syncthreads();  // the next matrix rewrites the tables and both published blocks
#ifdef MATINV_TILEPB_STAMPS
        if (blockIdx.x == 0 && item == blockIdx.x && l == 0 && (w == 0 || w == 1 || w == NT - 1))
            printf("tilepb NT=%d wave %d: total %llu  search %llu  aop %llu  gather4+zero %llu  mfma(own) %llu | gather16 %llu  update-mfma %llu  barrier-wait %llu (cycles, one matrix)\n",
                   NT, w, __builtin_amdgcn_s_memtime() - st_total, st_search, st_aop, st_gather, st_mfma, st_gat16, st_update, st_wait);
#endif
    }
}

// workgroups per CU the register budget is declared for: NT <= 8: four waves per SIMD (two to three matrices per CU), beyond
// that the matrix fills most of the CU's register file
template <int NT>
__global__ __launch_bounds__(64 * NT, NT > 8 ? 1 : 4) void matinv_gj_tilepb_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info,
                                                                               int n, unsigned batch, const int *in_count,
                                                                               const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) double panel2[16 * NT * 4];   // the owner's panel: one row per lane back
    __shared__ __attribute__((aligned(16))) double aopl[16 * NT * 4];     // the owner's finished panel = its A operand
    __shared__ __attribute__((aligned(16))) double S2[2 * 16 * 16 * NT];  // the published n x 16 block, double buffered
    __shared__ __attribute__((aligned(16))) double bball[NT * 16 * 16];   // per wave: its part of the 16 pivot rows
    __shared__ int pvl[2 * 20];
    __shared__ unsigned char tab[512];
    gj_tilepb_body<double, NT>(Ain, Xout, info, n, batch, panel2, aopl, S2, bball, pvl, tab, in_count, in_list, hint_out);
}

template <int NT>
__global__ __launch_bounds__(64 * NT, NT > 8 ? 1 : 4) void matinv_gj_tilepb_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info,
                                                                               int n, unsigned batch, const int *in_count,
                                                                               const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) float panel2[16 * NT * 4];
    __shared__ __attribute__((aligned(16))) float aopl[16 * NT * 4];
    __shared__ __attribute__((aligned(16))) float S2[2 * 16 * 16 * NT];
    __shared__ __attribute__((aligned(16))) float bball[NT * 16 * 16];
    __shared__ int pvl[2 * 20];
    __shared__ unsigned char tab[512];
    gj_tilepb_body<float, NT>(Ain, Xout, info, n, batch, panel2, aopl, S2, bball, pvl, tab, in_count, in_list, hint_out);
}

// in_count / in_list != nullptr: work-list form (one round of resident workgroups; usually empty)
template <class T>
static hipError_t launch_tilepb(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                                const int *in_count = nullptr, const int *in_list = nullptr, hint_t *hint_out = nullptr)
{
    if (n <= 64 || n > tilepw_limit(sizeof(T) == 8)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const int nt = (n + 15) / 16;
    const unsigned per_cu = nt > 8 ? 1u : (unsigned)(16 / nt);  // resident workgroups per CU at four waves per SIMD
    const unsigned cap = in_list ? 256u * per_cu : 256u * per_cu * tile_grid_rounds();
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    const unsigned b = (unsigned)batch;
#define TPB_LAUNCH(NT_)                                                                                                \
    if constexpr (sizeof(T) == 8) {                                                                                    \
        if constexpr (NT_ <= 12) hipLaunchKernelGGL((matinv_gj_tilepb_f64<NT_>), dim3(grid), dim3(64 * NT_), 0, stream, A, X, info, n, b, in_count, in_list, hint_out); \
    } else {                                                                                                           \
        hipLaunchKernelGGL((matinv_gj_tilepb_f32<NT_>), dim3(grid), dim3(64 * NT_), 0, stream, A, X, info, n, b, in_count, in_list, hint_out);        \
    }
    switch (nt) {
    case 5: TPB_LAUNCH(5) break;
    case 6: TPB_LAUNCH(6) break;
    case 7: TPB_LAUNCH(7) break;
    case 8: TPB_LAUNCH(8) break;
    case 9: TPB_LAUNCH(9) break;
    case 10: TPB_LAUNCH(10) break;
    case 11: TPB_LAUNCH(11) break;
    case 12: TPB_LAUNCH(12) break;
    case 13: TPB_LAUNCH(13) break;
    case 14: TPB_LAUNCH(14) break;
    case 15: TPB_LAUNCH(15) break;
    default: TPB_LAUNCH(16) break;
    }
#undef TPB_LAUNCH
    return hipGetLastError();
}

}  // namespace matinv
