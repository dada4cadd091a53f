// tilepw_impl.hpp (instantiated by tilepw_kernels.hip for f64 and tilepw_f32_kernels.hip for f32) -- the pivoting MFMA tile
// Gauss-Jordan of tilep_impl.hpp / tilep4_impl.hpp (read those headers first) for 128 < n <= 192 (f64) / 256 (f32): ONE
// WAVEFRONT PER TILE COLUMN, NT = 9 .. 16 wavefronts per matrix, one workgroup per CU -- the matrix fills most of the CU's
// register file (12 x 12 fp64 tiles = 1 152 of 2 048 VGPRs per lane).
//
// Same structure as the four-wave kernel: the owner of the pivot columns of a block is the one wave that searches it (R =
// ceil(n / 64) rows per lane) and publishes the finished panel, the pivots and the singular flag in a double-buffered LDS
// block, one workgroup barrier per block step; every wave gathers its part of the four pivot rows from its own registers
// through a private LDS strip and reads its A operand from the published panel. Differences:
//   * the kernel works on W = A itself, not on A^T (loads and stores are 32-byte segments instead of 128-byte ones; at these
//     sizes the kernel is far from memory-bound): the row search then IS the oracle's partial pivoting, so a singular
//     matrix is finished here with the oracle's info code (first column without a usable pivot + 1) and a NaN-filled output --
//     there is no kernel behind this one that serves every such n;
//   * each wave has one tile column, so all of its MFMAs of a block step run before the next panel is staged (nothing to
//     interleave with the search; the waves of the other workgroup phases fill in).
// Replaces, for general matrices of these sizes, the 2 n / 32 + 2 launches of the blocked path (blocked_gj_kernels.hip).
#pragma once
#include "tilep_impl.hpp"
#include "gather_tree.inc"

namespace matinv {

// gather_zero_tile_row for a wave that holds ONE tile column; see tilep_impl.hpp
template <class T, int NT, int TI>
__device__ __forceinline__ void gather_zero_tile_row1(typename TileGeo<T>::vec4 (&acc)[NT][1], unsigned addr, int pos,
                                                     unsigned long long mask)
{
    unsigned long long save;
    unsigned tmp;
#define TPW_WZ64(R) "ds_write_b64 %[addr], %[a" #R "]\n\tv_mov_b64_e32 %[a" #R "], 0\n\t"
#define TPW_WZ32(R) "ds_write_b32 %[addr], %[a" #R "]\n\tv_mov_b32_e32 %[a" #R "], 0\n\t"
#define TPW_BODY(WZ)                                                                                                   \
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
#define TPW_OPERANDS                                                                                                   \
    [a0] "+v"(acc[TI][0][0]), [a1] "+v"(acc[TI][0][1]), [a2] "+v"(acc[TI][0][2]), [a3] "+v"(acc[TI][0][3]),            \
        [save] "=&s"(save), [tmp] "=&s"(tmp)
    if constexpr (sizeof(T) == 8)
        asm volatile(TPW_BODY(TPW_WZ64) : TPW_OPERANDS : [addr] "v"(addr), [pos] "s"(pos), [mask] "s"(mask), [ti] "n"(TI) : "scc", "memory");
    else
        asm volatile(TPW_BODY(TPW_WZ32) : TPW_OPERANDS : [addr] "v"(addr), [pos] "s"(pos), [mask] "s"(mask), [ti] "n"(TI) : "scc", "memory");
#undef TPW_OPERANDS
#undef TPW_BODY
#undef TPW_WZ32
#undef TPW_WZ64
}

// The gather of one pivot row for a wave that holds ONE tile column of NT tile rows: blocks of four tile rows, each ONE asm
// statement that finds the (tile row, register) slot with a binary branch tree (gather_tree.inc; r03 -- the per-tile-row blocks
// of gather_zero_tile_row1 above cost ~75 cycles each when skipped: ~600 cycles per pivot row at 8 tile rows).
// MATINV_GATHER_LINEAR (compile time) keeps the r02 form for A/B measurements.
template <class T, int NT, int TI>
struct GatherAllRows {
    static __device__ __forceinline__ void run(typename TileGeo<T>::vec4 (&acc)[NT][1], unsigned addr, int pos, unsigned long long mask)
    {
#ifdef MATINV_GATHER_LINEAR
        gather_zero_tile_row1<T, NT, TI>(acc, addr, pos, mask);
        if constexpr (TI + 1 < NT) GatherAllRows<T, NT, TI + 1>::run(acc, addr, pos, mask);
#else
        static_assert(TI % 4 == 0, "blocks of four tile rows");
        if constexpr (NT - TI >= 4) gather_tree_1x4<4 * TI>(acc[TI][0], acc[TI + 1][0], acc[TI + 2][0], acc[TI + 3][0], addr, pos, mask);
        else if constexpr (NT - TI == 3) gather_tree_1x3<4 * TI>(acc[TI][0], acc[TI + 1][0], acc[TI + 2][0], addr, pos, mask);
        else if constexpr (NT - TI == 2) gather_tree_1x2<4 * TI>(acc[TI][0], acc[TI + 1][0], addr, pos, mask);
        else gather_tree_1x1<4 * TI>(acc[TI][0], addr, pos, mask);
        if constexpr (TI + 4 < NT) GatherAllRows<T, NT, TI + 4>::run(acc, addr, pos, mask);
#endif
    }
};

template <class T, int NT>
__device__ __forceinline__ void gj_tilepw_body(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n, unsigned batch, T *panel2,
                                               T *bball, unsigned char *tab, T *aopl, int *pvl, const int *in_count, const int *in_list,
                                               hint_t *hint_out)
{
    static_assert(NT >= 5 && NT <= 16, "one wavefront per tile column: 64 < n <= 256");
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int W = NT;
    constexpr int R = (N + 63) / 64;  // rows per lane in the search
    unsigned char *const rowaddr = tab, *const coladdr = tab + 256;
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;          // wave-uniform: this wave's tile column
    T *const bbuf = bball + w * (4 * 16);    // [4 pivots][16 columns], private to the wave
    typedef __attribute__((address_space(3))) T *lds_ptr;
    const unsigned bb_lane = (unsigned)(size_t)(lds_ptr)(bbuf + (l & 15));

    // work-list form (the matrices the natural-order kernel of this size rejected): in_list[0 .. *in_count); its length goes
    // back to the launcher's natural / pivot guess through pinned host memory (see tilep_impl.hpp)
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
        T aop[NT], bop;

        auto turn = [&](auto tKc, int rKn, auto firstc) {
            constexpr int tKn = decltype(tKc)::value;  // = the owner wave
            constexpr bool first = decltype(firstc)::value != 0;
            const bool panel_lane = G::blk(c) == rKn;
            const bool is_owner = w == tKn;  // wave-uniform: the ONE wave that searches this block
            T *const pbuf = panel2;          // touched by the owner only
            const int par = (4 * tKn + rKn) & 1;
            T *const abuf = aopl + par * (N * 4);
            int *const pbufi = pvl + par * 8;
            if (!first) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) acc[ti][0] = G::mfma(aop[ti], bop, acc[ti][0]);
            }
            if (is_owner && panel_lane) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pbuf[(16 * ti + G::trow(r, q)) * 4 + G::piv(c)] = acc[ti][0][r];
            }
            wave_lds_sync();  // the panel never leaves the owner wave
            T a[R][4];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                vec4 v = {};
                if (is_owner && lr + 64 * rr < N) v = *reinterpret_cast<const vec4 *>(&pbuf[(lr + 64 * rr) * 4]);
                a[rr][0] = v[0], a[rr][1] = v[1], a[rr][2] = v[2], a[rr][3] = v[3];
            }
            int pv = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!is_owner) continue;  // the other waves wait at the barrier below
                // largest |.| over the unused rows; lowest row on ties (rows of one lane set first, then the next set)
                unsigned key[R], kmax = 0;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    key[rr] = used[rr] ? 0u : magkey(a[rr][t]);
                    kmax = key[rr] > kmax ? key[rr] : kmax;
                }
                const unsigned mx = wave_max_u32(kmax);
                if (key_bad(T(0), mx) && bad == 0) bad = 16 * tKn + G::pcol(rKn, t) + 1;  // no usable pivot in this column
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
            // The owner publishes the finished panel (row i = Aop[i, 0:4]), the four pivot slots, the singular flag and the
            // permutation tables (double-buffered); ONE workgroup barrier; every wave picks up its A operand and the pivots.
            if (is_owner) {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    vec4 v;
                    v[0] = a[rr][0], v[1] = a[rr][1], v[2] = a[rr][2], v[3] = a[rr][3];
                    if (lr + 64 * rr < N) *reinterpret_cast<vec4 *>(&abuf[(lr + 64 * rr) * 4]) = v;
                }
                if (lr < 4) {
                    pbufi[lr] = pv;
                    const int j = 16 * tKn + G::pcol(rKn, lr);
                    coladdr[j] = (unsigned char)pv;
                    rowaddr[pv] = (unsigned char)j;
                }
                if (lr == 4) pbufi[4] = bad;
            }
            __syncthreads();
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) aop[ti] = abuf[(16 * ti + c) * 4 + q];
            pv = pbufi[lr & 3];  // lane t (and t + 4, ...) holds the slot of pivot t
            if (bad == 0) bad = pbufi[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int s = __builtin_amdgcn_readlane(pv, t);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) used[rr] = used[rr] || (lr + 64 * rr == s);
            }
            // B operand: this wave's part of the four pivot rows through its LDS strip (and zero it in C)
#pragma nounroll
            for (int t = 0; t < 4; ++t) {
                const int s = __builtin_amdgcn_readlane(pv, t);
                const int loc = s & 15;
                const int pos = 4 * (s >> 4) + G::slot_r(loc);
                const unsigned long long mask = 0xffffull << (16 * G::slot_q(loc));
                GatherAllRows<T, NT, 0>::run(acc, bb_lane + (unsigned)(t * 16 * (int)sizeof(T)), pos, mask);
            }
            // pivot columns (owner only): zero in C, I_4 in B
            {
                const unsigned long long zmask = __ballot((w == tKn) && panel_lane);
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
                                     : "+v"(acc[ti][0][0]), "+v"(acc[ti][0][1]), "+v"(acc[ti][0][2]), "+v"(acc[ti][0][3]), [save] "=&s"(save)
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
                                     : "+v"(acc[ti][0][0]), "+v"(acc[ti][0][1]), "+v"(acc[ti][0][2]), "+v"(acc[ti][0][3]), [save] "=&s"(save)
                                     : [mask] "s"(zmask)
                                     : "scc");
                }
            }
            wave_lds_sync();
            bop = bbuf[q * 16 + c];
            if (w == tKn) bop = panel_lane ? ((G::piv(c) == q) ? (T)1 : (T)0) : bop;
        };

        // ragged n: the all-padding blocks of the last tile column are not run (see tilep_impl.hpp); their table entries keep
        // the 0xff written here by wave 0 -- the wave that also writes the first turn's entries, every later turn's owner
        // writes behind a barrier
        const int last_blocks = G::real_blocks(nn - 16 * (NT - 1));
        if (w == 0) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) rowaddr[lr + 64 * rr] = coladdr[lr + 64 * rr] = (unsigned char)0xff;
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
        if constexpr (NT > 8) column(IntC<8>(), 0);
        if constexpr (NT > 9) column(IntC<9>(), 0);
        if constexpr (NT > 10) column(IntC<10>(), 0);
        if constexpr (NT > 11) column(IntC<11>(), 0);
        if constexpr (NT > 12) column(IntC<12>(), 0);
        if constexpr (NT > 13) column(IntC<13>(), 0);
        if constexpr (NT > 14) column(IntC<14>(), 0);
        if constexpr (NT > 15) column(IntC<15>(), 0);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti][0] = G::mfma(aop[ti], bop, acc[ti][0]);
        __syncthreads();  // the tables are complete; both panel buffers are free for the next matrix

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
            for (unsigned e = threadIdx.x; e < (unsigned)(nn * nn); e += 64u * W) X[e] = nan_of<T>();
            if (info && threadIdx.x == 0) info[mat] = bad;
        }
        __syncthreads();  // the next matrix rewrites the tables
    }
}

// workgroups per CU the register budget is declared for: NT > 8: one (the matrix fills most of the CU's register file);
// NT <= 8 (64 < n <= 128): 64 accumulator VGPRs per lane at most -> four waves per SIMD = two to three matrices per CU
constexpr int tilepw_waves_per_simd(int nt) { return nt > 8 ? 1 : 4; }

template <int NT>
__global__ __launch_bounds__(64 * NT, tilepw_waves_per_simd(NT)) void matinv_gj_tilepw_f64(BatchRef<const double> Ain, BatchRef<double> Xout, int *info, int n,
                                                                  unsigned batch, const int *in_count, const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) double panel2[16 * NT * 4];       // the owner's panel, one row per lane back
    __shared__ __attribute__((aligned(16))) double aopl[2 * 16 * NT * 4];     // the finished panel = A operand, double buffered
    __shared__ __attribute__((aligned(16))) double bball[NT * 4 * 16];
    __shared__ int pvl[16];
    __shared__ unsigned char tab[512];
    gj_tilepw_body<double, NT>(Ain, Xout, info, n, batch, panel2, bball, tab, aopl, pvl, in_count, in_list, hint_out);
}

template <int NT>
__global__ __launch_bounds__(64 * NT, tilepw_waves_per_simd(NT)) void matinv_gj_tilepw_f32(BatchRef<const float> Ain, BatchRef<float> Xout, int *info, int n,
                                                                  unsigned batch, const int *in_count, const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) float panel2[16 * NT * 4];       // the owner's panel, one row per lane back
    __shared__ __attribute__((aligned(16))) float aopl[2 * 16 * NT * 4];     // the finished panel = A operand, double buffered
    __shared__ __attribute__((aligned(16))) float bball[NT * 4 * 16];
    __shared__ int pvl[16];
    __shared__ unsigned char tab[512];
    gj_tilepw_body<float, NT>(Ain, Xout, info, n, batch, panel2, bball, tab, aopl, pvl, in_count, in_list, hint_out);
}

constexpr int tilepw_limit(bool f64) { return f64 ? 192 : 256; }

// in_count / in_list != nullptr: work-list form (one round of resident workgroups; usually empty)
template <class T>
static hipError_t launch_tilepw(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                                const int *in_count = nullptr, const int *in_list = nullptr, hint_t *hint_out = nullptr)
{
    if (n <= 64 || n > tilepw_limit(sizeof(T) == 8)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const int nt = (n + 15) / 16;
    const unsigned per_cu = nt > 8 ? 1u : (unsigned)(16 / nt);  // resident workgroups per CU at four waves per SIMD
    const unsigned cap = in_list ? 256u * per_cu : 256u * per_cu * tile_grid_rounds();
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    const unsigned b = (unsigned)batch;
#define TPW_LAUNCH(NT_)                                                                                                \
    if constexpr (sizeof(T) == 8) {                                                                                    \
        if constexpr (NT_ <= 12) hipLaunchKernelGGL((matinv_gj_tilepw_f64<NT_>), dim3(grid), dim3(64 * NT_), 0, stream, A, X, info, n, b, in_count, in_list, hint_out); \
    } else {                                                                                                           \
        hipLaunchKernelGGL((matinv_gj_tilepw_f32<NT_>), dim3(grid), dim3(64 * NT_), 0, stream, A, X, info, n, b, in_count, in_list, hint_out);        \
    }
    switch (nt) {
    case 5: TPW_LAUNCH(5) break;
    case 6: TPW_LAUNCH(6) break;
    case 7: TPW_LAUNCH(7) break;
    case 8: TPW_LAUNCH(8) break;
    case 9: TPW_LAUNCH(9) break;
    case 10: TPW_LAUNCH(10) break;
    case 11: TPW_LAUNCH(11) break;
    case 12: TPW_LAUNCH(12) break;
    case 13: TPW_LAUNCH(13) break;
    case 14: TPW_LAUNCH(14) break;
    case 15: TPW_LAUNCH(15) break;
    default: TPW_LAUNCH(16) break;
    }
#undef TPW_LAUNCH
    return hipGetLastError();
}

}  // namespace matinv
