// tile_screen.hpp -- the screening pass in front of the natural-order Gauss-Jordan tile kernels (tile_kernels.inc: n <= 64, one
// wavefront per matrix; tile4_impl.hpp: 64 < n <= 192 / 256, several), and the panel staging they share with it.
#pragma once
#include "tile_common.hpp"

namespace matinv {

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

// ---- screening pass (r04) ---------------------------------------------------------------------------------------------------
// Under the default policy (natural order first, rejects redone by the pivoting kernel) a batch of GENERAL matrices used to pay the
// whole natural-order sweep before being rejected: 1.6 + 1.9 ms per 100 k x 64^2 instead of 1.9. Nearly every such matrix already
// fails the acceptance test in its FIRST block step (the 6 LU multipliers of the first 4 x 4 pivot block and the n x 4 multipliers
// of the first panel: a U(0,1) matrix passes with probability ~1e-2), and that block's panel is the matrix as it was loaded. This
// kernel runs EXACTLY that test -- the loads of tile column 0 with gj_tile_body's own addressing (only the 16 lanes that hold the four
// pivot columns load: one 64-byte piece of every row, an eighth to a quarter of the matrix's cache lines), panel_to_lds, panel_solve,
// the same code on the same values, hence the same verdict bit for bit -- and sorts the batch into the work list of the pivoting kernel
// and an accept list that the natural-order kernel then takes instead of the whole batch. A matrix the screen rejects is one the
// natural-order kernel would have rejected after its first block step: which kernel inverts a matrix, and therefore every bit of
// the result, is the same with and without the screen. That is what allows the launcher to run it only when the previous natural-order
// launch of the same class rejected a quarter of its batch or more (the hint of tile_policy_record): launch history changes the
// speed, never the result. On an SPD batch the screen never runs, and the headline kernel is as it was.
// One atomic per matrix on ONE counter is what the first version of this kernel spent its time on (48 k returning atomics on one
// address: 0.5 ms -- as long as the natural-order sweep it was meant to save). Each wave therefore collects its verdicts in LDS and
// reserves list space once per 64 matrices.
__device__ __forceinline__ void screen_flush(int *buf, int cnt, int *count, int *list, int l)
{
    if (cnt == 0) return;  // wave-uniform
    int base = 0;
    if (l == 0) base = atomicAdd(count, cnt);
    base = __builtin_amdgcn_readfirstlane(base);
    if (l < cnt) list[base + l] = buf[l];
}

// ONEWAVE: the addressing of gj_tile_body (16-byte relabelling where it uses it); otherwise that of gj_tile4_body (plain)
template <class T, int NT, bool FULL, bool ONEWAVE = true>
__device__ __forceinline__ void gj_tile_screen_body(BatchRef<const T> Ain, int n_rt, unsigned batch, int *work_count, int *work_list,
                                                    int *accept_count, int *accept_list, T *panel, int *lists)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr bool PAIRED = ONEWAVE && FULL && (NT % 2 == 0);  // as in gj_tile_body
    const int l = threadIdx.x;
    int *const rej_buf = lists, *const acc_buf = lists + 64;
    int n_rej = 0, n_acc = 0;  // wave-uniform
    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const T *A = Ain.at_uniform(mat);
        int n = FULL ? N : n_rt;
        if (!FULL) asm volatile("" : "+s"(n));
        int q = l >> 4, c = l & 15;
        const unsigned lane_off = (unsigned)(G::trow(0, l >> 4) * n + (l & 15));
        asm volatile("" : "+v"(q), "+v"(c));
        const bool panel_lane = G::blk(c) == 0;
        vec4 acc[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T v = (T)0;
                if (panel_lane) {
                    if (PAIRED) {
                        // element 0 of the 16-byte access of gj_tile_body: tile column 0 of the pair (0, 1)
                        const unsigned lane_off2 = (unsigned)(2 * G::trow(0, l >> 4) * N + 2 * (l & 15));
                        const unsigned uoff = (unsigned)((32 * (ti >> 1) + 2 * G::trow(r, 0) + (ti & 1)) * N);
                        v = A[uoff + lane_off2];
                    } else {
                        const int row = 16 * ti + G::trow(r, q), col = c;
                        const unsigned uoff = (unsigned)((16 * ti + G::trow(r, 0)) * n);
                        const bool edge = !FULL && (ti == NT - 1 || NT == 1);
                        v = (!edge || (row < n && col < n)) ? A[uoff + lane_off] : ((row == col) ? (T)1 : (T)0);
                    }
                }
                acc[ti][0][r] = v;
            }
        unsigned long long bad = 0;
        T aop[NT];
        panel_to_lds<NT, T>(panel, acc, 0, q, c);
        wave_lds_sync();
        panel_solve<NT>(panel, 0, q, c, aop, bad);
        if (bad == 0) {
            if (l == 0) acc_buf[n_acc] = (int)mat;
            ++n_acc;
        } else {
            if (l == 0) rej_buf[n_rej] = (int)mat;
            ++n_rej;
        }
        wave_lds_sync();
        if (n_acc == 64) {
            screen_flush(acc_buf, n_acc, accept_count, accept_list, l);
            n_acc = 0;
            wave_lds_sync();
        }
        if (n_rej == 64) {
            screen_flush(rej_buf, n_rej, work_count, work_list, l);
            n_rej = 0;
            wave_lds_sync();
        }
    }
    screen_flush(acc_buf, n_acc, accept_count, accept_list, l);
    screen_flush(rej_buf, n_rej, work_count, work_list, l);
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, 8) void matinv_gj_tile_screen_f64(BatchRef<const double> Ain, int n_rt, unsigned batch, int *work_count,
                                                                  int *work_list, int *accept_count, int *accept_list)
{
    __shared__ __attribute__((aligned(16))) double panel[16 * NT * 4];
    __shared__ int lists[128];
    gj_tile_screen_body<double, NT, FULL>(Ain, n_rt, batch, work_count, work_list, accept_count, accept_list, panel, lists);
}
template <int NT, bool FULL>
__global__ __launch_bounds__(64, 8) void matinv_gj_tile_screen_f32(BatchRef<const float> Ain, int n_rt, unsigned batch, int *work_count,
                                                                  int *work_list, int *accept_count, int *accept_list)
{
    __shared__ __attribute__((aligned(16))) float panel[16 * NT * 4];
    __shared__ int lists[128];
    gj_tile_screen_body<float, NT, FULL>(Ain, n_rt, batch, work_count, work_list, accept_count, accept_list, panel, lists);
}

// the same for the several-wavefront kernels of tile4_impl.hpp (run-time n, plain addressing; one wavefront screens one matrix)
template <int NT>
__global__ __launch_bounds__(64, 4) void matinv_gj_tile4_screen_f64(BatchRef<const double> Ain, int n_rt, unsigned batch, int *work_count,
                                                                   int *work_list, int *accept_count, int *accept_list)
{
    __shared__ __attribute__((aligned(16))) double panel[16 * NT * 4];
    __shared__ int lists[128];
    gj_tile_screen_body<double, NT, false, false>(Ain, n_rt, batch, work_count, work_list, accept_count, accept_list, panel, lists);
}
template <int NT>
__global__ __launch_bounds__(64, 4) void matinv_gj_tile4_screen_f32(BatchRef<const float> Ain, int n_rt, unsigned batch, int *work_count,
                                                                   int *work_list, int *accept_count, int *accept_list)
{
    __shared__ __attribute__((aligned(16))) float panel[16 * NT * 4];
    __shared__ int lists[128];
    gj_tile_screen_body<float, NT, false, false>(Ain, n_rt, batch, work_count, work_list, accept_count, accept_list, panel, lists);
}

// NATURAL_FIRST policy: run the screening kernel in front of the natural-order kernel of this size class? (tile_kernels.inc)
bool tile_policy_use_screen(bool f64, int nt);

}  // namespace matinv
