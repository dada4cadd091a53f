// gp_tile_impl.hpp (instantiated by gp_tile_kernels.hip for f64 and gp_tile_f32_kernels.hip for f32: two translation units
// that compile in parallel) -- fused Gaussian-process scalars on the MFMA tile layout (f64, n <= 64):
//     mean = a^T (B + diag c)^-1 d        var = e - a^T (B + diag c)^-1 a
// Replaces addDiagonal + batched inverse + 2 x cublasSgemmBatched of /root/reference/src/gauss_bench.cu:127-265,
// 275-409 (and calcluateMeanCPU / calcluateVarianceCPU, src/gauss_cpu.c:41-72,174-206) with ONE kernel that never
// forms the inverse: it eliminates the n pivots of M = B + diag c from the bordered symmetric matrix
//     [ M    V ]      V = [a d]  (n x 2, zero padded to one 16-wide tile)
//     [ V^T  0 ]
// with the symmetric blocked sweep of matinv_spd_tile_f64 (same MFMA step, same LDS panel staging, lower-triangular
// tile storage) and reads the Schur complement -V^T M^-1 V out of the corner tile: mean = -G[0][1], var = e + G[0][0].
// Tile columns left of the current pivot block are dead and skipped (no inverse is wanted), so the work is that of a
// Cholesky factorisation with two right-hand sides; HBM traffic per item is the lower triangle of B plus three
// vectors in, one scalar out. Not SPD (a pivot <= 0) -> device work list -> matinv_gp_lds (info reported there).
#pragma once
#include <cstdio>

#include "tile_common.hpp"

namespace matinv {

template <class T, int NT, bool FULL>
__device__ __forceinline__ void gp_tile_body(const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out,
                                             int *info, int n_rt, unsigned batch, int *work_count, int *work_list, T *panel)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int N = 16 * NT;
    constexpr int NX = NT + 1;  // tile rows/cols of the bordered matrix; R = NT is the border
    constexpr int R = NT;
    constexpr int NKB = 4 * NT;
    const bool variance = (Ds == nullptr);
    const int l = threadIdx.x;

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        int n = FULL ? N : n_rt;  // run-time n opaque once per matrix, predicates on the edge tiles only: see gj_tile_body
        if (!FULL) asm volatile("" : "+s"(n));
        const T *B = Bs + (size_t)mat * n * n;
        const T *va = As + (size_t)mat * n;
        const T *vw = variance ? va : Ds + (size_t)mat * n;
        const T *vc = Cs + (size_t)mat * n;
        int q = l >> 4, c = l & 15;
        asm volatile("" : "+v"(q), "+v"(c));  // see matinv_gj_tile_f64

        vec4 acc[NX][NX];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                if (tj > ti) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                    const bool in = FULL || ti < NT - 1 || (row < n && col < n);  // tj <= ti: only the last tile row reaches beyond n
                    const int hi = row > col ? row : col, lo = row > col ? col : row;
                    // only the lower triangle of B is read (mirror position inside the diagonal tiles)
                    T v = in ? B[(unsigned)(lo * n + hi)] : ((row == col) ? (T)1 : (T)0);
                    if (ti == tj && row == col && in) v += vc[row];  // addDiagonal, gauss_bench.cu:38-43
                    acc[ti][tj][r] = v;
                }
            }
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int col = 16 * tj + c;
            const bool in = FULL || tj < NT - 1 || col < n;
            const T u = in ? va[col] : (T)0, w = in ? vw[col] : (T)0;
            acc[R][tj][0] = (q == 0) ? u : (q == 1) ? w : (T)0;  // border rows trow(0,0) (a) and trow(0,1) (d); the others are zero
            acc[R][tj][1] = (T)0, acc[R][tj][2] = (T)0, acc[R][tj][3] = (T)0;
        }
        acc[R][R] = vec4{(T)0, (T)0, (T)0, (T)0};

        unsigned long long bad = 0;
        T aop[NX], bop[NX];
        spd_panel_to_lds<NX, T>(panel, acc, 0, q, c);
        wave_lds_sync();
        {
            PanelSolve<NX, true, T> ps0;
#pragma unroll
            for (int s = 0; s < PanelSolve<NX, true, T>::NSTAGE; ++s) ps0.stage(s, panel, 0, q, c, aop, bop, bad);
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int tK = kb >> 2;
            spd_prep_operands<NX, T>(acc, bop, kb, q, c);
            if (kb + 1 < NKB) {
                const int tn = (kb + 1) >> 2;
                // (a) the tile column the next panel is read from (rows above it are dead)
#pragma unroll
                for (int ti = 0; ti < NX; ++ti) {
                    if (ti < tn) continue;
                    acc[ti][tn] = G::mfma(aop[ti], bop[tn], acc[ti][tn]);
                }
                // (b) the other LIVE lower tiles (tile column >= tK), pinned between the pieces of the next panel
                constexpr int NS = PanelSolve<NX, true, T>::NSTAGE;
                int nb = 0;  // number of (b) tiles: folds to a literal
#pragma unroll
                for (int ti = 0; ti < NX; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NX; ++tj)
                        if (tj <= ti && tj >= tK && tj != tn) ++nb;
                T aop_next[NX], bop_next[NX];
                PanelSolve<NX, true, T> ps;
                int count = 0, ev = 0;
                auto run_events = [&](bool flush) {
#pragma unroll
                    for (int e = 0; e < NS + 1; ++e) {
                        const int lead = nb < 2 ? nb : 2;
                        const int thr = (e == 0) ? lead : lead + ((nb - lead) * e) / NS;
                        if (e == ev && (flush || thr <= count)) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (e == 0) {
                                wave_lds_sync();
                                spd_panel_to_lds<NX, T>(panel, acc, kb + 1, q, c);
                                wave_lds_sync();
                            } else if (e - 1 < 6 || e - 1 - 6 >= tn) {  // tile rows above the next pivot block are dead
                                ps.stage(e - 1, panel, kb + 1, q, c, aop_next, bop_next, bad);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            ++ev;
                        }
                    }
                };
                run_events(false);
#pragma unroll
                for (int ti = 0; ti < NX; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NX; ++tj) {
                        if (tj > ti || tj < tK || tj == tn) continue;
                        acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
                        ++count;
                        run_events(false);
                    }
                run_events(true);
#pragma unroll
                for (int ti = 0; ti < NX; ++ti) { aop[ti] = aop_next[ti]; bop[ti] = bop_next[ti]; }
            } else {
                // last pivot block: only the corner matters
                acc[R][R] = G::mfma(aop[R], bop[R], acc[R][R]);
            }
        }

        if (bad == 0) {
            // corner tile G = -V^T M^-1 V, register 0, lane group q = 0 (row of a): G[a][a] at column trow(0,0) = lane 0,
            // G[a][d] at column trow(0,1)
            const T g = acc[R][R][0];
            if (variance) {
                if (l == 0) out[mat] = Es[mat] + g;
            } else {
                if (l == G::trow(0, 1)) out[mat] = -g;
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
__global__ __launch_bounds__(64, NT >= 6 ? 1 : 2) void matinv_gp_tile_f64(const double *As, const double *Bs, const double *Cs,
                                                           const double *Ds, const double *Es, double *out, int *info,
                                                           int n_rt, unsigned batch, int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) double panel[(16 * NT + 16) * 4];
    gp_tile_body<double, NT, FULL>(As, Bs, Cs, Ds, Es, out, info, n_rt, batch, work_count, work_list, panel);
}

template <int NT, bool FULL>
__global__ __launch_bounds__(64, NT >= 6 ? 2 : 3) void matinv_gp_tile_f32(const float *As, const float *Bs, const float *Cs,
                                                           const float *Ds, const float *Es, float *out, int *info,
                                                           int n_rt, unsigned batch, int *work_count, int *work_list)
{
    __shared__ __attribute__((aligned(16))) float panel[(16 * NT + 16) * 4];
    gp_tile_body<float, NT, FULL>(As, Bs, Cs, Ds, Es, out, info, n_rt, batch, work_count, work_list, panel);
}

template <class T>
hipError_t launch_gp_tile(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                          int *info, hipStream_t stream)
{
    if (!gp_tile_supports(sizeof(T) == 8, n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) {
        (void)scratch_free(ws, stream);
        return e;
    }
    const int nt = (n + 15) / 16;
    const unsigned grid = (unsigned)(batch < 256u * 8u * tile_grid_rounds() ? batch : 256u * 8u * tile_grid_rounds());
    const unsigned b = (unsigned)batch;
#define GP_LAUNCH(NT_)                                                                                                \
    if constexpr (sizeof(T) == 8) {                                                                                   \
        if (n == 16 * NT_)                                                                                            \
            hipLaunchKernelGGL((matinv_gp_tile_f64<NT_, true>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gp_tile_f64<NT_, false>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
    } else {                                                                                                          \
        if (n == 16 * NT_)                                                                                            \
            hipLaunchKernelGGL((matinv_gp_tile_f32<NT_, true>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gp_tile_f32<NT_, false>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
    }
#define GP_LAUNCH32(NT_)                                                                                              \
    if constexpr (sizeof(T) == 4) {                                                                                   \
        if (n == 16 * NT_)                                                                                            \
            hipLaunchKernelGGL((matinv_gp_tile_f32<NT_, true>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
        else                                                                                                          \
            hipLaunchKernelGGL((matinv_gp_tile_f32<NT_, false>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
    }
    switch (nt) {
    case 1: GP_LAUNCH(1) break;
    case 2: GP_LAUNCH(2) break;
    case 3: GP_LAUNCH(3) break;
    case 4: GP_LAUNCH(4) break;
    case 5: GP_LAUNCH(5) break;
    default: GP_LAUNCH(6) break;
    }
#undef GP_LAUNCH
#undef GP_LAUNCH32
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_gp_lds_worklist<T>(n, As, Bs, Cs, Ds, Es, out, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}
}  // namespace matinv
