// gp_tile_kernels.hip -- fused Gaussian-process scalars on the MFMA tile layout (f64, n <= 64):
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
#include "tile_common.hpp"

namespace matinv {

template <int NT, bool FULL>
__global__ __launch_bounds__(64, 2) void matinv_gp_tile_f64(const double *As, const double *Bs, const double *Cs,
                                                           const double *Ds, const double *Es, double *out, int *info,
                                                           int n_rt, unsigned batch, int *work_count, int *work_list)
{
    constexpr int N = 16 * NT;
    constexpr int NX = NT + 1;  // tile rows/cols of the bordered matrix; R = NT is the border
    constexpr int R = NT;
    constexpr int NKB = 4 * NT;
    const int n = FULL ? N : n_rt;
    const bool variance = (Ds == nullptr);
    __shared__ __attribute__((aligned(16))) double panel[(N + 16) * 4];
    const int l = threadIdx.x;

    for (unsigned mat = blockIdx.x; mat < batch; mat += gridDim.x) {
        const double *B = Bs + (size_t)mat * n * n;
        const double *va = As + (size_t)mat * n;
        const double *vw = variance ? va : Ds + (size_t)mat * n;
        const double *vc = Cs + (size_t)mat * n;
        int q = l >> 4, c = l & 15;
        asm volatile("" : "+v"(q), "+v"(c));  // see matinv_gj_tile_f64

        v4d acc[NX][NX];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                if (tj > ti) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + q, col = 16 * tj + c;
                    const bool in = FULL || (row < n && col < n);
                    const int hi = row > col ? row : col, lo = row > col ? col : row;
                    // only the lower triangle of B is read (mirror position inside the diagonal tiles)
                    double v = in ? B[(unsigned)(lo * n + hi)] : ((row == col) ? 1.0 : 0.0);
                    if (ti == tj && row == col && in) v += vc[row];  // addDiagonal, gauss_bench.cu:38-43
                    acc[ti][tj][r] = v;
                }
            }
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int col = 16 * tj + c;
            const bool in = FULL || col < n;
            const double u = in ? va[col] : 0.0, w = in ? vw[col] : 0.0;
            acc[R][tj][0] = (q == 0) ? u : (q == 1) ? w : 0.0;  // border rows 0 (a) and 1 (d); rows 2..15 are zero
            acc[R][tj][1] = 0.0, acc[R][tj][2] = 0.0, acc[R][tj][3] = 0.0;
        }
        acc[R][R] = v4d{0.0, 0.0, 0.0, 0.0};

        unsigned long long bad = 0;
        double aop[NX], bop[NX];
        spd_panel_to_lds<NX>(panel, acc, 0, q, c);
        wave_lds_sync();
        {
            PanelSolve<NX, true> ps0;
#pragma unroll
            for (int s = 0; s < PanelSolve<NX, true>::NSTAGE; ++s) ps0.stage(s, panel, 0, q, c, aop, bop, bad);
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int tK = kb >> 2;
            spd_prep_operands<NX>(acc, bop, kb, q, c);
            if (kb + 1 < NKB) {
                const int tn = (kb + 1) >> 2;
                // (a) the tile column the next panel is read from (rows above it are dead)
#pragma unroll
                for (int ti = 0; ti < NX; ++ti) {
                    if (ti < tn) continue;
                    acc[ti][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tn], acc[ti][tn], 0, 0, 0);
                }
                // (b) the other LIVE lower tiles (tile column >= tK), pinned between the pieces of the next panel
                constexpr int NS = PanelSolve<NX, true>::NSTAGE;
                int nb = 0;  // number of (b) tiles: folds to a literal
#pragma unroll
                for (int ti = 0; ti < NX; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NX; ++tj)
                        if (tj <= ti && tj >= tK && tj != tn) ++nb;
                double aop_next[NX], bop_next[NX];
                PanelSolve<NX, true> ps;
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
                                spd_panel_to_lds<NX>(panel, acc, kb + 1, q, c);
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
                        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc[ti][tj], 0, 0, 0);
                        ++count;
                        run_events(false);
                    }
                run_events(true);
#pragma unroll
                for (int ti = 0; ti < NX; ++ti) { aop[ti] = aop_next[ti]; bop[ti] = bop_next[ti]; }
            } else {
                // last pivot block: only the corner matters
                acc[R][R] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[R], bop[R], acc[R][R], 0, 0, 0);
            }
        }

        if (bad == 0) {
            // corner tile G = -V^T M^-1 V: G[0][0] at lane (q=0, c=0), G[0][1] at lane (q=0, c=1), register 0
            const double g = acc[R][R][0];
            if (variance) {
                if (l == 0) out[mat] = Es[mat] + g;
            } else {
                if (l == 1) out[mat] = -g;
            }
            if (info && l == 0) info[mat] = 0;
        } else if (l == 0) {
            const int slot = atomicAdd(work_count, 1);
            work_list[slot] = (int)mat;
        }
        wave_lds_sync();
    }
}

bool gp_tile_supports_f64(int n) { return n >= 1 && n <= 64; }

hipError_t launch_gp_tile_f64(int n, const double *As, const double *Bs, const double *Cs, const double *Ds,
                              const double *Es, double *out, size_t batch, int *info, hipStream_t stream)
{
    if (!gp_tile_supports_f64(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int nt = (n + 15) / 16;
    const unsigned grid = (unsigned)(batch < 256u * 8u * 4u ? batch : 256u * 8u * 4u);
    const unsigned b = (unsigned)batch;
#define GP_LAUNCH(NT_)                                                                                                \
    if (n == 16 * NT_)                                                                                                \
        hipLaunchKernelGGL((matinv_gp_tile_f64<NT_, true>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1); \
    else                                                                                                              \
        hipLaunchKernelGGL((matinv_gp_tile_f64<NT_, false>), dim3(grid), dim3(64), 0, stream, As, Bs, Cs, Ds, Es, out, info, n, b, ws, ws + 1)
    switch (nt) {
    case 1: GP_LAUNCH(1); break;
    case 2: GP_LAUNCH(2); break;
    case 3: GP_LAUNCH(3); break;
    default: GP_LAUNCH(4); break;
    }
#undef GP_LAUNCH
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_gp_lds_worklist<double>(n, As, Bs, Cs, Ds, Es, out, ws, ws + 1, info, stream);
    hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

const char *name_gp_tile_f64(int n)
{
    const bool full = (n % 16) == 0;
    switch ((n + 15) / 16) {
    case 1: return full ? "matinv_gp_tile_f64<1, true>" : "matinv_gp_tile_f64<1, false>";
    case 2: return full ? "matinv_gp_tile_f64<2, true>" : "matinv_gp_tile_f64<2, false>";
    case 3: return full ? "matinv_gp_tile_f64<3, true>" : "matinv_gp_tile_f64<3, false>";
    default: return full ? "matinv_gp_tile_f64<4, true>" : "matinv_gp_tile_f64<4, false>";
    }
}

}  // namespace matinv
