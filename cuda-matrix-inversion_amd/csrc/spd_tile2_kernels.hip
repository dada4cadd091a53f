// spd_tile2_kernels.hip -- the two-wavefront lower-triangle SPD sweep for 112 < n <= 128, fp64 (spd_tile2_impl.hpp): Cholesky
// entry point and fused mean / variance.
#include "spd_tile2_impl.hpp"

namespace matinv {

// MATINV_SPD_TILE2=0: these sizes stay on the four-wavefront kernel that sweeps all tiles (A/B switch)
bool spd_tile2_supports(bool f64, int n)
{
    static const bool on = [] { const char *s = getenv("MATINV_SPD_TILE2"); return !(s && *s == '0'); }();
    return on && f64 && n > 112 && n <= 128;
}

template <bool GP>
static hipError_t launch_tile2(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream, int *ws,
                               Spd2Gp<double> gp)
{
    const unsigned resident = 256u * 4u;  // four workgroups of two waves per CU
    const unsigned cap = resident * tile_grid_rounds();
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    hipLaunchKernelGGL((matinv_spd_tile2_f64<GP>), dim3(grid), dim3(128), 0, stream, A, X, info, n, (unsigned)batch, ws, ws + 1, gp);
    return hipGetLastError();
}

hipError_t launch_spd_tile2(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    if (n <= 112 || n > 128) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e == hipSuccess) e = launch_tile2<false>(n, A, X, batch, info, stream, ws, Spd2Gp<double>());
    if (e == hipSuccess) e = launch_chol_lds_worklist<double>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

hipError_t launch_gp_spd_tile2(int n, const double *As, const double *Bs, const double *Cs, const double *Ds, const double *Es, double *out,
                               size_t batch, int *info, hipStream_t stream)
{
    if (n <= 112 || n > 128) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    BatchRef<const double> A{Bs, (size_t)n * n, nullptr};
    BatchRef<double> X{nullptr, 0, nullptr};
    if (e == hipSuccess) e = launch_tile2<true>(n, A, X, batch, info, stream, ws, Spd2Gp<double>{As, Cs, Ds, Es, out});
    if (e == hipSuccess) e = launch_gp_lds_worklist<double>(n, As, Bs, Cs, Ds, Es, out, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

const char *name_spd_tile2(bool gp) { return gp ? "matinv_spd_tile2_f64<true>" : "matinv_spd_tile2_f64<false>"; }

}  // namespace matinv
