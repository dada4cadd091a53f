// spd_tile2_kernels.hip -- the two-wavefront lower-triangle SPD sweep, fp64 (spd_tile2_impl.hpp): Cholesky entry point and fused
// mean / variance for 112 < n <= 128 (8 x 8 tiles, instantiated here) and 128 < n <= 176 (9 ... 11 tiles per dimension: one translation
// unit per tile count, spd_tile2w{9,10,11}_kernels.hip, so that the fully unrolled sweeps compile in parallel). 12 x 12 tiles (39 + 3
// accumulator tiles per wave = 336 registers, + 24 for the A operand) do not fit 512 registers on two waves: hipcc spills 324 of them with
// AGPR-form MFMAs and crashes in its "Rewrite AGPR-Copy-MFMA" pass with VGPR-form ones (ROCm 7.2.0) -- 176 < n <= 192 runs the same body on
// THREE waves (spd_tile3w_kernels.hip).
#include "spd_tile2_impl.hpp"

namespace matinv {

bool spd_tile2_supports(bool f64, int n) { return f64 && n > 112 && n <= 192; }

template <bool GP>
static hipError_t launch_tile2(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream, int *ws,
                               Spd2Gp<double> gp)
{
    const int nt = (n + 15) / 16;
    // n <= 128: four workgroups of two waves per CU; beyond: one wave per SIMD (VGPRs + AGPRs), two workgroups per CU (12 x 12 tiles,
    // three waves: one)
    const unsigned resident = nt <= 8 ? 256u * 4u : (nt <= 11 ? 256u * 2u : 256u);
    const unsigned cap = resident * tile_grid_rounds();
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    switch (nt) {
    case 8: hipLaunchKernelGGL((matinv_spd_tile2_f64<GP>), dim3(grid), dim3(128), 0, stream, A, X, info, n, (unsigned)batch, ws, ws + 1, gp); break;
    case 9: return enqueue_spd_tile2w<9>(GP, n, A, X, grid, (unsigned)batch, info, ws, gp, stream);
    case 10: return enqueue_spd_tile2w<10>(GP, n, A, X, grid, (unsigned)batch, info, ws, gp, stream);
    case 11: return enqueue_spd_tile2w<11>(GP, n, A, X, grid, (unsigned)batch, info, ws, gp, stream);
    case 12: return enqueue_spd_tile3w<12>(GP, n, A, X, grid, (unsigned)batch, info, ws, gp, stream);
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_spd_tile2(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    if (n <= 112 || n > 192) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    if (e == hipSuccess) e = launch_tile2<false>(n, A, X, batch, info, stream, ws, Spd2Gp<double>());
    // n <= 128: items that are not positive definite go to the LDS Cholesky (it reports the column); beyond, the kernel finishes them
    if (e == hipSuccess && n <= 128) e = launch_chol_lds_worklist<double>(n, A, X, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

hipError_t launch_gp_spd_tile2(int n, const double *As, const double *Bs, const double *Cs, const double *Ds, const double *Es, double *out,
                               size_t batch, int *info, hipStream_t stream)
{
    if (n <= 112 || n > 192) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    int *ws = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&ws), (batch + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws, 0, sizeof(int), stream);
    BatchRef<const double> A{Bs, (size_t)n * n, nullptr};
    BatchRef<double> X{nullptr, 0, nullptr};
    if (e == hipSuccess) e = launch_tile2<true>(n, A, X, batch, info, stream, ws, Spd2Gp<double>{As, Cs, Ds, Es, out});
    if (e == hipSuccess && n <= 128) e = launch_gp_lds_worklist<double>(n, As, Bs, Cs, Ds, Es, out, ws, ws + 1, info, stream);
    hipError_t e2 = scratch_free(ws, stream);
    return e != hipSuccess ? e : e2;
}

const char *name_spd_tile2(bool gp, int n)
{
    static thread_local char buf[48];
    if (n <= 128) return gp ? "matinv_spd_tile2_f64<true>" : "matinv_spd_tile2_f64<false>";
    if (n > 176) return gp ? "matinv_spd_tile3w_f64<12, true>" : "matinv_spd_tile3w_f64<12, false>";
    snprintf(buf, sizeof buf, "matinv_spd_tile2w_f64<%d, %s>", (n + 15) / 16, gp ? "true" : "false");
    return buf;
}

}  // namespace matinv
