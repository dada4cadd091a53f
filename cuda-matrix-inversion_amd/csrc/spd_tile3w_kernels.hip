// spd_tile3w_kernels.hip -- the lower-triangle SPD sweep of 12 x 12 fp64 tiles on THREE wavefronts: Cholesky entry point and fused mean /
// variance (spd_tile2_impl.hpp, W = 3); compiled with VGPR-form MFMAs (Makefile).
#include "spd_tile2_impl.hpp"

namespace matinv {

template <>
hipError_t enqueue_spd_tile3w<12>(bool gp_mode, int n, BatchRef<const double> A, BatchRef<double> X, unsigned grid, unsigned batch, int *info,
                                  int *ws, Spd2Gp<double> gp, hipStream_t stream)
{
    if (gp_mode) hipLaunchKernelGGL((matinv_spd_tile3w_f64<12, true>), dim3(grid), dim3(192), 0, stream, A, X, info, n, batch, ws, ws + 1, gp);
    else hipLaunchKernelGGL((matinv_spd_tile3w_f64<12, false>), dim3(grid), dim3(192), 0, stream, A, X, info, n, batch, ws, ws + 1, gp);
    return hipGetLastError();
}

}  // namespace matinv
