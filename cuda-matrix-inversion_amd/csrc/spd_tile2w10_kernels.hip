// spd_tile2w10_kernels.hip -- the two-wavefront lower-triangle SPD sweep of 10 x 10 fp64 tiles (144 < n <= 160): Cholesky entry point and fused mean /
// variance (spd_tile2_impl.hpp); compiled with VGPR-form MFMAs, the AGPRs as parking space (Makefile).
#include "spd_tile2_impl.hpp"

namespace matinv {

template <>
hipError_t enqueue_spd_tile2w<10>(bool gp_mode, int n, BatchRef<const double> A, BatchRef<double> X, unsigned grid, unsigned batch, int *info,
                                  int *ws, Spd2Gp<double> gp, hipStream_t stream)
{
    if (gp_mode) hipLaunchKernelGGL((matinv_spd_tile2w_f64<10, true>), dim3(grid), dim3(128), 0, stream, A, X, info, n, batch, ws, ws + 1, gp);
    else hipLaunchKernelGGL((matinv_spd_tile2w_f64<10, false>), dim3(grid), dim3(128), 0, stream, A, X, info, n, batch, ws, ws + 1, gp);
    return hipGetLastError();
}

}  // namespace matinv
