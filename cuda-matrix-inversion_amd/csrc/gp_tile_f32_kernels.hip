// gp_tile_f32_kernels.hip -- fp32 instantiation of the one-wavefront fused pipeline kernels (gp_tile_impl.hpp)
#include "gp_tile_impl.hpp"

namespace matinv {

template hipError_t launch_gp_tile<float>(int, const float *, const float *, const float *, const float *, const float *,
                                          float *, size_t, int *, hipStream_t);

}  // namespace matinv
