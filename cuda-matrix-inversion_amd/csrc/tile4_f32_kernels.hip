// tile4_f32_kernels.hip -- fp32 instantiations of the several-wavefronts-per-matrix MFMA tile kernels (tile4_impl.hpp).
#include "tile4_impl.hpp"

namespace matinv {

template hipError_t launch_gj_tile4<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);
template hipError_t launch_spd_tile4<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);

}  // namespace matinv
