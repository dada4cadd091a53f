// tilep4_f32_kernels.hip -- fp32 instantiations of the four-wavefront pivoting MFMA tile kernels (tilep4_impl.hpp).
#include "tilep4_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilep4<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tilep4<float>(n, A, X, batch, info, stream);
}

}  // namespace matinv
