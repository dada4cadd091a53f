// tilep_f32_kernels.hip -- fp32 instantiations of the pivoting MFMA tile kernels (tilep_impl.hpp).
#include "tilep_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilep<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tilep<float>(n, A, X, batch, info, stream);
}

}  // namespace matinv
