// tilep4_f32_kernels.hip -- fp32 instantiations of the three- / four-wavefront pivoting MFMA tile kernels (tilep4_impl.hpp).
#include "tilep4_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilep4<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tilep4<float>(n, A, X, batch, info, stream);
}

template <>
hipError_t launch_gj_tilep4_worklist<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, const int *in_count,
                                           const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                           hint_t *hint_out, bool expect_many)
{
    return launch_tilep4_worklist<float>(n, A, X, batch, in_count, in_list, bad_count, bad_list, info, stream, hint_out, expect_many);
}

}  // namespace matinv
