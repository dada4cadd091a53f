// tilepb_f32_kernels.hip -- fp32 instantiations of the pivoting MFMA tile kernel that advances one tile column per workgroup
// barrier (tilepb_impl.hpp).
#include "tilepb_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilepb<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream,
                                   const int *in_count, const int *in_list, hint_t *hint_out)
{
    return launch_tilepb<float>(n, A, X, batch, info, stream, in_count, in_list, hint_out);
}

}  // namespace matinv
