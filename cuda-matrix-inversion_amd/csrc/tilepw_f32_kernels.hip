// tilepw_f32_kernels.hip -- fp32 instantiations of the pivoting MFMA tile kernel with one wavefront per tile column (tilepw_impl.hpp).
#include "tilepw_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilepw<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream)
{
    if (tilep_variant() == 2) return launch_gj_tilepb<float>(n, A, X, batch, info, stream);
    return launch_tilepw<float>(n, A, X, batch, info, stream);
}

template <>
hipError_t launch_gj_tilepw_worklist<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, const int *in_count,
                                            const int *in_list, int *info, hipStream_t stream, hint_t *hint_out)
{
    hipError_t e = tilep_variant() == 2 ? launch_gj_tilepb<float>(n, A, X, batch, info, stream, in_count, in_list, hint_out)
                                        : launch_tilepw<float>(n, A, X, batch, info, stream, in_count, in_list, hint_out);
    return e != hipSuccess ? e : debug_note_rejects(in_count, stream);
}

}  // namespace matinv
