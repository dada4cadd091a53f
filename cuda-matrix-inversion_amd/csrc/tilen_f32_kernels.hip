// tilen_f32_kernels.hip -- fp32 instantiations of the second-generation natural-order MFMA tile kernels (tilen_impl.hpp).
#include "tilen_impl.hpp"

namespace matinv {

template <>
hipError_t enqueue_gj_tilen<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream,
                                   int *work_count, int *work_list)
{
    return enqueue_tilen<float>(n, A, X, batch, info, stream, work_count, work_list);
}

}  // namespace matinv
