// tileq_f32_kernels.hip -- fp32 instantiations of the pivoting MFMA tile kernel with fixed pivot rows and searched pivot
// columns (tileq_impl.hpp).
#include "tileq_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tileq<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream,
                                  const int *in_count, const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list)
{
    hipError_t e = launch_tileq<float>(n, A, X, batch, info, stream, in_count, in_list, hint_out, bad_count, bad_list);
    return (e != hipSuccess || !in_count) ? e : debug_note_rejects(in_count, stream);
}

}  // namespace matinv
