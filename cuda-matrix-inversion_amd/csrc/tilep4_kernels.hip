// tilep4_kernels.hip -- fp64 instantiations of the four-wavefront pivoting MFMA tile kernels (tilep4_impl.hpp).
#include "tilep4_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilep4<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tilep4<double>(n, A, X, batch, info, stream);
}

template <>
hipError_t launch_gj_tilep4_worklist<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, const int *in_count,
                                            const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                            int *hint_out)
{
    return launch_tilep4_worklist<double>(n, A, X, batch, in_count, in_list, bad_count, bad_list, info, stream, hint_out);
}

const char *name_gj_tilep4(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tilep4_%s<%d, %s>", f64 ? "f64" : "f32", (n + 15) / 16, (n % 16) == 0 ? "true" : "false");
    return buf;
}

}  // namespace matinv
