// tilep4_kernels.hip -- fp64 instantiations of the four-wavefront pivoting MFMA tile kernels (tilep4_impl.hpp).
#include "tilep4_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilep4<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tilep4<double>(n, A, X, batch, info, stream);
}

const char *name_gj_tilep4(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tilep4_%s<%d, %s>", f64 ? "f64" : "f32", (n + 15) / 16, (n % 16) == 0 ? "true" : "false");
    return buf;
}

}  // namespace matinv
