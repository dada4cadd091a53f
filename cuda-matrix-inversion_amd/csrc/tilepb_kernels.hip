// tilepb_kernels.hip -- fp64 instantiations of the pivoting MFMA tile kernel that advances one tile column (16 pivots) per
// workgroup barrier (tilepb_impl.hpp).
#include "tilepb_impl.hpp"

namespace matinv {

template <>
hipError_t launch_gj_tilepb<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream,
                                    const int *in_count, const int *in_list, hint_t *hint_out)
{
    return launch_tilepb<double>(n, A, X, batch, info, stream, in_count, in_list, hint_out);
}

const char *name_gj_tilepb(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tilepb_%s<%d>", f64 ? "f64" : "f32", (n + 15) / 16);
    return buf;
}

}  // namespace matinv
