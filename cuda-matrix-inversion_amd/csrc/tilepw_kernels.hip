// tilepw_kernels.hip -- fp64 instantiations of the pivoting MFMA tile kernel with one wavefront per tile column (tilepw_impl.hpp).
#include "tilepw_impl.hpp"

namespace matinv {

bool tilepw_supports(bool f64, int n) { return n > 128 && n <= tilepw_limit(f64); }

template <>
hipError_t launch_gj_tilepw<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    if (tilep_variant() == 2) return launch_gj_tilepb<double>(n, A, X, batch, info, stream);
    return launch_tilepw<double>(n, A, X, batch, info, stream);
}

template <>
hipError_t launch_gj_tilepw_worklist<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, const int *in_count,
                                            const int *in_list, int *info, hipStream_t stream, hint_t *hint_out)
{
    hipError_t e = tilep_variant() == 2 ? launch_gj_tilepb<double>(n, A, X, batch, info, stream, in_count, in_list, hint_out)
                                        : launch_tilepw<double>(n, A, X, batch, info, stream, in_count, in_list, hint_out);
    return e != hipSuccess ? e : debug_note_rejects(in_count, stream);
}

const char *name_gj_tilepw(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tilepw_%s<%d>", f64 ? "f64" : "f32", (n + 15) / 16);
    return buf;
}

}  // namespace matinv
