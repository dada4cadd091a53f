// tilep_kernels.hip -- fp64 instantiations of the pivoting MFMA tile kernels (tilep_impl.hpp) and the family's helpers.
#include "tilep_impl.hpp"

namespace matinv {

bool tilep_supports(int n) { return n >= 1 && n <= 128; }

template <>
hipError_t launch_gj_tilep<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    return launch_tilep<double>(n, A, X, batch, info, stream);
}

const char *name_gj_tilep(bool f64, int n)
{
    if (n > 64) return name_gj_tilep4(f64, n);
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tilep_%s<%d, %s>", f64 ? "f64" : "f32", (n + 15) / 16, (n % 16) == 0 ? "true" : "false");
    return buf;
}

template <>
hipError_t launch_gj_tilep_worklist<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, const int *in_count,
                                           const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream, hint_t *hint_out, bool expect_many)
{
    return launch_tilep_worklist<double>(n, A, X, batch, in_count, in_list, bad_count, bad_list, info, stream, hint_out, expect_many);
}

}  // namespace matinv
