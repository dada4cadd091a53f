// tilep4_kernels.hip -- fp64 instantiations of the three- / four-wavefront pivoting MFMA tile kernels (tilep4_impl.hpp), general
// 64 < n <= 128.
// Measured against this kernel and not kept (r02 - r04; DESIGN.md, appendix "experiments that lost"): one wavefront per tile column
// with one searching wave and a barrier every 4 columns, the same with one barrier per tile column (16 pivots), the ONE-wavefront
// pivoting kernel on VGPRs + AGPRs for 64 < n <= 96, four wavefronts also at 5 x 5 / 6 x 6 tiles, and -- r04 -- the kernel with fixed
// pivot rows and searched pivot columns that now serves 128 < n (tileq_impl.hpp): at these sizes within 2 - 7 % of this one in fp64,
// 1.7 x slower in fp32.
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
                                            hint_t *hint_out, bool expect_many)
{
    return launch_tilep4_worklist<double>(n, A, X, batch, in_count, in_list, bad_count, bad_list, info, stream, hint_out, expect_many);
}

const char *name_gj_tilep4(bool f64, int n)
{
    static thread_local char buf[48];
    const int nt = (n + 15) / 16;
    snprintf(buf, sizeof buf, "matinv_gj_tilep%d_%s<%d, %s>", nt <= 6 ? 3 : 4, f64 ? "f64" : "f32", nt, (n % 16) == 0 ? "true" : "false");
    return buf;
}

}  // namespace matinv
