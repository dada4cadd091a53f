// tileq_kernels.hip -- fp64 instantiations of the pivoting MFMA tile kernel with fixed pivot rows and searched pivot columns
// (tileq_impl.hpp): one wavefront per tile column, general 128 < n <= 192 (fp64) / 256 (fp32).
#include "tileq_impl.hpp"

namespace matinv {

bool tileq_supports(bool f64, int n) { return n > 128 && n <= tileq_limit(f64); }

template <>
hipError_t launch_gj_tileq<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream,
                                   const int *in_count, const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list)
{
    hipError_t e = launch_tileq<double>(n, A, X, batch, info, stream, in_count, in_list, hint_out, bad_count, bad_list);
    return (e != hipSuccess || !in_count) ? e : debug_note_rejects(in_count, stream);
}

const char *name_gj_tileq(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tileqw_%s<%d, false>", f64 ? "f64" : "f32", (n + 15) / 16);
    return buf;
}

}  // namespace matinv
