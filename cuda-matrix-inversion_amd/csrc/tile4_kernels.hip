// tile4_kernels.hip -- fp64 instantiations of the several-wavefronts-per-matrix MFMA tile kernels (tile4_impl.hpp) and the
// family's non-template helpers.
#include "tile4_impl.hpp"

namespace matinv {

bool tile4_supports(int n) { return n > 64 && n <= 128; }
// the SPD sweep / fused pipeline with one wavefront per tile column: 128 < n <= 192 (f64) / 256 (f32)
bool tile4_wide_supports(bool f64, int n) { return n > 128 && n <= t4_wide_limit(f64); }

template hipError_t launch_gj_tile4<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_spd_tile4<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);

const char *name_tile4(bool f64, bool spd, int n)
{
    // as rocprofv3 prints the instantiation (default template arguments spelled out)
    static thread_local char buf[64];
    snprintf(buf, sizeof buf, "matinv_gj_tile4_%s<%d, %s, %d, %s>", f64 ? "f64" : "f32", (n + 15) / 16,
             ((n % 16) == 0 && n <= 128) ? "true" : "false",
             t4_waves(f64, (n + 15) / 16), spd ? "true" : "false");
    return buf;
}

}  // namespace matinv
