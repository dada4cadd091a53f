// gp_tile_kernels.hip -- fp64 instantiation of the one-wavefront fused pipeline kernels (gp_tile_impl.hpp) + the helpers
#include "gp_tile_impl.hpp"

namespace matinv {

// one wavefront holds the bordered lower triangle of up to 7 x 7 tiles (n <= 96; in fp64 the last size spills 292 B per lane
// and still beats the several-wavefront kernel: 2.0e7 vs 1.8e7 items/s at 96 x 96)
bool gp_tile_supports(bool, int n) { return n >= 1 && n <= 96; }

template hipError_t launch_gp_tile<double>(int, const double *, const double *, const double *, const double *,
                                           const double *, double *, size_t, int *, hipStream_t);

const char *name_gp_tile(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gp_tile_%s<%d, %s>", f64 ? "f64" : "f32", (n + 15) / 16, (n % 16) == 0 ? "true" : "false");
    return buf;
}

}  // namespace matinv
