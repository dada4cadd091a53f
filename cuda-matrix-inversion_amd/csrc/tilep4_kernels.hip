// tilep4_kernels.hip -- fp64 instantiations of the four-wavefront pivoting MFMA tile kernels (tilep4_impl.hpp).
#include "tilep4_impl.hpp"

namespace matinv {

// MATINV_TILEP_WAVES (A/B switch): "col" = 64 < n <= 128 on the one-wavefront-per-tile-column kernel with a barrier every 4
// columns (tilepw_impl.hpp; measured slower than the four-wave kernel of this file at every size: 3.6e6 against 4.1e6 inv/s at
// 128^2 f64, 8.2e6 against 1.28e7 at 72^2), "blk" = every 64 < n <= 192 / 256 on the kernel with one barrier per tile column
// (tilepb_impl.hpp).
// (r03, measured and removed: the ONE-wavefront pivoting kernel of tilep_impl.hpp for 64 < n <= 96 with two rows per lane and the
// 25 / 36 accumulator tiles in AGPRs -- one wave per SIMD, as launch_spd_tile does for the symmetric sweep. The pivot-row gather
// is inline asm on the accumulator registers with "v" constraints, so hipcc shuttles whole tile rows between AGPRs and VGPRs
// around every block: 156 B / 1 KB of scratch per lane at 5 x 5 / 6 x 6 tiles, 1.23e7 / 9.8e6 / 3.3e6 inv/s at 72^2 / 80^2 / 96^2
// against 1.32e7 / 1.18e7 / 8.4e6 here.)
int tilep_variant()
{
    static const int v = []() {
        const char *s = getenv("MATINV_TILEP_WAVES");
        return !s ? 0 : (s[0] == 'c' ? 1 : (s[0] == 'b' ? 2 : 0));
    }();
    return v;
}

template <>
hipError_t launch_gj_tilep4<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream)
{
    if (tilep_variant() == 2) return launch_gj_tilepb<double>(n, A, X, batch, info, stream);
    if (tilep_variant() == 1) return launch_gj_tilepw<double>(n, A, X, batch, info, stream);
    return launch_tilep4<double>(n, A, X, batch, info, stream);
}

template <>
hipError_t launch_gj_tilep4_worklist<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, const int *in_count,
                                            const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                            hint_t *hint_out)
{
    if (tilep_variant() == 2) return launch_gj_tilepb<double>(n, A, X, batch, info, stream, in_count, in_list, hint_out);
    if (tilep_variant() == 1) return launch_gj_tilepw_worklist<double>(n, A, X, batch, in_count, in_list, info, stream, hint_out);
    return launch_tilep4_worklist<double>(n, A, X, batch, in_count, in_list, bad_count, bad_list, info, stream, hint_out);
}

const char *name_gj_tilep4(bool f64, int n)
{
    static thread_local char buf[48];
    const int nt = (n + 15) / 16;
    snprintf(buf, sizeof buf, "matinv_gj_tilep%d_%s<%d, %s>", (nt <= 6 && tilep_three_waves()) ? 3 : 4, f64 ? "f64" : "f32", nt,
             (n % 16) == 0 ? "true" : "false");
    return buf;
}

}  // namespace matinv
