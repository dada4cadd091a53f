// tilen_kernels.hip -- fp64 instantiations of the second-generation natural-order MFMA tile kernels (tilen_impl.hpp).
#include "tilen_impl.hpp"

namespace matinv {

template <>
hipError_t enqueue_gj_tilen<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream,
                                    int *work_count, int *work_list)
{
    return enqueue_tilen<double>(n, A, X, batch, info, stream, work_count, work_list);
}

// Which natural-order kernel serves n <= 64: the r01 kernel of tile_kernels.inc unless MATINV_TILE_NATURAL=new. Measured A/B
// on one box (tools/ab_natural.sh, 100 k x 64^2 f64, median of 21 launches): r01 kernel 1.525 / 1.535 ms, this one 1.519 /
// 1.548 ms; 32^2: 1.383 vs 1.435 ms; f32 64^2: 1.503 vs 1.534 ms. 20 % fewer VALU instructions and three waves per SIMD
// instead of two buy nothing: at 4.3 TB/s both kernels sit at 0.95 of what a plain device copy reaches on the box
// (4.5 TB/s) -- the headline kernel is bound by the achievable HBM bandwidth of its access pattern, not by issue.
bool tile_natural_old()
{
    static const bool v = []() {
        const char *s = getenv("MATINV_TILE_NATURAL");
        return !(s && !strcmp(s, "new"));
    }();
    return v;
}

const char *name_gj_tilen(bool f64, int n)
{
    static thread_local char buf[48];
    snprintf(buf, sizeof buf, "matinv_gj_tilen_%s<%d, %s>", f64 ? "f64" : "f32", (n + 15) / 16, (n % 16) == 0 ? "true" : "false");
    return buf;
}

}  // namespace matinv
