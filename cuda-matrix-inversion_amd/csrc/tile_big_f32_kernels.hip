// tile_big_f32_kernels.hip -- fp32 one-wavefront symmetric sweeps of 7 x 7 / 8 x 8 lower tiles (Cholesky entry point, 96 < n <= 128)
// and the fused mean / variance on them (tile_kernels.inc); a translation unit of their own so that the fp32 half builds in parallel.
#define MATINV_TILE_PART 35
#include "tile_kernels.inc"
