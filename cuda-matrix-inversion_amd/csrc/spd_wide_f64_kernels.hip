// fp64 one-wavefront symmetric sweep of 7 x 7 lower tiles (96 < n <= 112), inverse and fused mean / variance: see tile_kernels.inc
#define MATINV_TILE_PART 65
#include "tile_kernels.inc"
