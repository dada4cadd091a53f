// fp32 fused mean / variance on the one-wavefront symmetric sweep beyond 8 x 8 lower tiles (128 < n <= 160): see tile_kernels.inc
#define MATINV_TILE_PART 34
#include "tile_kernels.inc"
