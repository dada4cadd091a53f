// fp32 one-wavefront symmetric sweep beyond 8 x 8 lower tiles (128 < n <= 176): see tile_kernels.inc
#define MATINV_TILE_PART 33
#include "tile_kernels.inc"
