// fp64 half of the one-wavefront MFMA tile family + its non-template helpers: see tile_kernels.inc
#define MATINV_TILE_PART 64
#include "tile_kernels.inc"
