// fp32 half of the one-wavefront MFMA tile family: see tile_kernels.inc
#define MATINV_TILE_PART 32
#include "tile_kernels.inc"
