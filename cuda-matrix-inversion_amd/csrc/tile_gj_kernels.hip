// tile_gj_kernels.hip -- fp64 natural-order Gauss-Jordan MFMA tile kernels, n <= 64, with their screening pass (tile_kernels.inc).
#define MATINV_TILE_PART 66
#include "tile_kernels.inc"
