// tile_gj_f32_kernels.hip -- fp32 natural-order Gauss-Jordan MFMA tile kernels, n <= 64, with their screening pass (tile_kernels.inc).
#define MATINV_TILE_PART 36
#include "tile_kernels.inc"
