// tile_kernels.hip -- placeholder until the MFMA-tile family lands.
#include "common.hpp"
namespace matinv {
template <class T> bool tile_family_supports(int) { return false; }
template <class T>
hipError_t launch_gj_tile(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t) { return hipErrorInvalidValue; }
template bool tile_family_supports<double>(int);
template bool tile_family_supports<float>(int);
template hipError_t launch_gj_tile<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_gj_tile<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);
const char *name_gj_tile(bool, int) { return ""; }
}
