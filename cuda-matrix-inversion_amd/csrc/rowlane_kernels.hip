// rowlane_kernels.hip -- placeholder until the register-resident small-n family lands.
#include "common.hpp"
namespace matinv {
template <class T> bool rowlane_family_supports(int) { return false; }
template <class T>
hipError_t launch_gj_rowlane(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t) { return hipErrorInvalidValue; }
template bool rowlane_family_supports<double>(int);
template bool rowlane_family_supports<float>(int);
template hipError_t launch_gj_rowlane<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_gj_rowlane<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);
const char *name_gj_rowlane(bool, int) { return ""; }
}
