// common.hpp -- shared declarations of the HIP translation units of libmatinv_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/matinv.h"

namespace matinv {

// (batch << 32) | rejected of a natural-order launch, stored by the work-list kernel behind it into pinned host memory
typedef unsigned long long hint_t;

// thread-local error message behind matinv_last_error(); returns `code` (abi.hip)
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int fail_hip(hipError_t e, const char *what);

// Device scratch of the launchers: blocks cached per (device, stream), reused on their own stream only (scratch.hip)
hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t stream);
hipError_t scratch_free(void *p, hipStream_t stream);
void scratch_retire_stream(hipStream_t stream);  // after synchronising a stream that is about to be destroyed
void blocked_gp_release_graphs();                // the cached launch-chain graphs of blocked_gp_kernels.hip (they hold scratch pointers)
void scratch_release_device();                   // after hipDeviceSynchronize: hipFree everything not in use

// Where matrix k of a batch lives: either base + k*stride, or table[k] (the reference's
// "array of device pointers" form, /root/reference/src/helper.cu:103-118).
template <class T>
struct BatchRef {
    T *base;
    size_t stride;
    T *const *table;  // device-resident pointer table, or nullptr
    __device__ __forceinline__ T *at(size_t k) const { return table ? table[k] : base + k * stride; }
    // Same for a wave-uniform k: the table entry is moved to SGPRs at once, so that on the (usual) strided path the result
    // never lives in a VGPR that a pending vector load targets -- otherwise hipcc guards the merge with s_waitcnt vmcnt(0)
    // on BOTH paths, draining every outstanding load / store / LDS-DMA of the wave at each matrix boundary.
    __device__ __forceinline__ T *at_uniform(size_t k) const
    {
        if (table) {
            const unsigned long long p = reinterpret_cast<unsigned long long>(table[k]);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
            // rebuilt from integers: say explicitly that it is a global-memory pointer, or every access through it
            // (and, after the merge, through the strided path too) degrades to flat_load / flat_store
            typedef __attribute__((address_space(1))) T *global_ptr;
            return (T *)(global_ptr)(((unsigned long long)hi << 32) | lo);
        }
        return base + k * stride;
    }
};

template <class T>
__device__ __forceinline__ T max_finite();
template <>
__device__ __forceinline__ double max_finite<double>() { return 1.7976931348623157e308; }
template <>
__device__ __forceinline__ float max_finite<float>() { return 3.402823466e38f; }
template <class T>
__device__ __forceinline__ T nan_of();
template <>
__device__ __forceinline__ double nan_of<double>() { return __longlong_as_double(0x7ff8000000000000LL); }
template <>
__device__ __forceinline__ float nan_of<float>() { return __int_as_float(0x7fc00000); }

// Host-side launchers; each returns hipSuccess / an error, or hipErrorInvalidValue when the
// family cannot serve (n, dtype). All are asynchronous on `stream`.
template <class T>
hipError_t launch_gj_lds(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
// pivoted LDS kernel over a device-side work list (count + indices), used as the fallback of the fast families
template <class T>
hipError_t launch_gj_lds_worklist(int n, BatchRef<const T> A, BatchRef<T> X, const int *work_count, const int *work_list,
                                  int *info, hipStream_t stream);
template <class T>
hipError_t launch_chol_lds_worklist(int n, BatchRef<const T> A, BatchRef<T> X, const int *work_count,
                                    const int *work_list, int *info, hipStream_t stream);
// phases: bit 0 factor, bit 1 triangular inverse, bit 2 L^-T L^-1 (7 = full inverse)
template <class T>
hipError_t launch_chol_lds(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream,
                           int phases);
// Ds == nullptr selects the variance form (needs Es); otherwise the mean form.
template <class T>
hipError_t launch_gp_lds(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                         int *info, hipStream_t stream);
template <class T>
bool lds_family_supports(int n);

template <class T>
hipError_t launch_gj_rowlane(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
// the same kernel as the Cholesky entry point for n <= 16 (lower triangle only, natural positive pivots)
template <class T>
hipError_t launch_spd_rowlane(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
const char *name_spd_rowlane(bool f64, int n);
// fused GP scalars on the same kernel, n <= 16
template <class T>
hipError_t launch_gp_rowlane(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                             int *info, hipStream_t stream);
template <class T>
bool rowlane_family_supports(int n);
// natural-order pass of the Gauss-Jordan entry point for 16 < n <= 32: the ROWLANE design with two rows per lane
// (rowlane2_kernels.hip); rejected matrices go to work_list like those of the natural-order tile kernels
bool rowlane2_supports(int n);
bool rowlane2_natural_use(bool f64, int n);
template <class T>
hipError_t enqueue_gj_rowlane2(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, int *work_count,
                               int *work_list);
const char *name_gj_rowlane2(bool f64, int n);
// r03: the same kernel as the fused mean / variance (positive pivots verified, a^T M^-1 d folded out of the registers)
template <class T>
hipError_t launch_gp_rowlane2(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch, int *info,
                              hipStream_t stream);
bool rowlane2_gp_use(bool f64, int n);  // where it replaces the MFMA tile pipeline kernel (16 < n <= 25; MATINV_ROWLANE2_GP=0: nowhere)
const char *name_gp_rowlane2(bool f64, int n);
template <class T>
hipError_t enqueue_gp_rowlane2(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch, int *info,
                               hipStream_t stream, int *work_count, int *work_list);

// blocked multi-launch Cholesky for the fused GP scalars at large n (blocked_gp_kernels.hip)
template <class T>
hipError_t launch_gp_blocked(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                             int *info, hipStream_t stream);

// blocked SPD inverse for large n on the same panel / update kernels (identity border + symmetric rank-n product)
bool blocked_inverse_supports(int n);
template <class T>
hipError_t launch_chol_blocked(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);

// rounds of resident workgroups in the grids of the MFMA-tile kernels (each workgroup strides over the batch);
// MATINV_TILE_GRID_MULT overrides the default for A/B measurements (tile_kernels.hip)
unsigned tile_grid_rounds();

// blocked Gauss-Jordan with partial pivoting for large general matrices (blocked_gj_kernels.hip)
bool blocked_gj_supports(int n);
int blocked_gj_two_level_min();
size_t blocked_workspace_cap();  // bytes; MATINV_BLOCKED_WS_MB overrides the 16 GiB default
template <class T>
hipError_t launch_gj_blocked(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);

// GLOBAL family (global_kernels.hip): any n <= 1024, working copy in global memory
template <class T>
bool global_family_supports(int n);
template <class T>
hipError_t launch_gj_global(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <class T>
hipError_t launch_chol_global(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <class T>
hipError_t launch_gp_global(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                            int *info, hipStream_t stream);
const char *name_gj_global(bool f64);
const char *name_chol_global(bool f64);

// ROW family (row_kernels.hip): true partial pivoting, one matrix per wavefront, n <= 64
template <class T>
bool row_family_supports(int n);
template <class T>
hipError_t launch_gj_row(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <class T>
hipError_t launch_gj_row_worklist(int n, BatchRef<const T> A, BatchRef<T> X, const int *work_count, const int *work_list,
                                  int *info, hipStream_t stream);
const char *name_gj_row(bool f64, int n);

template <class T>
hipError_t launch_gj_tile(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <class T>
bool tile_family_supports(int n);
// four-wave variant of the tile family, 64 < n <= 128, f64 and f32 (tile4_kernels.hip); the spd entry is the same
// kernel with lower-triangle loads and positivity-checked pivots (the Cholesky contract)
bool tile4_supports(int n);
bool tile4_wide_supports(bool f64, int n);  // SPD sweep and fused pipeline only: 128 < n <= 192 (f64) / 256 (f32)
template <class T>
hipError_t launch_gj_tile4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <class T>
hipError_t launch_spd_tile4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
const char *name_tile4(bool f64, bool spd, int n);
// SPD (symmetric blocked sweep) on the tile layout, f64, n <= 64
template <class T>
hipError_t launch_spd_tile(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <class T>
bool spd_tile_supports(int n);
const char *name_spd_tile(bool f64, int n);

// TILEP family (tilep_kernels.hip): the MFMA tile Gauss-Jordan with true partial pivoting inside the kernel, n <= 64
bool tilep_supports(int n);
template <class T>
hipError_t launch_gj_tilep(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <>
hipError_t launch_gj_tilep<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream);
template <>
hipError_t launch_gj_tilep<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream);
// work-list form: inverts in_list[0 .. *in_count) (device memory); singular ones go on to the ROW kernel through
// (bad_count, bad_list), zeroed by the caller; hint_out (pinned host memory, may be null) receives the list length
template <class T>
hipError_t launch_gj_tilep_worklist(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, const int *in_count, const int *in_list,
                                    int *bad_count, int *bad_list, int *info, hipStream_t stream, hint_t *hint_out, bool expect_many = false);
template <>
hipError_t launch_gj_tilep_worklist<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, const int *in_count,
                                            const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream, hint_t *hint_out,
                                            bool expect_many);
template <>
hipError_t launch_gj_tilep_worklist<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, const int *in_count,
                                           const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream, hint_t *hint_out,
                                           bool expect_many);
const char *name_gj_tilep(bool f64, int n);
// four wavefronts per matrix, 64 < n <= 128 (tilep4_kernels.hip)
template <class T>
hipError_t launch_gj_tilep4(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream);
template <>
hipError_t launch_gj_tilep4<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream);
template <>
hipError_t launch_gj_tilep4<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream);
template <class T>
hipError_t launch_gj_tilep4_worklist(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, const int *in_count, const int *in_list,
                                     int *bad_count, int *bad_list, int *info, hipStream_t stream, hint_t *hint_out, bool expect_many = false);
template <>
hipError_t launch_gj_tilep4_worklist<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, const int *in_count,
                                             const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                             hint_t *hint_out, bool expect_many);
template <>
hipError_t launch_gj_tilep4_worklist<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, const int *in_count,
                                            const int *in_list, int *bad_count, int *bad_list, int *info, hipStream_t stream,
                                            hint_t *hint_out, bool expect_many);
const char *name_gj_tilep4(bool f64, int n);
// r04: fixed pivot rows, searched pivot columns -- no run-time register index (tileq_kernels.hip): one wavefront per tile column,
// general 128 < n <= 192 (f64) / 256 (f32). in_count / in_list: work-list form (nullptr: the whole batch); bad_count / bad_list: unused
// at these sizes (the kernel finishes a singular matrix itself), kept for the n <= 128 instantiations of tools/tileq_stamps.hip
template <class T>
hipError_t launch_gj_tileq(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream, const int *in_count,
                           const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list);
template <>
hipError_t launch_gj_tileq<double>(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream,
                                   const int *in_count, const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list);
template <>
hipError_t launch_gj_tileq<float>(int n, BatchRef<const float> A, BatchRef<float> X, size_t batch, int *info, hipStream_t stream,
                                  const int *in_count, const int *in_list, hint_t *hint_out, int *bad_count, int *bad_list);
const char *name_gj_tileq(bool f64, int n);
// general 128 < n <= 192 (f64) / 256 (f32): TILEQ (tileq_kernels.hip)
bool tileq_supports(bool f64, int n);
// Adaptive choice between the natural-order (verified) tile kernel and the pivoting one (tile_kernels.hip): see gj_tile_policy
struct TileStats {
    unsigned long long natural_launches, pivot_launches, last_rejected, last_batch;
};
TileStats tile_stats();
bool tile_policy_use_pivot(bool f64, int nt);
hint_t *tile_policy_record(bool f64, int nt, size_t batch);
int gj_policy();                // a matinv_gj_policy (include/matinv.h)
int set_gj_policy(int policy);  // returns the previous one

const char *name_gj_rowlane(bool f64, int n);
const char *name_gj_tile(bool f64, int n);
template <class T>
hipError_t launch_gp_lds_worklist(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out,
                                  const int *work_count, const int *work_list, int *info, hipStream_t stream);
// fused GP scalars on the MFMA tile layout, one wavefront per item: f64 n <= 80, f32 n <= 96 (gp_tile_kernels.hip)
bool gp_tile_supports(bool f64, int n);
template <class T>
hipError_t launch_gp_tile(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                          int *info, hipStream_t stream);
const char *name_gp_tile(bool f64, int n);
// fused GP scalars on the SPD sweep of the Cholesky entry point (inverse in registers, folded, never stored): the sizes the
// bordered form above no longer holds in one wavefront -- f64 80 < n <= 96, f32 96 < n <= 112 (tile_kernels.inc)
// test hook (scratch.hip): MATINV_DEBUG_REJECTS=1 -> launchers with a work list add its final count to a running total
hipError_t debug_note_rejects(const int *work_count, hipStream_t stream);
long long debug_rejects(bool reset);
bool gp_spd_tile_supports(bool f64, int n);
int spd_onewave_max(bool f64);  // largest n of the one-wavefront symmetric sweep (fp64 112, fp32 160)
// fp32 9 x 9 / 10 x 10 lower tiles on one wavefront (spd_wide_f32_kernels.hip, gp_spd_wide_f32_kernels.hip): the kernel launch only, ws = [count, list...]
hipError_t enqueue_spd_tile_wide_f32(int n, BatchRef<const float> A, BatchRef<float> X, unsigned grid, unsigned b, int *info, int *ws,
                                     hipStream_t stream);
// fp64 7 x 7 lower tiles on one wavefront (spd_wide_f64_kernels.hip, compiled with VGPR-form MFMAs)
hipError_t enqueue_spd_tile_wide_f64(int n, BatchRef<const double> A, BatchRef<double> X, unsigned grid, unsigned b, int *info, int *ws,
                                     hipStream_t stream);
hipError_t enqueue_gp_spd_tile_wide_f64(int n, const double *As, const double *Bs, const double *Cs, const double *Ds, const double *Es,
                                        double *out, unsigned grid, unsigned b, int *info, int *ws, hipStream_t stream);
hipError_t enqueue_gp_spd_tile_wide_f32(int n, const float *As, const float *Bs, const float *Cs, const float *Ds, const float *Es, float *out,
                                        unsigned grid, unsigned b, int *info, int *ws, hipStream_t stream);
// two (176 < n <= 192: three) wavefronts per matrix, lower tiles only, fp64 112 < n <= 192 (r04: beyond 128 too): Cholesky entry point and fused pipeline (spd_tile2_kernels.hip);
bool spd_tile2_supports(bool f64, int n);
hipError_t launch_spd_tile2(int n, BatchRef<const double> A, BatchRef<double> X, size_t batch, int *info, hipStream_t stream);
hipError_t launch_gp_spd_tile2(int n, const double *As, const double *Bs, const double *Cs, const double *Ds, const double *Es, double *out,
                               size_t batch, int *info, hipStream_t stream);
const char *name_spd_tile2(bool gp, int n);
template <class T>
hipError_t launch_gp_spd_tile(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                              int *info, hipStream_t stream);
template <>
hipError_t launch_gp_spd_tile<double>(int n, const double *As, const double *Bs, const double *Cs, const double *Ds, const double *Es,
                                      double *out, size_t batch, int *info, hipStream_t stream);
template <>
hipError_t launch_gp_spd_tile<float>(int n, const float *As, const float *Bs, const float *Cs, const float *Ds, const float *Es,
                                     float *out, size_t batch, int *info, hipStream_t stream);
const char *name_gj_lds(bool f64);
const char *name_chol_lds(bool f64);
const char *name_gp_lds(bool f64);

}  // namespace matinv
