// abi.hip -- the C ABI of libmatinv_hip.so: include/matinv.h (native, device-resident, stream-aware)
// and the 17 reference-named entry points of include/inverse_gpu.h in fp64 and (suffix _f32) fp32.
//
// No CPU fallback lives here: every entry point ends in a HIP kernel launch or fails loudly.
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/matinv.h"
#include "common.hpp"

using namespace matinv;

namespace {
thread_local char g_err[512] = "";
}

namespace matinv {
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    return fail(MATINV_ERR_HIP, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}
}  // namespace matinv

namespace {

// The library is built for gfx950 only; refuse anything else instead of faulting in the launch.
int check_device()
{
    static thread_local int checked_dev = -1;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(MATINV_ERR_NO_DEVICE, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev == checked_dev) return MATINV_OK;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MATINV_ERR_NO_DEVICE, "device %d is %s; libmatinv_hip is built for gfx950 (MI355X) only", dev,
                    prop.gcnArchName);
    checked_dev = dev;
    return MATINV_OK;
}

// Staging buffers of the host-pointer entry points: the library's scratch cache (scratch.hip), so that the reference-style
// "allocate, copy, run, copy, free inside every call" (batched_invert.cu:120-176) stops paying hipMalloc / hipFree (about 0.5 ms
// per call) after the first call.
hipError_t staging_alloc(void **p, size_t bytes) { return scratch_alloc(p, bytes, nullptr); }
void staging_free(void *p)
{
    if (p) (void)scratch_free(p, nullptr);
}

template <class T>
int select_auto(int algo, int n)
{
    if (algo == MATINV_ALGO_GAUSS_JORDAN) {
        if (rowlane_family_supports<T>(n)) return MATINV_KERNEL_ROWLANE;
        if (tile_family_supports<T>(n)) return MATINV_KERNEL_TILE;
        if (tileq_supports(sizeof(T) == 8, n)) return MATINV_KERNEL_TILEP;  // one wavefront per tile column, pivoting
        if (blocked_gj_supports(n)) return MATINV_KERNEL_BLOCKED;  // beyond n = 128 it beats the LDS kernel at every size measured
    } else if (rowlane_family_supports<T>(n)) {
        return MATINV_KERNEL_ROWLANE;  // several SPD matrices per wavefront
    } else if (spd_tile_supports<T>(n)) {
        return MATINV_KERNEL_TILE;
    } else if (blocked_inverse_supports(n)) {
        return MATINV_KERNEL_BLOCKED;  // SPD inverse beyond the four-wave kernel
    }
    if (lds_family_supports<T>(n)) return MATINV_KERNEL_LDS;
    if (global_family_supports<T>(n)) return MATINV_KERNEL_GLOBAL;
    return MATINV_ERR_UNSUPPORTED;
}

template <class T>
int inverse_dispatch(int algo, int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *dInfo, hipStream_t stream,
                     int kernel, int chol_phases = 7)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (algo != MATINV_ALGO_GAUSS_JORDAN && algo != MATINV_ALGO_CHOLESKY)
        return fail(MATINV_ERR_ARG, "unknown algorithm %d", algo);
    if (batch == 0) return MATINV_OK;
    if (batch > 0x7fffffffu) return fail(MATINV_ERR_ARG, "batch %zu exceeds the grid limit; split the call", batch);
    int rc = check_device();
    if (rc) return rc;
    if (kernel == MATINV_KERNEL_AUTO) {
        // the Cholesky sub-phase entry points (factor only, ...) exist in the LDS family only
        kernel = (algo == MATINV_ALGO_CHOLESKY && chol_phases != 7 && lds_family_supports<T>(n)) ? (int)MATINV_KERNEL_LDS
                                                                                                 : select_auto<T>(algo, n);
        if (kernel < 0)
            return fail(MATINV_ERR_UNSUPPORTED, "n=%d exceeds every kernel family built in (limit 1024)", n);
    }
    hipError_t e = hipErrorInvalidValue;
    switch (kernel) {
    case MATINV_KERNEL_LDS:
        if (!lds_family_supports<T>(n)) return fail(MATINV_ERR_UNSUPPORTED, "LDS family: n=%d does not fit 160 KiB", n);
        e = (algo == MATINV_ALGO_GAUSS_JORDAN) ? launch_gj_lds<T>(n, A, X, batch, dInfo, stream)
                                               : launch_chol_lds<T>(n, A, X, batch, dInfo, stream, chol_phases);
        break;
    case MATINV_KERNEL_ROWLANE:
        if (!rowlane_family_supports<T>(n) || (algo == MATINV_ALGO_CHOLESKY && chol_phases != 7))
            return fail(MATINV_ERR_UNSUPPORTED, "rowlane family serves full inversions with n <= 16 only (n=%d)", n);
        e = (algo == MATINV_ALGO_CHOLESKY) ? launch_spd_rowlane<T>(n, A, X, batch, dInfo, stream)
                                           : launch_gj_rowlane<T>(n, A, X, batch, dInfo, stream);
        break;
    case MATINV_KERNEL_TILE:
        if (algo == MATINV_ALGO_CHOLESKY) {
            if (!spd_tile_supports<T>(n) || chol_phases != 7)
                return fail(MATINV_ERR_UNSUPPORTED, "tile family serves the full SPD inverse with n <= 192 (f64) / 256 (f32) only (n=%d)", n);
            e = launch_spd_tile<T>(n, A, X, batch, dInfo, stream);
            break;
        }
        if (!tile_family_supports<T>(n))
            return fail(MATINV_ERR_UNSUPPORTED, "tile family serves Gauss-Jordan with n <= 192 (f64) / 256 (f32) only (n=%d)", n);
        e = launch_gj_tile<T>(n, A, X, batch, dInfo, stream);
        break;
    case MATINV_KERNEL_GLOBAL:
        if (!global_family_supports<T>(n) || chol_phases != 7)
            return fail(MATINV_ERR_UNSUPPORTED, "global family serves full inversions with n <= 1024 only (n=%d)", n);
        e = (algo == MATINV_ALGO_GAUSS_JORDAN) ? launch_gj_global<T>(n, A, X, batch, dInfo, stream)
                                               : launch_chol_global<T>(n, A, X, batch, dInfo, stream);
        break;
    case MATINV_KERNEL_BLOCKED:
        if (algo == MATINV_ALGO_GAUSS_JORDAN) {
            if (!blocked_gj_supports(n)) return fail(MATINV_ERR_UNSUPPORTED, "blocked family: n=%d exceeds the limit 1024", n);
            e = launch_gj_blocked<T>(n, A, X, batch, dInfo, stream);
            break;
        }
        if (!blocked_inverse_supports(n) || chol_phases != 7)
            return fail(MATINV_ERR_UNSUPPORTED, "blocked family serves the full SPD inverse with n <= 1024 only (n=%d)", n);
        e = launch_chol_blocked<T>(n, A, X, batch, dInfo, stream);
        break;
    case MATINV_KERNEL_TILEP:
        if (algo != MATINV_ALGO_GAUSS_JORDAN || !(tilep_supports(n) || tileq_supports(sizeof(T) == 8, n)))
            return fail(MATINV_ERR_UNSUPPORTED, "pivoting tile family serves Gauss-Jordan with n <= 192 (f64) / 256 (f32) only (n=%d)", n);
        e = n > 128 ? launch_gj_tileq<T>(n, A, X, batch, dInfo, stream, nullptr, nullptr, nullptr, nullptr, nullptr)
                    : (n > 64 ? launch_gj_tilep4<T>(n, A, X, batch, dInfo, stream) : launch_gj_tilep<T>(n, A, X, batch, dInfo, stream));
        break;
    case MATINV_KERNEL_ROW:
        if (algo != MATINV_ALGO_GAUSS_JORDAN || !row_family_supports<T>(n))
            return fail(MATINV_ERR_UNSUPPORTED, "row family serves Gauss-Jordan with n <= 64 only (n=%d)", n);
        e = launch_gj_row<T>(n, A, X, batch, dInfo, stream);
        break;
    default:
        return fail(MATINV_ERR_ARG, "unknown kernel family %d", kernel);
    }
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return MATINV_OK;
}

template <class T>
int inverse_strided(int algo, int n, const void *dA, size_t strideA, void *dAinv, size_t strideInv, size_t batch,
                    int *dInfo, void *stream, int kernel)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (batch && (!dA || !dAinv)) return fail(MATINV_ERR_ARG, "null device pointer");
    if (batch > 1 && (strideA < (size_t)n * n || strideInv < (size_t)n * n))
        return fail(MATINV_ERR_ARG, "stride smaller than n*n");
    BatchRef<const T> A{static_cast<const T *>(dA), strideA, nullptr};
    BatchRef<T> X{static_cast<T *>(dAinv), strideInv, nullptr};
    return inverse_dispatch<T>(algo, n, A, X, batch, dInfo, static_cast<hipStream_t>(stream), kernel);
}

double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// MATINV_DETAILED_LOGGING=1 reproduces the reference's log=1 build (Makefile:115-117 there): one
// `name,batch,n,ms,ns\r\n` line per phase (include/timer.h:8-9), same key names as the reference's
// TIMER_LOG calls (e.g. src/gauss/batched_invert.cu:114-118,169-171).
bool detailed_logging()
{
    static int v = -1;
    if (v < 0) {
        const char *s = getenv("MATINV_DETAILED_LOGGING");
        v = (s && *s && *s != '0') ? 1 : 0;
    }
    return v == 1;
}

void timer_log(const char *prefix, const char *phase, size_t batch, int n, double ms)
{
    printf("%s_%s,%zu,%d,%.4f,%lu\r\n", prefix, phase, batch, n, ms, (unsigned long)(ms * 1e6));
}

// Large host batches: the caller's two buffers are page-locked in place (hipHostRegister) and the batch is cut into chunks
// that rotate over a few streams -- chunk i+1 uploads while chunk i is inverted and chunk i-1 downloads (PCIe is full
// duplex; the simple path below does H2D, kernel, D2H strictly one after the other, through the runtime's pageable
// staging). Device memory is chunk-sized, so a host batch larger than HBM also works.
// Returns MATINV_ERR_UNSUPPORTED (without touching the output) when the buffers cannot be registered; the caller then
// takes the simple path.
template <class T>
int inverse_host_pipelined(int algo, int n, const T *hA, T *hAinv, size_t batch, int *info, int kernel, double &ms_total,
                           bool registered = false)
{
    constexpr int NS = 3;  // streams / chunk slots in flight
    const size_t mat = (size_t)n * n, bytes_mat = mat * sizeof(T);
    size_t chunk = (size_t)(48u << 20) / bytes_mat;  // ~48 MiB per chunk
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const double t0 = now_ms();
    if (!registered) {
        if (hipHostRegister(const_cast<T *>(hA), batch * bytes_mat, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            return MATINV_ERR_UNSUPPORTED;
        }
        if (hipHostRegister(hAinv, batch * bytes_mat, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostUnregister(const_cast<T *>(hA));
            return MATINV_ERR_UNSUPPORTED;
        }
    }
    hipStream_t st[NS] = {};
    T *dA[NS] = {}, *dX[NS] = {};
    int *dI[NS] = {};
    hipError_t e = hipSuccess;
    int rc = MATINV_OK;
    for (int s = 0; s < NS && e == hipSuccess; ++s) {
        e = hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        if (e == hipSuccess) e = scratch_alloc(reinterpret_cast<void **>(&dA[s]), chunk * bytes_mat, st[s]);
        if (e == hipSuccess) e = scratch_alloc(reinterpret_cast<void **>(&dX[s]), chunk * bytes_mat, st[s]);
        if (e == hipSuccess && info) e = scratch_alloc(reinterpret_cast<void **>(&dI[s]), chunk * sizeof(int), st[s]);
    }
    size_t slot = 0;
    for (size_t off = 0; off < batch && e == hipSuccess && rc == MATINV_OK; off += chunk, ++slot) {
        const int s = (int)(slot % NS);
        const size_t cnt = (batch - off < chunk) ? batch - off : chunk;
        // the slot's previous chunk has fully left the device once its stream is idle (same stream: in order anyway)
        e = hipMemcpyAsync(dA[s], hA + off * mat, cnt * bytes_mat, hipMemcpyHostToDevice, st[s]);
        if (e != hipSuccess) break;
        rc = inverse_strided<T>(algo, n, dA[s], mat, dX[s], mat, cnt, info ? dI[s] : nullptr, st[s], kernel);
        if (rc != MATINV_OK) break;
        e = hipMemcpyAsync(hAinv + off * mat, dX[s], cnt * bytes_mat, hipMemcpyDeviceToHost, st[s]);
        if (e == hipSuccess && info)
            e = hipMemcpyAsync(info + off, dI[s], cnt * sizeof(int), hipMemcpyDeviceToHost, st[s]);
    }
    for (int s = 0; s < NS; ++s) {
        if (st[s]) {
            hipError_t es = hipStreamSynchronize(st[s]);
            if (e == hipSuccess) e = es;
        }
        if (dA[s]) (void)scratch_free(dA[s], st[s]);
        if (dX[s]) (void)scratch_free(dX[s], st[s]);
        if (dI[s]) (void)scratch_free(dI[s], st[s]);
        if (st[s]) {
            scratch_retire_stream(st[s]);  // synchronised above: its blocks may serve any stream of this device from now on
            (void)hipStreamDestroy(st[s]);
        }
    }
    if (!registered) {
        (void)hipHostUnregister(const_cast<T *>(hA));
        (void)hipHostUnregister(hAinv);
    }
    ms_total = now_ms() - t0;
    if (rc != MATINV_OK) return rc;
    if (e != hipSuccess) return fail_hip(e, "pipelined host<->device transfer");
    return MATINV_OK;
}

template <class T>
int inverse_host_multi(int algo, int n, const void *hA, void *hAinv, size_t batch, int *info, int nshards, int kernel);

// MATINV_DEVICES=N (> 1): the host-pointer entry points shard over N devices (matinv_inverse_batched_host_multi)
int env_devices()
{
    static const int v = []() {
        const char *s = getenv("MATINV_DEVICES");
        return s && *s ? atoi(s) : 0;
    }();
    return v;
}

// host_mode (set by the multi-device driver for its per-shard calls): HOST_REGISTERED = both host buffers are page-locked for every
// device already; HOST_STAGED_ONLY = its up-front page-locking failed -- go straight to the staged copies (a per-shard
// hipHostRegister would collide with the neighbours' on shared pages: ADVICE r03)
enum { HOST_DEFAULT = 0, HOST_REGISTERED = 1, HOST_STAGED_ONLY = 2 };
template <class T>
int inverse_host(int algo, int n, const void *hA, void *hAinv, size_t batch, int *info, const char *log_prefix,
                 int kernel = MATINV_KERNEL_AUTO, bool single_device = false, int host_mode = HOST_DEFAULT)
{
    const bool registered = host_mode == HOST_REGISTERED;
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (batch == 0) return MATINV_OK;
    if (!hA || !hAinv) return fail(MATINV_ERR_ARG, "null host pointer");
    if (!single_device && env_devices() > 1) {
        const double t0 = now_ms();
        const int mrc = inverse_host_multi<T>(algo, n, hA, hAinv, batch, info, env_devices(), kernel);
        // MATINV_DEVICES > 1 under MATINV_DETAILED_LOGGING: one line for the whole sharded call (the shards overlap: no phase split)
        if (mrc == MATINV_OK && detailed_logging() && log_prefix) timer_log(log_prefix, "multi_total", batch, n, now_ms() - t0);
        return mrc;
    }
    int rc = check_device();
    if (rc) return rc;
    const size_t elems = (size_t)n * n * batch;  // size_t: 1M x 64 x 64 overflows the reference's int index
    {
        // pipelined path for batches worth it (>= 128 MiB per direction); MATINV_HOST_PIPELINE=0/1 forces it off/on
        static const int mode = []() {
            const char *s = getenv("MATINV_HOST_PIPELINE");
            return s && *s ? atoi(s) : -1;
        }();
        const bool big = elems * sizeof(T) >= ((size_t)128 << 20);
        if (hA != hAinv && host_mode != HOST_STAGED_ONLY && (registered || mode == 1 || (mode < 0 && big))) {
            double ms = 0;
            const int prc = inverse_host_pipelined<T>(algo, n, static_cast<const T *>(hA), static_cast<T *>(hAinv), batch, info,
                                                      kernel, ms, registered);
            if (prc != MATINV_ERR_UNSUPPORTED) {
                if (prc == MATINV_OK && detailed_logging() && log_prefix) timer_log(log_prefix, "pipelined_total", batch, n, ms);
                return prc;
            }
        }
    }
    T *dA = nullptr, *dX = nullptr;
    int *dInfo = nullptr;
    hipError_t e;
    const bool log = detailed_logging() && log_prefix;
    if ((e = staging_alloc(reinterpret_cast<void **>(&dA), elems * sizeof(T))) != hipSuccess) return fail_hip(e, "hipMalloc(As)");
    if ((e = staging_alloc(reinterpret_cast<void **>(&dX), elems * sizeof(T))) != hipSuccess) {
        staging_free(dA);
        return fail_hip(e, "hipMalloc(aInvs)");
    }
    if (info && (e = staging_alloc(reinterpret_cast<void **>(&dInfo), batch * sizeof(int))) != hipSuccess) {
        staging_free(dA);
        staging_free(dX);
        return fail_hip(e, "hipMalloc(info)");
    }
    double t0 = now_ms();
    e = hipMemcpy(dA, hA, elems * sizeof(T), hipMemcpyHostToDevice);
    double t1 = now_ms();
    if (e == hipSuccess) {
        rc = inverse_strided<T>(algo, n, dA, (size_t)n * n, dX, (size_t)n * n, batch, dInfo, nullptr, kernel);
        if (rc == MATINV_OK && log) e = hipDeviceSynchronize();
    }
    double t2 = now_ms();
    if (e == hipSuccess && rc == MATINV_OK) e = hipMemcpy(hAinv, dX, elems * sizeof(T), hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc == MATINV_OK && info)
        e = hipMemcpy(info, dInfo, batch * sizeof(int), hipMemcpyDeviceToHost);
    double t3 = now_ms();
    staging_free(dA);
    staging_free(dX);
    staging_free(dInfo);
    if (rc != MATINV_OK) return rc;
    if (e != hipSuccess) return fail_hip(e, "host<->device copy");
    if (log) {
        timer_log(log_prefix, "mem_htod", batch, n, t1 - t0);
        timer_log(log_prefix, "ker", batch, n, t2 - t1);
        timer_log(log_prefix, "mem_dtoh", batch, n, t3 - t2);
    }
    return MATINV_OK;
}

// contiguous block partition of [0, batch): ceil(batch / nshards) per shard, rounded up to the per-wavefront packing factor of the
// small-n kernels (8 matrices per wave for n <= 8, 4 for n <= 16) so that no wavefront straddles two shards; trailing shards may be
// short or empty (shard.partition in Python)
void shard_range(size_t batch, int nshards, int n, int g, size_t &lo, size_t &hi)
{
    const size_t mult = n <= 8 ? 8 : (n <= 16 ? 4 : 1);
    size_t per = (batch + (size_t)nshards - 1) / (size_t)nshards;
    per = (per + mult - 1) / mult * mult;
    lo = std::min(batch, (size_t)g * per);
    hi = std::min(batch, lo + per);
}

int gfx950_devices(std::vector<int> &devs)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return fail(MATINV_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    for (int d = 0; d < count; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) devs.push_back(d);
    }
    if (devs.empty()) return fail(MATINV_ERR_NO_DEVICE, "no gfx950 (MI355X) device visible");
    return MATINV_OK;
}

// The batch over several devices of this process: contiguous blocks (the partition of shard.py / SURVEY 8e: ceil(batch /
// nshards) rounded up to the per-wavefront packing factor), one host thread per shard with hipSetDevice, each running the
// single-device host path on its slice -- the pipelined one (upload, kernel and download of 48 MiB chunks rotating over three
// streams) over its own host link when the slices are large. Nothing is exchanged between devices: results go back to host
// memory. Shards beyond the device count share devices round robin ("virtual shards": what a one-GPU box can test).
template <class T>
int inverse_host_multi(int algo, int n, const void *hA, void *hAinv, size_t batch, int *info, int nshards, int kernel)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (batch == 0) return MATINV_OK;
    if (!hA || !hAinv) return fail(MATINV_ERR_ARG, "null host pointer");
    std::vector<int> devs;
    int rc = gfx950_devices(devs);
    if (rc) return rc;
    if (nshards <= 0) nshards = (int)devs.size();
    if (nshards > 1024) return fail(MATINV_ERR_ARG, "nshards %d: at most 1024", nshards);
    const size_t mat = (size_t)n * n;
    int home = 0;
    (void)hipGetDevice(&home);
    // page-lock both buffers ONCE for every device (adjacent shards share pages: per-shard registration would collide);
    // small batches and aliasing buffers go through the plain staged copies instead
    const bool pin = hA != hAinv && batch * mat * sizeof(T) >= ((size_t)64 << 20);
    bool pinned = false;
    if (pin) {
        if (hipHostRegister(const_cast<void *>(hA), batch * mat * sizeof(T), hipHostRegisterPortable) == hipSuccess) {
            if (hipHostRegister(hAinv, batch * mat * sizeof(T), hipHostRegisterPortable) == hipSuccess) pinned = true;
            else (void)hipHostUnregister(const_cast<void *>(hA));
        }
        if (!pinned) (void)hipGetLastError();
    }
    struct Result {
        int rc = MATINV_OK;
        std::string msg;
    };
    std::vector<Result> res((size_t)nshards);
    std::vector<std::thread> workers;
    for (int g = 0; g < nshards; ++g) {
        size_t lo = 0, hi = 0;
        shard_range(batch, nshards, n, g, lo, hi);
        if (hi == lo) continue;
        const int dev = devs[(size_t)g % devs.size()];
        workers.emplace_back([=, &res]() {
            hipError_t e = hipSetDevice(dev);
            int r = e == hipSuccess ? inverse_host<T>(algo, n, static_cast<const T *>(hA) + lo * mat, static_cast<T *>(hAinv) + lo * mat,
                                                      hi - lo, info ? info + lo : nullptr, nullptr, kernel, true,
                                                      pinned ? HOST_REGISTERED : (pin ? HOST_STAGED_ONLY : HOST_DEFAULT))
                                    : fail_hip(e, "hipSetDevice");
            res[(size_t)g].rc = r;
            if (r != MATINV_OK) res[(size_t)g].msg = g_err;  // the worker's thread-local message
        });
    }
    for (auto &t : workers) t.join();
    if (pinned) {
        (void)hipHostUnregister(const_cast<void *>(hA));
        (void)hipHostUnregister(hAinv);
    }
    (void)hipSetDevice(home);
    for (int g = 0; g < nshards; ++g)
        if (res[(size_t)g].rc != MATINV_OK) return fail(res[(size_t)g].rc, "shard %d of %d: %s", g, nshards, res[(size_t)g].msg.c_str());
    return MATINV_OK;
}

// Reference *_batched_device form: HOST-resident tables of DEVICE pointers (filled by batchedCudaMalloc,
// /root/reference/src/helper.cu:103-118, so normally equally spaced by the pitch). Equally spaced tables are
// passed as base + stride; anything else is staged into a device-side table that lives until the launch retires.
template <class T>
int inverse_table(int algo, int n, T *const *hostIn, T *const *hostOut, size_t batch, int chol_phases = 7,
                  int kernel = MATINV_KERNEL_AUTO)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (batch == 0) return MATINV_OK;
    if (!hostIn || !hostOut) return fail(MATINV_ERR_ARG, "null pointer table");
    auto uniform = [&](T *const *tab, size_t &stride) {
        stride = (size_t)n * n;
        if (batch == 1) return true;
        if (tab[1] <= tab[0]) return false;
        stride = (size_t)(tab[1] - tab[0]);
        if (stride < (size_t)n * n) return false;
        for (size_t i = 2; i < batch; ++i)
            if (tab[i] != tab[0] + i * stride) return false;
        return true;
    };
    size_t sIn = 0, sOut = 0;
    const bool uIn = uniform(hostIn, sIn), uOut = uniform(hostOut, sOut);
    T **dTabIn = nullptr, **dTabOut = nullptr;
    hipError_t e = hipSuccess;
    if (!uIn) {
        if ((e = hipMalloc(&dTabIn, batch * sizeof(T *))) != hipSuccess) return fail_hip(e, "hipMalloc(table)");
        e = hipMemcpy(dTabIn, hostIn, batch * sizeof(T *), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !uOut) {
        if ((e = hipMalloc(&dTabOut, batch * sizeof(T *))) == hipSuccess)
            e = hipMemcpy(dTabOut, hostOut, batch * sizeof(T *), hipMemcpyHostToDevice);
    }
    int rc = MATINV_OK;
    if (e != hipSuccess) rc = fail_hip(e, "pointer table staging");
    if (rc == MATINV_OK) {
        BatchRef<const T> A{hostIn[0], sIn, uIn ? nullptr : const_cast<const T *const *>(dTabIn)};
        BatchRef<T> X{hostOut[0], sOut, uOut ? nullptr : dTabOut};
        rc = inverse_dispatch<T>(algo, n, A, X, batch, nullptr, nullptr, kernel, chol_phases);
    }
    if (dTabIn || dTabOut) {
        // rare path: the table must outlive the asynchronous launch
        (void)hipDeviceSynchronize();
        if (dTabIn) (void)hipFree(dTabIn);
        if (dTabOut) (void)hipFree(dTabOut);
    }
    return rc;
}

template <class T>
int gp_dispatch(int n, const void *a, const void *B, const void *c, const void *d, const void *e_, void *out,
                size_t batch, int *dInfo, void *stream, bool variance)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (batch == 0) return MATINV_OK;
    if (!a || !B || !c || !out || (variance ? !e_ : !d)) return fail(MATINV_ERR_ARG, "null device pointer");
    if (batch > 0x7fffffffu) return fail(MATINV_ERR_ARG, "batch %zu exceeds the grid limit; split the call", batch);
    int rc = check_device();
    if (rc) return rc;
    if (rowlane_family_supports<T>(n)) {
        static const bool use_rowlane = []() {
            const char *s = getenv("MATINV_GP_ROWLANE");  // A/B switch for profiling; default on
            return !(s && *s == '0');
        }();
        if (use_rowlane) {
            hipError_t er = launch_gp_rowlane<T>(n, static_cast<const T *>(a), static_cast<const T *>(B), static_cast<const T *>(c),
                                                 variance ? nullptr : static_cast<const T *>(d), static_cast<const T *>(e_),
                                                 static_cast<T *>(out), batch, dInfo, static_cast<hipStream_t>(stream));
            if (er != hipSuccess) return fail_hip(er, "kernel launch");
            return MATINV_OK;
        }
    }
    if (rowlane2_gp_use(sizeof(T) == 8, n)) {  // 16 < n <= 25: the two-rows-per-lane kernel, inverse folded in registers (r03)
        hipError_t er = launch_gp_rowlane2<T>(n, static_cast<const T *>(a), static_cast<const T *>(B), static_cast<const T *>(c),
                                              variance ? nullptr : static_cast<const T *>(d), static_cast<const T *>(e_),
                                              static_cast<T *>(out), batch, dInfo, static_cast<hipStream_t>(stream));
        if (er != hipSuccess) return fail_hip(er, "kernel launch");
        return MATINV_OK;
    }
    // MFMA tile kernels, one or two wavefronts per item, lower tiles only: fp64 112 < n <= 176 on two wavefronts, 176 < n <= 192 on three (spd_tile2_impl.hpp),
    // the SPD sweep with the bilinear form folded out of the accumulators where the bordered form no longer fits one wavefront
    // (fp64 80 < n <= 112, fp32 96 < n <= 160), the bordered sweep below that. (r01 - r03 kept A/B switches to the older kernels these
    // replaced -- MATINV_GP_TILE / _GP_SPD_TILE / _GP_TILE4 and a several-wavefront all-tiles pipeline kernel: gone with it in r04.)
    if constexpr (sizeof(T) == 8) {
        if (spd_tile2_supports(true, n)) {
            hipError_t e = launch_gp_spd_tile2(n, static_cast<const double *>(a), static_cast<const double *>(B), static_cast<const double *>(c),
                                               variance ? nullptr : static_cast<const double *>(d), static_cast<const double *>(e_),
                                               static_cast<double *>(out), batch, dInfo, static_cast<hipStream_t>(stream));
            if (e != hipSuccess) return fail_hip(e, "kernel launch");
            return MATINV_OK;
        }
    }
    if (gp_spd_tile_supports(sizeof(T) == 8, n)) {
        hipError_t e = launch_gp_spd_tile<T>(n, static_cast<const T *>(a), static_cast<const T *>(B), static_cast<const T *>(c),
                                             variance ? nullptr : static_cast<const T *>(d), static_cast<const T *>(e_),
                                             static_cast<T *>(out), batch, dInfo, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail_hip(e, "kernel launch");
        return MATINV_OK;
    }
    if (gp_tile_supports(sizeof(T) == 8, n)) {
        hipError_t e = launch_gp_tile<T>(n, static_cast<const T *>(a), static_cast<const T *>(B),
                                         static_cast<const T *>(c), variance ? nullptr : static_cast<const T *>(d),
                                         static_cast<const T *>(e_), static_cast<T *>(out), batch, dInfo,
                                         static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail_hip(e, "kernel launch");
        return MATINV_OK;
    }
    // measured: the LDS kernel wins while two workgroups fit a CU (f32 up to n = 137: 4.7e6 vs 3.6e6 items/s at 130) and, in
    // f64, over its whole range (2.0e6 vs 1.75e6 at 130); with one f32 workgroup per CU the blocked path wins (3.1e6 vs 1.6e6
    // at 160)
    if (!lds_family_supports<T>(n) || (sizeof(T) == 4 && n > 137)) {
        if (!global_family_supports<T>(n)) return fail(MATINV_ERR_UNSUPPORTED, "pipeline: n=%d exceeds the limit 1024", n);
        static const bool use_blocked = []() {
            const char *s = getenv("MATINV_GP_BLOCKED");  // A/B switch for profiling; default on
            return !(s && *s == '0');
        }();
        if (use_blocked) {
            hipError_t eb = launch_gp_blocked<T>(n, static_cast<const T *>(a), static_cast<const T *>(B),
                                                 static_cast<const T *>(c), variance ? nullptr : static_cast<const T *>(d),
                                                 static_cast<const T *>(e_), static_cast<T *>(out), batch, dInfo,
                                                 static_cast<hipStream_t>(stream));
            if (eb != hipSuccess) return fail_hip(eb, "kernel launch");
            return MATINV_OK;
        }
        if (!lds_family_supports<T>(n)) {
            hipError_t eg = launch_gp_global<T>(n, static_cast<const T *>(a), static_cast<const T *>(B),
                                                static_cast<const T *>(c), variance ? nullptr : static_cast<const T *>(d),
                                                static_cast<const T *>(e_), static_cast<T *>(out), batch, dInfo,
                                                static_cast<hipStream_t>(stream));
            if (eg != hipSuccess) return fail_hip(eg, "kernel launch");
            return MATINV_OK;
        }
    }
    hipError_t e = launch_gp_lds<T>(n, static_cast<const T *>(a), static_cast<const T *>(B), static_cast<const T *>(c),
                                    variance ? nullptr : static_cast<const T *>(d), static_cast<const T *>(e_),
                                    static_cast<T *>(out), batch, dInfo, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return MATINV_OK;
}

// Host-pointer form of the fused pipeline (what gauss_bench's calcluateMean / calcluateVariance time,
// /root/reference/src/gauss_bench.cu:127-265): allocate, H2D, one kernel, D2H of batch scalars, free.
template <class T>
int gp_host(int n, const void *hA, const void *hB, const void *hC, const void *hX, void *hOut, size_t batch, int *info,
            bool variance)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    if (batch == 0) return MATINV_OK;
    if (!hA || !hB || !hC || !hX || !hOut) return fail(MATINV_ERR_ARG, "null host pointer");
    int rc = check_device();
    if (rc) return rc;
    const size_t vec = (size_t)n * batch, mat = vec * n, xlen = variance ? batch : vec;
    const bool log = detailed_logging();
    const char *key = variance ? "calculate_variance_gpu" : "calculate_mean_gpu";
    T *dA = nullptr, *dB = nullptr, *dC = nullptr, *dX = nullptr, *dOut = nullptr;
    int *dInfo = nullptr;
    hipError_t e = hipSuccess;
    auto alloc = [&](T **p, size_t count) {
        if (e == hipSuccess) e = staging_alloc(reinterpret_cast<void **>(p), count * sizeof(T));
    };
    alloc(&dA, vec), alloc(&dB, mat), alloc(&dC, vec), alloc(&dX, xlen), alloc(&dOut, batch);
    if (e == hipSuccess && info) e = staging_alloc(reinterpret_cast<void **>(&dInfo), batch * sizeof(int));
    const double t0 = now_ms();
    auto h2d = [&](T *d, const void *h, size_t count) {
        if (e == hipSuccess) e = hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice);
    };
    h2d(dA, hA, vec), h2d(dB, hB, mat), h2d(dC, hC, vec), h2d(dX, hX, xlen);
    const double t1 = now_ms();
    if (e == hipSuccess) {
        rc = gp_dispatch<T>(n, dA, dB, dC, variance ? nullptr : dX, variance ? dX : nullptr, dOut, batch, dInfo, nullptr,
                            variance);
        if (rc == MATINV_OK && log) e = hipDeviceSynchronize();
    }
    const double t2 = now_ms();
    if (e == hipSuccess && rc == MATINV_OK) e = hipMemcpy(hOut, dOut, batch * sizeof(T), hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc == MATINV_OK && info) e = hipMemcpy(info, dInfo, batch * sizeof(int), hipMemcpyDeviceToHost);
    const double t3 = now_ms();
    staging_free(dA), staging_free(dB), staging_free(dC), staging_free(dX), staging_free(dOut);
    staging_free(dInfo);
    if (rc != MATINV_OK) return rc;
    if (e != hipSuccess) return fail_hip(e, "pipeline host<->device");
    if (log) {
        // the reference's six keys (src/gauss_bench.cu:151-156,250-255; results/generate_plots.m:69-76 sums them): the
        // fused kernel does add + inv + mul + dot in one launch, so its time is booked under _inv and the others are 0
        timer_log(key, "mem_htod", batch, n, t1 - t0);
        timer_log(key, "add", batch, n, 0.0);
        timer_log(key, "inv", batch, n, t2 - t1);
        timer_log(key, "mul", batch, n, 0.0);
        timer_log(key, "dot", batch, n, 0.0);
        timer_log(key, "mem_dtoh", batch, n, t3 - t2);
    }
    return MATINV_OK;
}

template <class T>
int lu_kernel(int n)
{
    return (n > 16 && (tilep_supports(n) || tileq_supports(sizeof(T) == 8, n))) ? (int)MATINV_KERNEL_TILEP : (int)MATINV_KERNEL_AUTO;
}

// Reference error contract: message on stderr, then exit (include/helper_gpu.h:9-18, helper_cpu.h:12-21 there).
void die_on(int rc, const char *fn)
{
    if (rc == MATINV_OK) return;
    fprintf(stderr, "ENSURE FAILED %s\r\n%s\r\n", fn, g_err);
    if (errno) perror("possible reason for failure from ERRNO");
    (void)hipDeviceReset();
    exit(EXIT_FAILURE);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

int matinv_abi_version(void) { return 2; }  // 2 (r03): + matinv_queue_*, matinv_batched_*, matinv_set_gj_policy, *_host_multi, matinv_allgather_*

int matinv_set_gj_policy(int policy)
{
    if (policy != MATINV_GJ_NATURAL_FIRST && policy != MATINV_GJ_PIVOT && policy != MATINV_GJ_ADAPTIVE)
        return fail(MATINV_ERR_ARG, "unknown Gauss-Jordan policy %d", policy);
    return set_gj_policy(policy);
}

int matinv_shard_range(size_t batch, int nshards, int n, int g, size_t *lo, size_t *hi)
{
    if (nshards < 1 || g < 0 || g >= nshards || n < 1 || !lo || !hi) return fail(MATINV_ERR_ARG, "matinv_shard_range: bad argument");
    shard_range(batch, nshards, n, g, *lo, *hi);
    return MATINV_OK;
}

int matinv_device_count(void)
{
    std::vector<int> devs;
    int rc = gfx950_devices(devs);
    return rc ? rc : (int)devs.size();
}

int matinv_inverse_batched_host_multi(int algo, int dtype, int n, const void *hA, void *hAinv, size_t batch, int *info, int nshards)
{
    if (dtype == MATINV_F64) return inverse_host_multi<double>(algo, n, hA, hAinv, batch, info, nshards, MATINV_KERNEL_AUTO);
    if (dtype == MATINV_F32) return inverse_host_multi<float>(algo, n, hA, hAinv, batch, info, nshards, MATINV_KERNEL_AUTO);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

int matinv_release_cache(void)
{
    int rc = check_device();
    if (rc) return rc;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail_hip(e, "release cache");
    blocked_gp_release_graphs();
    scratch_release_device();
    return MATINV_OK;
}
int matinv_stream_retire(void *stream)
{
    if (!stream) return fail(MATINV_ERR_ARG, "matinv_stream_retire: the null stream is never retired");
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "stream retire");
    scratch_retire_stream(static_cast<hipStream_t>(stream));
    return MATINV_OK;
}
long long matinv_debug_rejects(int reset) { return debug_rejects(reset != 0); }
const char *matinv_last_error(void) { return g_err; }

// batchedCudaMalloc (/root/reference/src/helper.cu:103-118): ONE pitched device block, host table of row pointers.
int matinv_batched_malloc(void **devArrayPtr, size_t *pitch, size_t arraySize, int batchSize)
{
    if (!devArrayPtr || !pitch || batchSize < 0) return fail(MATINV_ERR_ARG, "matinv_batched_malloc: bad argument");
    if (batchSize == 0 || arraySize == 0) {
        *pitch = arraySize;
        return MATINV_OK;
    }
    int rc = check_device();
    if (rc) return rc;
    char *p = nullptr;
    hipError_t e = hipMallocPitch(reinterpret_cast<void **>(&p), pitch, arraySize, (size_t)batchSize);
    if (e != hipSuccess) return fail_hip(e, "hipMallocPitch");
    for (int i = 0; i < batchSize; ++i) devArrayPtr[i] = p + (size_t)i * *pitch;
    return MATINV_OK;
}

int matinv_batched_free(void **devArrayPtr)
{
    if (!devArrayPtr || !devArrayPtr[0]) return MATINV_OK;
    hipError_t e = hipFree(devArrayPtr[0]);
    if (e != hipSuccess) return fail_hip(e, "hipFree");
    devArrayPtr[0] = nullptr;
    return MATINV_OK;
}

// cudaMemcpy2D as the reference uses it on those blocks (src/gauss_bench.cu:165-170,244): blocking, default stream
int matinv_memcpy_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, int toDevice)
{
    if (width == 0 || height == 0) return MATINV_OK;
    if (!dst || !src || dpitch < width || spitch < width) return fail(MATINV_ERR_ARG, "matinv_memcpy_2d: bad argument");
    hipError_t e = hipMemcpy2D(dst, dpitch, src, spitch, width, height, toDevice ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy2D");
    return MATINV_OK;
}

// how the adaptive Gauss-Jordan dispatch of the tile family went (tile_kernels.inc "natural order or pivot search?")
int matinv_tile_stats(unsigned long long *natural_launches, unsigned long long *pivot_launches, unsigned long long *last_rejected,
                      unsigned long long *last_batch)
{
    const TileStats s = tile_stats();
    if (natural_launches) *natural_launches = s.natural_launches;
    if (pivot_launches) *pivot_launches = s.pivot_launches;
    if (last_rejected) *last_rejected = s.last_rejected;
    if (last_batch) *last_batch = s.last_batch;
    return MATINV_OK;
}

int matinv_device_synchronize(void)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail_hip(e, "hipDeviceSynchronize");
    return MATINV_OK;
}

int matinv_inverse_batched_ex(int algo, int dtype, int n, const void *dA, size_t strideA, void *dAinv,
                              size_t strideInv, size_t batch, int *dInfo, void *stream, int kernel)
{
    if (dtype == MATINV_F64)
        return inverse_strided<double>(algo, n, dA, strideA, dAinv, strideInv, batch, dInfo, stream, kernel);
    if (dtype == MATINV_F32)
        return inverse_strided<float>(algo, n, dA, strideA, dAinv, strideInv, batch, dInfo, stream, kernel);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

int matinv_inverse_batched(int algo, int dtype, int n, const void *dA, size_t strideA, void *dAinv, size_t strideInv,
                           size_t batch, int *dInfo, void *stream)
{
    return matinv_inverse_batched_ex(algo, dtype, n, dA, strideA, dAinv, strideInv, batch, dInfo, stream,
                                     MATINV_KERNEL_AUTO);
}

int matinv_select_kernel(int algo, int dtype, int n)
{
    if (n < 1) return fail(MATINV_ERR_ARG, "n must be >= 1 (got %d)", n);
    int k = (dtype == MATINV_F64) ? select_auto<double>(algo, n) : select_auto<float>(algo, n);
    if (k < 0) return fail(MATINV_ERR_UNSUPPORTED, "n=%d exceeds every kernel family built in", n);
    return k;
}

const char *matinv_kernel_name(int algo, int dtype, int n, int kernel)
{
    const bool f64 = dtype == MATINV_F64;
    if (kernel == MATINV_KERNEL_AUTO) kernel = matinv_select_kernel(algo, dtype, n);
    switch (kernel) {
    case MATINV_KERNEL_LDS: return algo == MATINV_ALGO_CHOLESKY ? name_chol_lds(f64) : name_gj_lds(f64);
    case MATINV_KERNEL_ROWLANE: return algo == MATINV_ALGO_CHOLESKY ? name_spd_rowlane(f64, n) : name_gj_rowlane(f64, n);
    case MATINV_KERNEL_TILE: return algo == MATINV_ALGO_CHOLESKY ? name_spd_tile(f64, n) : name_gj_tile(f64, n);
    case MATINV_KERNEL_ROW: return name_gj_row(f64, n);
    case MATINV_KERNEL_TILEP: return n > 128 ? name_gj_tileq(f64, n) : name_gj_tilep(f64, n);
    case MATINV_KERNEL_GLOBAL: return algo == MATINV_ALGO_CHOLESKY ? name_chol_global(f64) : name_gj_global(f64);
    case MATINV_KERNEL_BLOCKED:
        if (algo == MATINV_ALGO_GAUSS_JORDAN) {
            // the kernel that dominates: the rank-128 MFMA update of the two-level scheme, the rank-32 VALU update below its threshold
            if (n < blocked_gj_two_level_min()) return f64 ? "matinv_bgj_update1<double>" : "matinv_bgj_update1<float>";
            return f64 ? "matinv_bgj_update_mfma<double, false, 2, 2>" : "matinv_bgj_update_mfma<float, false, 2, 2>";
        }
        return f64 ? "matinv_bgp_update<double>" : "matinv_bgp_update<float>";  // the trailing update: most of the time
    default: return "";
    }
}

int matinv_mean_batched(int dtype, int n, const void *dAs, const void *dBs, const void *dCs, const void *dDs,
                        void *dMeans, size_t batch, int *dInfo, void *stream)
{
    if (dtype == MATINV_F64) return gp_dispatch<double>(n, dAs, dBs, dCs, dDs, nullptr, dMeans, batch, dInfo, stream, false);
    if (dtype == MATINV_F32) return gp_dispatch<float>(n, dAs, dBs, dCs, dDs, nullptr, dMeans, batch, dInfo, stream, false);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

int matinv_variance_batched(int dtype, int n, const void *dAs, const void *dBs, const void *dCs, const void *dEs,
                            void *dVars, size_t batch, int *dInfo, void *stream)
{
    if (dtype == MATINV_F64) return gp_dispatch<double>(n, dAs, dBs, dCs, nullptr, dEs, dVars, batch, dInfo, stream, true);
    if (dtype == MATINV_F32) return gp_dispatch<float>(n, dAs, dBs, dCs, nullptr, dEs, dVars, batch, dInfo, stream, true);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

int matinv_mean_batched_host(int dtype, int n, const void *hAs, const void *hBs, const void *hCs, const void *hDs,
                             void *hMeans, size_t batch, int *info)
{
    if (dtype == MATINV_F64) return gp_host<double>(n, hAs, hBs, hCs, hDs, hMeans, batch, info, false);
    if (dtype == MATINV_F32) return gp_host<float>(n, hAs, hBs, hCs, hDs, hMeans, batch, info, false);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

int matinv_variance_batched_host(int dtype, int n, const void *hAs, const void *hBs, const void *hCs, const void *hEs,
                                 void *hVars, size_t batch, int *info)
{
    if (dtype == MATINV_F64) return gp_host<double>(n, hAs, hBs, hCs, hEs, hVars, batch, info, true);
    if (dtype == MATINV_F32) return gp_host<float>(n, hAs, hBs, hCs, hEs, hVars, batch, info, true);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

int matinv_inverse_batched_host(int algo, int dtype, int n, const void *hA, void *hAinv, size_t batch, int *info)
{
    if (dtype == MATINV_F64) return inverse_host<double>(algo, n, hA, hAinv, batch, info, nullptr);
    if (dtype == MATINV_F32) return inverse_host<float>(algo, n, hA, hAinv, batch, info, nullptr);
    return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
}

// ------------------------------------------------------------------------------------------------
// The 17 reference names (include/inverse_gpu.h), fp64 under the plain name and fp32 under <name>_f32.
// `handle` is ignored exactly as the reference's hand-written kernels ignore it.
#define GJ MATINV_ALGO_GAUSS_JORDAN
#define CH MATINV_ALGO_CHOLESKY
#define KAUTO(T, n) MATINV_KERNEL_AUTO
// the reference's LU entry points promise partial pivoting (cublas getrfBatched, src/gauss/inverse_gpu.cu:24-33): where the
// MFMA tile family serves n they go straight to its pivoting kernel, whatever the Gauss-Jordan policy says
#define KLU(T, n) lu_kernel<T>(n)
#define REF_GPU(name, suffix, T, algo, logkey, KSEL)                                                                 \
    void name##suffix(void *handle, int n, T *As, T *aInvs, int batchSize)                                          \
    {                                                                                                                \
        (void)handle;                                                                                                \
        die_on(batchSize < 0 ? fail(MATINV_ERR_ARG, "negative batchSize") : inverse_host<T>(algo, n, As, aInvs,      \
               (size_t)batchSize, nullptr, logkey, KSEL(T, n)), #name);                                              \
    }
#define REF_DEV(name, suffix, T, algo, in, out, phases, KSEL)                                                        \
    void name##suffix(void *handle, int N, T **devAs, T **devAInvs, int batchSize)                                  \
    {                                                                                                                \
        (void)handle; (void)devAs; (void)devAInvs;                                                                   \
        die_on(batchSize < 0 ? fail(MATINV_ERR_ARG, "negative batchSize") : inverse_table<T>(algo, N, in, out,       \
               (size_t)batchSize, phases, KSEL(T, N)), #name);                                                       \
    }
// log keys are the reference's TIMER_INIT names for each entry point
#define REF_ALL(suffix, T)                                                                                           \
    REF_GPU(inverse_gauss_batched_gpu, suffix, T, GJ, "inverse_gauss_batched_gpu", KAUTO)             /* batched_invert.cu:99 */ \
    REF_GPU(inverse_lu_cuda_batched_gpu, suffix, T, GJ, "inverse_lu_cuda_batched_gpu", KLU)         /* gauss/inverse_gpu.cu:60 */ \
    REF_GPU(inverse_cholesky_batched_gpu, suffix, T, CH, "decompose_cholesky_batched_gpu", KAUTO)     /* inverse_cholesky_gpu.cu:397 */ \
    REF_GPU(inverse_cholesky_mm_batched_gpu, suffix, T, CH, "decompose_cholesky_mm_batched_gpu", KAUTO) /* :627 */          \
    REF_GPU(inverse_cholesky_mm2_batched_gpu, suffix, T, CH, "cholesky_mm2_batched_gpu", KAUTO)       /* :699 */            \
    REF_GPU(inverse_cholesky_stride_batched_gpu, suffix, T, CH, "inverse_cholesky_stride_batched_gpu", KAUTO) /* :189 */    \
    REF_DEV(inverse_gauss_batched_device, suffix, T, GJ, devAs, devAInvs, 7, KAUTO)   /* declared inverse_gpu.h:10, never defined there */ \
    REF_DEV(inverse_lu_cuda_batched_device, suffix, T, GJ, devAs, devAInvs, 7, KLU) /* gauss/inverse_gpu.cu:16; input kept intact here */ \
    REF_DEV(inverse_cholesky_batched_device, suffix, T, CH, devAs, devAInvs, 7, KAUTO)        /* :323 */                    \
    REF_DEV(inverse_cholesky_mm_batched_device, suffix, T, CH, devAs, devAInvs, 7, KAUTO)     /* :608 */                    \
    REF_DEV(inverse_cholesky_mm2_batched_device, suffix, T, CH, devAs, devAInvs, 7, KAUTO)    /* :693 */                    \
    /* stride family works in place on devAInvs (the *_gpu wrapper copies As there first, :213-216) */              \
    REF_DEV(inverse_cholesky_stride_batched_device, suffix, T, CH, devAInvs, devAInvs, 7, KAUTO)     /* :182 */             \
    REF_DEV(decompose_cholesky_stride_batched_device, suffix, T, CH, devAInvs, devAInvs, 1, KAUTO)   /* :96  */             \
    REF_DEV(inverse_upper_stride_batched_device, suffix, T, CH, devAInvs, devAInvs, 2, KAUTO)        /* :137 */             \
    REF_DEV(multiply_upper_stride_batched_device, suffix, T, CH, devAInvs, devAInvs, 4, KAUTO)       /* :175 */             \
    /* decompose: factor in place in devAs, strict upper triangle zeroed (:357-369, :616-624) */                    \
    REF_DEV(decompose_cholesky_batched_device, suffix, T, CH, devAs, devAs, 1, KAUTO)                                       \
    REF_DEV(decompose_cholesky_mm_batched_device, suffix, T, CH, devAs, devAs, 1, KAUTO)

REF_ALL(, double)
REF_ALL(_f32, float)

}  // extern "C"
