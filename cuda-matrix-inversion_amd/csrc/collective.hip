// collective.hip -- result reassembly across GPUs over RCCL (xGMI), behind the C ABI (include/matinv.h, matinv_comm_* /
// matinv_allgather_*). The inversion path has exactly one exchange step -- every rank's shard of A^-1 gathered on every rank
// (the reference is single-device) -- so this file wraps ncclAllGather and nothing else.
//
// librccl is opened at the first call (dlopen), not linked: a host that never gathers (the host-pointer entry points return
// their results to host memory) needs no RCCL, and inside a PyTorch process the copy of librccl that torch has loaded already
// is reused instead of a second one.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/matinv.h"
#include "common.hpp"

using namespace matinv;

namespace {

// the few declarations of <rccl/rccl.h> this file needs (ABI-stable NCCL 2 interface)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;    // ncclSuccess = 0
typedef int ncclDataType_t;  // ncclFloat32 = 7, ncclFloat64 = 8
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // a copy that is loaded already (torch's) first, then the ROCm installation's
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *nm : names)
            if (!r.handle) r.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *p : paths)
            if (!r.handle) r.handle = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            snprintf(r.why, sizeof r.why, "librccl not found: %s", dlerror());
            return;
        }
#define RCCL_SYM(field, name)                                                                                          \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                                              \
    if (!r.field && !r.why[0]) snprintf(r.why, sizeof r.why, "librccl lacks %s", name)
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        RCCL_SYM(CommInitRank, "ncclCommInitRank");
        RCCL_SYM(CommInitAll, "ncclCommInitAll");
        RCCL_SYM(CommDestroy, "ncclCommDestroy");
        RCCL_SYM(AllGather, "ncclAllGather");
        RCCL_SYM(GroupStart, "ncclGroupStart");
        RCCL_SYM(GroupEnd, "ncclGroupEnd");
        RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    });
    return &r;
}

int need_rccl(Rccl *&r)
{
    r = rccl();
    if (r->why[0]) return fail(MATINV_ERR_UNSUPPORTED, "RCCL unavailable: %s", r->why);
    return MATINV_OK;
}

int fail_nccl(Rccl *r, ncclResult_t e, const char *what)
{
    return fail(MATINV_ERR_HIP, "%s: %s (%d)", what, r->GetErrorString ? r->GetErrorString(e) : "rccl error", (int)e);
}

int nccl_type(int dtype, ncclDataType_t &t)
{
    if (dtype == MATINV_F64) t = kNcclFloat64;
    else if (dtype == MATINV_F32) t = kNcclFloat32;
    else return fail(MATINV_ERR_ARG, "unknown dtype %d", dtype);
    return MATINV_OK;
}

// communicators of the single-process form, created once per device list
struct LocalComms {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};
std::mutex g_local_mu;
std::vector<LocalComms> g_local;

}  // namespace

extern "C" {

int matinv_comm_unique_id(void *id128)
{
    if (!id128) return fail(MATINV_ERR_ARG, "null id buffer");
    Rccl *r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e) return fail_nccl(r, e, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof id);
    return MATINV_OK;
}

int matinv_comm_init_rank(void **comm, int nranks, const void *id128, int rank)
{
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(MATINV_ERR_ARG, "matinv_comm_init_rank: bad argument");
    Rccl *r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t e = r->CommInitRank(&c, nranks, id, rank);  // on the calling thread's current device
    if (e) return fail_nccl(r, e, "ncclCommInitRank");
    *comm = c;
    return MATINV_OK;
}

int matinv_comm_destroy(void *comm)
{
    if (!comm) return MATINV_OK;
    Rccl *r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclResult_t e = r->CommDestroy(static_cast<ncclComm_t>(comm));
    if (e) return fail_nccl(r, e, "ncclCommDestroy");
    return MATINV_OK;
}

// ONE ncclAllGather: `count` elements from every rank, nranks * count received, on `stream` (asynchronous)
int matinv_allgather_shards(void *comm, int dtype, const void *dSend, void *dRecv, size_t count, void *stream)
{
    if (!comm || (count && (!dSend || !dRecv))) return fail(MATINV_ERR_ARG, "matinv_allgather_shards: null argument");
    Rccl *r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclDataType_t t;
    if ((rc = nccl_type(dtype, t))) return rc;
    if (count == 0) return MATINV_OK;
    ncclResult_t e = r->AllGather(dSend, dRecv, count, t, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream));
    if (e) return fail_nccl(r, e, "ncclAllGather");
    return MATINV_OK;
}

// single process, ndev devices: dSend[g] (count elements on devices[g]) gathered into dRecv[g] (ndev * count elements on
// devices[g]) for every g; returns when all of it has completed. The gather runs on the library's own streams: it is ordered
// behind the work that produces dSend[g] through `producers` -- producers[g] = the stream (on devices[g]) whose work so far writes
// dSend[g]: an event recorded there is waited for on the gather's stream (a null entry = the device's null stream). With
// producers == nullptr every device is synchronised on entry instead (r03 took neither precaution: a caller that inverted
// asynchronously and gathered at once could have shipped a half-written shard -- ADVICE r03).
static int allgather_local_impl(int ndev, const int *devices, int dtype, const void *const *dSend, void *const *dRecv, size_t count,
                                void *const *producers)
{
    if (ndev < 1 || !devices || !dSend || !dRecv) return fail(MATINV_ERR_ARG, "matinv_allgather_local: bad argument");
    Rccl *r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclDataType_t t;
    if ((rc = nccl_type(dtype, t))) return rc;
    if (count == 0) return MATINV_OK;
    int home = 0;
    (void)hipGetDevice(&home);
    std::lock_guard<std::mutex> lock(g_local_mu);
    LocalComms *lc = nullptr;
    for (auto &c : g_local)
        if ((int)c.devices.size() == ndev && !memcmp(c.devices.data(), devices, sizeof(int) * ndev)) lc = &c;
    if (!lc) {
        LocalComms c;
        c.devices.assign(devices, devices + ndev);
        c.comms.assign(ndev, nullptr);
        c.streams.assign(ndev, nullptr);
        ncclResult_t e = r->CommInitAll(c.comms.data(), ndev, devices);
        if (e) return fail_nccl(r, e, "ncclCommInitAll");
        hipError_t he = hipSuccess;
        for (int g = 0; g < ndev && he == hipSuccess; ++g) {
            he = hipSetDevice(devices[g]);
            if (he == hipSuccess) he = hipStreamCreateWithFlags(&c.streams[g], hipStreamNonBlocking);
        }
        if (he != hipSuccess) {
            // nothing half-built stays behind: the streams created so far and the communicators go
            for (int g = 0; g < ndev; ++g) {
                if (c.streams[g] && hipSetDevice(devices[g]) == hipSuccess) (void)hipStreamDestroy(c.streams[g]);
                if (c.comms[g]) (void)r->CommDestroy(c.comms[g]);
            }
            (void)hipSetDevice(home);
            return fail_hip(he, "stream for the all-gather");
        }
        g_local.push_back(c);
        lc = &g_local.back();
    }
    // order the gather behind the producers of the shards
    hipError_t he = hipSuccess;
    for (int g = 0; g < ndev && he == hipSuccess; ++g) {
        he = hipSetDevice(devices[g]);
        if (he != hipSuccess) break;
        if (!producers) {
            he = hipDeviceSynchronize();
        } else {
            hipEvent_t ev = nullptr;
            he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventRecord(ev, static_cast<hipStream_t>(producers[g]));
            if (he == hipSuccess) he = hipStreamWaitEvent(lc->streams[g], ev, 0);
            if (ev) (void)hipEventDestroy(ev);  // deferred by the runtime until the recorded work has completed
        }
    }
    if (he != hipSuccess) {
        (void)hipSetDevice(home);
        return fail_hip(he, "ordering the all-gather behind its producers");
    }
    ncclResult_t e = r->GroupStart();
    for (int g = 0; g < ndev && !e; ++g) e = r->AllGather(dSend[g], dRecv[g], count, t, lc->comms[g], lc->streams[g]);
    ncclResult_t e2 = r->GroupEnd();
    if (!e) e = e2;
    for (int g = 0; g < ndev; ++g) {
        hipError_t h1 = hipSetDevice(devices[g]);
        if (h1 == hipSuccess) h1 = hipStreamSynchronize(lc->streams[g]);
        if (he == hipSuccess) he = h1;
    }
    (void)hipSetDevice(home);
    if (e) return fail_nccl(r, e, "ncclAllGather (group)");
    if (he != hipSuccess) return fail_hip(he, "all-gather completion");
    return MATINV_OK;
}

int matinv_allgather_local(int ndev, const int *devices, int dtype, const void *const *dSend, void *const *dRecv, size_t count)
{
    return allgather_local_impl(ndev, devices, dtype, dSend, dRecv, count, nullptr);
}

int matinv_allgather_local_after(int ndev, const int *devices, int dtype, const void *const *dSend, void *const *dRecv, size_t count,
                                 void *const *producer_streams)
{
    if (!producer_streams) return fail(MATINV_ERR_ARG, "matinv_allgather_local_after: null producer stream array");
    return allgather_local_impl(ndev, devices, dtype, dSend, dRecv, count, producer_streams);
}

}  // extern "C"
