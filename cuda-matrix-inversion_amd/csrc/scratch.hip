// scratch.hip -- device scratch memory of the launchers (work lists, the workspaces of the blocked paths, the staging buffers of
// the host-pointer entry points, the queue's gather batches): a small caching allocator keyed by (device, stream).
//
// A block belongs to the stream it was first handed out on and is only ever handed out again on THAT stream: whatever used it
// before has been enqueued there earlier, so stream order alone makes the reuse safe -- no events, no cross-stream hand-over.
// (r01/r02 used hipMallocAsync / hipFreeAsync on the device's default memory pool. With several host threads launching on one
// device -- the multi-device host path run with more shards than devices -- two in-flight launches on different streams were
// seen to receive the SAME work-list block: rejected matrices of one launch vanished from its list when the other launch zeroed
// "its" counters. r03: host/multi_test.c, 64 x 9000 x 3 shards, one run in three.)
// Blocks whose stream the library itself retires (the three streams of a pipelined host call, after synchronising them) move
// to a per-device list that any stream may draw from. matinv_release_cache() returns everything that is not in use.
#include <stdio.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace matinv {

namespace {
struct Block {
    void *ptr;
    size_t bytes;
    bool in_use;
};
struct Key {
    int dev;
    hipStream_t stream;
    bool operator==(const Key &o) const { return dev == o.dev && stream == o.stream; }
};
struct KeyHash {
    size_t operator()(const Key &k) const { return std::hash<const void *>()(k.stream) * 31u + (size_t)k.dev; }
};
std::mutex g_mu;
std::unordered_map<Key, std::vector<Block>, KeyHash> g_owned;           // blocks of a live stream
std::unordered_map<int, std::vector<Block>> g_retired;                  // per device: free blocks of retired streams
std::unordered_map<const void *, Key> g_where;                          // block in use -> its owner

// the smallest free block that holds `bytes` without wasting more than half of itself (a 16 GiB workspace is not spent on a
// 12 KiB work list)
int best_fit(std::vector<Block> &v, size_t bytes)
{
    int best = -1;
    for (int i = 0; i < (int)v.size(); ++i)
        if (!v[i].in_use && v[i].bytes >= bytes && v[i].bytes <= std::max<size_t>(2 * bytes, bytes + (1u << 20)) &&
            (best < 0 || v[i].bytes < v[best].bytes))
            best = i;
    return best;
}

// free (hipFree) every cached block of `dev` that is not in use; the caller has made sure nothing on the device still reads them
void drop_free_blocks_locked(int dev)
{
    for (auto &kv : g_owned) {
        if (kv.first.dev != dev) continue;
        auto &v = kv.second;
        for (auto &b : v)
            if (!b.in_use) (void)hipFree(b.ptr);
        v.erase(std::remove_if(v.begin(), v.end(), [](const Block &b) { return !b.in_use; }), v.end());
    }
    auto it = g_retired.find(dev);
    if (it != g_retired.end()) {
        for (auto &b : it->second) (void)hipFree(b.ptr);
        it->second.clear();
    }
}
}  // namespace

// the device a stream belongs to (the null stream: the calling thread's current device). r03 keyed the cache by the CURRENT device:
// a queue destroyed while another device was current retired nothing, and a block could be filed under the wrong device.
static hipError_t device_of(hipStream_t stream, int *dev)
{
    if (stream) {
        hipError_t e = hipStreamGetDevice(stream, dev);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    return hipGetDevice(dev);
}

hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t stream)
{
    *p = nullptr;
    if (bytes == 0) bytes = 1;
    bytes = (bytes + 255) & ~(size_t)255;
    int dev = 0, cur = 0;
    hipError_t e = device_of(stream, &dev);
    if (e == hipSuccess) e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    const Key key{dev, stream};
    std::lock_guard<std::mutex> lock(g_mu);
    auto &mine = g_owned[key];
    int i = best_fit(mine, bytes);
    if (i >= 0) {
        mine[i].in_use = true;
        *p = mine[i].ptr;
        g_where[*p] = key;
        return hipSuccess;
    }
    auto &ret = g_retired[dev];
    i = best_fit(ret, bytes);
    if (i >= 0) {  // everything that used it has completed (its stream was synchronised before it was retired)
        Block b = ret[i];
        ret.erase(ret.begin() + i);
        b.in_use = true;
        mine.push_back(b);
        *p = b.ptr;
        g_where[*p] = key;
        return hipSuccess;
    }
    void *q = nullptr;
    if (cur != dev) e = hipSetDevice(dev);  // the block lives on the stream's device, whatever device the caller is on
    if (e == hipSuccess) e = hipMalloc(&q, bytes);
    if (e == hipErrorOutOfMemory) {
        // hand the cache back to the driver and try once more
        (void)hipGetLastError();
        if (hipDeviceSynchronize() == hipSuccess) {
            drop_free_blocks_locked(dev);
            e = hipMalloc(&q, bytes);
        }
    }
    if (cur != dev) (void)hipSetDevice(cur);
    if (e != hipSuccess) return e;
    g_owned[key].push_back(Block{q, bytes, true});
    g_where[q] = key;
    *p = q;
    return hipSuccess;
}

// stream-ordered "free": the block may be handed out again on its owner stream at once (work enqueued there so far precedes
// whatever the next user enqueues)
hipError_t scratch_free(void *p, hipStream_t /*stream*/)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lock(g_mu);
    auto w = g_where.find(p);
    if (w == g_where.end()) return hipErrorInvalidValue;
    auto &v = g_owned[w->second];
    for (auto &b : v)
        if (b.ptr == p) b.in_use = false;
    g_where.erase(w);
    return hipSuccess;
}

// the caller has synchronised `stream` and is about to destroy it: its free blocks become available to every stream of the device
void scratch_retire_stream(hipStream_t stream)
{
    int dev = 0;
    if (device_of(stream, &dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_owned.find(Key{dev, stream});
    if (it == g_owned.end()) return;
    auto &v = it->second;
    auto &ret = g_retired[dev];
    for (auto &b : v)
        if (!b.in_use) ret.push_back(b);
    v.erase(std::remove_if(v.begin(), v.end(), [](const Block &b) { return !b.in_use; }), v.end());
    if (v.empty()) g_owned.erase(it);
}

// matinv_release_cache(): the current device has been synchronised by the caller
void scratch_release_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_mu);
    drop_free_blocks_locked(dev);
}

// ---- test hook: how many matrices did the first-pass kernels hand to their fallback? -----------------------------------------
// A first-pass kernel that wrongly rejects everything is invisible in the results (the fallback computes them) and shows only as
// a rate 10x below its neighbours (r03: the fp32 symmetric sweep of 9 x 9 tiles built with its accumulators in AGPRs did exactly
// that). MATINV_DEBUG_REJECTS=1 makes every launcher that owns a work list read its count back (one stream synchronisation per
// launch: a test mode, never a production one); matinv_debug_rejects() returns the running total.
static std::atomic<long long> g_rejects{0};
bool debug_rejects_on()
{
    static const bool on = []() {
        const char *s = getenv("MATINV_DEBUG_REJECTS");
        return s && *s && *s != '0';
    }();
    return on;
}
hipError_t debug_note_rejects(const int *work_count, hipStream_t stream)
{
    if (!debug_rejects_on() || !work_count) return hipSuccess;
    int c = 0;
    hipError_t e = hipMemcpyAsync(&c, work_count, sizeof c, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) g_rejects.fetch_add(c);
    return e;
}
long long debug_rejects(bool reset) { return reset ? g_rejects.exchange(0) : g_rejects.load(); }

}  // namespace matinv
