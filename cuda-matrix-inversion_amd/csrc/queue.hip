// queue.hip -- size-binned multi-queue for mixed-size Gaussian-process items, in C behind the C ABI (include/matinv.h,
// matinv_queue_*). The reference only sketches it ("use multiple queues for different sizes: 32, 128, 512, 1024",
// /root/reference/README.md:41-44) and never built it; BASELINE.json configs[4] asks for it.
//
// An item is (a, B, c, d[, e]) of some n <= the largest bin; items arrive as CHUNKS of `count` equally sized items that lie
// back to back in device memory (count = 1 is allowed). submit() only records the chunk (five pointers, n, count, first
// ticket) in the queue of the smallest bin that holds n: no device work, no per-item host work. flush() turns every
// non-empty bin into launches of the fused mean / variance kernels; the largest pending bin runs on one HIP stream, the other
// bins on a second one, so the long launch chain of the large matrices overlaps the rest:
//   * chunks of one bin are grouped by their EXACT n -- the kernels take n at run time and pad a matrix to their tile size
//     with an identity block in registers ("device-side padding"), so nothing is padded or copied in memory to reach a bin
//     size (the sketch's pad-to-the-bin policy would make an n = 40 item cost what a 128 x 128 one does);
//   * inside a group, chunks that happen to follow each other in memory are merged into runs; a group that is ONE run is
//     computed where it lies (zero copy), otherwise its chunks are gathered into one staging batch by ONE segmented-copy
//     kernel (table of (src, dst, bytes) segments) and computed with one launch;
//   * results go straight to means[ticket ..] when a launch's items carry consecutive tickets, else through a small staging
//     array and the same segmented-copy kernel.
// Host work per flush is O(chunks), one pinned-memory table upload per flush.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/matinv.h"
#include "common.hpp"

using namespace matinv;

namespace {

struct Seg {
    const char *src;
    char *dst;
    unsigned long long bytes;
};

// one workgroup column (blockIdx.y) per segment, blockIdx.x slices of it; 16-byte words when everything is aligned
__global__ __launch_bounds__(256) void matinv_segcopy(const Seg *table)
{
    const Seg s = table[blockIdx.y];
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nth = (unsigned long long)gridDim.x * blockDim.x;
    if ((((unsigned long long)s.src | (unsigned long long)s.dst | s.bytes) & 15ull) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(s.src);
        uint4 *dst = reinterpret_cast<uint4 *>(s.dst);
        for (unsigned long long i = tid; i < s.bytes / 16; i += nth) dst[i] = src[i];
    } else {
        const unsigned *src = reinterpret_cast<const unsigned *>(s.src);  // every array here is made of 4- or 8-byte scalars
        unsigned *dst = reinterpret_cast<unsigned *>(s.dst);
        for (unsigned long long i = tid; i < s.bytes / 4; i += nth) dst[i] = src[i];
    }
}

struct Chunk {
    int n;
    size_t count, ticket;
    const char *a, *B, *c, *d, *e;  // e may be null
};

}  // namespace

struct matinv_queue {
    int dtype;
    size_t esz;
    std::vector<int> bins;
    std::vector<std::vector<Chunk>> q;  // per bin
    std::vector<hipStream_t> streams;   // two ordinary non-blocking streams (see matinv_queue_create)
    std::vector<hipEvent_t> done;       // per bin: its work of the last flush
    hipEvent_t fork = nullptr, tables_uploaded = nullptr;
    size_t tickets = 0;
    bool any_e = false, all_e = true;
    Seg *host_tab = nullptr;  // pinned
    size_t host_cap = 0;
    char last_error[256] = "";
};

namespace {

int qfail(matinv_queue *q, int code, const char *msg, hipError_t e = hipSuccess)
{
    if (e != hipSuccess) snprintf(q->last_error, sizeof q->last_error, "%s: %s", msg, hipGetErrorString(e));
    else snprintf(q->last_error, sizeof q->last_error, "%s", msg);
    return code;
}

int bin_index(const matinv_queue *q, int n)
{
    for (size_t i = 0; i < q->bins.size(); ++i)
        if (n <= q->bins[i]) return (int)i;
    return -1;
}

}  // namespace

extern "C" {

int matinv_queue_create(matinv_queue **out, int dtype, const int *bins, int nbins)
{
    if (!out || (dtype != MATINV_F64 && dtype != MATINV_F32) || nbins < 0 || (nbins > 0 && !bins)) return MATINV_ERR_ARG;
    static const int default_bins[4] = {32, 128, 512, 1024};  // README.md:41-44 of the reference
    if (nbins == 0) bins = default_bins, nbins = 4;
    matinv_queue *q = new matinv_queue();
    q->dtype = dtype;
    q->esz = dtype == MATINV_F64 ? 8 : 4;
    q->bins.assign(bins, bins + nbins);
    std::sort(q->bins.begin(), q->bins.end());
    if (q->bins.front() < 1 || q->bins.back() > 1024) {
        delete q;
        return MATINV_ERR_ARG;
    }
    q->q.resize(nbins);
    q->streams.assign(1, nullptr);  // the queue's own stream; a second one is created by the first flush that forks (see flush)
    q->done.assign(1, nullptr);
    hipError_t e = hipEventCreateWithFlags(&q->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&q->tables_uploaded, hipEventDisableTiming);
    for (int i = 0; i < (int)q->streams.size() && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&q->streams[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&q->done[i], hipEventDisableTiming);
    }
    // r04: ONE ordinary non-blocking stream per queue is created here (r03: one per bin, four) and given its first command. HIP
    // multiplexes the streams of a process onto four hardware queues -- a new stream goes to the least used one -- and two launch chains
    // that share a hardware queue do not overlap; nor does a chain with a stream that waits for another chain. Whether the flushes a
    // producer keeps in flight overlapped therefore depended on every stream the process had created before (r03's bench.py tried four
    // stream sets and kept the best, 0.72 against 0.95 ms per step -- VERDICT r03 #6). Two ways to flush now (matinv_queue_flush):
    //   * on the queue's OWN stream (matinv_queue_stream): the whole flush runs in that stream, largest bin first -- no fork, no join, no
    //     stream of one flush ever waits for another stream. Four queues created before the process creates other streams sit on the four
    //     hardware queues, and their flushes overlap completely: bench.py's mixed workload 0.93 / 0.59 / 0.47 / 0.41 ms per step with
    //     1 / 2 / 3 / 4 flushes in flight (5: 0.52, 8: 0.41). The form for throughput.
    //   * on any other stream (the caller's, or the null stream): the chain of the largest pending bin forks into a second stream of
    //     the queue (created at the first such flush), the other bins into the own stream, both joined back -- 0.72 ms for one flush
    //     by itself. The form for latency; with several in flight the waits collide on the hardware queues as described.
    // Measured and NOT kept: streams created with a CU mask (hipExtStreamCreateWithCUMask, every CU enabled), which get a hardware queue
    // each. With the forking form they made the overlap independent of the process's history (0.46 ms per step with two flushes in
    // flight) -- and host/queue_test.c, extended by a flush on such a stream, hung in 17 of 52 runs when variances were asked for
    // together with a blocked size (none of 14 without variances; not the HIP-graph replay, not the block LDL^T form, not the
    // null-stream flush before it; and not the blocking semantics such streams have -- the call takes no flags: with ordinary BLOCKING
    // streams in their place 0 of 12 runs hung, with CU-mask streams 7 of 12). The cause inside the runtime was not found.
    for (int i = 0; i < (int)q->streams.size() && e == hipSuccess; ++i) {
        e = hipEventRecord(q->done[i], q->streams[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(q->streams[i]);
    }
    if (e != hipSuccess) {
        matinv_queue_destroy(q);
        return MATINV_ERR_HIP;
    }
    *out = q;
    return MATINV_OK;
}

void *matinv_queue_stream(matinv_queue *q)
{
    if (!q || q->streams.empty()) return nullptr;
    return q->streams[0];
}

int matinv_queue_destroy(matinv_queue *q)
{
    if (!q) return MATINV_OK;
    for (hipStream_t s : q->streams)
        if (s) {
            (void)hipStreamSynchronize(s);
            scratch_retire_stream(s);  // its cached gather batches may serve any stream of the device from now on
            (void)hipStreamDestroy(s);
        }
    for (hipEvent_t ev : q->done)
        if (ev) (void)hipEventDestroy(ev);
    if (q->fork) (void)hipEventDestroy(q->fork);
    if (q->tables_uploaded) {
        (void)hipEventSynchronize(q->tables_uploaded);
        (void)hipEventDestroy(q->tables_uploaded);
    }
    if (q->host_tab) (void)hipHostFree(q->host_tab);
    delete q;
    return MATINV_OK;
}

const char *matinv_queue_last_error(const matinv_queue *q) { return q ? q->last_error : "null queue"; }

int matinv_queue_submit(matinv_queue *q, int n, const void *dAs, const void *dBs, const void *dCs, const void *dDs,
                        const void *dEs, size_t count, size_t *first_ticket)
{
    if (!q) return MATINV_ERR_ARG;
    if (n < 1 || !dAs || !dBs || !dCs || !dDs) return qfail(q, MATINV_ERR_ARG, "matinv_queue_submit: bad argument");
    const int b = bin_index(q, n);
    if (b < 0) return qfail(q, MATINV_ERR_UNSUPPORTED, "matinv_queue_submit: n exceeds the largest bin");
    if (first_ticket) *first_ticket = q->tickets;
    if (count == 0) return MATINV_OK;
    q->q[b].push_back(Chunk{n, count, q->tickets, static_cast<const char *>(dAs), static_cast<const char *>(dBs),
                            static_cast<const char *>(dCs), static_cast<const char *>(dDs), static_cast<const char *>(dEs)});
    q->tickets += count;
    q->any_e = q->any_e || dEs != nullptr;
    q->all_e = q->all_e && dEs != nullptr;
    return MATINV_OK;
}

// `chunks` chunks in one call: arrays of chunks entries each (Es may be NULL as a whole or per entry); first_tickets may be NULL
int matinv_queue_submit_chunks(matinv_queue *q, size_t chunks, const int *n, const void *const *dAs, const void *const *dBs,
                               const void *const *dCs, const void *const *dDs, const void *const *dEs, const size_t *count,
                               size_t *first_tickets)
{
    if (!q) return MATINV_ERR_ARG;
    if (chunks && (!n || !dAs || !dBs || !dCs || !dDs || !count)) return qfail(q, MATINV_ERR_ARG, "matinv_queue_submit_chunks: bad argument");
    for (size_t i = 0; i < chunks; ++i) {
        const int rc = matinv_queue_submit(q, n[i], dAs[i], dBs[i], dCs[i], dDs[i], dEs ? dEs[i] : nullptr, count[i],
                                           first_tickets ? first_tickets + i : nullptr);
        if (rc != MATINV_OK) return rc;
    }
    return MATINV_OK;
}

int matinv_queue_pending(const matinv_queue *q, size_t *items, size_t *per_bin)
{
    if (!q) return MATINV_ERR_ARG;
    if (items) *items = q->tickets;
    if (per_bin)
        for (size_t b = 0; b < q->q.size(); ++b) {
            per_bin[b] = 0;
            for (const Chunk &ch : q->q[b]) per_bin[b] += ch.count;
        }
    return MATINV_OK;
}

int matinv_queue_bins(const matinv_queue *q, int *bins, int cap)
{
    if (!q) return MATINV_ERR_ARG;
    for (int i = 0; i < cap && i < (int)q->bins.size(); ++i) bins[i] = q->bins[i];
    return (int)q->bins.size();
}

// Launch plan of one group (one n inside one bin)
struct Launch {
    int bin, n;
    size_t items;
    const char *a, *B, *c, *d, *e;  // where the kernel reads (the chunks themselves, or staging)
    char *m_out, *v_out;            // where it writes (means + ticket, or staging)
    size_t in_seg0, in_segs;        // gather segments (indices into the table), 0 = zero copy
    size_t out_seg0, out_segs;      // scatter segments of the results
    char *staging;                  // to free after the launch (stream ordered)
};

int matinv_queue_flush(matinv_queue *q, void *dMeans, void *dVariances, void *stream_)
{
    if (!q) return MATINV_ERR_ARG;
    if (q->tickets == 0) return MATINV_OK;
    if (!dMeans) return qfail(q, MATINV_ERR_ARG, "matinv_queue_flush: null result pointer");
    if (dVariances && !q->all_e) return qfail(q, MATINV_ERR_ARG, "matinv_queue_flush: variances asked for, but some items carry no e");
    hipStream_t user = static_cast<hipStream_t>(stream_);
    const size_t esz = q->esz;
    const bool want_var = dVariances != nullptr;
    std::vector<Seg> tab;
    std::vector<Launch> plan;
    hipError_t e = hipSuccess;
    int rc = MATINV_OK;
    // The two forms of a flush (see matinv_queue_create): serial, everything in the queue's own stream when that is the stream the flush
    // is issued on; or forked -- two streams, not one per bin: the largest pending bin (a latency-bound chain of many small launches)
    // beside everything else. (Measured on configs[4]'s mix, r02: one stream per bin 2.05 ms per step -- the cross-queue dependencies of
    // four forks and four joins cost more than the overlap buys; 2.8 ms with 8 hardware queues -- two streams 1.3-1.6 ms.)
    int top = -1;
    for (int b = (int)q->bins.size() - 1; b >= 0 && top < 0; --b)
        if (!q->q[b].empty()) top = b;
    const bool serial = user != nullptr && user == q->streams[0];
    if (!serial && q->streams.size() < 2) {  // the first forking flush: the chain stream
        hipStream_t s2 = nullptr;
        hipEvent_t d2 = nullptr;
        e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d2, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (s2) (void)hipStreamDestroy(s2);
            return qfail(q, MATINV_ERR_HIP, "matinv_queue_flush: second stream", e);
        }
        q->streams.push_back(s2);
        q->done.push_back(d2);
    }
    const std::vector<hipStream_t> &S = q->streams;
    auto slot_of = [&](int b) { return (serial || b != top) ? 0 : 1; };
    auto stream_of = [&](int b) { return S[slot_of(b)]; };

    // ---- plan (host, O(chunks))
    for (int b = (int)q->bins.size() - 1; b >= 0 && e == hipSuccess; --b) {
        std::vector<Chunk> &chunks = q->q[b];
        if (chunks.empty()) continue;
        std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk &x, const Chunk &y) {
            return x.n != y.n ? x.n > y.n : x.B < y.B;
        });
        for (size_t i = 0; i < chunks.size();) {
            size_t j = i;
            const int n = chunks[i].n;
            size_t items = 0;
            bool one_run = true, tickets_in_order = true;
            for (; j < chunks.size() && chunks[j].n == n; ++j) {
                if (j > i) {
                    const Chunk &p = chunks[j - 1], &c = chunks[j];
                    const size_t vb = p.count * n * esz;
                    one_run = one_run && c.a == p.a + vb && c.B == p.B + vb * n && c.c == p.c + vb && c.d == p.d + vb &&
                              (!want_var || c.e == p.e + p.count * esz);
                    tickets_in_order = tickets_in_order && c.ticket == p.ticket + p.count;
                }
                items += chunks[j].count;
            }
            Launch L{b, n, items, chunks[i].a, chunks[i].B, chunks[i].c, chunks[i].d, chunks[i].e,
                     static_cast<char *>(dMeans) + chunks[i].ticket * esz,
                     want_var ? static_cast<char *>(dVariances) + chunks[i].ticket * esz : nullptr, 0, 0, 0, 0, nullptr};
            const size_t vec = items * n * esz, mat = vec * n, sc = items * esz;
            const size_t in_bytes = one_run ? 0 : 3 * vec + mat + (want_var ? sc : 0);
            const size_t out_bytes = tickets_in_order ? 0 : sc * (want_var ? 2 : 1);
            if (in_bytes + out_bytes) {
                // 256-byte aligned sub-buffers of one stream-ordered allocation
                auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
                const size_t total = one_run ? al(sc) * 2 : al(vec) * 3 + al(mat) + al(sc) * 3;
                e = scratch_alloc(reinterpret_cast<void **>(&L.staging), total, stream_of(b));
                if (e != hipSuccess) break;
                char *p = L.staging;
                if (!one_run) {
                    char *sa = p, *sB = sa + al(vec), *sc_ = sB + al(mat), *sd = sc_ + al(vec), *se = sd + al(vec);
                    p = se + al(sc);
                    L.in_seg0 = tab.size();
                    size_t off = 0;
                    for (size_t k = i; k < j; ++k) {
                        const Chunk &c = chunks[k];
                        const size_t vb = c.count * n * esz;
                        tab.push_back(Seg{c.a, sa + off, vb});
                        tab.push_back(Seg{c.B, sB + off * n, vb * n});
                        tab.push_back(Seg{c.c, sc_ + off, vb});
                        tab.push_back(Seg{c.d, sd + off, vb});
                        if (want_var) tab.push_back(Seg{c.e, se + off / n, c.count * esz});
                        off += vb;
                    }
                    L.in_segs = tab.size() - L.in_seg0;
                    L.a = sa, L.B = sB, L.c = sc_, L.d = sd, L.e = want_var ? se : nullptr;
                }
                if (!tickets_in_order) {
                    char *sm = p, *sv = p + al(sc);
                    L.out_seg0 = tab.size();
                    size_t off = 0;
                    for (size_t k = i; k < j; ++k) {
                        const Chunk &c = chunks[k];
                        tab.push_back(Seg{sm + off, static_cast<char *>(dMeans) + c.ticket * esz, c.count * esz});
                        if (want_var) tab.push_back(Seg{sv + off, static_cast<char *>(dVariances) + c.ticket * esz, c.count * esz});
                        off += c.count * esz;
                    }
                    L.out_segs = tab.size() - L.out_seg0;
                    L.m_out = sm, L.v_out = want_var ? sv : nullptr;
                }
            }
            plan.push_back(L);
            i = j;
        }
    }

    // ---- segment tables: one upload per flush (pinned staging owned by the queue)
    Seg *dev_tab = nullptr;
    if (e == hipSuccess && !tab.empty()) {
        if (q->host_cap < tab.size()) {
            (void)hipEventSynchronize(q->tables_uploaded);
            if (q->host_tab) (void)hipHostFree(q->host_tab);
            q->host_cap = tab.size() * 2;
            e = hipHostMalloc(reinterpret_cast<void **>(&q->host_tab), q->host_cap * sizeof(Seg), hipHostMallocDefault);
            if (e != hipSuccess) q->host_tab = nullptr, q->host_cap = 0;
        } else {
            (void)hipEventSynchronize(q->tables_uploaded);  // the previous flush's upload has long finished
        }
        if (e == hipSuccess) {
            memcpy(q->host_tab, tab.data(), tab.size() * sizeof(Seg));
            e = scratch_alloc(reinterpret_cast<void **>(&dev_tab), tab.size() * sizeof(Seg), user);
            if (e == hipSuccess) e = hipMemcpyAsync(dev_tab, q->host_tab, tab.size() * sizeof(Seg), hipMemcpyHostToDevice, user);
            if (e == hipSuccess) e = hipEventRecord(q->tables_uploaded, user);
        }
    }

    // ---- launches: every bin on its own stream, forked from and joined back into the caller's stream
    if (e == hipSuccess) e = hipEventRecord(q->fork, user);
    bool forked[2] = {false, false};
    for (size_t li = 0; li < plan.size() && e == hipSuccess && rc == MATINV_OK; ++li) {
        const Launch &L = plan[li];
        hipStream_t s = stream_of(L.bin);
        if (!forked[slot_of(L.bin)]) {
            if (s != user) e = hipStreamWaitEvent(s, q->fork, 0);
            forked[slot_of(L.bin)] = true;
            if (e != hipSuccess) break;
        }
        // gridDim.y = segments of one launch: at most 65 535 (thousands of one-item submits of one n reach that), so a long
        // table goes out in several launches
        auto segcopy = [&](size_t seg0, size_t segs, size_t typical_bytes) {
            unsigned slices = (unsigned)std::min<size_t>(64, std::max<size_t>(1, typical_bytes / (256 * 16 * 4)));
            hipError_t es = hipSuccess;
            for (size_t done = 0; done < segs && es == hipSuccess; done += 65535) {
                const unsigned part = (unsigned)std::min<size_t>(65535, segs - done);
                hipLaunchKernelGGL(matinv_segcopy, dim3(slices, part), dim3(256), 0, s, dev_tab + seg0 + done);
                es = hipGetLastError();
            }
            return es;
        };
        if (L.in_segs) e = segcopy(L.in_seg0, L.in_segs, L.items * L.n * L.n * esz / std::max<size_t>(1, L.in_segs / 4));
        if (e != hipSuccess) break;
        rc = matinv_mean_batched(q->dtype, L.n, L.a, L.B, L.c, L.d, L.m_out, L.items, nullptr, s);
        if (rc == MATINV_OK && want_var) rc = matinv_variance_batched(q->dtype, L.n, L.a, L.B, L.c, L.e, L.v_out, L.items, nullptr, s);
        if (rc != MATINV_OK) {
            snprintf(q->last_error, sizeof q->last_error, "%s", matinv_last_error());
            break;
        }
        if (L.out_segs) e = segcopy(L.out_seg0, L.out_segs, 0);
        if (L.staging) {
            hipError_t ef = scratch_free(L.staging, s);
            plan[li].staging = nullptr;
            if (e == hipSuccess) e = ef;
        }
    }
    // a failed flush: the staging batches of the launches that never ran (or failed half way) go back to the pool too
    for (Launch &L : plan)
        if (L.staging) (void)scratch_free(L.staging, stream_of(L.bin)), L.staging = nullptr;
    for (int slot = 0; slot < 2; ++slot) {
        if (!forked[slot] || S[slot] == user) continue;
        hipError_t e2 = hipEventRecord(q->done[slot], S[slot]);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(user, q->done[slot], 0);
        if (e == hipSuccess) e = e2;
    }
    for (size_t b = 0; b < q->bins.size(); ++b) q->q[b].clear();
    if (dev_tab) {
        hipError_t e2 = scratch_free(dev_tab, user);  // after the join: every launch that reads it has been ordered before
        if (e == hipSuccess) e = e2;
    }
    q->tickets = 0;
    q->any_e = false, q->all_e = true;
    if (rc != MATINV_OK) return rc;
    if (e != hipSuccess) return qfail(q, MATINV_ERR_HIP, "matinv_queue_flush", e);
    return MATINV_OK;
}

}  // extern "C"
