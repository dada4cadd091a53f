/*
 * matinv.h -- native C ABI of the MI355X batched small-matrix inversion engine
 * (libmatinv_hip.so). Plain pointers and sizes only; no C++/torch types.
 *
 * This is the device-resident, stream-aware core that the reference-named
 * entry points of inverse_gpu.h are thin wrappers over. A matrix is n x n,
 * column-major (element (r,c) at c*n + r), and matrix k of a batch starts at
 * base + k*stride elements (stride >= n*n; the reference uses a pitched
 * allocation, /root/reference/src/helper.cu:103-118, so a stride is needed to
 * accept its device tables without a copy).
 *
 * All functions return MATINV_OK (0) or a negative matinv_status and never
 * call exit(); matinv_last_error() describes the last failure of the calling
 * thread. The reference-named wrappers keep the reference's fatal behaviour
 * (include/helper_gpu.h:9-18 there) on top of this.
 */
#ifndef MATINV_H_INCLUDED
#define MATINV_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MATINV_OK = 0,
    MATINV_ERR_ARG = -1,         /* bad n / batch / pointer / stride                   */
    MATINV_ERR_UNSUPPORTED = -2, /* n > 1024, or the forced kernel family cannot serve it */
    MATINV_ERR_HIP = -3,         /* a HIP runtime call failed (message in last_error)  */
    MATINV_ERR_NO_DEVICE = -4    /* no gfx950 device visible                           */
} matinv_status;

typedef enum {
    MATINV_F64 = 0,
    MATINV_F32 = 1
} matinv_dtype;

typedef enum {
    MATINV_ALGO_GAUSS_JORDAN = 0, /* general matrices, partial pivoting. Replaces pivotRow/normalizeRow/
                                     transform_matrix, src/gauss/batched_invert.cu:17-95, and the cuBLAS LU
                                     path src/gauss/inverse_gpu.cu:16-58                                   */
    MATINV_ALGO_CHOLESKY = 1      /* SPD matrices (only the lower triangle is read), full symmetric result.
                                     Replaces the four Cholesky families of src/inverse_cholesky_gpu.cu. LDS family:
                                     literal L L^T, L^-1, L^-T L^-1; tile family: the square-root-free symmetric
                                     blocked sweep (same Schur complements, pivots = squares of diag(L))       */
} matinv_algo;

/* Kernel families; MATINV_KERNEL_AUTO picks by (algo, dtype, n). The others force one family and
 * fail with MATINV_ERR_UNSUPPORTED when it cannot handle the request (used by tests and bench). */
typedef enum {
    MATINV_KERNEL_AUTO = 0,
    MATINV_KERNEL_LDS = 1,     /* one workgroup per matrix, matrix resident in LDS, any n up to the LDS limit */
    MATINV_KERNEL_ROWLANE = 2, /* n <= 16: 64/npad matrices per wavefront, one row per lane, DPP broadcasts     */
    MATINV_KERNEL_TILE = 3     /* 16x16 MFMA accumulator tiles, f64 and f32: blocked Gauss-Jordan with one wavefront per matrix
                                  (n <= 64) or four (64 < n <= 128); for MATINV_ALGO_CHOLESKY the symmetric blocked sweep on
                                  lower-triangular tiles (n <= 64) / the four-wave sweep with positivity-checked pivots */,
    MATINV_KERNEL_ROW = 4,     /* n <= 64: one matrix per wavefront, row per lane, classical partial pivoting with the pivot
                                  row broadcast through v_readlane; the pivoting path behind the tile family */
    MATINV_KERNEL_GLOBAL = 5,  /* any n <= 1024 (the reference's limit): one 1024-thread workgroup per matrix, working copy
                                  in global memory; the functional path for matrices that do not fit on chip */
    MATINV_KERNEL_TILEP = 7,   /* the MFMA tile Gauss-Jordan with TRUE partial pivoting inside the kernel (pivot search on the
                                  LDS-staged panel, one row per lane; rows never move, the permutation is folded into the store
                                  addresses): general matrices at MFMA speed. One wavefront per matrix (n <= 64), four
                                  (n <= 128), one per tile column (n <= 192 in f64, 256 in f32; automatic there) */
    MATINV_KERNEL_BLOCKED = 6  /* any n <= 1024, global-memory working copies, two launches per panel over the whole batch.
                                  MATINV_ALGO_CHOLESKY: blocked right-looking Cholesky, A^-1 = L^-T L^-1 as one symmetric product
                                  (automatic beyond n = 128). MATINV_ALGO_GAUSS_JORDAN: blocked Gauss-Jordan with partial
                                  pivoting, panel of 32 columns eliminated with one thread per row (automatic beyond the LDS limit) */
} matinv_kernel;

/* Gauss-Jordan, 16 < n <= 192 (f64) / 256 (f32): which MFMA tile kernel a launch uses (process-wide, atomic; default
 * MATINV_GJ_NATURAL_FIRST, or the environment's MATINV_GJ_POLICY=natural|pivot|adaptive read at the first launch).
 *   NATURAL_FIRST  the natural-order kernel (pivots verified, not searched: the fast path of dominant / SPD batches); a
 *                  matrix it rejects is redone by the pivoting kernel in the same stream. The result of a matrix depends
 *                  on that matrix alone: repeatable bit for bit, identical for a sharded and an unsharded batch.
 *   PIVOT          straight to the kernel with partial pivoting inside the MFMA sweep (general matrices at full speed; also
 *                  per-matrix deterministic). The reference's LU names (inverse_lu_cuda_batched_*) always take it.
 *   ADAPTIVE       chooses per launch from the reject count of the last completed natural-order launch of that size on
 *                  that device: fastest on streams of like batches, but a matrix that both kernels accept may get either
 *                  kernel's bits depending on what ran before. */
typedef enum {
    MATINV_GJ_NATURAL_FIRST = 0,
    MATINV_GJ_PIVOT = 1,
    MATINV_GJ_ADAPTIVE = 2
} matinv_gj_policy;
int matinv_set_gj_policy(int policy); /* returns the previous policy, or MATINV_ERR_ARG */

/* THREAD SAFETY. Every entry point may be called concurrently from several host threads, on the same or on different devices
 * (the device is the calling thread's current HIP device; matinv_last_error() is per thread). The library keeps no state
 * between calls except: the GJ policy above (atomic), the ADAPTIVE policy's per-device counters (atomic), and the staging
 * memory cached per (device, stream) inside the library (mutex-protected; matinv_release_cache). A matinv_queue is NOT internally locked: one
 * thread at a time per queue. Launches on one stream execute in issue order, as HIP defines. */

/* Invert `batch` matrices that are already resident in device memory.
 *   dA      in : batch matrices, matrix k at dA + k*strideA (elements). Never written.
 *   dAinv   out: matrix k at dAinv + k*strideInv. May be exactly dA with strideInv == strideA (in place:
 *                every kernel reads a whole matrix before writing it); partial overlap is undefined.
 *   dInfo   out: optional int[batch] (device). 0 = ok; k+1 = no usable pivot at elimination step k
 *                (Gauss-Jordan: singular) or leading minor k+1 not positive (Cholesky: not SPD).
 *                For info != 0 the output matrix is filled with NaN (never left half-written).
 *   stream     : hipStream_t as void* (NULL = default stream). Asynchronous.
 */
int matinv_inverse_batched(int algo, int dtype, int n, const void *dA, size_t strideA, void *dAinv,
                           size_t strideInv, size_t batch, int *dInfo, void *stream);

/* Same, forcing a kernel family (matinv_kernel). */
int matinv_inverse_batched_ex(int algo, int dtype, int n, const void *dA, size_t strideA, void *dAinv,
                              size_t strideInv, size_t batch, int *dInfo, void *stream, int kernel);

/* Which family MATINV_KERNEL_AUTO resolves to (a matinv_kernel), or a negative status. */
int matinv_select_kernel(int algo, int dtype, int n);

/* Name of the __global__ function a (algo, dtype, n, kernel) request launches -- the name rocprofv3 reports. */
const char *matinv_kernel_name(int algo, int dtype, int n, int kernel);

/* Fused Gaussian-process pipeline, device-resident (replaces calcluateMean / calcluateVariance,
 * src/gauss_bench.cu:127-265,275-409: addDiagonal + batched inverse + two gemmBatched):
 *   means[k] = a_k^T (B_k + diag(c_k))^-1 d_k
 *   vars[k]  = e_k - a_k^T (B_k + diag(c_k))^-1 a_k
 * a,c,d: batch*n; B: batch*n*n (SPD, column-major, stride n*n); e, out: batch scalars. No input is modified
 * and the inverse is never written to memory.
 */
int matinv_mean_batched(int dtype, int n, const void *dAs, const void *dBs, const void *dCs, const void *dDs,
                        void *dMeans, size_t batch, int *dInfo, void *stream);
int matinv_variance_batched(int dtype, int n, const void *dAs, const void *dBs, const void *dCs, const void *dEs,
                            void *dVars, size_t batch, int *dInfo, void *stream);

/* Host-pointer convenience used by the reference-named *_gpu wrappers: allocate, H2D, invert, D2H, free.
 * `info` is an optional host int[batch]. Synchronous. */
int matinv_inverse_batched_host(int algo, int dtype, int n, const void *hA, void *hAinv, size_t batch, int *info);

/* The same over SEVERAL devices of this process (one node): the batch is cut into `nshards` contiguous blocks of
 * ceil(batch / nshards) matrices (rounded up to the per-wavefront packing of the small-n kernels: 8 for n <= 8, 4 for n <= 16),
 * shard g runs on device g mod (number of gfx950 devices) on its own host thread with its own streams and its own host link
 * -- no collective: the result goes back to host memory. nshards <= 0: one shard per visible device. More shards than
 * devices is allowed ("virtual shards": several per device). The bits of every matrix are those of the single-device call.
 * The reference-named *_batched_gpu entry points take this path with nshards = MATINV_DEVICES when that variable is > 1. */
int matinv_inverse_batched_host_multi(int algo, int dtype, int n, const void *hA, void *hAinv, size_t batch, int *info,
                                      int nshards);
int matinv_device_count(void); /* visible gfx950 devices, or a negative status */
/* The block of shard g of nshards: [*lo, *hi) (pure host arithmetic, no device needed; the same partition as shard.py). */
int matinv_shard_range(size_t batch, int nshards, int n, int g, size_t *lo, size_t *hi);

/* Reassembly of device-resident shards over RCCL (librccl is opened at the first call: no link-time dependency). Every rank
 * contributes `count` elements (pad the short tail shard) and receives nranks * count.
 *   one process per GPU: matinv_comm_unique_id on rank 0 -> ship the 128 bytes to the others by any means ->
 *                        matinv_comm_init_rank everywhere -> matinv_allgather_shards(comm, ...) on each rank's stream;
 *   one process, several GPUs: matinv_allgather_local(ndev, devices, ...) (communicators created once and cached). It runs on
 *                        the library's own streams and returns when the gather has completed. Input readiness:
 *                        matinv_allgather_local synchronises every listed device on entry (whatever was enqueued there before the
 *                        call has finished when the gather reads dSend[g]); matinv_allgather_local_after takes, per device, the
 *                        stream whose work so far produces dSend[g] (0 = that device's null stream) and makes the gather wait
 *                        for an event recorded on it instead -- no host-side synchronisation of the producers. */
int matinv_comm_unique_id(void *id128);
int matinv_comm_init_rank(void **comm, int nranks, const void *id128, int rank);
int matinv_comm_destroy(void *comm);
int matinv_allgather_shards(void *comm, int dtype, const void *dSend, void *dRecv, size_t count, void *stream);
int matinv_allgather_local(int ndev, const int *devices, int dtype, const void *const *dSend, void *const *dRecv, size_t count);
int matinv_allgather_local_after(int ndev, const int *devices, int dtype, const void *const *dSend, void *const *dRecv, size_t count,
                                 void *const *producer_streams);

/* Host-pointer form of the fused pipeline (what gauss_bench times): H2D, one kernel, D2H of `batch` scalars.
 * Inputs are NOT modified (the reference CPU path destroys Bs and Cs, include/gauss_cpu.h:42 there). Synchronous. */
int matinv_mean_batched_host(int dtype, int n, const void *hAs, const void *hBs, const void *hCs, const void *hDs,
                             void *hMeans, size_t batch, int *info);
int matinv_variance_batched_host(int dtype, int n, const void *hAs, const void *hBs, const void *hCs, const void *hEs,
                                 void *hVars, size_t batch, int *info);

/* The reference's batch allocator (batchedCudaMalloc, src/helper.cu:103-118; declared include/helper_gpu.h:4): ONE pitched
 * device allocation of batchSize rows of arraySize BYTES; devArrayPtr (host array of batchSize pointers, caller-owned)
 * receives the row addresses, *pitch the row pitch in bytes (a multiple of 256, so of sizeof(double)). The tables are what
 * the *_batched_device entry points of inverse_gpu.h take. matinv_batched_free releases the block behind devArrayPtr[0]. */
int matinv_batched_malloc(void **devArrayPtr, size_t *pitch, size_t arraySize, int batchSize);
int matinv_batched_free(void **devArrayPtr);
/* cudaMemcpy2D as the reference uses it on those blocks (src/gauss_bench.cu:165-170,244): width bytes x height rows,
 * blocking. toDevice != 0: host -> device, else device -> host. Lets a plain-C caller stage data without HIP headers. */
int matinv_memcpy_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, int toDevice);
int matinv_device_synchronize(void);
/* Counters of the tile family's Gauss-Jordan dispatch since load: launches that started with the natural-order kernel /
 * went straight to the pivoting kernel, and -- MATINV_GJ_ADAPTIVE only -- the reject count / batch of the last natural-order
 * launch whose count has come back (call matinv_device_synchronize first for an exact figure). Any pointer may be NULL. */
int matinv_tile_stats(unsigned long long *natural_launches, unsigned long long *pivot_launches, unsigned long long *last_rejected,
                      unsigned long long *last_batch);

/* Size-binned multi-queue for mixed-size pipeline items (the reference sketches it, README.md:41-44: "use multiple queues
 * for different sizes: 32, 128, 512, 1024"; BASELINE configs[4]). Items of any n <= the largest bin are submitted as
 * chunks of `count` equally sized items lying back to back in device memory (As, Cs, Ds: count*n; Bs: count*n*n
 * column-major; Es: count scalars or NULL); submit only records the chunk and hands out consecutive tickets. flush runs
 * the largest pending bin on one HIP stream and the other bins on a second (forked from and joined back into `stream`; or
 * everything in the queue's own stream: matinv_queue_stream), one launch of the fused mean (and variance) kernel per distinct n of a bin -- the kernels pad a matrix to their tile size in registers, nothing is padded in memory --
 * gathering a group's chunks with one segmented-copy kernel unless they already form one contiguous run, and writes
 * means[ticket] (and variances[ticket] when dVariances != NULL, which needs every item to carry e). Asynchronous; the
 * item memory must stay valid until the work in `stream` has completed. bins == NULL / nbins == 0: {32, 128, 512, 1024}. */
typedef struct matinv_queue matinv_queue;
int matinv_queue_create(matinv_queue **q, int dtype, const int *bins, int nbins);
int matinv_queue_submit(matinv_queue *q, int n, const void *dAs, const void *dBs, const void *dCs, const void *dDs,
                        const void *dEs, size_t count, size_t *first_ticket);
/* the same for `chunks` chunks in one call (arrays of `chunks` entries; dEs NULL = no item carries e) */
int matinv_queue_submit_chunks(matinv_queue *q, size_t chunks, const int *n, const void *const *dAs, const void *const *dBs,
                               const void *const *dCs, const void *const *dDs, const void *const *dEs, const size_t *count,
                               size_t *first_tickets);
int matinv_queue_pending(const matinv_queue *q, size_t *items, size_t *per_bin);
int matinv_queue_bins(const matinv_queue *q, int *bins, int cap);
int matinv_queue_flush(matinv_queue *q, void *dMeans, void *dVariances, void *stream);
/* The queue's own stream (a hipStream_t; non-blocking), created with the queue. A flush issued ON it (pass it as `stream`) runs in it from
 * end to end, largest bin first: no fork, no join -- nothing in such a flush waits for another stream. HIP multiplexes the streams of a
 * process onto four hardware queues (a new stream goes to the least used one): four queues created at start-up, before the process
 * creates other streams, and used alternately overlap their flushes completely (bench.py, mixed workload: 0.93 / 0.59 / 0.47 / 0.41 ms
 * per step with 1 / 2 / 3 / 4 flushes in flight). A flush issued on any OTHER stream (the null stream included) forks the launch chain
 * of its largest bin into a second stream of the queue beside the other bins and joins both back: the shorter latency for one flush by
 * itself (0.75 ms), and at the mercy of the process's other streams when several are in flight. The caller orders its own work after
 * a flush on the own stream with an event recorded on it. */
void *matinv_queue_stream(matinv_queue *q);
int matinv_queue_destroy(matinv_queue *q);
const char *matinv_queue_last_error(const matinv_queue *q);

const char *matinv_last_error(void);
int matinv_abi_version(void);

/* The host-pointer entry points (and the work lists / workspaces of the kernels) take their device scratch memory from a cache
 * inside the library, keyed by (device, stream) -- a block is reused on the stream it was first handed out on only, so
 * stream order makes the reuse safe -- and leave it there between calls: the reference's "allocate and free inside every
 * call" (src/gauss/batched_invert.cu:120-176) without its cost. The device's default memory pool is not touched. An
 * allocation that fails for lack of memory empties the cache and is tried once more. This call synchronises the current
 * device and hands everything that is not in use back to the driver.
 * Returns MATINV_OK, or MATINV_ERR_HIP / MATINV_ERR_NO_DEVICE. Never needed for correctness. */
int matinv_release_cache(void);
/* A caller that is about to destroy one of ITS streams on which it has called this library: synchronises the stream and hands the
 * scratch blocks cached for it (work lists, blocked-path workspaces: they are kept per (device, stream) and reused on that stream
 * only) to the other streams of its device. Without it they stay cached until matinv_release_cache() or memory runs out. */
int matinv_stream_retire(void *stream);

/* Test hook. With MATINV_DEBUG_REJECTS=1 in the environment when the library is loaded, every launcher whose first-pass kernel
 * hands rejected matrices (needs row exchanges / not positive definite / singular) to a second kernel through a work list reads
 * that list's length back after the launch (one stream synchronisation per launch: a test mode) and adds it to a running total.
 * Returns the total; reset != 0 also zeroes it. Without the environment variable: always 0. The results never show whether
 * the fast kernel or its fallback produced them -- the tests use this to pin "a well-conditioned batch is never rejected". */
long long matinv_debug_rejects(int reset);

#ifdef __cplusplus
}
#endif
#endif
