/*
 * helper_gpu.h -- the GPU-side helper of the reference that the hot path's callers use
 * (/root/reference/include/helper_gpu.h:4, src/helper.cu:103-118), over libmatinv_hip.so.
 *
 *   batchedCudaMalloc(devArrayPtr, &pitch, arraySize, batchSize)
 * keeps the reference's name, argument order and meaning (one pitched allocation, host table of row pointers;
 * returns 0 on success like cudaSuccess) so that src/gauss_bench.cu:160-167 and src/inverse_cholesky_gpu.cu:207-208
 * read unchanged. gpuErrchk keeps the reference's fatal behaviour (include/helper_gpu.h:9-18 there).
 */
#ifndef HEADER_HELPER_GPU_INCLUDED
#define HEADER_HELPER_GPU_INCLUDED

#include <stdio.h>
#include <stdlib.h>

#include "matinv.h"
#include "types.h"

static inline int batchedCudaMalloc(Array *devArrayPtr, size_t *pitch, size_t arraySize, int batchSize)
{
    return matinv_batched_malloc((void **)devArrayPtr, pitch, arraySize, batchSize);
}

#define gpuErrchk(ans)                                                                                   \
    do {                                                                                                 \
        int matinv_rc_ = (ans);                                                                          \
        if (matinv_rc_ != 0) {                                                                           \
            fprintf(stderr, "GPUassert: %s %s %d\n", matinv_last_error(), __FILE__, __LINE__);           \
            exit(matinv_rc_);                                                                            \
        }                                                                                                \
    } while (0)

#endif
