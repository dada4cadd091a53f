/*
 * device_table_test.c -- calls the reference's DEVICE-TABLE entry point exactly as its only caller does
 * (/root/reference/src/gauss_bench.cu:68-78 batchedInverse -> inverse_lu_cuda_batched_device, tables from
 * batchedCudaMalloc :160-167, cudaMemcpy2D staging :165-170,244), in plain C over libmatinv_hip.so:
 *
 *   device_table_test N BATCH      matrices that NEED row exchanges but are well conditioned (a diagonally dominant
 *                                  matrix with its rows rotated by 1 + k mod (N-1)), checked against the host LU inverse
 *
 * prints "device_table_test n=.. batch=.. pitch=.. max_abs_err=.." and exits 0 when the error is within the tolerance.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/types.h"
#include "../../include/helper_cpu.h"
#include "../../include/helper_gpu.h"
#include "../../include/inverse_cpu.h"
#include "../../include/inverse_gpu.h"

#ifdef MATINV_DATATYPE_FLOAT
#define DEV_LU inverse_lu_cuda_batched_device_f32
#define TOL 1e-4
#else
#define DEV_LU inverse_lu_cuda_batched_device
#define TOL 1e-12
#endif

int main(int argc, char const *argv[])
{
    ensure(argc >= 3, "Usage: device_table_test N BATCH");
    const int n = atoi(argv[1]), batchSize = atoi(argv[2]);
    ensure(n >= 1 && batchSize >= 1, "N and BATCH must be positive");
    const size_t sizeOfMatrix = sizeof(DataType) * (size_t)n * n;
    Array As = (Array)malloc(sizeOfMatrix * batchSize), Invs = (Array)malloc(sizeOfMatrix * batchSize);
    Array Ref = (Array)malloc(sizeOfMatrix * batchSize);
    ensure(As && Invs && Ref, "out of memory");
    unsigned long long s = 0x5EEDull;
    for (int k = 0; k < batchSize; ++k) {
        const int shift = n > 1 ? 1 + k % (n - 1) : 0;
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const double u = (double)(s >> 11) / 9007199254740992.0;
                /* element (r, c) of the rotated matrix = element ((r + shift) mod n, c) of R + n I */
                const size_t i = (size_t)k * n * n + (size_t)c * n + r;
                As[i] = (DataType)(u + (((r + shift) % n) == c ? (double)n : 0.0));
                Ref[i] = As[i];
            }
    }
    /* the caller side of src/gauss_bench.cu:160-170 */
    Array *devAs = (Array *)malloc(sizeof(Array) * batchSize), *devAInvs = (Array *)malloc(sizeof(Array) * batchSize);
    size_t pitchAs, pitchAInvs;
    gpuErrchk(batchedCudaMalloc(devAs, &pitchAs, sizeOfMatrix, batchSize));
    gpuErrchk(batchedCudaMalloc(devAInvs, &pitchAInvs, sizeOfMatrix, batchSize));
    gpuErrchk(matinv_memcpy_2d(devAs[0], pitchAs, As, sizeOfMatrix, sizeOfMatrix, batchSize, 1));
    DEV_LU(NULL, n, devAs, devAInvs, batchSize); /* asynchronous; the blocking copy below waits (gauss_bench.cu:244) */
    gpuErrchk(matinv_memcpy_2d(Invs, sizeOfMatrix, devAInvs[0], pitchAInvs, sizeOfMatrix, batchSize, 0));
    gpuErrchk(matinv_batched_free((void **)devAs));
    gpuErrchk(matinv_batched_free((void **)devAInvs));

    inverse_lu_blas_omp(Ref, n, batchSize); /* host LU with partial pivoting, in place (src/inverse.c:74-107) */
    double err = 0;
    for (size_t i = 0; i < (size_t)n * n * batchSize; ++i) {
        const double d = fabs((double)Invs[i] - (double)Ref[i]);
        if (!(d <= err)) err = d;
    }
    printf("device_table_test n=%d batch=%d pitch=%zu max_abs_err=%.3e\n", n, batchSize, pitchAs, err);
    free(As), free(Invs), free(Ref), free(devAs), free(devAInvs);
    return err < TOL ? 0 : 1;
}
