/*
 * inverse_gpu.h -- THE DROP-IN BOUNDARY. Same 17 extern "C" names and argument
 * lists as /root/reference/include/inverse_gpu.h:7-31, implemented by
 * libmatinv_hip.so on MI355X (no CUDA, no cuBLAS).
 *
 * Include after types.h. `cublasHandle_t` is only carried through for source
 * compatibility: every hand-written kernel of the reference ignores it
 * (src/gauss/batched_invert.cu:99-104) and so does this library; pass NULL.
 * Define MATINV_HAVE_CUBLAS_HANDLE before including if your translation unit
 * already has a cublasHandle_t typedef.
 *
 * Two families, as in the reference:
 *   *_batched_gpu(handle, n, As, aInvs, batchSize)       host pointers, synchronous
 *       (called at src/inverse_bench.c:144,168,191,214). As is batchSize*n*n contiguous scalars, each matrix
 *       column-major; aInvs is caller-allocated and fully overwritten. UNLIKE the reference's Cholesky entry
 *       points (src/inverse_cholesky_gpu.cu:442,672,747) `As` is never written.
 *   *_batched_device(handle, n, devAs, devAInvs, batchSize)   HOST-resident tables of DEVICE pointers,
 *       asynchronous on the default stream (called at src/gauss_bench.cu:77). devAs[i] need not be equally
 *       spaced. The input batch is left intact.
 * Errors keep the reference's contract (include/helper_gpu.h:9-18, include/helper_cpu.h:12-21): message on
 * stderr and exit(EXIT_FAILURE). Singular / non-SPD input does not abort: the affected output matrix is
 * filled with NaN (the reference leaves it undefined, src/gauss/batched_invert.cu:29-31).
 *
 * Which native kernel serves which name:
 *   inverse_gauss_*, inverse_lu_cuda_*            -> Gauss-Jordan with partial pivoting (MATINV_ALGO_GAUSS_JORDAN)
 *   inverse_cholesky_{,mm_,mm2_,stride_}*         -> Cholesky inverse (MATINV_ALGO_CHOLESKY)
 *   decompose_cholesky_{,mm_,stride_}batched_device, inverse_upper_stride_*, multiply_upper_stride_*
 *       sub-phase entry points that neither reference CLI calls; see the .hip file for what each returns.
 */
#ifndef HEADER_INVERSE_GPU_INCLUDED
#define HEADER_INVERSE_GPU_INCLUDED

#ifndef MATINV_HAVE_CUBLAS_HANDLE
#define MATINV_HAVE_CUBLAS_HANDLE
typedef void *cublasHandle_t;
#endif

#ifdef MATINV_DATATYPE_FLOAT
#define MATINV_REFNAME(x) x##_f32
#define inverse_gauss_batched_gpu inverse_gauss_batched_gpu_f32
#define inverse_lu_cuda_batched_gpu inverse_lu_cuda_batched_gpu_f32
#define inverse_gauss_batched_device inverse_gauss_batched_device_f32
#define inverse_lu_cuda_batched_device inverse_lu_cuda_batched_device_f32
#define inverse_cholesky_stride_batched_gpu inverse_cholesky_stride_batched_gpu_f32
#define inverse_cholesky_stride_batched_device inverse_cholesky_stride_batched_device_f32
#define decompose_cholesky_stride_batched_device decompose_cholesky_stride_batched_device_f32
#define inverse_upper_stride_batched_device inverse_upper_stride_batched_device_f32
#define multiply_upper_stride_batched_device multiply_upper_stride_batched_device_f32
#define inverse_cholesky_batched_device inverse_cholesky_batched_device_f32
#define decompose_cholesky_batched_device decompose_cholesky_batched_device_f32
#define inverse_cholesky_mm_batched_device inverse_cholesky_mm_batched_device_f32
#define decompose_cholesky_mm_batched_device decompose_cholesky_mm_batched_device_f32
#define inverse_cholesky_batched_gpu inverse_cholesky_batched_gpu_f32
#define inverse_cholesky_mm_batched_gpu inverse_cholesky_mm_batched_gpu_f32
#define inverse_cholesky_mm2_batched_device inverse_cholesky_mm2_batched_device_f32
#define inverse_cholesky_mm2_batched_gpu inverse_cholesky_mm2_batched_gpu_f32
#endif

#ifdef __cplusplus
extern "C" {
#endif // __cplusplus
void inverse_gauss_batched_gpu(cublasHandle_t handle, int n, Array As, Array aInvs, int batchSize);
void inverse_lu_cuda_batched_gpu(cublasHandle_t handle, int n, Array As, Array aInvs, int batchSize);

void inverse_gauss_batched_device(cublasHandle_t handle, int n, Array *devAs, Array *devAInvs, int batchSize);
void inverse_lu_cuda_batched_device(cublasHandle_t handle, int n, Array *devAs, Array *devAInvs, int batchSize);

void inverse_cholesky_stride_batched_gpu(cublasHandle_t handle, int n, Array As, Array aInvs, int batchSize);

void inverse_cholesky_stride_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);
void decompose_cholesky_stride_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);
void inverse_upper_stride_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);
void multiply_upper_stride_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);

void inverse_cholesky_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);
void decompose_cholesky_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);

void inverse_cholesky_mm_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);
void decompose_cholesky_mm_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);

void inverse_cholesky_batched_gpu(cublasHandle_t handle, int n, Array As, Array aInvs, int batchSize);
void inverse_cholesky_mm_batched_gpu(cublasHandle_t handle, int n, Array As, Array aInvs, int batchSize);

void inverse_cholesky_mm2_batched_device(cublasHandle_t handle, int N, Array *devAs, Array *devAInvs, int batchSize);
void inverse_cholesky_mm2_batched_gpu(cublasHandle_t handle, int n, Array As, Array aInvs, int batchSize);

#ifdef __cplusplus
}
#endif // __cplusplus

#endif
