// How many independent VALU instructions fit in the shadow of one v_mfma_f64_16x16x4_f64 of the SAME wave on gfx950?
// One or two waves per SIMD, each running { MFMA ; K x v_fmac_f64 } back to back; prints cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_same_wave.hip -o build/mfma_valu_same_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 256, ITERS = 2048;

#define F1 "v_fmac_f64_e32 %0, %8, %8\n\t"
#define F2 F1 "v_fmac_f64_e32 %1, %8, %8\n\t"
#define F4 F2 "v_fmac_f64_e32 %2, %8, %8\n\tv_fmac_f64_e32 %3, %8, %8\n\t"
#define F8 F4 "v_fmac_f64_e32 %4, %8, %8\n\tv_fmac_f64_e32 %5, %8, %8\n\tv_fmac_f64_e32 %6, %8, %8\n\tv_fmac_f64_e32 %7, %8, %8\n\t"
#define VALU(STR) asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));

template <int K>
__global__ __launch_bounds__(1024) void k(double *out, long long *cyc, double seed)
{
    __shared__ double pad[12 * 1024];  // 96 KB: one workgroup per CU
    if (threadIdx.x == 0) pad[0] = seed;
    __syncthreads();
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = seed + threadIdx.x, b = 1e-9 * a, m = 1e-9 * seed;
    double a0 = a, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#define ONE(C)                                                       \
    C = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C, 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                               \
    if (K == 4) { VALU(F4) }                                         \
    if (K == 8) { VALU(F8) }                                         \
    if (K == 12) { VALU(F8) VALU(F4) }                               \
    if (K == 16) { VALU(F8) VALU(F8) }                               \
    if (K == 24) { VALU(F8) VALU(F8) VALU(F8) }                      \
    __builtin_amdgcn_sched_barrier(0);
        ONE(c0) ONE(c1) ONE(c2) ONE(c3)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + pad[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K>
void run(int waves)
{
    static double *o = nullptr;
    static long long *c = nullptr;
    static long long h[BLOCKS * 16];
    if (!o && (hipMalloc(&o, (size_t)BLOCKS * 1024 * 8) != hipSuccess || hipMalloc(&c, sizeof h) != hipSuccess)) { printf("alloc failed\n"); return; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<K>, dim3(BLOCKS), dim3(64 * waves), 0, 0, o, c, 1.0);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k<K>, dim3(BLOCKS), dim3(64 * waves), 0, 0, o, c, 1.0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = 10.0 * BLOCKS * waves * (ITERS * 4.0) * 2048 / (ms * 1e-3) / 1e12;
    if (hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
    double s = 0;
    for (int b = 0; b < BLOCKS; ++b)
        for (int w = 0; w < waves; ++w) s += h[b * 16 + w];
    printf("K=%2d VALU per MFMA, %d wave(s)/SIMD: %7.1f cycles per MFMA per wave  (SIMD: %6.1f per MFMA)   wall: %.3f ms per launch = %6.1f TFLOP/s fp64 MFMA\n",
           K, waves / 4, s / BLOCKS / waves / (ITERS * 4.0), s / BLOCKS / waves / (ITERS * 4.0) / (waves / 4), ms / 10, tflops);
}

int main()
{
    for (int waves : {4, 8, 12, 16}) {
        run<0>(waves); run<4>(waves); run<8>(waves); run<12>(waves); run<16>(waves); run<24>(waves);
    }
    return 0;
}
