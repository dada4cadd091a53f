// Cycles per wave-instruction (one wave per SIMD, independent accumulators) of the fp64 building blocks of the
// rowlane kernel: v_fmac_f64 (plain), v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp, 2 x v_mov_b32_dpp, v_mul_f64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(double *out, long long *cyc, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    double m = 1e-9 * seed;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {
            REP16(asm volatile("v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %9\n\tv_fmac_f64_e32 %2, %8, %9\n\tv_fmac_f64_e32 %3, %8, %9\n\t"
                               "v_fmac_f64_e32 %4, %8, %9\n\tv_fmac_f64_e32 %5, %8, %9\n\tv_fmac_f64_e32 %6, %8, %9\n\tv_fmac_f64_e32 %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(m));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_fmac_f64_dpp %0, %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %2, %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %4, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %6, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 2) {
            REP16(asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b64_dpp %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b64_dpp %5, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {
            REP16(asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                               "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else {
            int *p0 = (int *)&a0, *p1 = (int *)&a1;
            REP16(asm volatile("v_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                               : "+v"(p0[0]), "+v"(p0[1]), "+v"(p1[0]), "+v"(p1[1]));)
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, int waves_per_block)
{
    double *o; long long *c; long long h[1024];
    // 1024 blocks x up to 1024 threads of output, one cycle count per block
    if (hipMalloc(&o, (size_t)1024 * 1024 * 8) != hipSuccess || hipMalloc(&c, sizeof h) != hipSuccess) { printf("alloc failed\n"); return; }
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64 * waves_per_block), 0, 0, o, c, 1.0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64 * waves_per_block), 0, 0, o, c, 1.0);
    hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    // 64 iterations x 16 reps x 8 instructions per wave
    printf("%-28s %d wave(s)/SIMD: %.2f cycles per wave-instruction (per wave)\n", name, waves_per_block / 4 ? waves_per_block / 4 : 1, s / 1024 / (64.0 * 16 * 8));
    hipFree(o); hipFree(c);
}
int main()
{
    for (int w : {4, 8, 16}) {
        run<0>("v_fmac_f64", w); run<1>("v_fmac_f64_dpp newbcast", w); run<2>("v_mov_b64_dpp newbcast", w);
        run<3>("v_mul_f64", w); run<4>("v_mov_b32_dpp newbcast", w);
    }
    return 0;
}
