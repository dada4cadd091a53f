// Prints the C/D lane->(row,col) map of v_mfma_f32_16x16x4_f32 and v_mfma_f64_16x16x4_f64 (used to derive the tile
// layouts of tile_kernels.hip). D[i][j] = (i+1)*(100+j) identifies the element each lane/register holds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(float *of, double *od)
{
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    float a = (kk == 0) ? (float)(i + 1) : 0.f, b = (kk == 0) ? (float)(100 + i) : 0.f;
    v4f cf = {0, 0, 0, 0};
    cf = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, cf, 0, 0, 0);
    v4d cd = {0, 0, 0, 0};
    cd = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, (double)b, cd, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { of[l * 4 + r] = cf[r]; od[l * 4 + r] = cd[r]; }
}
int main()
{
    float *df; double *dd; float hf[256]; double hd[256];
    hipMalloc(&df, sizeof hf); hipMalloc(&dd, sizeof hd);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, df, dd);
    hipMemcpy(hf, df, sizeof hf, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int okf = 1, okd = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            int q = l >> 4, c = l & 15;
            // hypotheses: f32 row = 4q + r ; f64 row = 4r + q ; col = c
            if (hf[l * 4 + r] != (float)((4 * q + r + 1) * (100 + c))) okf = 0;
            if (hd[l * 4 + r] != (double)((4 * r + q + 1) * (100 + c))) okd = 0;
        }
    printf("f32 16x16x4: row = 4*(lane>>4) + reg, col = lane&15 : %s\n", okf ? "CONFIRMED" : "WRONG");
    printf("f64 16x16x4: row = 4*reg + (lane>>4), col = lane&15 : %s\n", okd ? "CONFIRMED" : "WRONG");
    if (!okf) for (int l = 0; l < 64; l += 5) printf("lane %d: %g %g %g %g\n", l, hf[4*l], hf[4*l+1], hf[4*l+2], hf[4*l+3]);
    return !(okf && okd);
}
