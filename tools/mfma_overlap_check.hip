// Does v_mfma_f32_16x16x4_f32 give the right answer on gfx950 when its destination PARTIALLY overlaps its C operand?
// (hipcc allocates such overlaps for 128-bit MFMA results -- 284 of them in the 9 x 9-tile fp32 symmetric sweep once the accumulators
// live in AGPRs.)  hipcc --offload-arch=gfx950 -O2 tools/mfma_overlap_check.hip -o /tmp/mfma_overlap_check && /tmp/mfma_overlap_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k(const float *a, const float *b, const float *c, float *out)
{
    const int l = threadIdx.x;
    const float av = a[l], bv = b[l];
    float c0 = c[4 * l], c1 = c[4 * l + 1], c2 = c[4 * l + 2], c3 = c[4 * l + 3];
    float r[6][4];
    // 0: VGPR, disjoint
    asm volatile("v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n v_mov_b32 v12, %6\n v_mov_b32 v13, %7\n s_nop 4\n"
                 "v_mfma_f32_16x16x4_f32 v[20:23], %8, %9, v[10:13]\n s_nop 15\n s_nop 15\n"
                 "v_mov_b32 %0, v20\n v_mov_b32 %1, v21\n v_mov_b32 %2, v22\n v_mov_b32 %3, v23\n"
                 : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[0][2]), "=&v"(r[0][3]) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(av), "v"(bv)
                 : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23");
    // 1: VGPR, dst = src - 2
    asm volatile("v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n v_mov_b32 v14, %6\n v_mov_b32 v15, %7\n s_nop 4\n"
                 "v_mfma_f32_16x16x4_f32 v[10:13], %8, %9, v[12:15]\n s_nop 15\n s_nop 15\n"
                 "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13\n"
                 : "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[1][2]), "=&v"(r[1][3]) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(av), "v"(bv)
                 : "v10", "v11", "v12", "v13", "v14", "v15");
    // 2: VGPR, dst = src + 2
    asm volatile("v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n v_mov_b32 v12, %6\n v_mov_b32 v13, %7\n s_nop 4\n"
                 "v_mfma_f32_16x16x4_f32 v[12:15], %8, %9, v[10:13]\n s_nop 15\n s_nop 15\n"
                 "v_mov_b32 %0, v12\n v_mov_b32 %1, v13\n v_mov_b32 %2, v14\n v_mov_b32 %3, v15\n"
                 : "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[2][2]), "=&v"(r[2][3]) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(av), "v"(bv)
                 : "v10", "v11", "v12", "v13", "v14", "v15");
    // 3: AGPR, disjoint
    asm volatile("v_accvgpr_write_b32 a10, %4\n v_accvgpr_write_b32 a11, %5\n v_accvgpr_write_b32 a12, %6\n v_accvgpr_write_b32 a13, %7\n s_nop 4\n"
                 "v_mfma_f32_16x16x4_f32 a[20:23], %8, %9, a[10:13]\n s_nop 15\n s_nop 15\n"
                 "v_accvgpr_read_b32 %0, a20\n v_accvgpr_read_b32 %1, a21\n v_accvgpr_read_b32 %2, a22\n v_accvgpr_read_b32 %3, a23\n"
                 : "=&v"(r[3][0]), "=&v"(r[3][1]), "=&v"(r[3][2]), "=&v"(r[3][3]) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(av), "v"(bv)
                 : "a10", "a11", "a12", "a13", "a20", "a21", "a22", "a23");
    // 4: AGPR, dst = src - 2 (the form in the kernel: a[2:5] <- a[4:7])
    asm volatile("v_accvgpr_write_b32 a12, %4\n v_accvgpr_write_b32 a13, %5\n v_accvgpr_write_b32 a14, %6\n v_accvgpr_write_b32 a15, %7\n s_nop 4\n"
                 "v_mfma_f32_16x16x4_f32 a[10:13], %8, %9, a[12:15]\n s_nop 15\n s_nop 15\n"
                 "v_accvgpr_read_b32 %0, a10\n v_accvgpr_read_b32 %1, a11\n v_accvgpr_read_b32 %2, a12\n v_accvgpr_read_b32 %3, a13\n"
                 : "=&v"(r[4][0]), "=&v"(r[4][1]), "=&v"(r[4][2]), "=&v"(r[4][3]) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(av), "v"(bv)
                 : "a10", "a11", "a12", "a13", "a14", "a15");
    // 5: AGPR, dst = src + 2
    asm volatile("v_accvgpr_write_b32 a10, %4\n v_accvgpr_write_b32 a11, %5\n v_accvgpr_write_b32 a12, %6\n v_accvgpr_write_b32 a13, %7\n s_nop 4\n"
                 "v_mfma_f32_16x16x4_f32 a[12:15], %8, %9, a[10:13]\n s_nop 15\n s_nop 15\n"
                 "v_accvgpr_read_b32 %0, a12\n v_accvgpr_read_b32 %1, a13\n v_accvgpr_read_b32 %2, a14\n v_accvgpr_read_b32 %3, a15\n"
                 : "=&v"(r[5][0]), "=&v"(r[5][1]), "=&v"(r[5][2]), "=&v"(r[5][3]) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(av), "v"(bv)
                 : "a10", "a11", "a12", "a13", "a14", "a15");
    for (int v = 0; v < 6; ++v)
        for (int i = 0; i < 4; ++i) out[(v * 64 + l) * 4 + i] = r[v][i];
}

int main()
{
    float ha[64], hb[64], hc[256], ho[6 * 256];
    for (int i = 0; i < 64; ++i) ha[i] = 0.25f * (i % 7) - 0.5f, hb[i] = 0.125f * (i % 5) + 0.25f;
    for (int i = 0; i < 256; ++i) hc[i] = (float)(i % 13) - 6.0f;
    float *a, *b, *c, *o;
    hipMalloc(&a, sizeof ha), hipMalloc(&b, sizeof hb), hipMalloc(&c, sizeof hc), hipMalloc(&o, sizeof ho);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice), hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c, o);
    if (hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
    const char *names[6] = {"vgpr disjoint", "vgpr dst=src-2", "vgpr dst=src+2", "agpr disjoint", "agpr dst=src-2", "agpr dst=src+2"};
    for (int v = 0; v < 6; ++v) {
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += ho[v * 256 + i] != ho[i];
        printf("%-16s mismatches vs disjoint VGPR form: %d / 256\n", names[v], bad);
    }
    return 0;
}
