// Do fp64 MFMAs and VALU work of ANOTHER wave on the same SIMD overlap on gfx950?  One 512-thread workgroup per CU
// (96 KB of LDS keeps a second one out): waves 0-3 (one per SIMD) issue v_mfma_f64_16x16x4_f64 back to back, waves 4-7
// (their SIMD partners) issue a VALU stream of a chosen kind. Each kind is timed alone and together (s_memtime, per wave).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_overlap.hip -o build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
constexpr int BLOCKS = 256, ITERS = 64;

template <int PARTNER>  // 0 none, 1 v_fmac_f64, 2 v_fmac_f32, 4 MFMA as well, 3/5..9 cndmask, add_u32, mov, xor, mul_f32, mov_dpp
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, double seed, int run_mfma, int partner_prio)
{
    __shared__ double pad[12 * 1024];  // 96 KB: one workgroup per CU
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) pad[0] = seed;
    __syncthreads();
    double r = 0;
    long long t0 = 0, t1 = 0;
    const bool mfma_wave = wave < 4;
    if (!mfma_wave && partner_prio) __builtin_amdgcn_s_setprio(3);
    if (mfma_wave ? (run_mfma != 0) : (PARTNER != 0)) {
        if (mfma_wave || PARTNER == 4) {
            v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            const double a = seed + threadIdx.x, b = 1e-9 * a;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < ITERS; ++it) {
                REP8(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);)
                REP8(c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);)
            }
            t1 = __builtin_amdgcn_s_memtime();
            r = c0[0] + c1[1] + c2[2] + c3[3];
        } else if (PARTNER == 1) {
            double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
            const double m = 1e-9 * seed;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < ITERS * 4; ++it) {
                REP8(asm volatile("v_fmac_f64_e32 %0, %8, %8\n\tv_fmac_f64_e32 %1, %8, %8\n\tv_fmac_f64_e32 %2, %8, %8\n\tv_fmac_f64_e32 %3, %8, %8\n\t"
                                  "v_fmac_f64_e32 %4, %8, %8\n\tv_fmac_f64_e32 %5, %8, %8\n\tv_fmac_f64_e32 %6, %8, %8\n\tv_fmac_f64_e32 %7, %8, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
            }
            t1 = __builtin_amdgcn_s_memtime();
            r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        } else if (PARTNER == 2) {
            float a0 = (float)seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
            const float m = 1e-9f * (float)seed;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < ITERS * 4; ++it) {
                REP8(asm volatile("v_fmac_f32_e32 %0, %8, %8\n\tv_fmac_f32_e32 %1, %8, %8\n\tv_fmac_f32_e32 %2, %8, %8\n\tv_fmac_f32_e32 %3, %8, %8\n\t"
                                  "v_fmac_f32_e32 %4, %8, %8\n\tv_fmac_f32_e32 %5, %8, %8\n\tv_fmac_f32_e32 %6, %8, %8\n\tv_fmac_f32_e32 %7, %8, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
            }
            t1 = __builtin_amdgcn_s_memtime();
            r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        } else {
            int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
            const int m = (int)seed + 3;
#define INT8(OP)                                                                                                         \
    REP8(asm volatile(OP(%0) "\n\t" OP(%1) "\n\t" OP(%2) "\n\t" OP(%3) "\n\t" OP(%4) "\n\t" OP(%5) "\n\t" OP(%6) "\n\t" OP(%7) \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
#define INT8NC(OP)                                                                                                       \
    REP8(asm volatile(OP(%0) "\n\t" OP(%1) "\n\t" OP(%2) "\n\t" OP(%3) "\n\t" OP(%4) "\n\t" OP(%5) "\n\t" OP(%6) "\n\t" OP(%7) \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
#define OP_FMACF(x) "v_fmac_f32_e32 " #x ", %8, %8"
#define OP_FMAF(x) "v_fma_f32 " #x ", " #x ", %8, %8"
#define OP_CND(x) "v_cndmask_b32_e32 " #x ", " #x ", %8, vcc"
#define OP_ADD(x) "v_add_u32_e32 " #x ", " #x ", %8"
#define OP_MOV(x) "v_mov_b32_e32 " #x ", %8"
#define OP_XOR(x) "v_xor_b32_e32 " #x ", " #x ", %8"
#define OP_MULF(x) "v_mul_f32_e32 " #x ", " #x ", %8"
#define OP_DPP(x) "v_mov_b32_dpp " #x ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < ITERS * 4; ++it) {
                if (PARTNER == 3) { INT8(OP_CND) }
                else if (PARTNER == 5) { INT8NC(OP_ADD) }
                else if (PARTNER == 6) { INT8NC(OP_MOV) }
                else if (PARTNER == 7) { INT8NC(OP_XOR) }
                else if (PARTNER == 8) { INT8NC(OP_MULF) }
                else if (PARTNER == 10) { INT8NC(OP_FMACF) }
                else if (PARTNER == 11) { INT8NC(OP_FMAF) }
                else { INT8NC(OP_DPP) }
            }
            t1 = __builtin_amdgcn_s_memtime();
            r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        }
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = r + pad[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PARTNER>
void run(const char *name, int run_mfma, int partner_prio = 0)
{
    static double *o = nullptr;
    static long long *c = nullptr;
    static long long h[BLOCKS * 8];
    if (!o && (hipMalloc(&o, (size_t)BLOCKS * 512 * 8) != hipSuccess || hipMalloc(&c, sizeof h) != hipSuccess)) { printf("alloc failed\n"); return; }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<PARTNER>, dim3(BLOCKS), dim3(512), 0, 0, o, c, 1.0, run_mfma, partner_prio);
    if (hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
    double sm = 0, sp = 0;
    for (int b = 0; b < BLOCKS; ++b) { sm += h[b * 8 + 0]; sp += h[b * 8 + 4]; }
    const double n_mfma = ITERS * 32.0, n_valu = ITERS * 4 * 64.0;
    printf("%-34s mfma wave: %8.0f cyc (%6.1f per MFMA)   partner wave: %8.0f cyc (%5.2f per instr)\n", name, sm / BLOCKS,
           sm / BLOCKS / n_mfma, sp / BLOCKS, PARTNER == 4 ? sp / BLOCKS / n_mfma : sp / BLOCKS / n_valu);
}

int main()
{
    run<0>("MFMA alone", 1);
    run<1>("v_fmac_f64 alone", 0);
    run<1>("MFMA + v_fmac_f64 partner", 1);
    run<2>("v_fmac_f32 alone", 0);
    run<2>("MFMA + v_fmac_f32 partner", 1);
    run<3>("v_cndmask_b32 alone", 0);
    run<3>("MFMA + v_cndmask_b32 partner", 1);
    run<5>("v_add_u32 alone", 0);
    run<5>("MFMA + v_add_u32 partner", 1);
    run<6>("v_mov_b32 alone", 0);
    run<6>("MFMA + v_mov_b32 partner", 1);
    run<7>("v_xor_b32 alone", 0);
    run<7>("MFMA + v_xor_b32 partner", 1);
    run<8>("v_mul_f32 alone", 0);
    run<8>("MFMA + v_mul_f32 partner", 1);
    run<9>("v_mov_b32_dpp alone", 0);
    run<9>("MFMA + v_mov_b32_dpp partner", 1);
    run<5>("MFMA + v_add_u32 partner, prio 3", 1, 1);
    run<8>("MFMA + v_mul_f32 partner, prio 3", 1, 1);
    run<3>("MFMA + v_cndmask partner, prio 3", 1, 1);
    run<1>("MFMA + v_fmac_f64 partner, prio 3", 1, 1);
    run<10>("MFMA + v_fmac_f32 (int regs) partner", 1);
    run<11>("MFMA + v_fma_f32 VOP3 partner", 1);
    run<4>("MFMA partner alone", 0);
    run<4>("MFMA + MFMA partner", 1);
    return 0;
}
