// gpr_idx_check.hip -- does gfx950 still have the gfx9 VGPR index mode (s_set_gpr_idx_on / _off), what does hipcc emit for a
// wave-uniform run-time index into a 1024-bit register tuple, and what does one indexed "read a register pair, zero it" cost?
// (VERDICT r03 #1a: the pivot-row gather of the r03 pivoting tile kernels was a scalar branch tree over the register slot.)
//   hipcc --offload-arch=gfx950 -O3 -o gpr_idx_check tools/gpr_idx_check.hip && ./gpr_idx_check
// Cases: (a) C++ `v[idx]` on a 16 x double vector (hipcc emits s_set_gpr_idx_on itself: read the -S output),
//        (b) hand-written index mode on a tuple pinned to v[64:95] with a "{v[64:95]}" constraint, 32-bit moves,
//        (c) the same with v_mov_b64 (is the index applied to a 64-bit operand, and in which unit?),
//        (d) cycles (s_memtime) per indexed extract + zero, back to back, one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v16d __attribute__((ext_vector_type(16)));

__global__ void k_cxx(const double *in, double *out, int idx)
{
    v16d a;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + 64 * i];
    const int u = __builtin_amdgcn_readfirstlane(idx);
    const double x = a[u];
    a[u] = 0.0;
    out[threadIdx.x] = x;
    for (int i = 0; i < 16; ++i) out[64 + threadIdx.x + 64 * i] = a[i];
}

// (b): dword index 2 * idx relative to v64 / v65; the tuple is pinned to v[64:95] for the asm statement
__global__ void k_asm32(const double *in, double *out, int idx)
{
    v16d a;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + 64 * i];
    const int u = 2 * __builtin_amdgcn_readfirstlane(idx);
    unsigned lo, hi;
    asm volatile("s_set_gpr_idx_on %[u], 1\n\t"  // SRC0 relative
                 "v_mov_b32_e32 %[lo], v64\n\t"
                 "v_mov_b32_e32 %[hi], v65\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_set_gpr_idx_on %[u], 8\n\t"  // DST relative
                 "v_mov_b32_e32 v64, 0\n\t"
                 "v_mov_b32_e32 v65, 0\n\t"
                 "s_set_gpr_idx_off"
                 : [lo] "=&v"(lo), [hi] "=&v"(hi), "+{v[64:95]}"(a)
                 : [u] "s"(u)
                 : "m0");
    out[threadIdx.x] = __longlong_as_double(((long long)hi << 32) | lo);
    for (int i = 0; i < 16; ++i) out[64 + threadIdx.x + 64 * i] = a[i];
}

// (c): 64-bit moves under index mode
__global__ void k_asm64(const double *in, double *out, int idx)
{
    v16d a;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + 64 * i];
    const int u = 2 * __builtin_amdgcn_readfirstlane(idx);
    double x;
    asm volatile("s_set_gpr_idx_on %[u], 1\n\t"
                 "v_mov_b64_e32 %[x], v[64:65]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_set_gpr_idx_on %[u], 8\n\t"
                 "v_mov_b64_e32 v[64:65], 0\n\t"
                 "s_set_gpr_idx_off"
                 : [x] "=&v"(x), "+{v[64:95]}"(a)
                 : [u] "s"(u)
                 : "m0");
    out[threadIdx.x] = x;
    for (int i = 0; i < 16; ++i) out[64 + threadIdx.x + 64 * i] = a[i];
}

// (d): 256 dependent-free extract + zero pairs with a rotating index, stamped
__global__ void k_time(const double *in, double *out, long long *cycles, int idx)
{
    v16d a;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + 64 * i];
    int u = 2 * __builtin_amdgcn_readfirstlane(idx);
    double acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        unsigned lo, hi;
        asm volatile("s_set_gpr_idx_on %[u], 1\n\t"
                     "v_mov_b32_e32 %[lo], v64\n\t"
                     "v_mov_b32_e32 %[hi], v65\n\t"
                     "s_set_gpr_idx_off\n\t"
                     "s_set_gpr_idx_on %[u], 8\n\t"
                     "v_mov_b32_e32 v64, 0\n\t"
                     "v_mov_b32_e32 v65, 0\n\t"
                     "s_set_gpr_idx_off"
                     : [lo] "=&v"(lo), [hi] "=&v"(hi), "+{v[64:95]}"(a)
                     : [u] "s"(u)
                     : "m0");
        acc += __longlong_as_double(((long long)hi << 32) | lo);
        u = (u + 6) & 30;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
    out[threadIdx.x] = acc + a[3];
}

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)

static int check(const char *name, const double *h_in, const double *h_out, int idx)
{
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (h_out[l] != h_in[l + 64 * idx]) ++bad;
        for (int i = 0; i < 16; ++i) {
            const double want = i == idx ? 0.0 : h_in[l + 64 * i];
            if (h_out[64 + l + 64 * i] != want) ++bad;
        }
    }
    printf("%-44s idx %2d: %s\n", name, idx, bad ? "MISMATCH" : "ok");
    return bad;
}

int main()
{
    double h_in[1024], h_out[64 + 1024];
    for (int i = 0; i < 1024; ++i) h_in[i] = 1.0 + i;
    double *d_in, *d_out;
    long long *d_cyc, h_cyc = 0;
    CK(hipMalloc(&d_in, sizeof h_in));
    CK(hipMalloc(&d_out, sizeof h_out));
    CK(hipMalloc(&d_cyc, sizeof h_cyc));
    CK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
    int bad = 0;
    for (int idx : {0, 5, 15}) {
        hipLaunchKernelGGL(k_cxx, dim3(1), dim3(64), 0, 0, d_in, d_out, idx);
        CK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
        bad += check("(a) C++ v[idx] (hipcc's own index mode)", h_in, h_out, idx);
        hipLaunchKernelGGL(k_asm32, dim3(1), dim3(64), 0, 0, d_in, d_out, idx);
        CK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
        bad += check("(b) asm s_set_gpr_idx_on + v_mov_b32", h_in, h_out, idx);
        hipLaunchKernelGGL(k_asm64, dim3(1), dim3(64), 0, 0, d_in, d_out, idx);
        CK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
        const int b64 = check("(c) asm s_set_gpr_idx_on + v_mov_b64", h_in, h_out, idx);
        if (b64) printf("    (v_mov_b64 under index mode does not behave like two indexed v_mov_b32: use 32-bit moves)\n");
    }
    hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, d_in, d_out, d_cyc, 3);
    CK(hipMemcpy(&h_cyc, d_cyc, sizeof h_cyc, hipMemcpyDeviceToHost));
    printf("(d) indexed extract + zero of one fp64 register pair: %.1f cycles each (256 back to back, one wave, s_memtime units)\n",
           (double)h_cyc / 256.0);
    printf("%s\n", bad ? "index mode: some case FAILED" : "index mode: works on this device");
    return bad ? 1 : 0;
}
