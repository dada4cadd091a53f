// tileq_stamps.hip -- where a block step of the TILEQ kernels (csrc/tileq_impl.hpp) spends its cycles: the kernel compiled with
// -DMATINV_TILEQ_STAMPS sums s_memtime differences per phase and wave; this driver runs it on U(0,1) matrices and prints cycles
// per matrix and block step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -DMATINV_TILEQ_STAMPS -DSTAMP_NT=8 -DSTAMP_W=4 \
//         -DSTAMP_NC=2 -DSTAMP_OCC=2 [-DSTAMP_T=float] -Icuda-matrix-inversion_amd/csrc -o tileq_stamps tools/tileq_stamps.hip
// (its own kernel wrapper, so that waves per matrix, tile columns per wave and the occupancy the registers are budgeted for can be varied;
// without -DMATINV_TILEQ_STAMPS it only times the launch)
#include "tileq_impl.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef STAMP_T
#define STAMP_T double
#endif
#define STAMP_THREADS (64 * STAMP_W)

namespace matinv {
template <int NT, bool FULL>
__global__ __launch_bounds__(STAMP_THREADS, STAMP_OCC) void stamp_kernel(BatchRef<const STAMP_T> Ain, BatchRef<STAMP_T> Xout, int *info, int n_rt,
                                                                        unsigned batch, int *bad_count, int *bad_list, const int *in_count,
                                                                        const int *in_list, hint_t *hint_out)
{
    __shared__ __attribute__((aligned(16))) STAMP_T rowpanel[16 * NT * 4];
    __shared__ __attribute__((aligned(16))) STAMP_T bprime[16 * NT * 4];
    __shared__ __attribute__((aligned(16))) STAMP_T colpanel[16 * NT * 4];
    __shared__ unsigned char tab[512];
    __shared__ int meta[8];
    gj_tileq_body<STAMP_T, NT, STAMP_W, STAMP_NC, FULL>(Ain, Xout, info, n_rt, batch, rowpanel, bprime, colpanel, tab, meta, bad_count, bad_list,
                                                        in_count, in_list, hint_out);
}
}  // namespace matinv

int main(int argc, char **argv)
{
    using namespace matinv;
    typedef STAMP_T T;
    constexpr int NT = STAMP_NT, n = 16 * NT;
    const unsigned batch = argc > 1 ? (unsigned)atoi(argv[1]) : 12288u;
    const unsigned grid = argc > 2 ? (unsigned)atoi(argv[2]) : (batch < 1024u ? batch : 1024u);
    std::vector<T> h((size_t)batch * n * n);
    unsigned long long s = 0x5EEDull;
    for (auto &v : h) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        v = (T)((s >> 11) * (1.0 / 9007199254740992.0));
    }
    T *dA, *dX;
    int *dInfo, *dBad;
    if (hipMalloc(&dA, h.size() * sizeof(T)) != hipSuccess || hipMalloc(&dX, h.size() * sizeof(T)) != hipSuccess ||
        hipMalloc(&dInfo, batch * sizeof(int)) != hipSuccess || hipMalloc(&dBad, (batch + 1) * sizeof(int)) != hipSuccess)
        return 2;
    (void)hipMemcpy(dA, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    (void)hipMemset(dBad, 0, sizeof(int));
    BatchRef<const T> A{dA, (size_t)n * n, nullptr};
    BatchRef<T> X{dX, (size_t)n * n, nullptr};
    for (int rep = 0; rep < 3; ++rep) {
#ifdef MATINV_TILEQ_STAMPS
        unsigned long long zero[256] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(matinv_tileq_stamps), zero, sizeof zero);
#endif
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL((stamp_kernel<NT, true>), dim3(grid), dim3(STAMP_THREADS), 0, 0, A, X, dInfo, n, batch, dBad, dBad + 1,
                           (const int *)nullptr, (const int *)nullptr, (hint_t *)nullptr);
        if (hipDeviceSynchronize() != hipSuccess) {
            fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
            return 1;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep < 2) continue;
#ifndef MATINV_TILEQ_STAMPS
        printf("n = %d, batch %u, grid %u, %d waves x %d tile columns, budget %d: %.3f ms = %.3e inv/s\n", n, batch, grid, STAMP_W, STAMP_NC, STAMP_OCC, ms,
               batch / ms * 1e3);
#else
        unsigned long long st[256];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(matinv_tileq_stamps), sizeof st);
        static const char *names[15] = {"0 loads / idle", "1 look-ahead MFMA + rows->LDS", "2 barrier 1", "3 search (+MFMAs)", "4 -",
                                        "5 wait for pivots, col gather", "6 barrier 2", "7 operands", "8 last MFMAs", "9 stores", "10 other MFMAs (non-searcher)", "11 fine", "12 fine", "13 fine", "14 fine"};
        printf("n = %d, batch %u, grid %u: %.3f ms = %.3e inv/s; cycles (s_memtime) per matrix, summed over its %d block steps, per wave:\n", n, batch, grid,
               ms, batch / ms * 1e3, 4 * NT);
        const int waves = STAMP_THREADS / 64;
        for (int ph = 0; ph < 15; ++ph) {
            printf("  %-34s", names[ph]);
            for (int w = 0; w < waves; ++w) printf(" %9.0f", (double)st[w * 16 + ph] / batch);
            printf("\n");
        }
        for (int w = 0; w < waves; ++w) {
            double tot = 0;
            for (int ph = 0; ph < 15; ++ph) tot += (double)st[w * 16 + ph];
            printf("  wave %d total %.0f per matrix = %.0f per block step\n", w, tot / batch, tot / batch / (4 * NT));
        }
#endif
    }
    // residual of the first and the last matrix (the kernel variants this tool is built with are experiments)
    std::vector<T> x((size_t)n * n);
    double worst = 0;
    for (unsigned m : {0u, batch - 1}) {
        (void)hipMemcpy(x.data(), dX + (size_t)m * n * n, x.size() * sizeof(T), hipMemcpyDeviceToHost);
        const T *a = h.data() + (size_t)m * n * n;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0;
                for (int k = 0; k < n; ++k) acc += (double)a[k * n + i] * (double)x[j * n + k];  // column-major: (i, k) at k n + i
                const double d = acc - (i == j ? 1.0 : 0.0);
                if (!(d <= worst && -d <= worst)) worst = d < 0 ? -d : d;
            }
    }
    printf("max |A X - I| over matrices 0 and %u: %.2e\n", batch - 1, worst);
    return worst < (sizeof(T) == 8 ? 1e-8 : 5e-2) ? 0 : 1;
}
