// row_kernels.hip -- kernel family "ROW": Gauss-Jordan with TRUE partial pivoting for 16 < n <= 64, one matrix per
// wavefront, register resident, no LDS. It is the pivoting path behind the MFMA tile kernels: a matrix whose natural
// pivot order fails their acceptance test (general, non-dominant input) is appended to a device work list and inverted
// here, in the same stream.
//
// Lane i owns row i, register c holds column c (load a[c] = A[c*n + i]: 512 contiguous bytes per column at n = 64).
// Step k:   pivot = largest |a[i][k]| over the rows not used yet  (16-lane DPP max, 4 v_readlane + s_max across the
//                   rows of 16, ballot + s_ff1 for the lowest lane attaining it)
//           the pivot row stays where it is (IMPLICIT pivoting: no data moves); its entries reach the other lanes as
//           SCALAR operands: s = v_readlane(a[c], p), a[c] = fma(-m_i, s, a[c])   -- 2 readlanes + 1 FMA per column.
// That broadcast is the whole cost (the reason the tile kernels use the matrix cores instead), about 3x the FMAs, but it
// buys classical partial pivoting at ~15 k VALU instructions per 64 x 64 matrix instead of a trip through LDS.
// Row normalisation is deferred to one multiply per element at the end; the permutation that implicit pivoting leaves
// behind (output row = step at which the lane was pivot, output column of register j = pivot lane of step j) is folded
// into the store addresses, which stay 512-byte contiguous per column.
//
// Replaces pivotRow / normalizeRow / transform_matrix of /root/reference/src/gauss/batched_invert.cu:17-82 for the inputs
// that really need row exchanges (the reference only swaps on an exactly zero diagonal, :19-35).
#include "wave_util.hpp"

namespace matinv {


// One matrix by the calling wavefront. NP = 32 or 64 (register count); n <= NP.
template <class T, int NP>
__device__ __forceinline__ void gj_row_one(const T *A, T *X, int *info_slot, int n)
{
    int i = threadIdx.x & 63;
    // launder the lane id once per matrix: otherwise LICM hoists the NP lane masks (i == k) of the unrolled steps out of
    // the caller's batch loop, 2 SGPRs each, and the scalar file spills through VGPR lanes
    asm volatile("" : "+v"(i));
    const bool row_in = i < n;
    T a[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) a[c] = (row_in && c < n) ? A[c * n + i] : (T)0;

    bool used = !row_in;   // rows beyond n never take part
    int pivstep = 0;       // step at which this lane's row was the pivot = its row index in the result
    int pivlane = 0;       // lane j: pivot lane of step j = output column of register j
    T rowscale = (T)1;
    int bad = 0;

#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (k < n) {  // wave-uniform
            const unsigned key = used ? 0u : magkey(a[k]);
            const unsigned mx = wave_max_u32(key);
            if (key_bad(T(0), mx) && bad == 0) bad = k + 1;  // no non-zero finite candidate: singular
            const unsigned long long vote = __ballot(!used && key == mx);
            const int p = vote ? (int)__builtin_ctzll(vote) : 0;
            const T piv = lane_value(a[k], p);
            const T inv = rcp_full(piv);
            const bool me = (i == p);
            const T negm = me ? (T)0 : -(a[k] * inv);
            // Pivot row entries as scalar operands. The scalar register file is the scarce resource (~100 SGPRs): left to
            // itself hipcc hoists all 2*NP v_readlane of a step ahead of the FMAs and spills SGPRs through VGPR lanes, so
            // the order is pinned in groups of 4 columns.
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                // columns >= n hold zeros and stay zero (never stored): no per-column branch
                if (c != k) a[c] = fmat(negm, lane_value(a[c], p), a[c]);
                if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            a[k] = me ? (T)1 : negm;
            rowscale = me ? inv : rowscale;
            pivstep = me ? k : pivstep;
            used = used || me;
            pivlane = (i == k) ? p : pivlane;
        }
    }
    // result element (row pivstep_i, column pivlane_j) = a_i[j] * rowscale_i
    const bool fail = bad != 0;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        if (c < n) {
            const int col = __builtin_amdgcn_readlane(pivlane, c);
            if (row_in) X[col * n + (fail ? i : pivstep)] = fail ? nan_of<T>() : a[c] * rowscale;
        }
    }
    if (info_slot && i == 0) *info_slot = bad;
}

// work-list form: one wavefront per listed matrix, 4 wavefronts per workgroup
template <class T, int NP>
__global__ __launch_bounds__(256, 2) void matinv_gj_row_worklist(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n,
                                                                const int *work_count, const int *work_list)
{
    const int count = *work_count;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), stride = gridDim.x * 4;
    for (unsigned w = wave; w < (unsigned)count; w += stride) {
        const size_t k = (size_t)work_list[w];
        gj_row_one<T, NP>(Ain.at_uniform(k), Xout.at_uniform(k), info ? info + k : nullptr, n);
    }
}

// whole-batch form (MATINV_KERNEL_ROW)
template <class T, int NP>
__global__ __launch_bounds__(256, 2) void matinv_gj_row(BatchRef<const T> Ain, BatchRef<T> Xout, int *info, int n,
                                                       unsigned batch)
{
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), stride = gridDim.x * 4;
    for (unsigned k = wave; k < batch; k += stride)
        gj_row_one<T, NP>(Ain.at_uniform(k), Xout.at_uniform(k), info ? info + k : nullptr, n);
}

template <class T>
bool row_family_supports(int n) { return n >= 1 && n <= 64; }
template bool row_family_supports<double>(int);
template bool row_family_supports<float>(int);

template <class T>
hipError_t launch_gj_row_worklist(int n, BatchRef<const T> A, BatchRef<T> X, const int *work_count, const int *work_list,
                                  int *info, hipStream_t stream)
{
    if (!row_family_supports<T>(n)) return hipErrorInvalidValue;
    // the list length is only known on the device: a few rounds of resident blocks; blocks beyond the list exit at once
    const unsigned wl_rounds = tile_grid_rounds() < 8u ? tile_grid_rounds() : 8u;
    if (n <= 32)
        hipLaunchKernelGGL((matinv_gj_row_worklist<T, 32>), dim3(512 * wl_rounds), dim3(256), 0, stream, A, X, info, n, work_count, work_list);
    else
        hipLaunchKernelGGL((matinv_gj_row_worklist<T, 64>), dim3(512 * wl_rounds), dim3(256), 0, stream, A, X, info, n, work_count, work_list);
    return hipGetLastError();
}
template <class T>
hipError_t launch_gj_row(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!row_family_supports<T>(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const size_t blocks = (batch + 3) / 4;
    // 2 blocks of 4 waves are resident per CU; like the tile kernels (tile_grid_rounds) many rounds beat a static stride
    const unsigned cap = 256u * 2u * tile_grid_rounds();
    const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
    if (n <= 32) hipLaunchKernelGGL((matinv_gj_row<T, 32>), dim3(grid), dim3(256), 0, stream, A, X, info, n, (unsigned)batch);
    else hipLaunchKernelGGL((matinv_gj_row<T, 64>), dim3(grid), dim3(256), 0, stream, A, X, info, n, (unsigned)batch);
    return hipGetLastError();
}
#define INST(T)                                                                                                        \
    template hipError_t launch_gj_row_worklist<T>(int, BatchRef<const T>, BatchRef<T>, const int *, const int *, int *, \
                                                  hipStream_t);                                                       \
    template hipError_t launch_gj_row<T>(int, BatchRef<const T>, BatchRef<T>, size_t, int *, hipStream_t);
INST(double)
INST(float)
#undef INST

const char *name_gj_row(bool f64, int n)
{
    if (n <= 32) return f64 ? "matinv_gj_row<double, 32>" : "matinv_gj_row<float, 32>";
    return f64 ? "matinv_gj_row<double, 64>" : "matinv_gj_row<float, 64>";
}

}  // namespace matinv
