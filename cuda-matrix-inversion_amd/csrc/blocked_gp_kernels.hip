// blocked_gp_kernels.hip -- fused Gaussian-process scalars for LARGE n (beyond what fits in LDS; bins 512 / 1024 of
// BASELINE configs[4]):   mean = a^T (B + diag c)^-1 d,   var = e - a^T (B + diag c)^-1 a.
//
// Blocked right-looking Cholesky on a global-memory working copy, with the two vectors carried as two extra ROWS of the
// matrix (bordered trick: row n = a^T, row n+1 = d^T; after the factorisation they hold (L^-1 a)^T and (L^-1 d)^T, and the
// answer is their dot product -- no inverse, no separate triangular solves). Per panel of PB = 64 columns, TWO launches
// over the whole batch:
//   matinv_bgp_panel   one workgroup per 256 rows of a matrix: factor the PB x PB diagonal block in LDS (every
//                      workgroup of the matrix does this redundantly -- the block itself is never needed again, so it
//                      is not written back and there is no write/read race between them), then each thread solves ONE
//                      row of the column panel below it (incl. the two border rows) against it;
//   matinv_bgp_update  one workgroup per 64 x 64 tile of the trailing lower triangle (and of the border rows):
//                      W[I,J] -= L[I,K] L[J,K]^T, both panels staged through LDS in slabs of 32, 4 x 4 outputs per thread.
// so a batch of few large matrices still fills the chip (the GLOBAL family runs one workgroup per matrix). The launch
// count is 2 n / PB per batch -- this replaces the reference's 4N+1 launches per batch (src/inverse_cholesky_gpu.cu:
// 323-354) plus its two gemmBatched calls (src/gauss_bench.cu:210,232) for the sizes its design note calls "512, 1024"
// (README.md:41-44).
// (Tried and measured, not kept: ONE cooperative launch for small batches -- the same four bodies looped over by a co-resident
// grid with cooperative_groups grid.sync() between phases instead of 2 n / PB + 2 dependent launches. 8 x 1024^2 fp32:
// 5.27 ms against 1.27 ms for the launches; 32 x 512^2: 2.84 against 0.64 ms. A grid-wide barrier over ~500 workgroups costs
// far more here than the ~6 us gap between two dependent launches.)
#include <mutex>
#include <vector>

#include "chol_block.hpp"
#include "slab_mma.hpp"

namespace matinv {

constexpr int BGP_PB = 64;     // panel width
constexpr int BGP_KS = SLAB_KS;  // k-slab of the update kernel (LDS staging depth)
// waves per SIMD the two MFMA product kernels are built for (r03): 5 (<= 102 registers) against the 4 hipcc chose by itself:
// SPD inverse fp64 1024^2 2.15e4 -> 2.43e4 inv/s, 256^2 5.6e5 -> 6.2e5; 8-deep slabs at 6 waves: 2.48e4 / 6.1e5 (fp32 slightly slower)
#ifndef MATINV_BGP_OCC
#define MATINV_BGP_OCC 5
#endif
constexpr int BGP_TILE = 64;   // update tile edge
constexpr int BGP_THREADS = 256;
constexpr int BGP_LDS = SLAB_LDS;  // row stride of the LDS slabs [k][row] (slab_mma.hpp)
static_assert(BGP_KS == SLAB_KS, "the update and product kernels stage slab_mma's slabs");

// Leading dimension of a working copy with `rows` rows: padded so that one column is an ODD multiple of 256 bytes. With
// ld = 2 n = 2048 doubles consecutive columns of a panel are 16 KiB apart and a 64-column panel lands on two of the 128
// HBM channels; with the odd multiple the columns walk through all of them. MATINV_BGP_PAD=0: unpadded (A/B measurements).
template <class T>
__host__ __device__ __forceinline__ int bgp_ld_padded(int rows)
{
    int units = (int)(((size_t)rows * sizeof(T) + 255) / 256);
    units |= 1;
    return units * (256 / (int)sizeof(T));
}
static bool bgp_pad_on()
{
    static const bool on = [] { const char *s = getenv("MATINV_BGP_PAD"); return !(s && s[0] == '0'); }();
    return on;
}
template <class T>
static int bgp_ld(int rows) { return bgp_pad_on() ? bgp_ld_padded<T>(rows) : rows; }

// working copy layout per item: (n + 2) rows x n columns, COLUMN-major with leading dimension ld >= n + 2 (bgp_ld):
// element (r, c) at c*ld + r; rows n and n+1 are the border rows a^T and d^T (d = a for the variance).
template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_init(const T *As, const T *Bs, const T *Cs, const T *Ds, T *W,
                                                               int n, int ld, int *status)
{
    const size_t item = blockIdx.y;
    T *w = W + item * (size_t)ld * n;
    const T *B = Bs + item * (size_t)n * n;
    for (size_t e = (size_t)blockIdx.x * BGP_THREADS + threadIdx.x; e < (size_t)ld * n; e += (size_t)gridDim.x * BGP_THREADS) {
        const int c = (int)(e / ld), r = (int)(e - (size_t)c * ld);
        if (r >= n + 2 || r < c) continue;  // padding of the leading dimension; the strict upper triangle is never read
        T v;
        if (r < n) v = (r >= c) ? B[(size_t)c * n + r] + ((r == c) ? Cs[item * n + c] : (T)0) : (T)0;  // lower triangle only
        else if (r == n) v = As[item * n + c];
        else v = Ds ? Ds[item * n + c] : As[item * n + c];
        w[e] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) status[item] = 0;
}

template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_panel(T *W, int n, int ld, int row_end, int k0, int *status)
{
    constexpr int LD = BGP_PB + 1;
    __shared__ T L11[BGP_PB * LD];  // column-major, L11[c * LD + r]
    __shared__ T rinv[BGP_PB];
    const size_t item = blockIdx.y;
    const int pb = (n - k0 < BGP_PB) ? n - k0 : BGP_PB, t = threadIdx.x;
    T *w = W + item * (size_t)ld * n;
    if (status[item] != 0) return;  // an earlier panel found a non-positive pivot
    for (int e = t; e < BGP_PB * BGP_PB; e += BGP_THREADS) {
        const int c = e / BGP_PB, r = e - c * BGP_PB;
        // lower triangle; ragged last panel: identity padding
        L11[c * LD + r] = (r < pb && c < pb) ? ((r >= c) ? w[(size_t)(k0 + c) * ld + k0 + r] : (T)0) : (T)(r == c);
    }
    __syncthreads();
    int nb = BGP_PB;
    asm volatile("" : "+s"(nb));  // opaque size: with literal bounds hipcc unrolls the whole factorisation and spills
    const int bad = chol_factor_lds(L11, LD, nb, nb);
    if (bad) {  // block-uniform, and the same in every workgroup of this item
        if (t == 0 && blockIdx.x == 0) status[item] = k0 + bad;
        return;
    }
    if (t < BGP_PB) rinv[t] = (T)1 / L11[t * LD + t];
    __syncthreads();
    // one row per thread: x L11^T = row, right-looking (once x[c] is final it is eliminated from the later entries, so
    // the dependent chain is PB long and the inner updates are independent FMAs).
    // (Tried and measured, not kept: TWO rows per thread in fp32, sharing every L11 entry read from LDS and halving the
    // workgroups that repeat the factorisation: 170 VGPRs, two waves per SIMD -- fused pipeline 256^2 1.63 -> 2.72 ms,
    // 1024^2 3.10 -> 3.44 ms, the 8-item bin of the mixed queue 1.05 -> 1.47 ms.)
    const int r = k0 + pb + blockIdx.x * BGP_THREADS + t;
    if (r >= row_end) return;  // rows beyond row_end are still zero in these columns (identity border of the inversion)
    T x[BGP_PB];
#pragma unroll
    for (int c = 0; c < BGP_PB; ++c) {
        const T v = w[(size_t)(k0 + (c < pb ? c : pb - 1)) * ld + r];  // clamped address: no branch per element
        x[c] = (c < pb) ? v : (T)0;
    }
#pragma unroll
    for (int c = 0; c < BGP_PB; ++c) {
        x[c] *= rinv[c];
        // the column offset is made opaque and tied to x[c]: hipcc otherwise hoists all 2016 (read-only) LDS loads to the
        // top of the block and spills them
        int off = c * LD;
        asm volatile("" : "+v"(off), "+v"(x[c]));
#pragma unroll
        for (int j = c + 1; j < BGP_PB; ++j) x[j] = fma(-x[c], L11[off + j], x[j]);
    }
#pragma unroll
    for (int c = 0; c < BGP_PB; ++c)
        if (c < pb) w[(size_t)(k0 + c) * ld + r] = x[c];
}

// trailing update: W[I, J] -= L[I, K] L[J, K]^T for 64 x 64 tiles with jbeg <= J < jend, I >= J (rows up to row_end - 1);
// K = the kcnt panel columns from kbeg on (one panel of 64, or the two panels of a pair: see launch_gp_blocked)
// LDL = true (the block-LDL^T path below): the J side of the product is read from the raw copy of the panel in Sraw (per item
// BGP_PB columns of leading dimension ld, same row index as the working copy) instead of from the working copy itself.
template <class T, bool LDL = false>
__global__ __launch_bounds__(BGP_THREADS, MATINV_BGP_OCC) void matinv_bgp_update(T *W, int n, int ld, int row_end, int kbeg, int kcnt, int jbeg,
                                                                 int jend, const int *status, unsigned gx, unsigned gy, unsigned nb,
                                                                 const T *Sraw = nullptr)
{
    const XcdTile tile = xcd_tile_of(blockIdx.x, gx, gy, nb);
    if (!tile.valid) return;
    typedef TileGeo<T> G;
    __shared__ T Li[BGP_KS][BGP_LDS], Lj[BGP_KS][BGP_LDS];
    const size_t item = tile.z;
    if (status[item] != 0) return;
    const int pb = kcnt, k0 = kbeg;
    const int j0 = jbeg + tile.x * BGP_TILE, i0 = jbeg + tile.y * BGP_TILE;
    if (j0 >= jend || i0 >= row_end || i0 + BGP_TILE <= j0) return;  // outside, or strictly above the diagonal
    T *w = W + item * (size_t)ld * n;
    const int t = threadIdx.x, wv = t >> 6, q = (t >> 4) & 3, c = t & 15;
    // this wavefront's 32 x 32 part strictly above the diagonal: it only helps staging (wave-uniform)
    const bool live = i0 + 32 * (wv & 1) + 32 > j0 + 32 * (wv >> 1);
    typename G::vec4 acc[2][2] = {};
    // slabs of BGP_KS panel columns through LDS; the NEXT slab is fetched into registers while the current one is multiplied
    // (thread t fetches row t & 63 of columns (t >> 6) + 4 x; clamped addresses, so the loads are unconditional). Fetching TWO
    // slabs ahead was measured and lost: 130 VGPRs, three waves per SIMD instead of four, 1024^2 fp64 SPD inverse 23.7 ms against 21.4.
    const int lr = t & 63, lk = t >> 6;
    const bool in_i = i0 + lr < row_end, in_j = j0 + lr < jend;
    const T *wi = w + (in_i ? i0 + lr : row_end - 1);
    const T *wj = (LDL ? Sraw + item * (size_t)ld * (2 * BGP_PB) : w) + (in_j ? j0 + lr : jend - 1);
    T pi[BGP_KS / 4], pj[BGP_KS / 4];
    auto fetch = [&](int ks) {
#pragma unroll
        for (int x = 0; x < BGP_KS / 4; ++x) {
            const int k = ks + lk + 4 * x;
            const bool kin = k < pb;
            const size_t col = (size_t)(k0 + (kin ? k : pb - 1)) * ld;
            pi[x] = wi[col];  // raw: padding is zeroed when the slab is staged -- a select here would make the wave wait
            pj[x] = wj[LDL ? (size_t)(kin ? k : pb - 1) * ld : col];  // for the loads before the MFMAs they are meant to hide behind
        }
    };
    fetch(0);
    for (int ks = 0; ks < pb; ks += BGP_KS) {
        __syncthreads();
#pragma unroll
        for (int x = 0; x < BGP_KS / 4; ++x) {
            const bool kin = ks + lk + 4 * x < pb;
            Li[lk + 4 * x][lr] = (kin && in_i) ? pi[x] : (T)0;
            Lj[lk + 4 * x][lr] = (kin && in_j) ? pj[x] : (T)0;
        }
        __syncthreads();
        if (ks + BGP_KS < pb) fetch(ks + BGP_KS);
        if (live) slab_mma<T>(Lj, Li, wv, q, c, acc);
    }
    if (!live) return;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int J = j0 + 32 * (wv >> 1) + 16 * tj + G::trow(r, q), I = i0 + 32 * (wv & 1) + 16 * ti + c;
                if (I < row_end && J < jend && I >= J) w[(size_t)J * ld + I] -= acc[tj][ti][r];
            }
}

// (Tried and measured, not kept: both products -- this update and matinv_binv_syrk below -- on the matrix cores, 64 x 64 per
// workgroup / 32 x 32 per wavefront, computed transposed so that operands and the tile are 128-byte row segments, operands
// straight from global memory 16 columns ahead of their MFMAs. SPD inverse, fp64, ms per batch, MFMA against these kernels:
// 256^2 x 3051: 9.7 / 10.7; 512^2 x 762: 16.6 / 16.2; 1024^2 x 256: 38.8 / 32.9; fp32 1024^2: 18.0 / 14.8. With a panel of
// 64 columns the update is one read-modify-write of the trailing matrix per 64 multiply-adds per element: at the 2.4 TB/s
// this kernel moves, its flops are not what bounds it, and without LDS staging the operand loads of four independent
// wavefronts cost more than the matrix cores save. The lever is a wider panel, not the instruction.)
template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_bgp_finish(const T *W, const T *Es, T *out, int *info, int n, int ld,
                                                                 const int *status, int drow)
{
    __shared__ T part[BGP_THREADS / 64];
    const size_t item = blockIdx.x;
    const int t = threadIdx.x;
    const T *w = W + item * (size_t)ld * n;
    const int bad = status[item];
    T s = 0;
    if (!bad)
        for (int c = t; c < n; c += BGP_THREADS) s += w[(size_t)c * ld + n] * w[(size_t)c * ld + drow];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off);
    if ((t & 63) == 0) part[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
        T q = 0;
        for (int i = 0; i < BGP_THREADS / 64; ++i) q += part[i];
        out[item] = bad ? nan_of<T>() : (Es ? Es[item] - q : q);
        if (info) info[item] = bad;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Block LDL^T form of the same pipeline for LATENCY-BOUND launches (a handful of large items: the 512 / 1024 bins of the mixed queue,
// r04). Measured on 8 x 1024^2 fp32, the dependent chain of 2 n / 64 launches waits for the panel kernel above: 50 us per launch, of
// which the one-row-per-thread triangular solve is 34 us (2016 broadcast LDS reads per thread, four wavefronts on one LDS unit) and
// the factorisation of the diagonal block 10 us; carrying the identity as border rows of that factorisation to get L11^-T (and the
// solve as a product) cost as much as it saved (44 -> 40 us). Here no Cholesky factor is formed at all:
//     S[I,J] <- A[I,J] - sum_{k<J} S[I,k] G_k S[J,k]^T,   G_k = S[k,k]^-1,   answer = sum_k S[a,k] G_k S[d,k]^T
// (S[I,k] G_k S[J,k]^T = L[I,k] L[J,k]^T of the Cholesky form, so the Schur complements S are the same matrices). Per panel:
//   matinv_bldl_panel   one workgroup per 64 rows below the diagonal block (border rows included). Wave 0 inverts the 64 x 64 block
//                       with the symmetric MFMA sweep of the one-wavefront kernels (tile_kernels.inc: 16 block steps on 10 lower
//                       tiles held in registers; non-positive pivot = not positive definite, reported with its column), every
//                       wave then forms its 32 x 32 part of P = S[rows, k] G_k on the matrix cores, the S operand fetched from
//                       global memory into MFMA operand registers BEFORE the sweep starts. P replaces the panel in place; the raw
//                       panel S goes to a side buffer (the J side of the update) and the raw d-row to the extra row n + 2.
//   matinv_bgp_update<T, true>   W[I,J] -= P[I,k] S[J,k]^T, the kernel above with its J operand read from the side buffer.
// matinv_bgp_finish then takes the dot product of row n (P[a,:]) and row n + 2 (S[d,:]).
template <class T>
__device__ __forceinline__ void spd_invert64_wave(const T *S, int lds, T *Gs, int ldg, T *panel, int l, int &binfo)
{
    typedef TileGeo<T> G;
    typedef typename G::vec4 vec4;
    constexpr int NT = BGP_PB / 16;
    int q = l >> 4, c = l & 15;
    asm volatile("" : "+v"(q), "+v"(c));
    vec4 acc[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            if (tj > ti) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                const int hi = row > col ? row : col, lo = row > col ? col : row;
                acc[ti][tj][r] = S[lo * lds + hi];  // the block is symmetric: only its lower triangle is stored
            }
        }
    unsigned long long bad = 0;
    T aop[NT], bop[NT];
#pragma unroll
    for (int kb = 0; kb < 4 * NT; ++kb) {
        spd_panel_to_lds<NT, T>(panel, acc, kb, q, c);
        wave_lds_sync();
        PanelSolve<NT, true, T> ps;
        ps.binfo = &binfo;
#pragma unroll
        for (int s = 0; s < PanelSolve<NT, true, T>::NSTAGE; ++s) ps.stage(s, panel, kb, q, c, aop, bop, bad);
        spd_prep_operands<NT, T>(acc, bop, kb, q, c);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                if (tj > ti) continue;
                acc[ti][tj] = G::mfma(aop[ti], bop[tj], acc[ti][tj]);
            }
        wave_lds_sync();
    }
    // acc = -S^-1 in the lower tiles (diagonal tiles complete): both triangles go to Gs
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            if (tj > ti) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + G::trow(r, q), col = 16 * tj + c;
                Gs[row * ldg + col] = -acc[ti][tj][r];
                if (tj < ti) Gs[col * ldg + row] = -acc[ti][tj][r];
            }
        }
}

// (two waves per SIMD = at most 256 registers, all of them VGPRs: built for one wave per SIMD hipcc puts the fp32 accumulators of the
// sweep into AGPRs and its AGPR-form v_mfma_f32 computes garbage from the second block step on -- ROCm 7.2.0, see matinv_spd_tile_f32)
template <class T>
__global__ __launch_bounds__(BGP_THREADS, 2) void matinv_bldl_panel(T *W, T *Sraw, int n, int ld, int row_end, int k0, int *status, int scol)
{
    typedef TileGeo<T> G;
    constexpr int LD = BGP_PB + 1;
    __shared__ T Sd[BGP_PB * LD];  // the diagonal block, column-major Sd[c * LD + r], lower triangle
    __shared__ T Gs[BGP_PB * LD];  // its inverse, both triangles
    __shared__ T panel[4 * BGP_PB];
    __shared__ int sh_bad;
    const size_t item = blockIdx.y;
    const int pb = (n - k0 < BGP_PB) ? n - k0 : BGP_PB, t = threadIdx.x;
    T *w = W + item * (size_t)ld * n;
    T *sraw = Sraw + item * (size_t)ld * (2 * BGP_PB) + (size_t)scol * ld;  // scol = 0, or 64 for the second panel of a pair
    if (status[item] != 0) return;  // an earlier panel found a non-positive pivot
    const int wv = t >> 6, q = (t >> 4) & 3, c = t & 15;
    // this wavefront's rows of the panel (the I side of the product): 2 x 16 rows, all 64 panel columns, in MFMA operand layout
    const int r0 = k0 + pb + blockIdx.x * BGP_TILE + 32 * (wv & 1) + c;
    T b[2][BGP_PB / 4];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int r = r0 + 16 * ti;
        const T *wr = w + (r < row_end ? r : row_end - 1);
#pragma unroll
        for (int k4 = 0; k4 < BGP_PB / 4; ++k4) {
            const int k = 4 * k4 + q;
            b[ti][k4] = wr[(size_t)(k0 + (k < pb ? k : pb - 1)) * ld];  // clamped address: padding columns meet zeros of G
        }
    }
    for (int e = t; e < BGP_PB * BGP_PB; e += BGP_THREADS) {
        const int cc = e / BGP_PB, r = e - cc * BGP_PB;
        // lower triangle; ragged last panel: identity padding
        Sd[cc * LD + r] = (r < pb && cc < pb) ? ((r >= cc) ? w[(size_t)(k0 + cc) * ld + k0 + r] : (T)0) : (T)(r == cc);
    }
    if (t == 0) sh_bad = 0;
    __syncthreads();
    if (wv == 0) {  // wave-uniform
        int binfo = 0;
        spd_invert64_wave<T>(Sd, LD, Gs, LD, panel, t, binfo);
        if (t == 0) sh_bad = binfo;
    }
    __syncthreads();
    const int bad = sh_bad;
    if (bad) {  // block-uniform, and the same in every workgroup of this item
        if (t == 0 && blockIdx.x == 0) status[item] = k0 + bad;
        return;
    }
    // P[I][J] = sum_k S[I][k] G[k][J]; computed transposed as in slab_mma (the MFMA's A operand carries the J side), so that a lane
    // group holds 16 consecutive rows I of one column J
    const int jb = 32 * (wv >> 1) + c;
    typename G::vec4 acc[2][2] = {};
#pragma unroll
    for (int k4 = 0; k4 < BGP_PB / 4; ++k4) {
        const int k = 4 * k4 + q;
        const T a0 = Gs[jb * LD + k], a1 = Gs[(jb + 16) * LD + k];
        acc[0][0] = G::mfma(a0, b[0][k4], acc[0][0]);
        acc[0][1] = G::mfma(a0, b[1][k4], acc[0][1]);
        acc[1][0] = G::mfma(a1, b[0][k4], acc[1][0]);
        acc[1][1] = G::mfma(a1, b[1][k4], acc[1][1]);
    }
    // the raw panel: the J side of this panel's update, and the raw d-row for the final dot product
    if ((wv >> 1) == 0) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int I = r0 + 16 * ti;
#pragma unroll
            for (int k4 = 0; k4 < BGP_PB / 4; ++k4) {
                const int k = 4 * k4 + q;
                if (I < row_end && k < pb) {
                    if (I < n) sraw[(size_t)k * ld + I] = b[ti][k4];
                    if (I == n + 1) w[(size_t)(k0 + k) * ld + n + 2] = b[ti][k4];
                }
            }
        }
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int J = 32 * (wv >> 1) + 16 * tj + G::trow(r, q), I = r0 + 16 * ti;
                if (I < row_end && J < pb) w[(size_t)(k0 + J) * ld + I] = acc[tj][ti][r];
            }
}

// Few large items: the launch chain is what takes the time. Measured (fused mean, ms per batch, Cholesky form / this one): fp32 8 x 1024^2
// 0.97 / 0.50, 32 x 512^2 0.50 / 0.26, 64 x 1024^2 1.44 / 1.12, 128 x 512^2 0.63 / 0.46, 512 x 200^2 0.33 / 0.24; fp64 8 x 1024^2 1.15 /
// 0.69, 64 x 1024^2 1.87 / 1.82; beyond about a thousand 64-row blocks per batch the paired rank-128 updates of the Cholesky form win
// (fp32 762 x 512^2 1.78 / 2.14, 256 x 1024^2 3.11 / 3.82 -- and with this form's panels paired as well 1.80 / 1.97 and 3.11 / 3.33: one
// workgroup per 64 rows repeating the sweep costs more than the 256-row workgroups of the Cholesky panel). MATINV_BGP_LDL=0 / 1: never / always.
static bool bgp_ldl_pays(int n, unsigned b)
{
    static const int mode = [] { const char *s = getenv("MATINV_BGP_LDL"); return s ? atoi(s) : -1; }();
    if (mode == 0 || mode == 1) return mode == 1;
    return (size_t)b * ((size_t)(n + BGP_TILE - 1) / BGP_TILE) <= 1024;
}

static bool bgp_pairs_pay(int n, unsigned b);
template <class T>
static void bgp_ldl_chain(T *W, T *Sraw, int n, int ld, unsigned b, int *status, hipStream_t stream)
{
    const int rows = n + 2;
    auto panel = [&](int kb, int scol) {
        const int pb = (n - kb < BGP_PB) ? n - kb : BGP_PB;
        const unsigned chunks = (unsigned)((rows - (kb + pb) + BGP_TILE - 1) / BGP_TILE);  // >= 1: the border rows
        hipLaunchKernelGGL(matinv_bldl_panel<T>, dim3(chunks, b), dim3(BGP_THREADS), 0, stream, W, Sraw, n, ld, rows, kb, status, scol);
    };
    auto update = [&](int kbeg, int kcnt, int jbeg, int jend) {
        const unsigned gx = (unsigned)((jend - jbeg + BGP_TILE - 1) / BGP_TILE), gy = (unsigned)((rows - jbeg + BGP_TILE - 1) / BGP_TILE);
        hipLaunchKernelGGL((matinv_bgp_update<T, true>), dim3(xcd_tile_grid(gx, gy, b)), dim3(BGP_THREADS), 0, stream, W, n, ld, rows, kbeg, kcnt,
                           jbeg, jend, status, gx, gy, b, Sraw);
    };
    if (!bgp_pairs_pay(n, b)) {  // one update per panel (the latency-bound launches)
        for (int k0 = 0; k0 < n; k0 += BGP_PB) {
            panel(k0, 0);
            const int k1 = k0 + BGP_PB;
            if (k1 >= n) break;
            update(k0, BGP_PB, k1, n);
        }
        return;
    }
    // panels in PAIRS, as in the Cholesky form (bgp_pair): the raw copies of both panels side by side in the side buffer
    for (int k0 = 0; k0 < n; k0 += 2 * BGP_PB) {
        panel(k0, 0);
        const int k1 = k0 + BGP_PB;
        if (k1 >= n) break;
        const int k2 = (k1 + BGP_PB < n) ? k1 + BGP_PB : n;
        update(k0, BGP_PB, k1, k2);  // narrow: the columns of the second panel
        panel(k1, BGP_PB);
        if (k2 < n) update(k0, k2 - k0, k2, n);  // wide: both panels at once
    }
}

// one PAIR of 64-column panels starting at column k0 (the second one may be ragged or absent); rows1 / rows2 = number of rows
// the first / second panel's columns can be non-zero in
// Pairs pay when the update launches are throughput-bound (fused pipeline 256 x 1024^2 fp32: 5.35 -> 4.59 ms, 762 x 512^2: 2.84 ->
// 2.47 ms); a handful of items (the 8-item bins of the mixed queue) is bound by the dependent-launch chain, where the longer
// rank-128 tiles cost 3 %: those keep one update per panel.
static bool bgp_pairs_pay(int n, unsigned b)
{
    static const int mode = [] { const char *s = getenv("MATINV_BGP_PAIRS"); return s ? atoi(s) : -1; }();  // 0 / 1: never / always
    if (mode == 0 || mode == 1) return mode == 1;
    const size_t nt = (size_t)(n + BGP_TILE - 1) / BGP_TILE;
    return (size_t)b * nt * nt / 2 >= 4096;
}

template <class T>
static void bgp_pair(T *W, int n, int ld, int k0, int rows1, int rows2, unsigned b, int *status, hipStream_t stream)
{
    auto panel = [&](int kb, int rows_) {
        const int pb = (n - kb < BGP_PB) ? n - kb : BGP_PB;
        const unsigned chunks = (unsigned)((rows_ - (kb + pb) + BGP_THREADS - 1) / BGP_THREADS);  // >= 1: the border rows
        hipLaunchKernelGGL(matinv_bgp_panel<T>, dim3(chunks, b), dim3(BGP_THREADS), 0, stream, W, n, ld, rows_, kb, status);
    };
    auto update = [&](int kcnt, int jbeg, int jend, int rows_) {
        const unsigned gx = (unsigned)((jend - jbeg + BGP_TILE - 1) / BGP_TILE), gy = (unsigned)((rows_ - jbeg + BGP_TILE - 1) / BGP_TILE);
        hipLaunchKernelGGL(matinv_bgp_update<T>, dim3(xcd_tile_grid(gx, gy, b)), dim3(BGP_THREADS), 0, stream, W, n, ld, rows_, k0, kcnt,
                           jbeg, jend, status, gx, gy, b);
    };
    panel(k0, rows1);
    const int k1 = k0 + BGP_PB;
    if (k1 >= n) return;
    const int k2 = (k1 + BGP_PB < n) ? k1 + BGP_PB : n;
    if (!bgp_pairs_pay(n, b)) {  // one update per panel
        update(BGP_PB, k1, n, rows1);
        k0 = k1;
        panel(k1, rows2);
        if (k2 < n) update(k2 - k1, k2, n, rows2);
        return;
    }
    update(BGP_PB, k1, k2, rows1);  // narrow: the columns of the second panel
    panel(k1, rows2);
    if (k2 < n) update(k2 - k0, k2, n, rows2);  // wide: both panels at once
}

// ---- the launch chain of a latency-bound call as a HIP graph ----------------------------------------------------------------------
// A caller that streams the same buffers through the pipeline again and again (the size-binned queue: its gather batches and this
// file's workspaces come from per-stream caches, so a steady stream of flushes repeats the same pointers) issues the same 2 n / 64 + 2
// launches every time; enqueuing them is most of the host's work per flush (about 3.5 us each). The SECOND time a (device, shapes,
// pointers) key is seen its launches are captured into a graph while they are issued; from then on the call is one hipGraphLaunch. A
// key seen once costs nothing extra; 32 graphs are kept (oldest out). MATINV_BGP_GRAPH=0: never.
struct ChainKey {
    int dev, n, dtype, ld;
    unsigned b;
    const void *a, *B, *c, *d, *e;
    void *out, *info, *W, *S, *status;
    bool operator==(const ChainKey &o) const
    {
        return dev == o.dev && n == o.n && dtype == o.dtype && ld == o.ld && b == o.b && a == o.a && B == o.B && c == o.c && d == o.d &&
               e == o.e && out == o.out && info == o.info && W == o.W && S == o.S && status == o.status;
    }
};
struct ChainSlot {
    ChainKey key;
    hipGraphExec_t exec;  // nullptr: seen once, not captured yet
    unsigned long long stamp;
};
static std::mutex g_chain_mu;
static std::vector<ChainSlot> g_chains;
static unsigned long long g_chain_clock = 0;
static bool chain_graphs_on()
{
    static const bool on = [] { const char *s = getenv("MATINV_BGP_GRAPH"); return !(s && s[0] == '0'); }();
    return on;
}
// 0: launch directly; 1: launch directly AND capture (second sighting); 2: *exec is ready, replay it
static int chain_lookup(const ChainKey &k, hipGraphExec_t *exec)
{
    std::lock_guard<std::mutex> lock(g_chain_mu);
    for (auto &sl : g_chains)
        if (sl.key == k) {
            sl.stamp = ++g_chain_clock;
            if (sl.exec) {
                *exec = sl.exec;
                return 2;
            }
            return 1;
        }
    if (g_chains.size() >= 32) {
        size_t old = 0;
        for (size_t i = 1; i < g_chains.size(); ++i)
            if (g_chains[i].stamp < g_chains[old].stamp) old = i;
        if (g_chains[old].exec) (void)hipGraphExecDestroy(g_chains[old].exec);
        g_chains.erase(g_chains.begin() + old);
    }
    g_chains.push_back(ChainSlot{k, nullptr, ++g_chain_clock});
    return 0;
}
static void chain_store(const ChainKey &k, hipGraphExec_t exec)
{
    std::lock_guard<std::mutex> lock(g_chain_mu);
    for (auto &sl : g_chains)
        if (sl.key == k && !sl.exec) {
            sl.exec = exec;
            return;
        }
    (void)hipGraphExecDestroy(exec);  // evicted meanwhile, or another thread was faster
}
void blocked_gp_release_graphs()
{
    std::lock_guard<std::mutex> lock(g_chain_mu);
    for (auto &sl : g_chains)
        if (sl.exec) (void)hipGraphExecDestroy(sl.exec);
    g_chains.clear();
}

template <class T>
hipError_t launch_gp_blocked(int n, const T *As, const T *Bs, const T *Cs, const T *Ds, const T *Es, T *out, size_t batch,
                             int *info, hipStream_t stream)
{
    if (n < 1 || n > 4096) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const bool ldl = bgp_ldl_pays(n, (unsigned)(batch < 65535 ? batch : 65535));
    const int rows = n + 2, ld = bgp_ld<T>(ldl ? rows + 1 : rows);  // block LDL^T: one more row (the raw d-row)
    // chunks: grid.y / grid.z limit and a bounded workspace
    size_t chunk = blocked_workspace_cap() / ((size_t)ld * n * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > 65535) chunk = 65535;
    if (chunk > batch) chunk = batch;
    T *W = nullptr;
    int *status = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&W), chunk * (size_t)ld * n * sizeof(T), stream);
    if (e != hipSuccess) return e;
    e = scratch_alloc(reinterpret_cast<void **>(&status), chunk * sizeof(int), stream);
    if (e != hipSuccess) { (void)scratch_free(W, stream); return e; }
    T *Sraw = nullptr;
    if (ldl) {
        e = scratch_alloc(reinterpret_cast<void **>(&Sraw), chunk * (size_t)ld * (2 * BGP_PB) * sizeof(T), stream);
        if (e != hipSuccess) { (void)scratch_free(W, stream); (void)scratch_free(status, stream); return e; }
    }
    // the launch chain as a graph (see above): one chunk, a stream of its own that is not being captured by the caller
    int graph_mode = 0;
    hipGraphExec_t exec = nullptr;
    ChainKey key{};
    if (ldl && chunk == batch && stream && chain_graphs_on()) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        int dev = 0;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone && hipGetDevice(&dev) == hipSuccess) {
            key = ChainKey{dev, n, (int)sizeof(T), ld, (unsigned)batch, As, Bs, Cs, Ds, Es, out, info, W, Sraw, status};
            graph_mode = chain_lookup(key, &exec);
        } else {
            (void)hipGetLastError();
        }
    }
    if (graph_mode == 2) {
        e = hipGraphLaunch(exec, stream);
        hipError_t e2 = scratch_free(W, stream), e3 = scratch_free(status, stream);
        if (Sraw) (void)scratch_free(Sraw, stream);
        return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
    }
    if (graph_mode == 1 && hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        graph_mode = 0;
    }
    for (size_t first = 0; first < batch; first += chunk) {
        const unsigned b = (unsigned)((batch - first < chunk) ? batch - first : chunk);
        const T *a_ = As + first * n, *B_ = Bs + first * (size_t)n * n, *c_ = Cs + first * n, *d_ = Ds ? Ds + first * n : nullptr;
        hipLaunchKernelGGL(matinv_bgp_init<T>, dim3(64, b), dim3(BGP_THREADS), 0, stream, a_, B_, c_, d_, W, n, ld, status);
        // Panels of 64 columns applied in PAIRS: after the first panel only the next 64 columns are updated (narrow launch),
        // the second panel is factored, and everything behind the pair takes both panels in ONE rank-128 update -- the same
        // number of launches as panel / update per 64 columns, half the read-modify-write traffic on the trailing matrix.
        if (ldl) {
            bgp_ldl_chain<T>(W, Sraw, n, ld, b, status, stream);
        } else {
            for (int k0 = 0; k0 < n; k0 += 2 * BGP_PB) bgp_pair<T>(W, n, ld, k0, rows, rows, b, status, stream);
        }
        hipLaunchKernelGGL(matinv_bgp_finish<T>, dim3(b), dim3(BGP_THREADS), 0, stream, W, (Ds || !Es) ? nullptr : Es + first,
                           out + first, info ? info + first : nullptr, n, ld, status, ldl ? n + 2 : n + 1);
    }
    e = hipGetLastError();
    if (graph_mode == 1) {  // the launches above were recorded, not run: instantiate, keep, run
        hipGraph_t graph = nullptr;
        hipError_t ec = hipStreamEndCapture(stream, &graph);
        if (ec == hipSuccess && e == hipSuccess) ec = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (ec == hipSuccess && e == hipSuccess) {
            e = hipGraphLaunch(exec, stream);
            chain_store(key, exec);
        } else if (e == hipSuccess) {
            e = ec;
        }
    }
    hipError_t e2 = scratch_free(W, stream), e3 = scratch_free(status, stream);
    if (Sraw) (void)scratch_free(Sraw, stream);
    return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
}
template hipError_t launch_gp_blocked<double>(int, const double *, const double *, const double *, const double *,
                                              const double *, double *, size_t, int *, hipStream_t);
template hipError_t launch_gp_blocked<float>(int, const float *, const float *, const float *, const float *, const float *,
                                             float *, size_t, int *, hipStream_t);

// ------------------------------------------------------------------------------------------------------------------
// SPD INVERSE for large n (MATINV_ALGO_CHOLESKY beyond the four-wave kernel, n <= 1024) on the same two kernels: the border
// is the n x n IDENTITY (ld = 2n). After the factorisation border row i holds (L^-1 e_i)^T, i.e. the border block is
// Y = L^-T, and A^-1 = L^-T L^-1 = Y Y^T is one symmetric rank-n product (matinv_binv_syrk). Border row i is still zero in
// the columns of a panel until the panel reaches column i, so the panel and update launches stop at row n + k0 + pb:
// n^3/3 flops each for the factor, the triangular inverse and the product -- the three phases of
// /root/reference/src/inverse_cholesky_cpu.c:17-85 in blocked form. Only the lower triangle of A is read.
template <class T>
__global__ __launch_bounds__(BGP_THREADS) void matinv_binv_init(BatchRef<const T> Ain, size_t first, T *W, int n, int ld, int *status)
{
    const size_t item = blockIdx.y;
    T *w = W + item * (size_t)ld * n;
    const T *A = Ain.at(first + item);
    for (size_t e = (size_t)blockIdx.x * BGP_THREADS + threadIdx.x; e < (size_t)ld * n; e += (size_t)gridDim.x * BGP_THREADS) {
        const int c = (int)(e / ld), r = (int)(e - (size_t)c * ld);
        if (r >= 2 * n || r < c) continue;  // padding of the leading dimension; the strict upper triangle is never read
        w[e] = (r < n) ? ((r >= c) ? A[(size_t)c * n + r] : (T)0) : ((r - n == c) ? (T)1 : (T)0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) status[item] = 0;
}

// X[i][j] = sum_{c >= max(i,j)} Y[i][c] Y[j][c]; one workgroup per 64 x 64 tile with j0 <= i0, mirrored on write
template <class T>
__global__ __launch_bounds__(BGP_THREADS, MATINV_BGP_OCC) void matinv_binv_syrk(const T *W, BatchRef<T> Xout, size_t first, int *info, int n,
                                                                int ld, const int *status, unsigned g, unsigned nb)
{
    const XcdTile tile = xcd_tile_of(blockIdx.x, g, g, nb);
    if (!tile.valid) return;
    typedef TileGeo<T> G;
    __shared__ T Yi[BGP_KS][BGP_LDS], Yj[BGP_KS][BGP_LDS];
    const size_t item = tile.z;
    const int j0 = tile.x * BGP_TILE, i0 = tile.y * BGP_TILE;
    if (j0 > i0) return;
    const int t = threadIdx.x, wv = t >> 6, q = (t >> 4) & 3, c = t & 15;
    const T *w = W + item * (size_t)ld * n;
    T *X = Xout.at(first + item);
    const int bad = status[item];
    typename G::vec4 acc[2][2] = {};
    if (!bad) {  // block-uniform
        // as in matinv_bgp_update: the next slab of 32 columns is fetched into registers while the current one is multiplied
        const int lr = t & 63, lk = t >> 6;
        const bool in_i = i0 + lr < n, in_j = j0 + lr < n;
        const T *wi = w + n + (in_i ? i0 + lr : n - 1), *wj = w + n + (in_j ? j0 + lr : n - 1);
        T pi[BGP_KS / 4], pj[BGP_KS / 4];
        auto fetch = [&](int c0) {
#pragma unroll
            for (int x = 0; x < BGP_KS / 4; ++x) {
                const int cc = c0 + lk + 4 * x;
                const bool cin = cc < n;
                const size_t col = (size_t)(cin ? cc : n - 1) * ld;
                pi[x] = wi[col];  // raw; zeroed when staged (see matinv_bgp_update)
                pj[x] = wj[col];
            }
        };
        fetch(i0);
        for (int c0 = i0; c0 < n; c0 += BGP_KS) {  // Y[i][c] = 0 for c < i: start at the tile's first row
            __syncthreads();
#pragma unroll
            for (int x = 0; x < BGP_KS / 4; ++x) {
                const bool cin = c0 + lk + 4 * x < n;
                Yi[lk + 4 * x][lr] = (cin && in_i) ? pi[x] : (T)0;
                Yj[lk + 4 * x][lr] = (cin && in_j) ? pj[x] : (T)0;
            }
            __syncthreads();
            if (c0 + BGP_KS < n) fetch(c0 + BGP_KS);
            slab_mma<T>(Yj, Yi, wv, q, c, acc);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int J = j0 + 32 * (wv >> 1) + 16 * tj + G::trow(r, q), I = i0 + 32 * (wv & 1) + 16 * ti + c;
                if (I < n && J < n) {
                    const T x = bad ? nan_of<T>() : acc[tj][ti][r];
                    X[(size_t)J * n + I] = x;
                    if (i0 != j0) X[(size_t)I * n + J] = x;  // off-diagonal tiles fill their mirror image
                }
            }
    if (info && tile.x == 0 && tile.y == 0 && t == 0) info[first + item] = bad;
}

bool blocked_inverse_supports(int n) { return n >= 1 && n <= 1024; }

template <class T>
hipError_t launch_chol_blocked(int n, BatchRef<const T> A, BatchRef<T> X, size_t batch, int *info, hipStream_t stream)
{
    if (!blocked_inverse_supports(n)) return hipErrorInvalidValue;
    if (batch == 0) return hipSuccess;
    const int ld = bgp_ld<T>(2 * n);
    // chunks: grid.y / grid.z limit and a bounded workspace
    size_t chunk = blocked_workspace_cap() / ((size_t)ld * n * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > 65535) chunk = 65535;
    if (chunk > batch) chunk = batch;
    T *W = nullptr;
    int *status = nullptr;
    hipError_t e = scratch_alloc(reinterpret_cast<void **>(&W), chunk * (size_t)ld * n * sizeof(T), stream);
    if (e != hipSuccess) return e;
    e = scratch_alloc(reinterpret_cast<void **>(&status), chunk * sizeof(int), stream);
    if (e != hipSuccess) { (void)scratch_free(W, stream); return e; }
    for (size_t first = 0; first < batch; first += chunk) {
        const unsigned b = (unsigned)((batch - first < chunk) ? batch - first : chunk);
        hipLaunchKernelGGL(matinv_binv_init<T>, dim3(64, b), dim3(BGP_THREADS), 0, stream, A, first, W, n, ld, status);
        // border rows beyond n + (end of a panel) are still zero in that panel's columns: each launch stops there
        for (int k0 = 0; k0 < n; k0 += 2 * BGP_PB) {
            const int e1 = (k0 + BGP_PB < n) ? k0 + BGP_PB : n, e2 = (k0 + 2 * BGP_PB < n) ? k0 + 2 * BGP_PB : n;
            bgp_pair<T>(W, n, ld, k0, n + e1, n + e2, b, status, stream);
        }
        const unsigned g = (unsigned)((n + BGP_TILE - 1) / BGP_TILE);
        hipLaunchKernelGGL(matinv_binv_syrk<T>, dim3(xcd_tile_grid(g, g, b)), dim3(BGP_THREADS), 0, stream, W, X, first, info, n, ld, status,
                           g, b);
    }
    e = hipGetLastError();
    hipError_t e2 = scratch_free(W, stream), e3 = scratch_free(status, stream);
    return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
}
template hipError_t launch_chol_blocked<double>(int, BatchRef<const double>, BatchRef<double>, size_t, int *, hipStream_t);
template hipError_t launch_chol_blocked<float>(int, BatchRef<const float>, BatchRef<float>, size_t, int *, hipStream_t);

}  // namespace matinv
