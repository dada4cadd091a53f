// slab_mma.hpp -- the 64 x 64 tile product of the blocked large-n paths on the matrix cores, one LDS-staged slab at a time
// (blocked_gp_kernels.hip: Cholesky trailing update and the Y Y^T product).
#pragma once
#include "tile_common.hpp"

namespace matinv {

#ifndef MATINV_SLAB_KS
#define MATINV_SLAB_KS 16
#endif
constexpr int SLAB_KS = MATINV_SLAB_KS;   // columns per slab
constexpr int SLAB_LDS = 80;  // row stride of a slab S[k][row] in LDS: the four k-groups of an MFMA operand read land in disjoint banks

// One slab (SLAB_KS columns, staged in LDS as S[k][row]) of the 64 x 64 tile product D[J, I] += sum_k Sj[k][J] Si[k][I] on the matrix
// cores: wavefront wv owns the 32 x 32 part (J half wv >> 1, I half wv & 1) as 2 x 2 MFMA tiles. The tile is computed TRANSPOSED
// (the MFMA's A operand carries the J side): in the accumulator layout the 16 lanes of a group then hold 16 consecutive ROWS I
// of one column J -- the contiguous direction of the column-major working copies, so the tile itself is read and written in
// 128-byte segments. Per 4 columns a wavefront reads 4 x 512 B from LDS for 4 MFMAs; the 4 x 4-per-thread vector-ALU form this
// replaces read 4 KB per 16 FMAs and spent its time on the LDS (PMC, SPD inverse 1024^2: 52 % of the LDS cycles bank conflicts,
// vector ALU 10 % busy). Element (J, I) of the result: J = 32 (wv >> 1) + 16 tj + TileGeo::trow(r, q), I = 32 (wv & 1) + 16 ti + c
// in acc[tj][ti][r] of lane (q, c).
template <class T>
__device__ __forceinline__ void slab_mma(const T (*Sj)[SLAB_LDS], const T (*Si)[SLAB_LDS], int wv, int q, int c,
                                         typename TileGeo<T>::vec4 (&acc)[2][2])
{
    typedef TileGeo<T> G;
    const int jb = 32 * (wv >> 1) + c, ib = 32 * (wv & 1) + c;
#pragma unroll
    for (int k4 = 0; k4 < SLAB_KS; k4 += 4) {  // columns beyond the slab's real width are zero-filled by whoever staged it
        const T a0 = Sj[k4 + q][jb], a1 = Sj[k4 + q][jb + 16], b0 = Si[k4 + q][ib], b1 = Si[k4 + q][ib + 16];
        acc[0][0] = G::mfma(a0, b0, acc[0][0]);
        acc[0][1] = G::mfma(a0, b1, acc[0][1]);
        acc[1][0] = G::mfma(a1, b0, acc[1][0]);
        acc[1][1] = G::mfma(a1, b1, acc[1][1]);
    }
}

// XCD-aware tile order for the tile kernels of the blocked paths: workgroup h of a 1-D grid runs on XCD h % 8 (round-robin dispatch), and
// each XCD has its own L2. Tile h is therefore taken from the h % 8-th EIGHTH of the tile list (x fastest, then y, then item),
// so that the tiles of one tile row of one matrix -- which share their operand panel -- run on ONE XCD at about the same time
// and fetch it into that L2 once instead of into all eight.
struct XcdTile { unsigned x, y, z; bool valid; };
__device__ __forceinline__ XcdTile xcd_tile_of(unsigned h, unsigned gx, unsigned gy, unsigned b)
{
    const unsigned total = gx * gy * b, per = (total + 7) / 8;
    const unsigned lid = (h % 8) * per + h / 8;
    XcdTile t;
    t.valid = h / 8 < per && lid < total;
    t.x = lid % gx;
    t.y = (lid / gx) % gy;
    t.z = lid / (gx * gy);
    return t;
}
inline unsigned xcd_tile_grid(unsigned gx, unsigned gy, unsigned b) { return ((gx * gy * b + 7) / 8) * 8; }

}  // namespace matinv
