// wave_util.hpp -- wave-wide helpers shared by the lane-per-row pivot searches (row_kernels.hip, tilep_impl.hpp):
// DPP max-reduction over the 64 lanes, magnitude keys (top 32 bits of |x|), lane -> scalar broadcast, full-accuracy reciprocal.
#pragma once
#include "common.hpp"

namespace matinv {

constexpr int DPPR_QUAD_XOR1 = 0xB1, DPPR_QUAD_XOR2 = 0x4E, DPPR_ROW_MIRROR = 0x140, DPPR_ROW_HALF_MIRROR = 0x141;

template <int CTRL>
__device__ __forceinline__ unsigned dppu(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// maximum over the 64 lanes, wave-uniform (SGPR). Six DPP-modified v_max_u32 (quad swaps, half-row and row mirrors: every
// lane of a row of 16 then holds the row's maximum; row_bcast:15 / row_bcast:31 carry it across the rows into lane 63) and
// ONE v_readlane. Written through __builtin_amdgcn_update_dpp hipcc emits v_mov + v_mov_dpp + v_max per step and four
// v_readlane + a scalar max tree: 19 VALU instructions instead of 7. The s_nop 1 are the two wait states a DPP source
// needs after the VALU instruction that wrote it (hipcc pads nothing inside an asm block).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    unsigned m;
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %[v], %[v], %[v] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_readlane_b32 %[m], %[v], 63"
                 : [v] "+v"(v), [m] "=s"(m));
    return m;
}

__device__ __forceinline__ unsigned magkey(double v) { return (unsigned)(__double_as_longlong(v) >> 32) & 0x7fffffffu; }
__device__ __forceinline__ unsigned magkey(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ bool key_bad(double, unsigned k) { return k == 0u || k >= 0x7ff00000u; }
__device__ __forceinline__ bool key_bad(float, unsigned k) { return k == 0u || k >= 0x7f800000u; }

// value of lane `p` (wave-uniform) as a scalar
__device__ __forceinline__ double lane_value(double v, int p)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, p), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), p);
    return __longlong_as_double(((long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane_value(float v, int p)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), p));
}

__device__ __forceinline__ double rcp_full(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float rcp_full(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ double fmat(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmat(float a, float b, float c) { return __builtin_fmaf(a, b, c); }


}  // namespace matinv
