// fp32 -> three bf16 terms, exactly: x = hi + mid + lo with hi = x truncated to its upper 16 bits, mid = (x - hi) truncated,
// lo = x - hi - mid (at most 8 significant bits left: exact).  3 x 8 significand bits = the 24 of an fp32 value.
// Shared by the contractions that run on the bf16 matrix pipe (gemm_tn_x3.hip, gemm_rows_x3.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tp3d {

// four values -> four halfwords per plane (v_perm_b32 packs the upper halves of two dwords).
// Non-finite values need no special case: for x = +-inf or NaN, x - hi is NaN, so mid and lo are NaN and every output that
// involves x becomes NaN -- the outputs the fp32 product makes non-finite as well (inf * b or NaN * b is never finite); the
// finite / non-finite pattern of the result is that of the fp32 kernel, an infinity may read NaN.
// Truncation never rounds up, so values next to FLT_MAX do not overflow in the split.
__device__ __forceinline__ void x3_split(float4 v, uint2 &hi, uint2 &mid, uint2 &lo)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned r1[4], r2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a = x[c] - __uint_as_float(__float_as_uint(x[c]) & 0xffff0000u);
        const float b = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u);  // at most 8 significant bits left
        r1[c] = __float_as_uint(a);
        r2[c] = __float_as_uint(b);
    }
    constexpr unsigned UP = 0x07060302u;  // bytes 2,3 of the second source, then bytes 2,3 of the first
    hi = make_uint2(__builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), UP),
                    __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), UP));
    mid = make_uint2(__builtin_amdgcn_perm(r1[1], r1[0], UP), __builtin_amdgcn_perm(r1[3], r1[2], UP));
    lo = make_uint2(__builtin_amdgcn_perm(r2[1], r2[0], UP), __builtin_amdgcn_perm(r2[3], r2[2], UP));
}

}  // namespace tp3d
