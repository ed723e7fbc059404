// Shared pieces of the split-fp16 ("f16x3") arithmetic: s*x = hi + lo with hi = fp16(s*x), lo = fp16(s*x - hi), the
// power-of-two pre-scale chosen from a tensor's abs-max, and the pixel-blocked gradient layout that wgrad consumes
// (conv_f16_kernels.hip) and that the BN backward apply kernel can emit directly (elementwise_kernels.hip).
#pragma once
#include "dcn_common.h"

namespace dcnsplit {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pow2_scale(float absmax) {
    // largest power of two s with s * absmax <= 4096 (fp16 max is 65504: 16x headroom for the fp32->fp16 rounding), kept
    // inside 2^+-120 so that s and 1/s stay finite for denormal / overflowing abs-max values
    if (!(absmax > 0.f)) return 1.f;
    const float e = fminf(fmaxf(floorf(log2f(4096.f / absmax)), -120.f), 120.f);
    return exp2f(e);
}

// (x0, x1) -> packed hi = rn16(x), lo = rn16(x - hi): one v_cvt_pk_f16_f32 per pair for hi, two v_cvt_f32_f16 back,
// two subtractions, one v_cvt_pk_f16_f32 for lo (vector conversions keep hipcc from converting every element twice)
__device__ __forceinline__ void split2(float x0, float x1, h2& hi, h2& lo) {
    f32x2 v = {x0, x1};
    hi = __builtin_convertvector(v, h2);
    const f32x2 back = __builtin_convertvector(hi, f32x2);
    lo = __builtin_convertvector(v - back, h2);
}

__device__ __forceinline__ void split4_unscaled(float4 v, h4& hi, h4& lo) {
    h2 a, b, c, d;
    split2(v.x, v.y, a, b);
    split2(v.z, v.w, c, d);
    hi = h4{a[0], a[1], c[0], c[1]};
    lo = h4{b[0], b[1], d[0], d[1]};
}

__device__ __forceinline__ void split4(float4 v, float s, h4& hi, h4& lo) {
    split4_unscaled(make_float4(v.x * s, v.y * s, v.z * s, v.w * s), hi, lo);
}

// Writes the 4-pixel x 4-channel micro-tile v[pixel][channel] (already multiplied by nothing: `s` is applied here) of pixel
// quad q, channel quad cq into the pixel-blocked gradient tensor dq[q][4 sub-planes][c4n][2 channels x 4 pixels]:
// sub-plane 0 / 1 = hi of channels (0,1) / (2,3), 2 / 3 = the lo parts.
__device__ __forceinline__ void store_blocked_quad(u32x4* __restrict__ dq, int64_t q, int cq, int c4n, const float (&v)[4][4],
                                                   float s) {
    u32x2 hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h4 a, b;
        split4(make_float4(v[0][e], v[1][e], v[2][e], v[3][e]), s, a, b);
        hi[e] = __builtin_bit_cast(u32x2, a);
        lo[e] = __builtin_bit_cast(u32x2, b);
    }
    u32x4* o = dq + q * 4 * c4n + cq;
    o[0] = u32x4{hi[0][0], hi[0][1], hi[1][0], hi[1][1]};
    o[c4n] = u32x4{hi[2][0], hi[2][1], hi[3][0], hi[3][1]};
    o[2 * c4n] = u32x4{lo[0][0], lo[0][1], lo[1][0], lo[1][1]};
    o[3 * c4n] = u32x4{lo[2][0], lo[2][1], lo[3][0], lo[3][1]};
}

}  // namespace dcnsplit
