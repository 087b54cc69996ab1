// Split of fp32 values into 16-bit planes.  Shared by every kernel that produces a split operand.
//   NP = 3: bf16 planes, x = h + m + l, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)  (round to nearest even; exact)
//   NP = 2: fp16 planes, x = h + l,     h = fp16(x), l = fp16(x - h): 22 significand bits while the residual stays in
//           fp16's normal range, an absolute floor of 2^-25 below it.  The CALLER hands in x / s with a power-of-two
//           tensor scale s that keeps |x / s| <= 2^15 (fp16 overflows at 65504) and puts typical values well above
//           2^-3; the consumer multiplies the accumulator by s (exact).  NP = 2 therefore exists only where a rigorous
//           bound of the tensor is known (BatchNorm outputs and gradients, weights); everything else uses NP = 3.
//   NP = 1: ONE fp16 plane h = fp16(x / s) with the same tensor scales — plain fp16 operands (11 significand bits), one
//           MFMA per multiply-add, fp32 accumulation: the reduced-precision arithmetic of BASELINE configs[4]
//           (RPNET_CONV_MATH=f16), NOT fp32-equivalent.
#pragma once
#include "common.h"

namespace rpnet {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

__device__ __forceinline__ unsigned bf16_bits(float x) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);   // v_cvt_pk_bf16_f32: round to nearest even
}
__device__ __forceinline__ float bf16_val(unsigned b) { return __uint_as_float(b << 16); }

__device__ __forceinline__ unsigned f16_bits(float x) {
    return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x);   // v_cvt_f16_f32: round to nearest even
}
__device__ __forceinline__ float f16_val(unsigned b) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)b);
}

// power of two >= b, times 2^-15: the tensor scale that maps a bound b of |x| to <= 2^15 (b = 0 or tiny -> 2^-100)
__device__ __forceinline__ float pow2_scale(float b) {
    const unsigned u = __float_as_uint(fmaxf(b, 0.f));
    int e = (int)((u >> 23) & 255u) - 127 + ((u & 0x7fffffu) ? 1 : 0);
    e = e - 15;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return __uint_as_float((unsigned)(e + 127) << 23);
}

// v[8] -> NP planes of 8 16-bit values (16 bytes each)
template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[NP]) {
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (NP <= 2) {
            hb[q] = f16_bits(v[q]);
            if (NP == 2) mb[q] = f16_bits(v[q] - f16_val(hb[q]));
            continue;
        }
        hb[q] = bf16_bits(v[q]);
        const float r1 = v[q] - bf16_val(hb[q]);
        mb[q] = bf16_bits(r1);
        if (NP > 2) lb[q] = bf16_bits(r1 - bf16_val(mb[q]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[0][q] = hb[2 * q] | (hb[2 * q + 1] << 16);
        if (NP > 1) out[NP > 1 ? 1 : 0][q] = mb[2 * q] | (mb[2 * q + 1] << 16);
        if (NP > 2) out[NP - 1][q] = lb[2 * q] | (lb[2 * q + 1] << 16);
    }
}

// partial products of the planes, smallest first: NP = 3: lh hl mm mh hm hh (plane 0 = h, 1 = m, 2 = l), NP = 2: lh hl hh
// (plane 1 = l), NP = 1: hh
template <int NP> constexpr int nprod() { return NP == 3 ? 6 : (NP == 2 ? 3 : 1); }
template <int NP> constexpr int prod_a(int q) {
    return NP == 3 ? (q == 0 ? 2 : ((q == 2 || q == 3) ? 1 : 0)) : (NP == 2 ? (q == 0 ? 1 : 0) : 0);
}
template <int NP> constexpr int prod_b(int q) {
    return NP == 3 ? (q == 1 ? 2 : ((q == 2 || q == 4) ? 1 : 0)) : (NP == 2 ? (q == 1 ? 1 : 0) : 0);
}

// one partial product of split planes on the matrix pipe: NP = 3 -> bf16 planes, NP <= 2 -> fp16 planes
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
template <int NP>
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (NP <= 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

}  // namespace rpnet
