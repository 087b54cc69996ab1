// Split of fp32 values into bf16 planes: x = h + m (+ l), h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
// (round to nearest even; exact with three planes).  Shared by every kernel that produces a split operand.
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

// v[8] -> NP planes of 8 bf16 (16 bytes each)
template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[NP]) {
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        hb[q] = bf16_bits(v[q]);
        const float r1 = v[q] - bf16_val(hb[q]);
        mb[q] = bf16_bits(r1);
        if (NP > 2) lb[q] = bf16_bits(r1 - bf16_val(mb[q]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[0][q] = hb[2 * q] | (hb[2 * q + 1] << 16);
        out[1][q] = mb[2 * q] | (mb[2 * q + 1] << 16);
        if (NP > 2) out[NP - 1][q] = lb[2 * q] | (lb[2 * q + 1] << 16);
    }
}

}  // namespace rpnet
