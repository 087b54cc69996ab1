// Device helpers shared by the matcher kernels (matcher.hip) and the fused refinement-loop glue (refine.hip): the bilinear taps of
// F.interpolate(align_corners=False) and the lane-group reduction of the cosine match.
#pragma once
#include "common.h"

namespace rpnet {

// source taps of F.interpolate(mode='bilinear', align_corners=False) for destination index d
__device__ __forceinline__ void bl_taps(int d, float rscale, int in_size, int& i0, int& i1, float& w0, float& w1) {
    float src = rscale * ((float)d + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
    w1 = src - (float)i0;
    w0 = 1.f - w1;
}
// weight with which destination d reads source s
__device__ __forceinline__ float bl_weight(int d, int s, float rscale, int in_size) {
    int i0, i1; float w0, w1;
    bl_taps(d, rscale, in_size, i0, i1, w0, w1);
    return (s == i0 ? w0 : 0.f) + (s == i1 ? w1 : 0.f);
}

constexpr float kCosEps = 1e-8f;
constexpr int kMaxK = 4;

template <int L>  // lanes per pixel = C/4, power of two <= 64
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dproto[b][k][:] = sum over the nblk partial rows dpart[b][0 .. nblk)[k][:] (matcher.hip: cosine_dproto_final), host-side launcher
void launch_cosine_dproto_final(const float* dpart, float* dproto, int B, int nblk, int K, int C, hipStream_t stream);

}  // namespace rpnet
