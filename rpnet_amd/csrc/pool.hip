// 2x2 max-pooling, nearest-x2 up-sampling adjoint and mask average pooling (HBM-bound,
// NHWC float4).  Replaces nn.MaxPool2d(2,2) (net/unet.py:397,442-455), the autograd of
// nn.Upsample(scale_factor=2) (net/modules.py:66) and F.avg_pool2d(mask, scale)
// (net/rp_net.py:269-272,311).
#include "common.h"

namespace rpnet {

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ z, float* __restrict__ out,
                                                            int N, int Ho, int Wo, int C4) {
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int W = Wo * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const f32x4* src = reinterpret_cast<const f32x4*>(z) + (((size_t)n * Ho * 2 + oy * 2) * W + ox * 2) * C4 + c4;
        const f32x4 a = src[0], b = src[C4], c = src[(size_t)W * C4], d = src[(size_t)W * C4 + C4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], d[k]));
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// dz[window] = (first max of the window ? dpool : 0) + skip
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dpool,
                                                            const float* __restrict__ skip, float* __restrict__ dz,
                                                            int N, int Ho, int Wo, int C4) {
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int W = Wo * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const size_t o00 = (((size_t)n * Ho * 2 + oy * 2) * W + ox * 2) * C4 + c4;
        const size_t offs[4] = {o00, o00 + C4, o00 + (size_t)W * C4, o00 + (size_t)W * C4 + C4};
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const f32x4*>(z)[offs[q]];
        const f32x4 g = reinterpret_cast<const f32x4*>(dpool)[i];
        f32x4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int best = 0; float bv = v[0][k];
#pragma unroll
            for (int q = 1; q < 4; ++q) if (v[q][k] > bv) { bv = v[q][k]; best = q; }
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q][k] = (q == best) ? g[k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (skip) r[q] += reinterpret_cast<const f32x4*>(skip)[offs[q]];
            reinterpret_cast<f32x4*>(dz)[offs[q]] = r[q];
        }
    }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dyu, float* __restrict__ dx,
                                                             int N, int Ho, int Wo, int C4) {
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int W = Wo * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const f32x4* src = reinterpret_cast<const f32x4*>(dyu) + (((size_t)n * Ho * 2 + oy * 2) * W + ox * 2) * C4 + c4;
        reinterpret_cast<f32x4*>(dx)[i] = (src[0] + src[C4]) + (src[(size_t)W * C4] + src[(size_t)W * C4 + C4]);
    }
}

__global__ void mask_avgpool_kernel(const float* __restrict__ m, float* __restrict__ out, int B, int H, int W, int s) {
    const int h = H / s, w = W / s;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * h * w) return;
    const int ox = i % w, oy = (i / w) % h, b = i / (w * h);
    const float* p = m + ((size_t)b * H + oy * s) * W + ox * s;
    float acc = 0.f;
    for (int dy = 0; dy < s; ++dy)
        for (int dx = 0; dx < s; ++dx) acc += p[dy * W + dx];
    out[i] = acc / (float)(s * s);
}

static int pool_grid(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace rpnet

extern "C" int rpnet_maxpool2_fwd(const float* z, float* out, int N, int H, int W, int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(z && out, RPNET_ERR_ARG, "maxpool2_fwd: null pointer");
    RPNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, RPNET_ERR_SHAPE, "maxpool2_fwd: H=%d W=%d C=%d", H, W, C);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, z, out, N, H / 2, W / 2, C / 4);
    return check_launch("maxpool2_fwd");
}

extern "C" int rpnet_maxpool2_bwd(const float* z, const float* dpool, const float* skip, float* dz, int N, int H, int W,
                                  int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(z && dpool && dz, RPNET_ERR_ARG, "maxpool2_bwd: null pointer");
    RPNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, RPNET_ERR_SHAPE, "maxpool2_bwd: H=%d W=%d C=%d", H, W, C);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, z, dpool, skip, dz, N, H / 2, W / 2, C / 4);
    return check_launch("maxpool2_bwd");
}

extern "C" int rpnet_upsample2_bwd(const float* dyu, float* dx, int N, int H, int W, int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(dyu && dx, RPNET_ERR_ARG, "upsample2_bwd: null pointer");
    RPNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, RPNET_ERR_SHAPE, "upsample2_bwd: H=%d W=%d C=%d", H, W, C);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, dyu, dx, N, H / 2, W / 2, C / 4);
    return check_launch("upsample2_bwd");
}

extern "C" int rpnet_mask_avgpool(const float* mask, float* out, int B, int H, int W, int scale, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(mask && out, RPNET_ERR_ARG, "mask_avgpool: null pointer");
    RPNET_REQUIRE(scale >= 1 && H % scale == 0 && W % scale == 0, RPNET_ERR_SHAPE, "mask_avgpool: H=%d W=%d scale=%d", H, W, scale);
    const int total = B * (H / scale) * (W / scale);
    hipLaunchKernelGGL(mask_avgpool_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, mask, out, B, H, W, scale);
    return check_launch("mask_avgpool");
}
