// 2x2 max-pooling, nearest-x2 up-sampling adjoint and mask average pooling (HBM-bound,
// NHWC float4).  Replaces nn.MaxPool2d(2,2) (net/unet.py:397,442-455), the autograd of
// nn.Upsample(scale_factor=2) (net/modules.py:66) and F.avg_pool2d(mask, scale)
// (net/rp_net.py:269-272,311).
#include "common.h"

namespace rpnet {

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ z, float* __restrict__ out,
                                                            int N, int Ho, int Wo, int C4, const FastDiv fC, const FastDiv fW, const FastDiv fH) {
    RPNET_PASS_PRIORITY();
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int W = Wo * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        unsigned p, py, pn;
        const int c4 = (int)fC.divmod((unsigned)i, p);
        const int ox = (int)fW.divmod(p, py);
        const int oy = (int)fH.divmod(py, pn);
        const int n = (int)pn;
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
                                                            int N, int Ho, int Wo, int C4, const FastDiv fC, const FastDiv fW, const FastDiv fH) {
    RPNET_PASS_PRIORITY();
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int W = Wo * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        unsigned p, py, pn;
        const int c4 = (int)fC.divmod((unsigned)i, p);
        const int ox = (int)fW.divmod(p, py);
        const int oy = (int)fH.divmod(py, pn);
        const int n = (int)pn;
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
                                                             int N, int Ho, int Wo, int C4, const FastDiv fC, const FastDiv fW, const FastDiv fH) {
    RPNET_PASS_PRIORITY();
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int W = Wo * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        unsigned p, py, pn;
        const int c4 = (int)fC.divmod((unsigned)i, p);
        const int ox = (int)fW.divmod(p, py);
        const int oy = (int)fH.divmod(py, pn);
        const int n = (int)pn;
        const f32x4* src = reinterpret_cast<const f32x4*>(dyu) + (((size_t)n * Ho * 2 + oy * 2) * W + ox * 2) * C4 + c4;
        reinterpret_cast<f32x4*>(dx)[i] = (src[0] + src[C4]) + (src[(size_t)W * C4] + src[(size_t)W * C4 + C4]);
    }
}

__global__ void mask_avgpool_kernel(const float* __restrict__ m, float* __restrict__ out, int B, int H, int W, int s) {
    RPNET_PASS_PRIORITY();
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

// MaxPool2d(3, stride, padding=1): -inf padding, first maximum in (ky, kx) scan order
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int N, int H,
                                                            int W, int Ho, int Wo, int C4, int stride, const FastDiv fC, const FastDiv fW, const FastDiv fH) {
    RPNET_PASS_PRIORITY();
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        unsigned p, py, pn;
        const int c4 = (int)fC.divmod((unsigned)i, p);
        const int ox = (int)fW.divmod(p, py);
        const int oy = (int)fH.divmod(py, pn);
        const int n = (int)pn;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = reinterpret_cast<const f32x4*>(z)[(((size_t)n * H + iy) * W + ix) * C4 + c4];
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
        reinterpret_cast<f32x4*>(out)[i] = m;
    }
}

// gather form: input pixel (iy, ix) receives dpool of every window whose FIRST maximum it is
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dpool,
                                                            float* __restrict__ dz, int N, int H, int W, int Ho, int Wo,
                                                            int C4, int stride) {
    RPNET_PASS_PRIORITY();
    const size_t total = (size_t)N * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        const f32x4 me = reinterpret_cast<const f32x4*>(z)[i];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int oy_hi = min(Ho - 1, (iy + 1) / stride), ox_hi = min(Wo - 1, (ix + 1) / stride);
        for (int oy = max(0, oy_hi - 2); oy <= oy_hi; ++oy) {       // candidate windows; coverage checked below
            for (int ox = max(0, ox_hi - 2); ox <= ox_hi; ++ox) {
                const int y0 = oy * stride - 1, x0 = ox * stride - 1;
                if (iy < y0 || iy > y0 + 2 || ix < x0 || ix > x0 + 2) continue;
                const int myrank = (iy - y0) * 3 + (ix - x0);
                bool first[4] = {true, true, true, true};
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y0 + ky;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x0 + kx;
                        if (xx < 0 || xx >= W) continue;
                        const int rank = ky * 3 + kx;
                        if (rank == myrank) continue;
                        const f32x4 v = reinterpret_cast<const f32x4*>(z)[(((size_t)n * H + yy) * W + xx) * C4 + c4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)   // beaten by a larger value, or by an equal one scanned earlier
                            if (v[k] > me[k] || (v[k] == me[k] && rank < myrank)) first[k] = false;
                    }
                }
                const f32x4 g = reinterpret_cast<const f32x4*>(dpool)[(((size_t)n * Ho + oy) * Wo + ox) * C4 + c4];
#pragma unroll
                for (int k = 0; k < 4; ++k) if (first[k]) acc[k] += g[k];
            }
        }
        reinterpret_cast<f32x4*>(dz)[i] = acc;
    }
}

// dy = dz * [z > 0]; db partial column sums: partial[blk][C]
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                             float* __restrict__ dy, double* __restrict__ partial, size_t P,
                                                             int C) {
    RPNET_PASS_PRIORITY();
    __shared__ double red[256 * 4];
    const int t = threadIdx.x, C4 = C / 4, rows_it = 256 / C4;
    const int tc = t % C4, tr = t / C4;
    double s[4] = {0, 0, 0, 0};
    if (tr < rows_it)
        for (size_t r = (size_t)blockIdx.x * rows_it + tr; r < P; r += (size_t)gridDim.x * rows_it) {
            f32x4 g = reinterpret_cast<const f32x4*>(dz + r * C)[tc];
            if (z) {
                const f32x4 zv = reinterpret_cast<const f32x4*>(z + r * C)[tc];
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = zv[k] > 0.f ? g[k] : 0.f;
            }
            reinterpret_cast<f32x4*>(dy + r * C)[tc] = g;
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += g[k];
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[t * 4 + k] = s[k];
    __syncthreads();
    if (tr == 0) {
        for (int rr = 1; rr < rows_it; ++rr)
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += red[(rr * C4 + tc) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) partial[(size_t)blockIdx.x * C + tc * 4 + k] = s[k];
    }
}

__global__ __launch_bounds__(64) void bias_grad_final(const double* __restrict__ partial, float* __restrict__ db, int nblk, int C) {
    RPNET_PASS_PRIORITY();
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0;
    for (int b = lane; b < nblk; b += 64) s += partial[(size_t)b * C + c];
    s = wave_sum(s);
    if (lane == 0) db[c] = (float)s;
}

constexpr int kBiasBlocks = 512;

static int pool_grid(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace rpnet

extern "C" size_t rpnet_bias_relu_bwd_workspace_bytes(int C) { return (size_t)rpnet::kBiasBlocks * C * sizeof(double); }

extern "C" int rpnet_bias_relu_bwd(const float* dz, const float* z, float* dy, float* db, size_t P, int C, void* workspace,
                                   size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(dz && dy && db && workspace, RPNET_ERR_ARG, "bias_relu_bwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && C / 4 <= 256, RPNET_ERR_SHAPE, "bias_relu_bwd: C=%d", C);
    RPNET_REQUIRE(workspace_bytes >= rpnet_bias_relu_bwd_workspace_bytes(C), RPNET_ERR_WORKSPACE, "bias_relu_bwd: workspace");
    const int rows_it = 256 / (C / 4);
    int nb = (int)((P + rows_it - 1) / rows_it);
    if (nb > kBiasBlocks) nb = kBiasBlocks;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3(nb), dim3(256), 0, s, dz, z, dy, (double*)workspace, P, C);
    hipLaunchKernelGGL(bias_grad_final, dim3(C), dim3(64), 0, s, (const double*)workspace, db, nb, C);
    return check_launch("bias_relu_bwd");
}

extern "C" int rpnet_maxpool3_fwd(const float* z, float* out, int N, int H, int W, int C, int stride, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(z && out, RPNET_ERR_ARG, "maxpool3_fwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && (stride == 1 || stride == 2), RPNET_ERR_SHAPE, "maxpool3_fwd: C=%d stride=%d", C, stride);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    RPNET_REQUIRE(total < kIndex32, RPNET_ERR_SHAPE, "maxpool3_fwd: %zu elements do not fit the 32-bit index arithmetic", total);
    hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, z, out, N, H, W, Ho, Wo, C / 4, stride,
                       FastDiv(C / 4), FastDiv(Wo), FastDiv(Ho));
    return check_launch("maxpool3_fwd");
}

extern "C" int rpnet_maxpool3_bwd(const float* z, const float* dpool, float* dz, int N, int H, int W, int C, int stride,
                                  rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(z && dpool && dz, RPNET_ERR_ARG, "maxpool3_bwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && (stride == 1 || stride == 2), RPNET_ERR_SHAPE, "maxpool3_bwd: C=%d stride=%d", C, stride);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t total = (size_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, z, dpool, dz, N, H, W, Ho, Wo, C / 4, stride);
    return check_launch("maxpool3_bwd");
}

extern "C" int rpnet_maxpool2_fwd(const float* z, float* out, int N, int H, int W, int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(z && out, RPNET_ERR_ARG, "maxpool2_fwd: null pointer");
    RPNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, RPNET_ERR_SHAPE, "maxpool2_fwd: H=%d W=%d C=%d", H, W, C);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    RPNET_REQUIRE(total < kIndex32, RPNET_ERR_SHAPE, "maxpool2_fwd: %zu elements do not fit the 32-bit index arithmetic", total);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, z, out, N, H / 2, W / 2, C / 4, FastDiv(C / 4), FastDiv(W / 2), FastDiv(H / 2));
    return check_launch("maxpool2_fwd");
}

extern "C" int rpnet_maxpool2_bwd(const float* z, const float* dpool, const float* skip, float* dz, int N, int H, int W,
                                  int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(z && dpool && dz, RPNET_ERR_ARG, "maxpool2_bwd: null pointer");
    RPNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, RPNET_ERR_SHAPE, "maxpool2_bwd: H=%d W=%d C=%d", H, W, C);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    RPNET_REQUIRE(total < kIndex32, RPNET_ERR_SHAPE, "maxpool2_bwd: %zu elements do not fit the 32-bit index arithmetic", total);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, z, dpool, skip, dz, N, H / 2, W / 2, C / 4, FastDiv(C / 4), FastDiv(W / 2), FastDiv(H / 2));
    return check_launch("maxpool2_bwd");
}

extern "C" int rpnet_upsample2_bwd(const float* dyu, float* dx, int N, int H, int W, int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(dyu && dx, RPNET_ERR_ARG, "upsample2_bwd: null pointer");
    RPNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, RPNET_ERR_SHAPE, "upsample2_bwd: H=%d W=%d C=%d", H, W, C);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    RPNET_REQUIRE(total < kIndex32, RPNET_ERR_SHAPE, "upsample2_bwd: %zu elements do not fit the 32-bit index arithmetic", total);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, dyu, dx, N, H / 2, W / 2, C / 4, FastDiv(C / 4), FastDiv(W / 2), FastDiv(H / 2));
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

// out = sum of n (<= 16) equally sized fp32 tensors: autograd's fan-in of a tensor with many consumers (the query
// features feed 2 T convolutions, net/rp_net.py:275,283 inside the refinement loop :281-312) in ONE pass — (n + 1) x the
// tensor in HBM traffic instead of the 3 (n - 1) x of a chain of pairwise adds.
namespace rpnet {
struct SumSources {
    const float* p[16];
};
__global__ __launch_bounds__(256) void sum_n_kernel(const SumSources src, const int n, float* __restrict__ out, const size_t n4,
                                                     const size_t numel) {
    RPNET_PASS_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 acc = reinterpret_cast<const f32x4*>(src.p[0])[i];
        for (int k = 1; k < n; ++k) acc += reinterpret_cast<const f32x4*>(src.p[k])[i];
        reinterpret_cast<f32x4*>(out)[i] = acc;
    }
    if (blockIdx.x == 0) {
        const size_t e = n4 * 4 + threadIdx.x;       // tail (< 4 elements)
        if (e < numel) {
            float acc = src.p[0][e];
            for (int k = 1; k < n; ++k) acc += src.p[k][e];
            out[e] = acc;
        }
    }
}
}  // namespace rpnet

extern "C" int rpnet_sum_n(const float* const* srcs, int n, float* out, size_t numel, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(srcs && out, RPNET_ERR_ARG, "sum_n: null pointer");
    RPNET_REQUIRE(n >= 1 && n <= 16, RPNET_ERR_SHAPE, "sum_n: n=%d (1..16)", n);
    if (numel == 0) return RPNET_OK;
    SumSources s;
    for (int k = 0; k < 16; ++k) {
        s.p[k] = srcs[k < n ? k : 0];
        RPNET_REQUIRE(s.p[k] && ((size_t)s.p[k] & 15) == 0, RPNET_ERR_ARG, "sum_n: source %d null or not 16-byte aligned", k);
    }
    RPNET_REQUIRE(((size_t)out & 15) == 0, RPNET_ERR_ARG, "sum_n: output not 16-byte aligned");
    const size_t n4 = numel / 4;
    const int blocks = (int)(n4 / 256 < 1 ? 1 : (n4 / 256 > 8192 ? 8192 : n4 / 256));
    hipLaunchKernelGGL(sum_n_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, n, out, n4, numel);
    return check_launch("sum_n");
}
