// First U-Net layer, Cin = 1 (encoder.Conv1.conv.0, net/unet.py:407 with img_ch = 1):
// K = 9 is no GEMM — a direct convolution that is bound by writing the [N,H,W,Cout]
// output (forward) or reading dy (weight gradient).  One thread = one pixel x 4 output
// channels, so a pixel's Cout channels are one coalesced float4 row; the 9 x Cout filter
// sits in LDS.  The input image needs no gradient.
#include "common.h"

namespace rpnet {

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         const float* __restrict__ ep_scale,
                                                         const float* __restrict__ ep_shift, int N, int H, int W, int Cout,
                                                         float* __restrict__ out_absmax, double* __restrict__ stats_partial,
                                                         const int groups) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9][Cout] + bias [Cout] (+ the statistics reduction)
    const int t = threadIdx.x;
    for (int i = t; i < 9 * Cout; i += 256) { const int co = i / 9, tap = i - co * 9; wl[tap * Cout + co] = w[i]; }
    for (int i = t; i < Cout; i += 256) wl[9 * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int Q = Cout / 4, ppb = 256 / Q;
    const int q = t % Q, pl = t / Q;
    float amax = 0.f;
    const size_t M = (size_t)N * H * W;
    // stats_partial (train-mode BatchNorm statistics fused, as rpnet_conv_desc.stats_partial): grid (blocks, groups), a block
    // owns a CONTIGUOUS pixel range of one statistic group and leaves one (sum, sum of squares) row per channel;
    // otherwise a grid-stride loop over all pixels
    size_t p_lo = (size_t)blockIdx.x * ppb, p_hi = M, p_step = (size_t)gridDim.x * ppb;
    if (stats_partial) {
        const size_t Mg = M / groups, chunk = (Mg + gridDim.x - 1) / gridDim.x;
        p_lo = (size_t)blockIdx.y * Mg + (size_t)blockIdx.x * chunk;
        p_hi = min(p_lo + chunk, (size_t)(blockIdx.y + 1) * Mg);
        p_step = ppb;
    }
    double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
    for (size_t p = p_lo + pl; pl < ppb && p < p_hi; p += p_step) {
        const int ox = (int)(p % W), oy = (int)((p / W) % H);
        const size_t nb = p - (size_t)oy * W - ox;  // n*H*W
        f32x4 acc = *reinterpret_cast<const f32x4*>(&wl[9 * Cout + q * 4]);
#pragma unroll
        for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
            for (int kx = -1; kx <= 1; ++kx) {
                const int iy = oy + ky, ix = ox + kx;
                const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[nb + (size_t)iy * W + ix] : 0.f;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[((ky + 1) * 3 + kx + 1) * Cout + q * 4]);
                acc += xv * wv;
            }
        if (ep_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ep_scale + q * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ep_shift + q * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k] * sc[k] + sh[k], 0.f);
        }
        *reinterpret_cast<f32x4*>(y + p * Cout + q * 4) = acc;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
        if (stats_partial) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { ssum[k] += acc[k]; ssq[k] += (double)acc[k] * acc[k]; }
        }
    }
    if (stats_partial) {        // the ppb pixel lanes of a channel quad are added up through LDS (behind the filter)
        double* red = reinterpret_cast<double*>(wl + 10 * Cout + (10 * Cout & 1));      // 8-byte aligned
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[t * 8 + k] = ssum[k]; red[t * 8 + 4 + k] = ssq[k]; }
        __syncthreads();
        if (pl == 0) {
            for (int r = 1; r < ppb; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) { ssum[k] += red[(r * Q + q) * 8 + k]; ssq[k] += red[(r * Q + q) * 8 + 4 + k]; }
            double* o = stats_partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * Cout + q * 4) * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[k * 2] = ssum[k]; o[k * 2 + 1] = ssq[k]; }
        }
    }
    if (out_absmax) {       // as rpnet_conv_desc.out_absmax: ONE atomic per block, and only where it would raise the maximum
        // (a wave's atomic each from 16384 blocks serialised on the one address: 0.7 ms behind a 25 us kernel)
        __shared__ float wave_max[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((t & 63) == 0) wave_max[t >> 6] = amax;
        __syncthreads();
        if (t == 0) {
            amax = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
            // the stored value only grows, so a stale read can only cause a redundant atomic, never a lost maximum
            if (amax > __hip_atomic_load(out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(out_absmax), __float_as_uint(amax));
        }
    }
}

// partial[group][blk][Cout][9].  BN = true: dy is not read but made on the spot from the gradient dz of the layer's
// BatchNorm + ReLU output, its pre-BatchNorm tensor y and the coefficients of rpnet_bn_bwd's reduction pass,
// dy = scale (dz [z > 0] - c1 - xhat c2) — the 12 bytes per element of a separate apply pass (this layer has no input
// gradient, so its dy has no other reader) become 4
template <bool BN>
__global__ __launch_bounds__(256) void conv1_wgrad_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ ybn, const float* __restrict__ stats,
                                                            const float* __restrict__ coef, float* __restrict__ partial,
                                                            int N, int H, int W, int Cout, int groups) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [ppb][Cout][9]
    const int t = threadIdx.x;
    const int Q = Cout / 4, ppb = 256 / Q;
    const int q = t % Q, pl = t / Q;
    const int g = blockIdx.y;
    float acc[9][4];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[a][k] = 0.f;
    const size_t Mg = (size_t)(N / groups) * H * W, p_lo = (size_t)g * Mg;
    f32x4 sc = {}, sh = {}, mu = {}, is = {}, c1 = {}, c2 = {};
    if (BN) {
        const int GC = groups * Cout, o = g * Cout + q * 4;
        sc = *reinterpret_cast<const f32x4*>(stats + o);
        sh = *reinterpret_cast<const f32x4*>(stats + GC + o);
        mu = *reinterpret_cast<const f32x4*>(stats + 2 * GC + o);
        is = *reinterpret_cast<const f32x4*>(stats + 3 * GC + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { c1[k] = coef[(o + k) * 2]; c2[k] = coef[(o + k) * 2 + 1]; }
    }
    if (pl < ppb) {
        for (size_t pg = (size_t)blockIdx.x * ppb + pl; pg < Mg; pg += (size_t)gridDim.x * ppb) {
            const size_t p = p_lo + pg;
            const int ox = (int)(p % W), oy = (int)((p / W) % H);
            const size_t nb = p - (size_t)oy * W - ox;
            f32x4 gr = *reinterpret_cast<const f32x4*>(dy + p * Cout + q * 4);
            if (BN) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ybn + p * Cout + q * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dm = (v[k] * sc[k] + sh[k] > 0.f) ? gr[k] : 0.f;
                    gr[k] = sc[k] * (dm - c1[k] - (v[k] - mu[k]) * is[k] * c2[k]);
                }
            }
#pragma unroll
            for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
                for (int kx = -1; kx <= 1; ++kx) {
                    const int iy = oy + ky, ix = ox + kx;
                    const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[nb + (size_t)iy * W + ix] : 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[(ky + 1) * 3 + kx + 1][k] += xv * gr[k];
                }
        }
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[(pl * Cout + q * 4 + k) * 9 + a] = acc[a][k];
    }
    __syncthreads();
    for (int i = t; i < Cout * 9; i += 256) {
        float s = 0.f;
        for (int r = 0; r < ppb; ++r) s += red[r * Cout * 9 + i];
        partial[((size_t)g * gridDim.x + blockIdx.x) * Cout * 9 + i] = s;
    }
}

__global__ __launch_bounds__(64) void conv1_wgrad_final(const float* __restrict__ partial, float* __restrict__ dw, int nblk, int n) {
    const int i = blockIdx.x, lane = threadIdx.x;
    double s = 0;
    for (int b = lane; b < nblk; b += 64) s += partial[(size_t)b * n + i];
    s = wave_sum(s);
    if (lane == 0) dw[i] = (float)s;
}

constexpr int kConv1WgradBlocks = 1024;

}  // namespace rpnet

extern "C" int rpnet_conv1_stats_blocks(int N, int H, int W, int cout, int groups) {
    (void)cout;
    if (groups < 1 || N % groups) return 0;
    const size_t per_group = (size_t)(N / groups) * H * W;
    const size_t nb = (per_group + 511) / 512;               // >= 512 pixels per block: enough blocks to stream at HBM rate
    return (int)(nb > 2048 ? 2048 : nb);
}

extern "C" int rpnet_conv1_fwd(const float* x, const float* w, const float* bias, float* y, const float* ep_scale,
                               const float* ep_shift, int N, int H, int W, int cout, float* out_absmax,
                               double* stats_partial, int groups, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && w && y, RPNET_ERR_ARG, "conv1_fwd: null pointer");
    RPNET_REQUIRE(cout % 4 == 0 && cout <= 1024 && 256 % (cout / 4) == 0, RPNET_ERR_SHAPE, "conv1_fwd: cout=%d", cout);
    const int ppb = 256 / (cout / 4);
    const size_t M = (size_t)N * H * W;
    if (stats_partial) {
        const int nblk = rpnet_conv1_stats_blocks(N, H, W, cout, groups);
        RPNET_REQUIRE(nblk > 0 && !ep_scale, RPNET_ERR_ARG, "conv1_fwd: fused statistics need N %% groups == 0 and no epilogue affine");
        const size_t lds = (size_t)(10 * cout + 2) * sizeof(float) + (size_t)256 * 8 * sizeof(double);
        hipLaunchKernelGGL(conv1_fwd_kernel, dim3(nblk, groups), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, ep_scale,
                           ep_shift, N, H, W, cout, out_absmax, stats_partial, groups);
        return check_launch("conv1_fwd");
    }
    size_t nb = (M + ppb - 1) / ppb;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3((int)nb), dim3(256), (size_t)10 * cout * sizeof(float), (hipStream_t)stream, x,
                       w, bias, y, ep_scale, ep_shift, N, H, W, cout, out_absmax, (double*)nullptr, 1);
    return check_launch("conv1_fwd");
}

extern "C" size_t rpnet_conv1_wgrad_workspace_bytes(int N, int H, int W, int cout) {
    (void)N; (void)H; (void)W;
    return (size_t)rpnet::kConv1WgradBlocks * cout * 9 * sizeof(float);
}

static int conv1_wgrad_launch(const float* x, const float* dy, const float* ybn, const float* stats, const float* coef, float* dw,
                              int N, int H, int W, int cout, int groups, void* workspace, size_t workspace_bytes,
                              rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && dy && dw && workspace, RPNET_ERR_ARG, "conv1_wgrad: null pointer");
    RPNET_REQUIRE(cout % 4 == 0 && cout <= 256 && 256 % (cout / 4) == 0, RPNET_ERR_SHAPE, "conv1_wgrad: cout=%d", cout);
    RPNET_REQUIRE(groups >= 1 && N % groups == 0, RPNET_ERR_SHAPE, "conv1_wgrad: N=%d groups=%d", N, groups);
    RPNET_REQUIRE(workspace_bytes >= rpnet_conv1_wgrad_workspace_bytes(N, H, W, cout), RPNET_ERR_WORKSPACE, "conv1_wgrad: workspace");
    const int ppb = 256 / (cout / 4);
    const size_t Mg = (size_t)(N / groups) * H * W;
    int nb = (int)((Mg + ppb - 1) / ppb);
    if (nb > kConv1WgradBlocks / groups) nb = kConv1WgradBlocks / groups;
    if (nb < 1) nb = 1;
    hipStream_t s = (hipStream_t)stream;
    if (ybn)
        hipLaunchKernelGGL(conv1_wgrad_partial<true>, dim3(nb, groups), dim3(256), (size_t)ppb * cout * 9 * sizeof(float), s, x, dy,
                           ybn, stats, coef, (float*)workspace, N, H, W, cout, groups);
    else
        hipLaunchKernelGGL(conv1_wgrad_partial<false>, dim3(nb, groups), dim3(256), (size_t)ppb * cout * 9 * sizeof(float), s, x, dy,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)workspace, N, H, W, cout, groups);
    hipLaunchKernelGGL(conv1_wgrad_final, dim3(cout * 9), dim3(64), 0, s, (const float*)workspace, dw, nb * groups, cout * 9);
    return check_launch("conv1_wgrad");
}

extern "C" int rpnet_conv1_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int cout, void* workspace,
                                 size_t workspace_bytes, rpnet_stream_t stream) {
    return conv1_wgrad_launch(x, dy, nullptr, nullptr, nullptr, dw, N, H, W, cout, 1, workspace, workspace_bytes, stream);
}

extern "C" int rpnet_conv1_wgrad_bn(const float* x, const float* dz, const float* y, const float* stats, const float* coef,
                                    float* dw, int N, int H, int W, int cout, int groups, void* workspace,
                                    size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(y && stats && coef, RPNET_ERR_ARG, "conv1_wgrad_bn: null pointer");
    RPNET_REQUIRE(groups <= kConv1WgradBlocks, RPNET_ERR_SHAPE, "conv1_wgrad_bn: groups=%d", groups);
    return conv1_wgrad_launch(x, dz, y, stats, coef, dw, N, H, W, cout, groups, workspace, workspace_bytes, stream);
}
