// First U-Net layer, Cin = 1 (encoder.Conv1.conv.0, net/unet.py:407 with img_ch = 1):
// K = 9 is no GEMM — a direct convolution that is bound by writing the [N,H,W,Cout]
// output (forward) or reading dy (weight gradient).  One thread = one pixel x 4 output
// channels, so a pixel's Cout channels are one coalesced float4 row; the 9 x Cout filter
// sits in LDS.  The input image needs no gradient.
//
// Train mode on fp16 planes (round 4): the pre-BatchNorm tensor y of this layer is NEVER in memory.  It costs nine
// multiply-adds per value to make and eight bytes per value to write and read back, so the statistics launch
// (rpnet_conv1_fwd with y == NULL) only sums it, BatchNorm + ReLU (rpnet_conv1_bn_relu) makes it again and writes the
// operand planes of Conv1.conv.3, and the backward's reduction pass and weight gradient (rpnet_conv1_bn_bwd_partial,
// rpnet_conv1_wgrad_bn with y == NULL) make it again from the image.  conv1_quad is the ONE definition of a value of y
// (explicit fused multiply-adds, taps in row-major order): a value made twice is the same bits twice.
#include "common.h"
#include "split_bf16.h"

namespace rpnet {

// filter [9][Cout] + bias [Cout] in LDS (wl), as every kernel here keeps it
__device__ __forceinline__ void conv1_load_filter(float* wl, const float* __restrict__ w, const float* __restrict__ bias,
                                                  const int Cout) {
    const int t = threadIdx.x;
    for (int i = t; i < 9 * Cout; i += 256) { const int co = i / 9, tap = i - co * 9; wl[tap * Cout + co] = w[i]; }
    for (int i = t; i < Cout; i += 256) wl[9 * Cout + i] = bias ? bias[i] : 0.f;
}

// y[pixel (nb + oy W + ox)][4 q .. 4 q + 3], bias included
__device__ __forceinline__ f32x4 conv1_quad(const float* __restrict__ x, const float* wl, const int Cout, const int q,
                                            const size_t nb, const int oy, const int ox, const int H, const int W) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(&wl[9 * Cout + q * 4]);
#pragma unroll
    for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
        for (int kx = -1; kx <= 1; ++kx) {
            const int iy = oy + ky, ix = ox + kx;
            const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[nb + (size_t)iy * W + ix] : 0.f;
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[((ky + 1) * 3 + kx + 1) * Cout + q * 4]);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(xv, wv[k], acc[k]);
        }
    return acc;
}

// The same for the FOUR pixels (oy, ox0 .. ox0 + 3), ox0 % 4 == 0 = W % 4: one 3 x 6 window of the image (a 16-byte load and
// two edge values per row) instead of four 3 x 3 ones, the filter quad read once — each value still the fused multiply-add chain
// of conv1_quad in the same order: the same bits.  Round 5: the per-pixel form spent its time in address arithmetic and
// 4-byte loads (conv1_fwd_kernel 142 us for a 4 MB image, conv1_bn_relu_kernel 154 us against 55 us of HBM writes).
__device__ __forceinline__ void conv1_quad4(const float* __restrict__ x, const float* wl, const int Cout, const int q,
                                            const unsigned nb, const int oy, const int ox0, const int H, const int W, f32x4 (&out)[4]) {
    float xv[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        const bool rok = iy >= 0 && iy < H;
        const float* row = x + nb + (unsigned)(rok ? iy : 0) * (unsigned)W + ox0;
        const f32x4 mid = rok ? *reinterpret_cast<const f32x4*>(row) : f32x4{0.f, 0.f, 0.f, 0.f};
        xv[ky][0] = (rok && ox0 > 0) ? row[-1] : 0.f;
        xv[ky][1] = mid[0]; xv[ky][2] = mid[1]; xv[ky][3] = mid[2]; xv[ky][4] = mid[3];
        xv[ky][5] = (rok && ox0 + 4 < W) ? row[4] : 0.f;
    }
    const f32x4 b = *reinterpret_cast<const f32x4*>(&wl[9 * Cout + q * 4]);
    f32x4 wv[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wv[tap] = *reinterpret_cast<const f32x4*>(&wl[tap * Cout + q * 4]);
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        f32x4 acc = b;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(xv[ky][px + kx], wv[ky * 3 + kx][k], acc[k]);
        out[px] = acc;
    }
}

// strip s (four consecutive pixels 4 s .. 4 s + 3 of the N x H x W pixel sequence, W % 4 == 0) -> (image base, oy, ox0)
__device__ __forceinline__ void conv1_strip(const unsigned s, const int H, const int W, unsigned& nb, int& oy, int& ox0) {
    const unsigned p = s * 4u, row = p / (unsigned)W;
    ox0 = (int)(p - row * (unsigned)W);
    const unsigned n = row / (unsigned)H;
    oy = (int)(row - n * (unsigned)H);
    nb = n * (unsigned)H * (unsigned)W;
}

// conv1_fwd_kernel on strips of four pixels (W % 4 == 0, fewer than 2^31 pixels): thread = strip x 4 output channels
__global__ __launch_bounds__(256) void conv1_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          const float* __restrict__ ep_scale, const float* __restrict__ ep_shift, int N,
                                                          int H, int W, int Cout, float* __restrict__ out_absmax,
                                                          double* __restrict__ stats_partial, const int groups) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int t = threadIdx.x;
    conv1_load_filter(wl, w, bias, Cout);
    __syncthreads();
    const int Q = Cout / 4, spb = 256 / Q;
    const int q = t % Q, sl = t / Q;
    float amax = 0.f;
    const unsigned S = (unsigned)N * H * W / 4;            // strips
    unsigned s_lo = blockIdx.x * spb, s_hi = S, s_step = gridDim.x * spb;
    if (stats_partial) {      // a block owns a contiguous strip range of one statistic group (one partial row per block)
        const unsigned Sg = S / groups, chunk = (Sg + gridDim.x - 1) / gridDim.x;
        s_lo = blockIdx.y * Sg + blockIdx.x * chunk;
        s_hi = min(s_lo + chunk, (blockIdx.y + 1) * Sg);
        s_step = spb;
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (ep_scale) {
        sc = *reinterpret_cast<const f32x4*>(ep_scale + q * 4);
        sh = *reinterpret_cast<const f32x4*>(ep_shift + q * 4);
    }
    double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
    for (unsigned s = s_lo + sl; sl < spb && s < s_hi; s += s_step) {
        unsigned nb; int oy, ox0;
        conv1_strip(s, H, W, nb, oy, ox0);
        f32x4 v[4];
        conv1_quad4(x, wl, Cout, q, nb, oy, ox0, H, W, v);
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            f32x4 acc = v[px];
            if (ep_scale) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k] * sc[k] + sh[k], 0.f);
            }
            if (y) *reinterpret_cast<f32x4*>(y + ((size_t)s * 4 + px) * Cout + q * 4) = acc;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
            if (stats_partial) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { ssum[k] += acc[k]; ssq[k] += (double)acc[k] * acc[k]; }
            }
        }
    }
    if (stats_partial) {
        double* red = reinterpret_cast<double*>(wl + 10 * Cout + (10 * Cout & 1));
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[t * 8 + k] = ssum[k]; red[t * 8 + 4 + k] = ssq[k]; }
        __syncthreads();
        if (sl == 0) {
            for (int r = 1; r < spb; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) { ssum[k] += red[(r * Q + q) * 8 + k]; ssq[k] += red[(r * Q + q) * 8 + 4 + k]; }
            double* o = stats_partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * Cout + q * 4) * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[k * 2] = ssum[k]; o[k * 2 + 1] = ssq[k]; }
        }
    }
    if (out_absmax) {
        __shared__ float wave_max[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((t & 63) == 0) wave_max[t >> 6] = amax;
        __syncthreads();
        if (t == 0) {
            amax = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
            if (amax > __hip_atomic_load(out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(out_absmax), __float_as_uint(amax));
        }
    }
}

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         const float* __restrict__ ep_scale,
                                                         const float* __restrict__ ep_shift, int N, int H, int W, int Cout,
                                                         float* __restrict__ out_absmax, double* __restrict__ stats_partial,
                                                         const int groups) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9][Cout] + bias [Cout] (+ the statistics reduction)
    const int t = threadIdx.x;
    conv1_load_filter(wl, w, bias, Cout);
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
        f32x4 acc = conv1_quad(x, wl, Cout, q, nb, oy, ox, H, W);
        if (ep_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ep_scale + q * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ep_shift + q * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k] * sc[k] + sh[k], 0.f);
        }
        if (y) *reinterpret_cast<f32x4*>(y + p * Cout + q * 4) = acc;      // y == NULL: the statistics only
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
// RECOMP (with BN): ybn is not read either — y is made again from x and the filter (conv1_quad; wf, bf = weight, bias)
template <bool BN, bool RECOMP = false>
__global__ __launch_bounds__(256) void conv1_wgrad_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ ybn, const float* __restrict__ stats,
                                                            const float* __restrict__ coef, float* __restrict__ partial,
                                                            int N, int H, int W, int Cout, int groups,
                                                            const float* __restrict__ wf, const float* __restrict__ bf) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float red[];  // [ppb][Cout][9] (RECOMP: + the filter behind it)
    const int t = threadIdx.x;
    float* const wl = red + (256 / (Cout / 4)) * Cout * 9;
    if (RECOMP) {
        conv1_load_filter(wl, wf, bf, Cout);
        __syncthreads();
    }
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
                f32x4 v;
                if constexpr (RECOMP) v = conv1_quad(x, wl, Cout, q, nb, oy, ox, H, W);
                else v = *reinterpret_cast<const f32x4*>(ybn + p * Cout + q * 4);
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

// conv1_wgrad_partial on strips of four pixels (W % 4 == 0): the 3 x 6 image window of a strip serves its 36 tap products
template <bool BN, bool RECOMP>
__global__ __launch_bounds__(256) void conv1_wgrad_partial4(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ ybn, const float* __restrict__ stats,
                                                             const float* __restrict__ coef, float* __restrict__ partial, int N, int H,
                                                             int W, int Cout, int groups, const float* __restrict__ wf,
                                                             const float* __restrict__ bf) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float red[];  // [spb][Cout][9] (RECOMP: + the filter behind it)
    const int t = threadIdx.x;
    float* const wl = red + (256 / (Cout / 4)) * Cout * 9;
    if (RECOMP) {
        conv1_load_filter(wl, wf, bf, Cout);
        __syncthreads();
    }
    const int Q = Cout / 4, spb = 256 / Q;
    const int q = t % Q, sl = t / Q;
    const int g = blockIdx.y;
    float acc[9][4];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[a][k] = 0.f;
    const unsigned Sg = (unsigned)(N / groups) * H * W / 4, s_lo = (unsigned)g * Sg;
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
    if (sl < spb) {
        for (unsigned sg = blockIdx.x * spb + sl; sg < Sg; sg += gridDim.x * spb) {
            const unsigned s = s_lo + sg;
            unsigned nb; int oy, ox0;
            conv1_strip(s, H, W, nb, oy, ox0);
            float xv[3][6];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
                const bool rok = iy >= 0 && iy < H;
                const float* row = x + nb + (unsigned)(rok ? iy : 0) * (unsigned)W + ox0;
                const f32x4 mid = rok ? *reinterpret_cast<const f32x4*>(row) : f32x4{0.f, 0.f, 0.f, 0.f};
                xv[ky][0] = (rok && ox0 > 0) ? row[-1] : 0.f;
                xv[ky][1] = mid[0]; xv[ky][2] = mid[1]; xv[ky][3] = mid[2]; xv[ky][4] = mid[3];
                xv[ky][5] = (rok && ox0 + 4 < W) ? row[4] : 0.f;
            }
            f32x4 v4[4];
            if constexpr (BN && RECOMP) conv1_quad4(x, wl, Cout, q, nb, oy, ox0, H, W, v4);
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const size_t e = ((size_t)s * 4 + px) * Cout + q * 4;
                f32x4 gr = *reinterpret_cast<const f32x4*>(dy + e);
                if (BN) {
                    f32x4 v;
                    if constexpr (RECOMP) v = v4[px];
                    else v = *reinterpret_cast<const f32x4*>(ybn + e);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dm = (v[k] * sc[k] + sh[k] > 0.f) ? gr[k] : 0.f;
                        gr[k] = sc[k] * (dm - c1[k] - (v[k] - mu[k]) * is[k] * c2[k]);
                    }
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[ky * 3 + kx][k] += xv[ky][px + kx] * gr[k];
            }
        }
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[(sl * Cout + q * 4 + k) * 9 + a] = acc[a][k];
    }
    __syncthreads();
    for (int i = t; i < Cout * 9; i += 256) {
        float s = 0.f;
        for (int r = 0; r < spb; ++r) s += red[r * Cout * 9 + i];
        partial[((size_t)g * gridDim.x + blockIdx.x) * Cout * 9 + i] = s;
    }
}

// conv1_bn_relu_kernel on strips of four pixels (W % 4 == 0): thread = strip x 8 channels
template <int NP>
__global__ __launch_bounds__(256) void conv1_bn_relu4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ z,
                                                              unsigned short* __restrict__ zs, const int N, const int H, const int W,
                                                              const int Cout, const int groups, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float sqrt_n, float* __restrict__ s_out) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float wl[];
    __shared__ float red4[4];
    const int t = threadIdx.x;
    conv1_load_filter(wl, w, bias, Cout);
    float inv_s = 1.f;
    if (NP <= 2) {
        float m = 0.f;
        for (int c = t; c < Cout; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_n + fabsf(beta[c]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((t & 63) == 0) red4[t >> 6] = m;
        __syncthreads();
        const float sc = pow2_scale(fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3])));
        if (blockIdx.x == 0 && t == 0) *s_out = sc;
        inv_s = 1.f / sc;
    }
    __syncthreads();
    const int Q8 = Cout / 8, spb = 256 / Q8;
    const int q8 = t % Q8, sl = t / Q8;
    const unsigned S = (unsigned)N * H * W / 4, Sg = S / groups;
    const size_t plane = (size_t)S * 4 * Cout;
    for (unsigned s = blockIdx.x * spb + sl; sl < spb && s < S; s += gridDim.x * spb) {
        unsigned nb; int oy, ox0;
        conv1_strip(s, H, W, nb, oy, ox0);
        const int g = (int)(s / Sg);
        f32x4 a[2][4];
        conv1_quad4(x, wl, Cout, q8 * 2, nb, oy, ox0, H, W, a[0]);
        conv1_quad4(x, wl, Cout, q8 * 2 + 1, nb, oy, ox0, H, W, a[1]);
        f32x4 s4[2], h4[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            s4[hh] = *reinterpret_cast<const f32x4*>(scale + g * Cout + q8 * 8 + hh * 4);
            h4[hh] = *reinterpret_cast<const f32x4*>(shift + g * Cout + q8 * 8 + hh * 4);
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const size_t e = ((size_t)s * 4 + px) * Cout + q8 * 8;
            float v[8];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[k] = fmaxf(a[hh][px][k] * s4[hh][k] + h4[hh][k], 0.f); v[hh * 4 + k] = o[k]; }
                if (z) *reinterpret_cast<f32x4*>(z + e + hh * 4) = o;
            }
            if (NP <= 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= inv_s;
            }
            u32x4 pk[NP];
            split8<NP>(v, pk);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<u32x4*>(zs + pp * plane + e) = pk[pp];
        }
    }
}

// conv1_bn_bwd_partial_kernel on strips of four pixels (W % 4 == 0)
__global__ __launch_bounds__(256) void conv1_bn_bwd_partial4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, const float* __restrict__ dz,
                                                                     const float* __restrict__ stats, double* __restrict__ partial,
                                                                     const int N, const int H, const int W, const int Cout,
                                                                     const int groups) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int t = threadIdx.x;
    conv1_load_filter(wl, w, bias, Cout);
    __syncthreads();
    const int Q = Cout / 4, spb = 256 / Q;
    const int q = t % Q, sl = t / Q;
    const int g = blockIdx.y;
    const unsigned Sg = (unsigned)(N / groups) * H * W / 4, s_lo = (unsigned)g * Sg;
    const int GC = groups * Cout, o = g * Cout + q * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(stats + o);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(stats + GC + o);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + 2 * GC + o);
    const f32x4 is = *reinterpret_cast<const f32x4*>(stats + 3 * GC + o);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (sl < spb) {
        for (unsigned sg = blockIdx.x * spb + sl; sg < Sg; sg += gridDim.x * spb) {
            const unsigned s = s_lo + sg;
            unsigned nb; int oy, ox0;
            conv1_strip(s, H, W, nb, oy, ox0);
            f32x4 d[4], v[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) d[px] = *reinterpret_cast<const f32x4*>(dz + ((size_t)s * 4 + px) * Cout + q * 4);
            conv1_quad4(x, wl, Cout, q, nb, oy, ox0, H, W, v);
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dm = (v[px][k] * sc[k] + sh[k] > 0.f) ? d[px][k] : 0.f;
                    s1[k] += dm;
                    s2[k] += (double)dm * ((v[px][k] - mu[k]) * is[k]);
                }
        }
    }
    double* red = reinterpret_cast<double*>(wl + 10 * Cout + (10 * Cout & 1));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[t * 8 + k] = s1[k]; red[t * 8 + 4 + k] = s2[k]; }
    __syncthreads();
    if (sl == 0) {
        for (int r = 1; r < spb; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s1[k] += red[(r * Q + q) * 8 + k]; s2[k] += red[(r * Q + q) * 8 + 4 + k]; }
        double* out = partial + ((size_t)(g * gridDim.x + blockIdx.x) * Cout + q * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { out[k * 2] = s1[k]; out[k * 2 + 1] = s2[k]; }
    }
}

// BatchNorm + ReLU of the layer with y made on the spot: thread = one pixel x 8 channels; z (optional, fp32) and the NP
// operand planes of z / s, s = the rigorous tensor scale of rpnet_bn_relu (every block derives it from gamma, beta;
// block 0 publishes it).  Pixels are dealt to the blocks cyclically in groups of 256 / (Cout / 8).
template <int NP>
__global__ __launch_bounds__(256) void conv1_bn_relu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ z,
                                                             unsigned short* __restrict__ zs, const int N, const int H,
                                                             const int W, const int Cout, const int groups,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float sqrt_n, float* __restrict__ s_out) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float wl[];
    __shared__ float red4[4];
    const int t = threadIdx.x;
    conv1_load_filter(wl, w, bias, Cout);
    float inv_s = 1.f;
    if (NP <= 2) {
        float m = 0.f;
        for (int c = t; c < Cout; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_n + fabsf(beta[c]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((t & 63) == 0) red4[t >> 6] = m;
        __syncthreads();
        const float sc = pow2_scale(fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3])));
        if (blockIdx.x == 0 && t == 0) *s_out = sc;
        inv_s = 1.f / sc;
    }
    __syncthreads();
    const int Q8 = Cout / 8, ppb = 256 / Q8;
    const int q8 = t % Q8, pl = t / Q8;
    const size_t M = (size_t)N * H * W, Mg = M / groups, plane = M * Cout;
    for (size_t p = (size_t)blockIdx.x * ppb + pl; pl < ppb && p < M; p += (size_t)gridDim.x * ppb) {
        const int ox = (int)(p % W), oy = (int)((p / W) % H);
        const size_t nb = p - (size_t)oy * W - ox;
        const int g = (int)(p / Mg);
        float v[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 a = conv1_quad(x, wl, Cout, q8 * 2 + hh, nb, oy, ox, H, W);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(scale + g * Cout + q8 * 8 + hh * 4);
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(shift + g * Cout + q8 * 8 + hh * 4);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[k] = fmaxf(a[k] * s4[k] + h4[k], 0.f); v[hh * 4 + k] = o[k]; }
            if (z) *reinterpret_cast<f32x4*>(z + p * Cout + q8 * 8 + hh * 4) = o;
        }
        if (NP <= 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= inv_s;
        }
        u32x4 pk[NP];
        split8<NP>(v, pk);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<u32x4*>(zs + pp * plane + p * Cout + q8 * 8) = pk[pp];
    }
}

// reduction pass of the layer's BatchNorm backward with y made on the spot: partial[(g nblk + blk)][Cout][2] doubles
// (sum dz m, sum dz m xhat) as rpnet::bn_bwd_partial leaves them — for rpnet_bn_bwd(given_partial).  grid (nblk, groups);
// the pixel groups of a statistic group are dealt to its blocks cyclically.
__global__ __launch_bounds__(256) void conv1_bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, const float* __restrict__ dz,
                                                                    const float* __restrict__ stats, double* __restrict__ partial,
                                                                    const int N, const int H, const int W, const int Cout,
                                                                    const int groups) {
    RPNET_PASS_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float wl[];      // filter, then [256][8] doubles of the reduction
    const int t = threadIdx.x;
    conv1_load_filter(wl, w, bias, Cout);
    __syncthreads();
    const int Q = Cout / 4, ppb = 256 / Q;
    const int q = t % Q, pl = t / Q;
    const int g = blockIdx.y;
    const size_t Mg = (size_t)(N / groups) * H * W, p_lo = (size_t)g * Mg;
    const int GC = groups * Cout, o = g * Cout + q * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(stats + o);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(stats + GC + o);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + 2 * GC + o);
    const f32x4 is = *reinterpret_cast<const f32x4*>(stats + 3 * GC + o);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < ppb) {
        for (size_t pg = (size_t)blockIdx.x * ppb + pl; pg < Mg; pg += (size_t)gridDim.x * ppb) {
            const size_t p = p_lo + pg;
            const int ox = (int)(p % W), oy = (int)((p / W) % H);
            const size_t nb = p - (size_t)oy * W - ox;
            const f32x4 d = *reinterpret_cast<const f32x4*>(dz + p * Cout + q * 4);
            const f32x4 v = conv1_quad(x, wl, Cout, q, nb, oy, ox, H, W);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dm = (v[k] * sc[k] + sh[k] > 0.f) ? d[k] : 0.f;
                s1[k] += dm;
                s2[k] += (double)dm * ((v[k] - mu[k]) * is[k]);
            }
        }
    }
    double* red = reinterpret_cast<double*>(wl + 10 * Cout + (10 * Cout & 1));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[t * 8 + k] = s1[k]; red[t * 8 + 4 + k] = s2[k]; }
    __syncthreads();
    if (pl == 0) {
        for (int r = 1; r < ppb; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s1[k] += red[(r * Q + q) * 8 + k]; s2[k] += red[(r * Q + q) * 8 + 4 + k]; }
        double* out = partial + ((size_t)(g * gridDim.x + blockIdx.x) * Cout + q * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { out[k * 2] = s1[k]; out[k * 2 + 1] = s2[k]; }
    }
}

__global__ __launch_bounds__(64) void conv1_wgrad_final(const float* __restrict__ partial, float* __restrict__ dw, int nblk, int n) {
    RPNET_PASS_PRIORITY();
    const int i = blockIdx.x, lane = threadIdx.x;
    double s = 0;
    for (int b = lane; b < nblk; b += 64) s += partial[(size_t)b * n + i];
    s = wave_sum(s);
    if (lane == 0) dw[i] = (float)s;
}

constexpr int kConv1WgradBlocks = 1024;

// the four-pixel strip kernels: rows are whole strips, statistic groups too, 32-bit pixel arithmetic
static bool conv1_strips_ok(int N, int H, int W, int groups) {
    return W % 4 == 0 && groups >= 1 && N % groups == 0 && (size_t)N * H * W < ((size_t)1 << 31);
}

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
    RPNET_REQUIRE(x && w && (y || stats_partial), RPNET_ERR_ARG, "conv1_fwd: null pointer (y may be NULL only with stats_partial)");
    RPNET_REQUIRE(cout % 4 == 0 && cout <= 1024 && 256 % (cout / 4) == 0, RPNET_ERR_SHAPE, "conv1_fwd: cout=%d", cout);
    const int ppb = 256 / (cout / 4);
    const size_t M = (size_t)N * H * W;
    const bool strips = conv1_strips_ok(N, H, W, groups);
    if (stats_partial) {
        const int nblk = rpnet_conv1_stats_blocks(N, H, W, cout, groups);
        RPNET_REQUIRE(nblk > 0 && !ep_scale, RPNET_ERR_ARG, "conv1_fwd: fused statistics need N %% groups == 0 and no epilogue affine");
        const size_t lds = (size_t)(10 * cout + 2) * sizeof(float) + (size_t)256 * 8 * sizeof(double);
        if (strips)
            hipLaunchKernelGGL(conv1_fwd4_kernel, dim3(nblk, groups), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, ep_scale, ep_shift, N,
                               H, W, cout, out_absmax, stats_partial, groups);
        else
            hipLaunchKernelGGL(conv1_fwd_kernel, dim3(nblk, groups), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, ep_scale,
                               ep_shift, N, H, W, cout, out_absmax, stats_partial, groups);
        return check_launch("conv1_fwd");
    }
    size_t nb = (M + ppb - 1) / ppb;
    if (strips) nb = (nb + 3) / 4;
    if (nb > 16384) nb = 16384;
    if (strips)
        hipLaunchKernelGGL(conv1_fwd4_kernel, dim3((int)nb), dim3(256), (size_t)10 * cout * sizeof(float), (hipStream_t)stream, x, w, bias, y,
                           ep_scale, ep_shift, N, H, W, cout, out_absmax, (double*)nullptr, 1);
    else
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
                              rpnet_stream_t stream, const float* wf = nullptr, const float* bf = nullptr) {
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
    if (conv1_strips_ok(N, H, W, groups)) {
        const size_t lds0 = (size_t)ppb * cout * 9 * sizeof(float);
        if (wf)
            hipLaunchKernelGGL((conv1_wgrad_partial4<true, true>), dim3(nb, groups), dim3(256), lds0 + (size_t)10 * cout * sizeof(float), s, x, dy,
                               (const float*)nullptr, stats, coef, (float*)workspace, N, H, W, cout, groups, wf, bf);
        else if (ybn)
            hipLaunchKernelGGL((conv1_wgrad_partial4<true, false>), dim3(nb, groups), dim3(256), lds0, s, x, dy, ybn, stats, coef, (float*)workspace,
                               N, H, W, cout, groups, (const float*)nullptr, (const float*)nullptr);
        else
            hipLaunchKernelGGL((conv1_wgrad_partial4<false, false>), dim3(nb, groups), dim3(256), lds0, s, x, dy, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, (float*)workspace, N, H, W, cout, groups, (const float*)nullptr,
                               (const float*)nullptr);
    } else if (wf)
        hipLaunchKernelGGL((conv1_wgrad_partial<true, true>), dim3(nb, groups), dim3(256),
                           (size_t)(ppb * cout * 9 + 10 * cout) * sizeof(float), s, x, dy, (const float*)nullptr, stats, coef,
                           (float*)workspace, N, H, W, cout, groups, wf, bf);
    else if (ybn)
        hipLaunchKernelGGL((conv1_wgrad_partial<true, false>), dim3(nb, groups), dim3(256), (size_t)ppb * cout * 9 * sizeof(float), s, x, dy,
                           ybn, stats, coef, (float*)workspace, N, H, W, cout, groups, (const float*)nullptr, (const float*)nullptr);
    else
        hipLaunchKernelGGL((conv1_wgrad_partial<false, false>), dim3(nb, groups), dim3(256), (size_t)ppb * cout * 9 * sizeof(float), s, x, dy,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)workspace, N, H, W, cout, groups,
                           (const float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(conv1_wgrad_final, dim3(cout * 9), dim3(64), 0, s, (const float*)workspace, dw, nb * groups, cout * 9);
    return check_launch("conv1_wgrad");
}

extern "C" int rpnet_conv1_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int cout, void* workspace,
                                 size_t workspace_bytes, rpnet_stream_t stream) {
    return conv1_wgrad_launch(x, dy, nullptr, nullptr, nullptr, dw, N, H, W, cout, 1, workspace, workspace_bytes, stream);
}

extern "C" int rpnet_conv1_wgrad_bn(const float* x, const float* dz, const float* y, const float* stats, const float* coef,
                                    float* dw, int N, int H, int W, int cout, int groups, void* workspace,
                                    size_t workspace_bytes, const float* w, const float* bias, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE((y || w) && stats && coef, RPNET_ERR_ARG, "conv1_wgrad_bn: null pointer (y, or the filter w to make it again)");
    RPNET_REQUIRE(groups <= kConv1WgradBlocks, RPNET_ERR_SHAPE, "conv1_wgrad_bn: groups=%d", groups);
    return conv1_wgrad_launch(x, dz, y, stats, coef, dw, N, H, W, cout, groups, workspace, workspace_bytes, stream,
                              y ? nullptr : w, y ? nullptr : bias);
}

extern "C" int rpnet_conv1_bn_relu(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                                   float* z, void* z_split, int planes, const float* gamma, const float* beta,
                                   float* split_scale, int N, int H, int W, int cout, int groups, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && w && scale && shift && z_split, RPNET_ERR_ARG, "conv1_bn_relu: null pointer");
    RPNET_REQUIRE(cout % 8 == 0 && cout <= 1024 && 256 % (cout / 8) == 0, RPNET_ERR_SHAPE, "conv1_bn_relu: cout=%d", cout);
    RPNET_REQUIRE(groups >= 1 && N % groups == 0, RPNET_ERR_SHAPE, "conv1_bn_relu: N=%d groups=%d", N, groups);
    RPNET_REQUIRE(planes >= 1 && planes <= 3 && (planes == 3 || (gamma && beta && split_scale)), RPNET_ERR_ARG,
                  "conv1_bn_relu: planes=%d (fp16 planes need gamma, beta and the scale output)", planes);
    const size_t M = (size_t)N * H * W;
    const int ppb = 256 / (cout / 8);
    size_t nb = (M + (size_t)ppb * 4 - 1) / ((size_t)ppb * 4);      // four pixel groups per block and more
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    const float sqrt_n = sqrtf((float)((size_t)(N / groups) * H * W)) * 1.0001f;
    const size_t lds = (size_t)10 * cout * sizeof(float);
    const bool strips = conv1_strips_ok(N, H, W, groups);
    if (strips) nb = (nb + 3) / 4;
#define RPNET_C1BN(NP_)                                                                                                     \
    do { if (strips) hipLaunchKernelGGL(conv1_bn_relu4_kernel<NP_>, dim3((int)nb), dim3(256), lds, (hipStream_t)stream, x, w, bias, scale, shift, z, \
                       (unsigned short*)z_split, N, H, W, cout, groups, gamma, beta, sqrt_n, split_scale);                  \
    else hipLaunchKernelGGL(conv1_bn_relu_kernel<NP_>, dim3((int)nb), dim3(256), lds, (hipStream_t)stream, x, w, bias, scale, shift, z, \
                       (unsigned short*)z_split, N, H, W, cout, groups, gamma, beta, sqrt_n, split_scale); } while (0)
    if (planes == 3) RPNET_C1BN(3);
    else if (planes == 2) RPNET_C1BN(2);
    else RPNET_C1BN(1);
#undef RPNET_C1BN
    return check_launch("conv1_bn_relu");
}

extern "C" int rpnet_conv1_bn_bwd_rows(int N, int H, int W, int cout, int groups) {
    return rpnet_conv1_stats_blocks(N, H, W, cout, groups);
}

extern "C" int rpnet_conv1_bn_bwd_partial(const float* x, const float* w, const float* bias, const float* dz, const float* stats,
                                          double* partial, int N, int H, int W, int cout, int groups, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && w && dz && stats && partial, RPNET_ERR_ARG, "conv1_bn_bwd_partial: null pointer");
    RPNET_REQUIRE(cout % 4 == 0 && cout <= 1024 && 256 % (cout / 4) == 0, RPNET_ERR_SHAPE, "conv1_bn_bwd_partial: cout=%d", cout);
    const int nblk = rpnet_conv1_bn_bwd_rows(N, H, W, cout, groups);
    RPNET_REQUIRE(nblk > 0, RPNET_ERR_SHAPE, "conv1_bn_bwd_partial: N=%d groups=%d", N, groups);
    const size_t lds = (size_t)(10 * cout + 2) * sizeof(float) + (size_t)256 * 8 * sizeof(double);
    if (conv1_strips_ok(N, H, W, groups))
        hipLaunchKernelGGL(conv1_bn_bwd_partial4_kernel, dim3(nblk, groups), dim3(256), lds, (hipStream_t)stream, x, w, bias, dz, stats, partial,
                           N, H, W, cout, groups);
    else
        hipLaunchKernelGGL(conv1_bn_bwd_partial_kernel, dim3(nblk, groups), dim3(256), lds, (hipStream_t)stream, x, w, bias, dz, stats,
                           partial, N, H, W, cout, groups);
    return check_launch("conv1_bn_bwd_partial");
}
