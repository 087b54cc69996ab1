// Local-window correlation (the exact form of Correlation(), net/rp_net.py:153-181) and
// its gradients.  NHWC fp32.
//   corr[b,y,x, a*K + c] = <f1[b,y,x,:], f2[b, y + c - r, x + a - r, :]> / sqrt(C),  K = 2r+1,
// zero outside the image.  The reference materialises the all-pairs (hw)x(hw) matrix
// (67 MB per sample at 64x64) and grid_samples an 11x11 window out of it; here a block owns
// an 8x8 pixel tile, stages the f2 halo (8+2r)^2 and the f1 tile in LDS 32 channels at a
// time (float4 along C) and every thread accumulates ~K*K/4 window offsets for one pixel.
// The finished tile goes through LDS once more so that each pixel's window is written as
// one contiguous row (cstride floats, zero padded: the 1x1 conv that consumes it wants
// K = 128, see rpnet_pack_conv_weight).
#include "common.h"

namespace rpnet {

constexpr int CT = 8;        // pixel tile edge
constexpr int CC = 32;       // channels per LDS stage
constexpr int CSTR = CC + 4; // padded row (floats), keeps float4 alignment

template <int R>
__global__ __launch_bounds__(256) void local_corr_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ corr, int h, int w, int C, int cstride,
                                                              float inv_sqrt_c) {
    constexpr int K = 2 * R + 1, KK = K * K, HT = CT + 2 * R, NI = (KK + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* f2s = sm;                    // [HT*HT][CSTR]
    float* f1s = sm + HT * HT * CSTR;   // [64][CSTR]
    const int t = threadIdx.x;
    const int tiles_x = (w + CT - 1) / CT;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT, tx0 = (blockIdx.x % tiles_x) * CT;
    const int pix = t & 63, og = t >> 6;
    const int py = pix >> 3, px = pix & 7;

    int noff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int o = og * NI + i;
        const int a = o / K, c = o - a * K;  // a -> x offset, c -> y offset
        noff[i] = (o < KK) ? ((py + c) * HT + (px + a)) * CSTR : 0;
    }
    float acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = 0.f;

    const float* f1b = f1 + (size_t)b * h * w * C;
    const float* f2b = f2 + (size_t)b * h * w * C;
    for (int c0 = 0; c0 < C; c0 += CC) {
        __syncthreads();
        for (int e = t; e < HT * HT * (CC / 4); e += 256) {
            const int c4 = e & 7, hp = e >> 3;
            const int hy = hp / HT, hx = hp - hy * HT;
            const int y = ty0 + hy - R, x = tx0 + hx - R;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < h && x >= 0 && x < w) v = *reinterpret_cast<const f32x4*>(f2b + ((size_t)y * w + x) * C + c0 + c4 * 4);
            *reinterpret_cast<f32x4*>(&f2s[hp * CSTR + c4 * 4]) = v;
        }
        for (int e = t; e < 64 * (CC / 4); e += 256) {
            const int c4 = e & 7, p = e >> 3;
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < h && x < w) v = *reinterpret_cast<const f32x4*>(f1b + ((size_t)y * w + x) * C + c0 + c4 * 4);
            *reinterpret_cast<f32x4*>(&f1s[p * CSTR + c4 * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < CC / 4; ++c4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&f1s[pix * CSTR + c4 * 4]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&f2s[noff[i] + c4 * 4]);
                acc[i] += a[0] * bv[0] + a[1] * bv[1] + a[2] * bv[2] + a[3] * bv[3];
            }
        }
    }
    // stage the tile as [64][cstride] rows and write them out contiguously
    __syncthreads();
    float* outs = sm;  // 64 * cstride floats (host checks it fits)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int o = og * NI + i;
        if (o < KK) outs[pix * cstride + o] = acc[i] * inv_sqrt_c;
    }
    for (int e = t; e < 64 * (cstride - KK); e += 256) {
        const int p = e / (cstride - KK), o = KK + e - p * (cstride - KK);
        outs[p * cstride + o] = 0.f;
    }
    __syncthreads();
    float* cb = corr + (size_t)b * h * w * cstride;
    for (int e = t; e < 64 * cstride; e += 256) {
        const int p = e / cstride, o = e - p * cstride;
        const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
        if (y < h && x < w) cb[((size_t)y * w + x) * cstride + o] = outs[e];
    }
}

// dcT[b,q, o] = dcorr[b, q - off(o), o]  (0 outside): the window gradient seen from the f2 pixel
template <int R>
__global__ void corr_transpose_kernel(const float* __restrict__ dcorr, float* __restrict__ dct, int B, int h, int w,
                                      int cstride) {
    constexpr int K = 2 * R + 1, KK = K * K;
    const size_t total = (size_t)B * h * w * cstride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % cstride);
        size_t p = i / cstride;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int b = (int)(p / h);
        float v = 0.f;
        if (o < KK) {
            const int a = o / K, c = o - a * K;
            const int sy = y - (c - R), sx = x - (a - R);
            if (sy >= 0 && sy < h && sx >= 0 && sx < w) v = dcorr[(((size_t)b * h + sy) * w + sx) * cstride + o];
        }
        dct[i] = v;
    }
}

// df[b,p,ch] = inv_sqrt_c * sum_o g[b,p,o] * fo[b, p + sign*off(o), ch]
//   sign=+1, g = dcorr, fo = f2  -> d f1 ;  sign=-1, g = dcT, fo = f1 -> d f2
template <int R>
__global__ __launch_bounds__(256) void local_corr_bwd_kernel(const float* __restrict__ g, const float* __restrict__ fo,
                                                              float* __restrict__ df, int h, int w, int C, int cstride,
                                                              int sign, float inv_sqrt_c) {
    constexpr int K = 2 * R + 1, KK = K * K, HT = CT + 2 * R;
    constexpr int GSTR = KK + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* fs = sm;                   // [HT*HT][CSTR]
    float* gs = sm + HT * HT * CSTR;  // [64][GSTR]
    const int t = threadIdx.x;
    const int tiles_x = (w + CT - 1) / CT;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT, tx0 = (blockIdx.x % tiles_x) * CT;
    const int pix = t & 63, cg = t >> 6;  // 8 channels per thread inside the 32-channel stage
    const int py = pix >> 3, px = pix & 7;
    const float* gb = g + (size_t)b * h * w * cstride;
    const float* fb = fo + (size_t)b * h * w * C;
    float* dfb = df + (size_t)b * h * w * C;

    for (int e = t; e < 64 * KK; e += 256) {
        const int p = e / KK, o = e - p * KK;
        const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
        gs[p * GSTR + o] = (y < h && x < w) ? gb[((size_t)y * w + x) * cstride + o] : 0.f;
    }
    for (int c0 = 0; c0 < C; c0 += CC) {
        __syncthreads();
        for (int e = t; e < HT * HT * (CC / 4); e += 256) {
            const int c4 = e & 7, hp = e >> 3;
            const int hy = hp / HT, hx = hp - hy * HT;
            const int y = ty0 + hy - R, x = tx0 + hx - R;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < h && x >= 0 && x < w) v = *reinterpret_cast<const f32x4*>(fb + ((size_t)y * w + x) * C + c0 + c4 * 4);
            *reinterpret_cast<f32x4*>(&fs[hp * CSTR + c4 * 4]) = v;
        }
        __syncthreads();
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < K; ++a) {
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const float gv = gs[pix * GSTR + a * K + c];
                const int ny = py + R + sign * (c - R), nx = px + R + sign * (a - R);
                const float* src = &fs[(ny * HT + nx) * CSTR + cg * 8];
                a0 += gv * *reinterpret_cast<const f32x4*>(src);
                a1 += gv * *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
        const int y = ty0 + py, x = tx0 + px;
        if (y < h && x < w) {
            float* dst = dfb + ((size_t)y * w + x) * C + c0 + cg * 8;
            *reinterpret_cast<f32x4*>(dst) = a0 * inv_sqrt_c;
            *reinterpret_cast<f32x4*>(dst + 4) = a1 * inv_sqrt_c;
        }
    }
}

template <int R>
static int corr_fwd_launch(const float* f1, const float* f2, float* corr, int B, int h, int w, int C, int cstride, hipStream_t s) {
    constexpr int HT = CT + 2 * R;
    size_t lds = (size_t)(HT * HT + 64) * CSTR * sizeof(float);
    const size_t need_out = (size_t)64 * cstride * sizeof(float);
    if (need_out > lds) lds = need_out;
    const int tiles = cdiv(h, CT) * cdiv(w, CT);
    hipLaunchKernelGGL((local_corr_fwd_kernel<R>), dim3(tiles, B), dim3(256), lds, s, f1, f2, corr, h, w, C, cstride,
                       1.0f / sqrtf((float)C));
    return check_launch("local_corr_fwd");
}

}  // namespace rpnet

#define RPNET_CORR_DISPATCH(R_, ...)                         \
    switch (R_) {                                              \
        case 1: { constexpr int RR = 1; __VA_ARGS__; } break;         \
        case 2: { constexpr int RR = 2; __VA_ARGS__; } break;         \
        case 3: { constexpr int RR = 3; __VA_ARGS__; } break;         \
        case 4: { constexpr int RR = 4; __VA_ARGS__; } break;         \
        case 5: { constexpr int RR = 5; __VA_ARGS__; } break;         \
        default: rpnet::set_error("local_corr: radius %d not in 1..5", R_); return RPNET_ERR_SHAPE; \
    }

extern "C" int rpnet_local_corr_fwd(const float* f1, const float* f2, float* corr, int B, int h, int w, int C, int r,
                                    int cstride, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f1 && f2 && corr, RPNET_ERR_ARG, "local_corr_fwd: null pointer");
    RPNET_REQUIRE(C % CC == 0 && cstride >= (2 * r + 1) * (2 * r + 1) && cstride <= 160, RPNET_ERR_SHAPE,
                  "local_corr_fwd: C=%d (multiple of 32) cstride=%d", C, cstride);
    int rc = 0;
    RPNET_CORR_DISPATCH(r, rc = corr_fwd_launch<RR>(f1, f2, corr, B, h, w, C, cstride, (hipStream_t)stream));
    return rc;
}

extern "C" size_t rpnet_local_corr_bwd_workspace_bytes(int B, int h, int w, int cstride) {
    return (size_t)B * h * w * cstride * sizeof(float);
}

extern "C" int rpnet_local_corr_bwd(const float* f1, const float* f2, const float* dcorr, float* df1, float* df2, int B,
                                       int h, int w, int C, int r, int cstride, void* workspace, size_t workspace_bytes,
                                       rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f1 && f2 && dcorr && df1 && df2 && workspace, RPNET_ERR_ARG, "local_corr_bwd: null pointer");
    RPNET_REQUIRE(C % CC == 0 && cstride >= (2 * r + 1) * (2 * r + 1), RPNET_ERR_SHAPE, "local_corr_bwd: C=%d cstride=%d", C, cstride);
    RPNET_REQUIRE(workspace_bytes >= rpnet_local_corr_bwd_workspace_bytes(B, h, w, cstride), RPNET_ERR_WORKSPACE,
                  "local_corr_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* dct = (float*)workspace;
    const float isc = 1.0f / sqrtf((float)C);
    const int tiles = cdiv(h, CT) * cdiv(w, CT);
    const size_t total = (size_t)B * h * w * cstride;
    int nb = (int)((total + 255) / 256);
    if (nb > 16384) nb = 16384;
    RPNET_CORR_DISPATCH(r, {
        constexpr int K = 2 * RR + 1, HT = CT + 2 * RR;
        const size_t lds = (size_t)(HT * HT * CSTR + 64 * (K * K + 1)) * sizeof(float);
        // > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&local_corr_bwd_kernel<RR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((local_corr_bwd_kernel<RR>), dim3(tiles, B), dim3(256), lds, s, dcorr, f2, df1, h, w, C, cstride, +1, isc);
        hipLaunchKernelGGL((corr_transpose_kernel<RR>), dim3(nb), dim3(256), 0, s, dcorr, dct, B, h, w, cstride);
        hipLaunchKernelGGL((local_corr_bwd_kernel<RR>), dim3(tiles, B), dim3(256), lds, s, (const float*)dct, f1, df2, h, w, C, cstride, -1, isc);
    });
    return check_launch("local_corr_bwd");
}
