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

// forward: register-tiled so the kernel is VALU-bound, not LDS-bound.  A thread owns 4 horizontally
// adjacent pixels and ONE vertical offset (dy): per 4 channels it reads the 4 f1 pixels and the
// 4 + 2R f2 pixels of the shifted row once (float4 each) and does 4 * K * 4 FMAs (K = 2R+1
// horizontal offsets) — 0.1 LDS reads per FMA instead of 0.26 with one pixel per thread.
template <int R>
__global__ __launch_bounds__(256) void local_corr_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ corr, int h, int w, int C, int cstride,
                                                              float inv_sqrt_c) {
    constexpr int K = 2 * R + 1, KK = K * K, HT = CT + 2 * R, NX = 4 + 2 * R;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* f2s = sm;                    // [HT*HT][CSTR]
    float* f1s = sm + HT * HT * CSTR;   // [64][CSTR]
    const int t = threadIdx.x;
    const int tiles_x = (w + CT - 1) / CT;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT, tx0 = (blockIdx.x % tiles_x) * CT;
    // thread -> (quad: row py, pixels px0..px0+3; vertical offset index cdy); 16 quads x K offsets
    const int quad = t & 15, cdy = t >> 4;
    const int py = quad >> 1, px0 = (quad & 1) * 4;
    const bool active = cdy < K;

    float acc[4][K];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < K; ++a) acc[p][a] = 0.f;

    const float* f1b = f1 + (size_t)b * h * w * C;
    const float* f2b = f2 + (size_t)b * h * w * C;
    const int f1row = (py * 8 + px0) * CSTR;
    const int f2row = ((py + (active ? cdy : 0)) * HT + px0) * CSTR;
    for (int c0 = 0; c0 < C; c0 += CC) {
        __syncthreads();
        for (int e = t; e < HT * HT * (CC / 4); e += 256) {
            const int c4 = e & 7, hp = e >> 3;
            const int hy = hp / HT, hx = hp - hy * HT;
            const int y = ty0 + hy - R, x = tx0 + hx - R;
            const bool ok = y >= 0 && y < h && x >= 0 && x < w;     // clamped, unconditional load (no per-load wait)
            const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
            *reinterpret_cast<f32x4*>(&f2s[hp * CSTR + c4 * 4]) =
                *reinterpret_cast<const f32x4*>(f2b + ((size_t)yc * w + xc) * C + c0 + c4 * 4) * (ok ? 1.f : 0.f);
        }
        for (int e = t; e < 64 * (CC / 4); e += 256) {
            const int c4 = e & 7, p = e >> 3;
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            const bool ok = y < h && x < w;
            *reinterpret_cast<f32x4*>(&f1s[p * CSTR + c4 * 4]) =
                *reinterpret_cast<const f32x4*>(f1b + ((size_t)min(y, h - 1) * w + min(x, w - 1)) * C + c0 + c4 * 4) * (ok ? 1.f : 0.f);
        }
        __syncthreads();
        if (active) {
#pragma unroll 2
            for (int c4 = 0; c4 < CC / 4; ++c4) {
                f32x4 a1[4], b2[NX];
#pragma unroll
                for (int p = 0; p < 4; ++p) a1[p] = *reinterpret_cast<const f32x4*>(&f1s[f1row + p * CSTR + c4 * 4]);
#pragma unroll
                for (int j = 0; j < NX; ++j) b2[j] = *reinterpret_cast<const f32x4*>(&f2s[f2row + j * CSTR + c4 * 4]);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int a = 0; a < K; ++a) {
                        const f32x4 bv = b2[p + a];
                        acc[p][a] += a1[p][0] * bv[0] + a1[p][1] * bv[1] + a1[p][2] * bv[2] + a1[p][3] * bv[3];
                    }
            }
        }
    }
    // stage the tile as [64][cstride] rows and write them out contiguously
    __syncthreads();
    float* outs = sm;  // 64 * cstride floats (host checks it fits)
    if (active) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int a = 0; a < K; ++a) outs[(py * 8 + px0 + p) * cstride + a * K + cdy] = acc[p][a] * inv_sqrt_c;
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
                                      int cstride, const FastDiv fS, const FastDiv fW, const FastDiv fH) {
    RPNET_PASS_PRIORITY();
    constexpr int K = 2 * R + 1, KK = K * K;
    const size_t total = (size_t)B * h * w * cstride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        unsigned p, py, pb;
        const int o = (int)fS.divmod((unsigned)i, p);
        const int x = (int)fW.divmod(p, py);
        const int y = (int)fH.divmod(py, pb);
        const int b = (int)pb;
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
// Register-tiled like the forward: a thread owns 4 adjacent pixels x 4 channels; per vertical offset
// it reads the 4 x K window gradients (3 float4 per pixel from a [px][dy][dx] LDS image) and the
// 4 + 2R shifted feature pixels once and does 4 * K * 4 FMAs.  64 channels per LDS stage.
constexpr int BC = 64;          // channels per stage (backward)
constexpr int BSTR = BC + 4;    // padded feature row
// BCT: channels per stage — 64, or 32 for radius 6 / 7 whose halo (20^2 / 22^2 pixels) would not fit the LDS beside the window
// gradients at 64 (half of the threads then only help staging)
template <int R, int SIGN, int BCT = 64>
__global__ __launch_bounds__(256) void local_corr_bwd_kernel(const float* __restrict__ g, const float* __restrict__ fo,
                                                              float* __restrict__ df, int h, int w, int C, int cstride,
                                                              float inv_sqrt_c, const float* __restrict__ add) {
    constexpr int K = 2 * R + 1, HT = CT + 2 * R, NX = 4 + 2 * R, KA = (K + 3) & ~3;
    constexpr int BC = BCT, BSTR = BCT + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* fs = sm;                   // [HT*HT][BSTR]
    float* gs = sm + HT * HT * BSTR;  // [64][K][KA]
    const int t = threadIdx.x;
    const int tiles_x = (w + CT - 1) / CT;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT, tx0 = (blockIdx.x % tiles_x) * CT;
    const int quad = t & 15, cg = t >> 4;       // 16 quads x 16 channel groups of 4
    const int py = quad >> 1, px0 = (quad & 1) * 4;
    const float* gb = g + (size_t)b * h * w * cstride;
    const float* fb = fo + (size_t)b * h * w * C;
    float* dfb = df + (size_t)b * h * w * C;
    const float* addb = add ? add + (size_t)b * h * w * C : nullptr;   // a second gradient of the same tensor, summed here

    for (int e = t; e < 64 * K * KA; e += 256) {
        const int a = e % KA, c = (e / KA) % K, p = e / (KA * K);
        const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
        const bool ok = a < K && y < h && x < w;
        gs[e] = gb[((size_t)min(y, h - 1) * w + min(x, w - 1)) * cstride + min(a, K - 1) * K + c] * (ok ? 1.f : 0.f);
    }
    for (int c0 = 0; c0 < C; c0 += BC) {
        __syncthreads();
        for (int e = t; e < HT * HT * (BC / 4); e += 256) {
            const int c4 = e % (BC / 4), hp = e / (BC / 4);
            const int hy = hp / HT, hx = hp - hy * HT;
            const int y = ty0 + hy - R, x = tx0 + hx - R;
            const bool ok = y >= 0 && y < h && x >= 0 && x < w;
            const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
            *reinterpret_cast<f32x4*>(&fs[hp * BSTR + c4 * 4]) =
                *reinterpret_cast<const f32x4*>(fb + ((size_t)yc * w + xc) * C + c0 + c4 * 4) * (ok ? 1.f : 0.f);
        }
        __syncthreads();
        if (cg * 4 >= BC) continue;          // BCT = 32: channel groups 8 .. 15 have nothing to compute (uniform per wave)
        f32x4 acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < K; ++c) {
            const int ny = py + R + SIGN * (c - R);
            f32x4 fv[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) fv[j] = *reinterpret_cast<const f32x4*>(&fs[(ny * HT + px0 + j) * BSTR + cg * 4]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float gv[KA];
#pragma unroll
                for (int q = 0; q < KA / 4; ++q)
                    *reinterpret_cast<f32x4*>(&gv[q * 4]) = *reinterpret_cast<const f32x4*>(&gs[((py * 8 + px0 + p) * K + c) * KA + q * 4]);
#pragma unroll
                for (int a = 0; a < K; ++a) acc[p] += gv[a] * fv[SIGN > 0 ? p + a : p + (K - 1 - a)];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = ty0 + py, x = tx0 + px0 + p;
            if (y < h && x < w) {
                const size_t o = ((size_t)y * w + x) * C + c0 + cg * 4;
                f32x4 v = acc[p] * inv_sqrt_c;
                if (addb) v += *reinterpret_cast<const f32x4*>(addb + o);
                *reinterpret_cast<f32x4*>(dfb + o) = v;
            }
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
    if (lds > 64 * 1024)      // radius 6 / 7: > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&local_corr_fwd_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((local_corr_fwd_kernel<R>), dim3(tiles, B), dim3(256), lds, s, f1, f2, corr, h, w, C, cstride,
                       1.0f / sqrtf((float)C));
    return check_launch("local_corr_fwd");
}

// dcT for the split-bf16 backward (corr_split.hip)
int launch_corr_transpose(const float* dcorr, float* dct, int B, int h, int w, int cstride, int r, hipStream_t s) {
    const size_t total = (size_t)B * h * w * cstride;
    int nb = (int)((total + 255) / 256);
    if (nb > 16384) nb = 16384;
    if (r != 5) { set_error("corr_transpose: radius %d", r); return RPNET_ERR_SHAPE; }
    if (total >= kIndex32) { set_error("corr_transpose: %zu elements do not fit the 32-bit index arithmetic", total); return RPNET_ERR_SHAPE; }
    hipLaunchKernelGGL((corr_transpose_kernel<5>), dim3(nb), dim3(256), 0, s, dcorr, dct, B, h, w, cstride, FastDiv(cstride), FastDiv(w), FastDiv(h));
    return check_launch("corr_transpose");
}

}  // namespace rpnet

#define RPNET_CORR_DISPATCH(R_, ...)                         \
    switch (R_) {                                              \
        case 1: { constexpr int RR = 1; __VA_ARGS__; } break;         \
        case 2: { constexpr int RR = 2; __VA_ARGS__; } break;         \
        case 3: { constexpr int RR = 3; __VA_ARGS__; } break;         \
        case 4: { constexpr int RR = 4; __VA_ARGS__; } break;         \
        case 5: { constexpr int RR = 5; __VA_ARGS__; } break;         \
        case 6: { constexpr int RR = 6; __VA_ARGS__; } break;         \
        case 7: { constexpr int RR = 7; __VA_ARGS__; } break;         \
        default: rpnet::set_error("local_corr: radius %d not in 1..7", R_); return RPNET_ERR_SHAPE; \
    }

extern "C" int rpnet_local_corr_fwd(const float* f1, const float* f2, float* corr, int B, int h, int w, int C, int r,
                                    int cstride, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f1 && f2 && corr, RPNET_ERR_ARG, "local_corr_fwd: null pointer");
    RPNET_REQUIRE(C % CC == 0 && cstride >= (2 * r + 1) * (2 * r + 1) && cstride <= 256, RPNET_ERR_SHAPE,
                  "local_corr_fwd: C=%d (multiple of 32) cstride=%d", C, cstride);
    int rc = 0;
    RPNET_CORR_DISPATCH(r, rc = corr_fwd_launch<RR>(f1, f2, corr, B, h, w, C, cstride, (hipStream_t)stream));
    return rc;
}

extern "C" size_t rpnet_local_corr_bwd_workspace_bytes(int B, int h, int w, int cstride) {
    return (size_t)B * h * w * cstride * sizeof(float);
}

extern "C" int rpnet_local_corr_bwd(const float* f1, const float* f2, const float* dcorr, float* df1, float* df2, int B,
                                       int h, int w, int C, int r, int cstride, const float* df1_add, void* workspace,
                                       size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f1 && f2 && dcorr && df1 && df2 && workspace, RPNET_ERR_ARG, "local_corr_bwd: null pointer");
    RPNET_REQUIRE(C % CC == 0 && cstride >= (2 * r + 1) * (2 * r + 1), RPNET_ERR_SHAPE, "local_corr_bwd: C=%d cstride=%d", C, cstride);
    RPNET_REQUIRE(C % BC == 0, RPNET_ERR_SHAPE, "local_corr_bwd: C=%d must be a multiple of 64", C);
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
        constexpr int K = 2 * RR + 1, HT = CT + 2 * RR, KA = (K + 3) & ~3;
        constexpr int BCT = RR <= 5 ? 64 : 32;
        const size_t lds = (size_t)(HT * HT * (BCT + 4) + 64 * K * KA) * sizeof(float);
        // > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&local_corr_bwd_kernel<RR, 1, BCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&local_corr_bwd_kernel<RR, -1, BCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((local_corr_bwd_kernel<RR, 1, BCT>), dim3(tiles, B), dim3(256), lds, s, dcorr, f2, df1, h, w, C, cstride, isc, df1_add);
        hipLaunchKernelGGL((corr_transpose_kernel<RR>), dim3(nb), dim3(256), 0, s, dcorr, dct, B, h, w, cstride, FastDiv(cstride), FastDiv(w), FastDiv(h));
        hipLaunchKernelGGL((local_corr_bwd_kernel<RR, -1, BCT>), dim3(tiles, B), dim3(256), lds, s, (const float*)dct, f1, df2, h, w, C, cstride, isc, (const float*)nullptr);
    });
    return check_launch("local_corr_bwd");
}
