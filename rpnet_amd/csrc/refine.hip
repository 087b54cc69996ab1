// The glue between two refinement iterations of RP_Net.forward (net/rp_net.py:301-311) as ONE launch, and its backward as one
// launch + the prototype-gradient reduce.  Per iteration the reference runs, on the 64-channel output of cre.q:
//     BatchNorm + ReLU (net/rp_net.py:65-69)  ->  calDist x (1 + Wa) (:301, :353-363)  ->  stack, F.interpolate(bilinear) (:302-303)
//     ->  softmax(dim=1)[:, 1] (:308)  ->  > 0.5 unless soft_mask (:309-310)  ->  avg_pool2d(., 4) (:311)
// and the next iteration starts with x * mask and x * (1 - mask) (:283) in front of its two 3x3 convolutions.  Rounds 1 - 4 ran
// that as a serial chain of six small launches + two operand-split launches (profiles/r04_mfma_idle.txt: 170 - 190 us without
// MFMA work per iteration).  Here a block owns an 8 x 8 patch of feature pixels of one episode:
//   A  the patch + a one-pixel halo (all the x4 bilinear up-sampling reads): BatchNorm affine + ReLU (written for the patch),
//      cosine x scaler against the K prototypes, 16 lanes per pixel (float4 each, xor-shuffle sums) -> LDS, written for the patch;
//   B  the 32 x 32 logits of the patch from LDS (a thread = one row of four pixels, all K classes: 16-byte stores), softmax,
//      threshold, the 4 x 4 average -> the next mask;
//   C  (optional) the next CRE call's operand planes fp16((x * mask) / s), fp16((x * (1 - mask)) / s) of the patch's 64 feature
//      pixels (two planes each: x = h + l, or one / three bf16) — one read of x instead of two launches reading it twice.
// Every stage keeps the arithmetic and the summation order of the kernel it replaces (bn_relu_kernel, cosine_match_fwd_kernel,
// bilinear_up_fwd_kernel, softmax_thresh_pool4_kernel, split_f16_kernel / split_bf16_kernel): same bits, tests/test_gpu_ops.py.
// HBM / latency-bound: per block 29 KB of y in, 16 KB z + 8 KB logits out, + 64 KB of x in and 128 KB of planes out.
#include "matcher.h"
#include "split_bf16.h"

namespace rpnet {

constexpr int RT = 8;              // feature pixels per patch side
constexpr int RH = RT + 2;         // + halo
constexpr int RF = 64;             // channels of the matched features (cre.q: num_feat = 64, net/rp_net.py:49)
constexpr int RL = RF / 4;         // lanes per pixel

struct RefineFwd {
    const float* y; const float* bn_scale; const float* bn_shift; const float* proto;
    float* z; float* pred; float* logits; float* mask_next;
    const float* x; const float* x_scale; unsigned short* xk; unsigned short* xq;
    float scaler; int soft, K, h, w, C;
    size_t plane_elems;
};

template <int NP>
__global__ __launch_bounds__(256) void refine_glue_fwd_kernel(const RefineFwd a) {
    RPNET_PASS_PRIORITY();
    __shared__ float pred_s[kMaxK][RH * RH];
    __shared__ float sm[RT * RT][17];
    __shared__ float msk[RT * RT];
    const int t = threadIdx.x, l = t % RL, pl = t / RL;
    const int b = blockIdx.y, K = a.K, h = a.h, w = a.w, H = 4 * h, W = 4 * w;
    const int tiles_x = w / RT;
    const int ty0 = (blockIdx.x / tiles_x) * RT, tx0 = (blockIdx.x % tiles_x) * RT;

    // ---- A: BatchNorm + ReLU, cosine match (patch + halo)
    f32x4 p[kMaxK]; float pn[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
        p[k] = f32x4{0.f, 0.f, 0.f, 0.f}; pn[k] = 1.f;
        if (k < K) {
            p[k] = *reinterpret_cast<const f32x4*>(a.proto + ((size_t)b * K + k) * RF + l * 4);
            const float n2 = group_sum<RL>(p[k][0] * p[k][0] + p[k][1] * p[k][1] + p[k][2] * p[k][2] + p[k][3] * p[k][3]);
            pn[k] = fmaxf(sqrtf(n2), kCosEps);
        }
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.bn_scale) {
        sc = *reinterpret_cast<const f32x4*>(a.bn_scale + l * 4);
        sh = *reinterpret_cast<const f32x4*>(a.bn_shift + l * 4);
    }
    for (int pass = 0; pass * 16 < RH * RH; ++pass) {
        const int pos = min(pass * 16 + pl, RH * RH - 1);
        const int hy = pos / RH, hx = pos - hy * RH;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const int cy = min(max(gy, 0), h - 1), cx = min(max(gx, 0), w - 1);        // positions outside the image are never read in B
        const size_t q = ((size_t)b * h + cy) * w + cx;
        f32x4 v = *reinterpret_cast<const f32x4*>(a.y + q * RF + l * 4);
        const bool inner = hy >= 1 && hy <= RT && hx >= 1 && hx <= RT && pass * 16 + pl < RH * RH;
        if (a.bn_scale) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
            if (inner) *reinterpret_cast<f32x4*>(a.z + q * RF + l * 4) = v;
        }
        const float nf = fmaxf(sqrtf(group_sum<RL>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3])), kCosEps);
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) {
                const float d = group_sum<RL>(v[0] * p[k][0] + v[1] * p[k][1] + v[2] * p[k][2] + v[3] * p[k][3]);
                if (l == 0) {
                    const float pr = a.scaler * d / (nf * pn[k]);
                    pred_s[k][pos] = pr;
                    if (inner) a.pred[((size_t)b * K + k) * h * w + (size_t)cy * w + cx] = pr;
                }
            }
    }
    __syncthreads();

    // ---- B: bilinear x4 -> logits, softmax / threshold / 4 x 4 average -> next mask
    {
        const int fx = t & 7, dy = (t >> 3) & 3, fy = t >> 5;
        const int Y = (ty0 + fy) * 4 + dy, X0 = (tx0 + fx) * 4;
        const float rsy = (float)h / (float)H, rsx = (float)w / (float)W;
        int y0, y1; float wy0, wy1;
        bl_taps(Y, rsy, h, y0, y1, wy0, wy1);
        const int r0 = (y0 - (ty0 - 1)) * RH - (tx0 - 1), r1 = (y1 - (ty0 - 1)) * RH - (tx0 - 1);
        float lg[kMaxK][4];
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            int x0, x1; float wx0, wx1;
            bl_taps(X0 + dx, rsx, w, x0, x1, wx0, wx1);
#pragma unroll
            for (int k = 0; k < kMaxK; ++k)
                if (k < K) {
                    const float* ps = pred_s[k];
                    lg[k][dx] = wy0 * (wx0 * ps[r0 + x0] + wx1 * ps[r0 + x1]) + wy1 * (wx0 * ps[r1 + x0] + wx1 * ps[r1 + x1]);
                }
        }
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K)
                *reinterpret_cast<f32x4*>(a.logits + (((size_t)b * K + k) * H + Y) * W + X0) = f32x4{lg[k][0], lg[k][1], lg[k][2], lg[k][3]};
        if (a.mask_next) {
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                float mx = lg[0][dx];
#pragma unroll
                for (int k = 1; k < kMaxK; ++k)
                    if (k < K) mx = fmaxf(mx, lg[k][dx]);
                float den = 0.f;
#pragma unroll
                for (int k = 0; k < kMaxK; ++k)
                    if (k < K) den += expf(lg[k][dx] - mx);
                const float p1 = expf(lg[1][dx] - mx) / den;
                sm[fy * RT + fx][dy * 4 + dx] = a.soft ? p1 : (p1 > 0.5f ? 1.f : 0.f);
            }
        }
    }
    if (!a.mask_next) return;
    __syncthreads();
    if (t < RT * RT) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += sm[t][i];
        const float m = acc / 16.f;
        msk[t] = m;
        a.mask_next[((size_t)b * h + ty0 + t / RT) * w + tx0 + t % RT] = m;
    }
    if (!a.x) return;
    __syncthreads();

    // ---- C: operand planes of x * mask and x * (1 - mask) for the next call's two 3x3 convolutions
    {
        const int C8 = a.C / 8, ppp = 256 / C8;
        const float inv = NP <= 2 ? 1.f / *a.x_scale : 1.f;
        const int c8 = t % C8, pq = t / C8;
        for (int pi = pq; pi < RT * RT; pi += ppp) {
            const size_t row = ((size_t)b * h + ty0 + pi / RT) * w + tx0 + pi % RT;
            const size_t off = row * a.C + c8 * 8;
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.x + off), u1 = *reinterpret_cast<const f32x4*>(a.x + off + 4);
            const float f1 = msk[pi], f2 = 1.f - f1;
            float vk[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]}, vq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { vq[q] = __fmul_rn(vk[q], f2); vk[q] = __fmul_rn(vk[q], f1); }     // (x * f rounded once: no contraction with 1 - m)
            if (NP <= 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { vk[q] *= inv; vq[q] *= inv; }
            }
            u32x4 ok[NP], oq[NP];
            split8<NP>(vk, ok);
            split8<NP>(vq, oq);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                *reinterpret_cast<u32x4*>(a.xk + pp * a.plane_elems + off) = ok[pp];
                *reinterpret_cast<u32x4*>(a.xq + pp * a.plane_elems + off) = oq[pp];
            }
        }
    }
}

// Backward of stages A (cosine) and B (bilinear): dlogits [B][K][4h][4w] -> df [B][h][w][64] and the per-patch partial sums of the
// prototype gradient (dpart [B][patches][K][64], summed by cosine_dproto_final).  Same arithmetic per pixel as
// bilinear_up_bwd_kernel + cosine_match_bwd_kernel; the prototype gradient adds its pixels in patch order.
struct RefineBwd {
    const float* dlogits; const float* f; const float* proto; float* df; float* dpart;
    float scaler; int K, h, w;
};

// <= 128 registers per lane: a wave of this pass then fits on a SIMD beside a wave of the LDS-DMA convolution kernel (384 of the 512)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void refine_glue_bwd_kernel(const RefineBwd a) {
    RPNET_PASS_PRIORITY();
    __shared__ float dps[kMaxK][RT * RT];
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int t = threadIdx.x, b = blockIdx.y, K = a.K, h = a.h, w = a.w, H = 4 * h, W = 4 * w;
    const int tiles_x = w / RT;
    const int ty0 = (blockIdx.x / tiles_x) * RT, tx0 = (blockIdx.x % tiles_x) * RT;
    // ---- the adjoint of the bilinear up-sampling: four lanes share a feature pixel, each takes every fourth row of its window
    {
        const int pix = t >> 2, part = t & 3;
        const int y = ty0 + pix / RT, x = tx0 + pix % RT;
        const float rsy = (float)h / (float)H, rsx = (float)w / (float)W;
        const int Y0 = max(0, (y - 1) * 4 - 1), Y1 = min(H - 1, (y + 1) * 4 + 4 + 1);
        const int X0 = max(0, (x - 1) * 4 - 1), X1 = min(W - 1, (x + 1) * 4 + 4 + 1);
        // The window of a feature pixel is at most 15 x 15 logit gradients (rows / columns (y - 1) 4 - 1 .. (y + 1) 4 + 5), this lane's share
        // at most 4 rows.  Round 6: ALL of a lane's loads are issued before the first is used (fixed trip counts, zeros outside the window)
        // — the launch sits on the backward chain beside two streaming GEMM kernels, where a memory round trip takes microseconds: the
        // nested loops below it replaced were a chain of ~100 dependent 4-byte loads per lane, 100 - 190 us per launch in the step against
        // 26 alone (profiles/r06_timeline.txt).  Same products, same order of additions: same bits.
        constexpr int NRW = 4, NCW = 15;
        float wx[NCW], wyv[NRW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) wx[c] = (X0 + c <= X1) ? bl_weight(X0 + c, x, rsx, w) : 0.f;
#pragma unroll
        for (int r = 0; r < NRW; ++r) wyv[r] = (Y0 + part + 4 * r <= Y1) ? bl_weight(Y0 + part + 4 * r, y, rsy, h) : 0.f;
        for (int k = 0; k < K; ++k) {
            const float* g = a.dlogits + ((size_t)b * K + k) * H * W;
            float v[NRW][NCW];
#pragma unroll
            for (int r = 0; r < NRW; ++r) {
                const int Y = Y0 + part + 4 * r;
                const bool rowok = Y <= Y1 && wyv[r] != 0.f;
#pragma unroll
                for (int c = 0; c < NCW; ++c) v[r][c] = (rowok && X0 + c <= X1) ? g[(size_t)Y * W + X0 + c] : 0.f;
            }
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < NRW; ++r) {
                if (wyv[r] == 0.f) continue;
                float row = 0.f;
#pragma unroll
                for (int c = 0; c < NCW; ++c)
                    if (X0 + c <= X1) row += v[r][c] * wx[c];
                acc += wyv[r] * row;
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (part == 0) dps[k][pix] = acc;
        }
    }
    __syncthreads();
    // ---- cosine backward, 16 lanes per pixel
    const int l = t % RL, pl = t / RL;
    f32x4 p[kMaxK], pnrm[kMaxK], dp[kMaxK];
    float pn_raw[kMaxK], pn_c[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
        p[k] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[k] = p[k]; pnrm[k] = p[k]; pn_raw[k] = 1.f; pn_c[k] = 1.f;
        if (k < K) {
            p[k] = *reinterpret_cast<const f32x4*>(a.proto + ((size_t)b * K + k) * RF + l * 4);
            pn_raw[k] = sqrtf(group_sum<RL>(p[k][0] * p[k][0] + p[k][1] * p[k][1] + p[k][2] * p[k][2] + p[k][3] * p[k][3]));
            pn_c[k] = fmaxf(pn_raw[k], kCosEps);
            pnrm[k] = p[k] * (1.f / pn_c[k]);
        }
    }
    for (int pix = pl; pix < RT * RT; pix += 16) {
        const size_t q = ((size_t)b * h + ty0 + pix / RT) * w + tx0 + pix % RT;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.f + q * RF + l * 4);
        const float nf = sqrtf(group_sum<RL>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
        const float nfc = fmaxf(nf, kCosEps);
        const f32x4 fn = v * (1.f / nfc);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) {
                const float go = a.scaler * dps[k][pix];
                const float fp = group_sum<RL>(v[0] * pnrm[k][0] + v[1] * pnrm[k][1] + v[2] * pnrm[k][2] + v[3] * pnrm[k][3]);
                g += go * (pnrm[k] * (1.f / nfc));
                if (nf > kCosEps) g -= (go * fp / (nfc * nfc * nf)) * v;
                const float fnp = group_sum<RL>(fn[0] * p[k][0] + fn[1] * p[k][1] + fn[2] * p[k][2] + fn[3] * p[k][3]);
                dp[k] += go * (fn * (1.f / pn_c[k]));
                if (pn_raw[k] > kCosEps) dp[k] -= (go * fnp / (pn_c[k] * pn_c[k] * pn_raw[k])) * p[k];
            }
        *reinterpret_cast<f32x4*>(a.df + q * RF + l * 4) = g;
    }
    for (int k = 0; k < K; ++k) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[t * 4]) = dp[k];
        __syncthreads();
        if (pl == 0) {
            f32x4 r = dp[k];
            for (int gidx = 1; gidx < 16; ++gidx) r += *reinterpret_cast<const f32x4*>(&red[(gidx * RL + l) * 4]);
            *reinterpret_cast<f32x4*>(a.dpart + (((size_t)b * gridDim.x + blockIdx.x) * K + k) * RF + l * 4) = r;
        }
    }
}

}  // namespace rpnet

extern "C" int rpnet_refine_glue_supported(int K, int h, int w, int F, int C, int planes) {
    using namespace rpnet;
    if (F != RF || K < 2 || K > kMaxK || h % RT || w % RT || h < RT || w < RT) return 0;
    if (planes == 0) return 1;
    return (planes >= 1 && planes <= 3 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0) ? 1 : 0;
}

extern "C" int rpnet_refine_glue_fwd(const float* y, const float* bn_scale, const float* bn_shift, const float* proto, float scaler,
                                     float* z, float* pred, float* logits, float* mask_next, int soft, const float* x,
                                     const float* x_scale, void* xk_planes, void* xq_planes, int planes, int B, int K, int h, int w,
                                     int F, int C, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(y && proto && pred && logits, RPNET_ERR_ARG, "refine_glue_fwd: null pointer");
    RPNET_REQUIRE(!bn_scale || (bn_shift && z), RPNET_ERR_ARG, "refine_glue_fwd: the BatchNorm form needs shift and the output z");
    const bool want_planes = x != nullptr;
    RPNET_REQUIRE(!want_planes || (mask_next && xk_planes && xq_planes && (planes == 3 || x_scale)), RPNET_ERR_ARG,
                  "refine_glue_fwd: operand planes need the mask output, both plane buffers and (fp16) the tensor scale");
    RPNET_REQUIRE(rpnet_refine_glue_supported(K, h, w, F, C, want_planes ? planes : 0), RPNET_ERR_SHAPE,
                  "refine_glue_fwd: K=%d h=%d w=%d F=%d C=%d planes=%d", K, h, w, F, C, planes);
    RefineFwd a{y, bn_scale, bn_shift, proto, z, pred, logits, mask_next, x, x_scale, (unsigned short*)xk_planes,
                (unsigned short*)xq_planes, scaler, soft, K, h, w, C, (size_t)B * h * w * C};
    const dim3 grid((h / RT) * (w / RT), B);
    hipStream_t s = (hipStream_t)stream;
    if (want_planes && planes == 3) hipLaunchKernelGGL(refine_glue_fwd_kernel<3>, grid, dim3(256), 0, s, a);
    else if (want_planes && planes == 1) hipLaunchKernelGGL(refine_glue_fwd_kernel<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(refine_glue_fwd_kernel<2>, grid, dim3(256), 0, s, a);
    return check_launch("refine_glue_fwd");
}

extern "C" size_t rpnet_refine_glue_bwd_workspace_bytes(int B, int K, int h, int w, int F) {
    return (size_t)B * (h / rpnet::RT) * (w / rpnet::RT) * K * F * sizeof(float);
}

extern "C" int rpnet_refine_glue_bwd(const float* dlogits, const float* f, const float* proto, float scaler, float* df, float* dproto,
                                     int B, int K, int h, int w, int F, void* workspace, size_t workspace_bytes,
                                     rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(dlogits && f && proto && df && dproto && workspace, RPNET_ERR_ARG, "refine_glue_bwd: null pointer");
    RPNET_REQUIRE(rpnet_refine_glue_supported(K, h, w, F, 0, 0), RPNET_ERR_SHAPE, "refine_glue_bwd: K=%d h=%d w=%d F=%d", K, h, w, F);
    RPNET_REQUIRE(workspace_bytes >= rpnet_refine_glue_bwd_workspace_bytes(B, K, h, w, F), RPNET_ERR_WORKSPACE, "refine_glue_bwd: workspace");
    RefineBwd a{dlogits, f, proto, df, (float*)workspace, scaler, K, h, w};
    const int nt = (h / RT) * (w / RT);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(refine_glue_bwd_kernel, dim3(nt, B), dim3(256), 0, s, a);
    launch_cosine_dproto_final((const float*)workspace, dproto, B, nt, K, F, s);
    return check_launch("refine_glue_bwd");
}
