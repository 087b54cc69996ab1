// Prototype matcher: masked average pooling in adjoint-mask form, cosine matching,
// bilinear up-sampling of the logits and the inter-iteration mask glue.  HBM/latency-bound
// fp32 kernels; features are NHWC [B][hw][C], logits are the reference's NCHW [B][K][H][W].
// Replaces getFeatures / getPrototype / calDist / F.interpolate / softmax / threshold /
// avg_pool2d of net/rp_net.py:288-311,353-391 and their autograd.
#include "matcher.h"

namespace rpnet {

// am[b,k,y,x] = sum_{Y,X} mask_k[b,Y,X] * wy(Y->y) * wx(X->x)
__global__ void mask_adjoint_kernel(const float* __restrict__ masks, float* __restrict__ am, int B, int nmask, int H,
                                    int W, int h, int w) {
    RPNET_PASS_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nmask * h * w) return;
    const int x = i % w, y = (i / w) % h, k = (i / (w * h)) % nmask, b = i / (w * h * nmask);
    const float rsy = (float)h / (float)H, rsx = (float)w / (float)W;
    const int sy = H / h + 1, sx = W / w + 1;
    const int Y0 = max(0, (y - 1) * (H / h) - 1), Y1 = min(H - 1, (y + 1) * (H / h) + sy);
    const int X0 = max(0, (x - 1) * (W / w) - 1), X1 = min(W - 1, (x + 1) * (W / w) + sx);
    const float* m = masks + ((size_t)k * B + b) * H * W;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        const float wy = bl_weight(Y, y, rsy, h);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = X0; X <= X1; ++X) row += m[(size_t)Y * W + X] * bl_weight(X, x, rsx, w);
        acc += wy * row;
    }
    am[i] = acc;
}

__global__ __launch_bounds__(256) void mask_sum_kernel(const float* __restrict__ masks, float* __restrict__ msum, int B,
                                                        int nmask, int HW) {
    RPNET_PASS_PRIORITY();
    __shared__ double sm4[4];
    const int b = blockIdx.x % B, k = blockIdx.x / B;
    const float* m = masks + ((size_t)k * B + b) * HW;
    double s = 0;
    if ((HW & 3) == 0) {        // 16-byte loads, four independent partial sums per thread
        double s4[4] = {0, 0, 0, 0};
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(m)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] += v[q];
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) s += m[i];
    }
    s = block_sum256(s, sm4);
    if (threadIdx.x == 0) msum[b * nmask + k] = (float)s;
}

constexpr int kPoolSplit = 64;      // pixel chunks per episode (blocks = 64 x B: 16 left a batch-4 launch on 64 blocks)
constexpr int kMaxMask = 4;

// partial[b][s][k][C] = sum_{q in chunk s} f[b,q,:] * am[b,k,q]
__global__ __launch_bounds__(256) void masked_pool_partial(const float* __restrict__ f, const float* __restrict__ am,
                                                            float* __restrict__ partial, int nmask, int hw, int C) {
    RPNET_PASS_PRIORITY();
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int t = threadIdx.x, C4 = C / 4, rows = 256 / C4;
    const int c4 = t % C4, qg = t / C4;
    const int s = blockIdx.x, b = blockIdx.y;
    const int chunk = (hw + kPoolSplit - 1) / kPoolSplit;
    const int q0 = s * chunk, q1 = min(q0 + chunk, hw);
    f32x4 acc[kMaxMask];
#pragma unroll
    for (int k = 0; k < kMaxMask; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (qg < rows)
        for (int q = q0 + qg; q < q1; q += rows) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(f + ((size_t)b * hw + q) * C + c4 * 4);
#pragma unroll
            for (int k = 0; k < kMaxMask; ++k)
                if (k < nmask) acc[k] += v * am[((size_t)b * nmask + k) * hw + q];
        }
    for (int k = 0; k < nmask; ++k) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[t * 4]) = acc[k];
        __syncthreads();
        if (qg == 0) {
            f32x4 r = acc[k];
            for (int g = 1; g < rows; ++g) r += *reinterpret_cast<const f32x4*>(&red[(g * C4 + c4) * 4]);
            *reinterpret_cast<f32x4*>(partial + (((size_t)b * kPoolSplit + s) * nmask + k) * C + c4 * 4) = r;
        }
    }
}

__global__ void masked_pool_final(const float* __restrict__ partial, const float* __restrict__ msum,
                                  float* __restrict__ proto, int B, int nmask, int C) {
    RPNET_PASS_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nmask * C) return;
    const int c = i % C, k = (i / C) % nmask, b = i / (C * nmask);
    float s = 0.f;
    for (int j = 0; j < kPoolSplit; ++j) s += partial[(((size_t)b * kPoolSplit + j) * nmask + k) * C + c];
    proto[i] = s / (msum[b * nmask + k] + 1e-5f);
}

__global__ __launch_bounds__(256) void masked_pool_bwd_kernel(const float* __restrict__ dproto, const float* __restrict__ am,
                                                               const float* __restrict__ msum, float* __restrict__ df,
                                                               int B, int nmask, int hw, int C, int accumulate) {
    RPNET_PASS_PRIORITY();
    const int C4 = C / 4;
    const size_t total = (size_t)B * hw * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t bq = i / C4;
        const int q = (int)(bq % hw), b = (int)(bq / hw);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nmask; ++k) {
            const float wgt = am[((size_t)b * nmask + k) * hw + q] / (msum[b * nmask + k] + 1e-5f);
            r += wgt * *reinterpret_cast<const f32x4*>(dproto + ((size_t)b * nmask + k) * C + c4 * 4);
        }
        f32x4* dst = reinterpret_cast<f32x4*>(df) + i;
        *dst = accumulate ? (*dst + r) : r;
    }
}

// ---- cosine match: a group of C/4 lanes owns one pixel (float4 each), xor-shuffle reduce (helpers: matcher.h)
template <int L>
__global__ __launch_bounds__(256) void cosine_match_fwd_kernel(const float* __restrict__ f, const float* __restrict__ proto,
                                                                float* __restrict__ pred, int B, int K, int hw, float scaler) {
    RPNET_PASS_PRIORITY();
    constexpr int C = L * 4, PPB = 256 / L;
    const int t = threadIdx.x, l = t % L, pl = t / L;
    const int b = blockIdx.y;
    f32x4 p[kMaxK]; float pn[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
        p[k] = f32x4{0.f, 0.f, 0.f, 0.f}; pn[k] = 1.f;
        if (k < K) {
            p[k] = *reinterpret_cast<const f32x4*>(proto + ((size_t)b * K + k) * C + l * 4);
            const float n2 = group_sum<L>(p[k][0] * p[k][0] + p[k][1] * p[k][1] + p[k][2] * p[k][2] + p[k][3] * p[k][3]);
            pn[k] = fmaxf(sqrtf(n2), kCosEps);
        }
    }
    for (int q = blockIdx.x * PPB + pl; q < hw; q += gridDim.x * PPB) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(f + ((size_t)b * hw + q) * C + l * 4);
        const float nf = fmaxf(sqrtf(group_sum<L>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3])), kCosEps);
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) {
                const float d = group_sum<L>(v[0] * p[k][0] + v[1] * p[k][1] + v[2] * p[k][2] + v[3] * p[k][3]);
                if (l == 0) pred[((size_t)b * K + k) * hw + q] = scaler * d / (nf * pn[k]);
            }
    }
}

// df per pixel; dproto partials per block -> workspace [B][nblk][K][C]
template <int L>
__global__ __launch_bounds__(256) void cosine_match_bwd_kernel(const float* __restrict__ f, const float* __restrict__ proto,
                                                                const float* __restrict__ dpred, float* __restrict__ df,
                                                                float* __restrict__ dpart, int B, int K, int hw,
                                                                float scaler, int accumulate_df) {
    RPNET_PASS_PRIORITY();
    constexpr int C = L * 4, PPB = 256 / L;
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int t = threadIdx.x, l = t % L, pl = t / L;
    const int b = blockIdx.y;
    f32x4 p[kMaxK], pnrm[kMaxK], dp[kMaxK];
    float pn_raw[kMaxK], pn_c[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
        p[k] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[k] = p[k]; pnrm[k] = p[k]; pn_raw[k] = 1.f; pn_c[k] = 1.f;
        if (k < K) {
            p[k] = *reinterpret_cast<const f32x4*>(proto + ((size_t)b * K + k) * C + l * 4);
            pn_raw[k] = sqrtf(group_sum<L>(p[k][0] * p[k][0] + p[k][1] * p[k][1] + p[k][2] * p[k][2] + p[k][3] * p[k][3]));
            pn_c[k] = fmaxf(pn_raw[k], kCosEps);
            pnrm[k] = p[k] * (1.f / pn_c[k]);
        }
    }
    for (int q = blockIdx.x * PPB + pl; q < hw; q += gridDim.x * PPB) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(f + ((size_t)b * hw + q) * C + l * 4);
        const float nf = sqrtf(group_sum<L>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
        const float nfc = fmaxf(nf, kCosEps);
        const f32x4 fn = v * (1.f / nfc);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) {
                const float go = scaler * dpred[((size_t)b * K + k) * hw + q];
                // cos = <fn, pnrm>;  d cos / d f = pnrm/nfc - [nf > eps] <f,pnrm> f / (nfc^2 nf)
                const float fp = group_sum<L>(v[0] * pnrm[k][0] + v[1] * pnrm[k][1] + v[2] * pnrm[k][2] + v[3] * pnrm[k][3]);
                g += go * (pnrm[k] * (1.f / nfc));
                if (nf > kCosEps) g -= (go * fp / (nfc * nfc * nf)) * v;
                // d cos / d p = fn/pnc - [pn > eps] <fn,p> p / (pnc^2 pn)
                const float fnp = group_sum<L>(fn[0] * p[k][0] + fn[1] * p[k][1] + fn[2] * p[k][2] + fn[3] * p[k][3]);
                dp[k] += go * (fn * (1.f / pn_c[k]));
                if (pn_raw[k] > kCosEps) dp[k] -= (go * fnp / (pn_c[k] * pn_c[k] * pn_raw[k])) * p[k];
            }
        f32x4* dst = reinterpret_cast<f32x4*>(df + ((size_t)b * hw + q) * C + l * 4);
        *dst = accumulate_df ? (*dst + g) : g;
    }
    for (int k = 0; k < K; ++k) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[t * 4]) = dp[k];
        __syncthreads();
        if (pl == 0) {
            f32x4 r = dp[k];
            for (int gidx = 1; gidx < PPB; ++gidx) r += *reinterpret_cast<const f32x4*>(&red[(gidx * L + l) * 4]);
            *reinterpret_cast<f32x4*>(dpart + (((size_t)b * gridDim.x + blockIdx.x) * K + k) * C + l * 4) = r;
        }
    }
}

// one block per (episode, prototype): 256 / C row groups share the nblk partial rows, four loads in flight each
// (one thread per output with a serial loop over 256 rows took 16 us)
__global__ __launch_bounds__(256) void cosine_dproto_final(const float* __restrict__ dpart, float* __restrict__ dproto, int B, int nblk, int K, int C) {
    RPNET_PASS_PRIORITY();
    __shared__ float red[256];
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const int t = threadIdx.x;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cw = min(C - c0, 256), parts = 256 / cw;
        const int c = t % cw, part = t / cw;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (part < parts) {
            const float* p = dpart + ((size_t)b * nblk * K + k) * C + c0 + c;
            const size_t rs = (size_t)K * C;
            int j = part;
            for (; j + 3 * parts < nblk; j += 4 * parts) {
                s0 += p[(size_t)j * rs]; s1 += p[(size_t)(j + parts) * rs];
                s2 += p[(size_t)(j + 2 * parts) * rs]; s3 += p[(size_t)(j + 3 * parts) * rs];
            }
            for (; j < nblk; j += parts) s0 += p[(size_t)j * rs];
        }
        red[t] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (t < cw) {
            float s = red[t];
            for (int q = 1; q < parts; ++q) s += red[q * cw + t];
            dproto[((size_t)b * K + k) * C + c0 + t] = s;
        }
        __syncthreads();
    }
}

void launch_cosine_dproto_final(const float* dpart, float* dproto, int B, int nblk, int K, int C, hipStream_t stream) {
    hipLaunchKernelGGL(cosine_dproto_final, dim3(B * K), dim3(256), 0, stream, dpart, dproto, B, nblk, K, C);
}

// blocks per episode of the cosine-match backward: enough for ~8 blocks per CU whatever the batch (a fixed 32 left the
// machine a quarter full at batch 4: 72 us at 512^2), never more than one block per 64 pixels
static int cos_blocks(int B, int hw) {
    int nb = 2048 / (B > 0 ? B : 1);
    nb = nb < 32 ? 32 : (nb > 256 ? 256 : nb);
    const int cap = hw / 64 > 0 ? hw / 64 : 1;
    return nb < cap ? nb : cap;
}

__global__ void bilinear_up_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int H, int W) {
    RPNET_PASS_PRIORITY();
    const size_t total = (size_t)planes * H * W;
    const float rsy = (float)h / (float)H, rsx = (float)w / (float)W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const size_t pl = i / ((size_t)W * H);
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        bl_taps(Y, rsy, h, y0, y1, wy0, wy1);
        bl_taps(X, rsx, w, x0, x1, wx0, wx1);
        const float* p = in + pl * h * w;
        out[i] = wy0 * (wx0 * p[y0 * w + x0] + wx1 * p[y0 * w + x1]) + wy1 * (wx0 * p[y1 * w + x0] + wx1 * p[y1 * w + x1]);
    }
}

// gather form (no atomics): four lanes share a low-resolution pixel, each takes every fourth row of the window of
// high-resolution gradients that touch it, and a two-step shuffle adds them up
__global__ void bilinear_up_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int planes, int h, int w, int H, int W) {
    RPNET_PASS_PRIORITY();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, part = t & 3;
    const bool live = i < planes * h * w;
    const int ii = live ? i : 0;
    const int x = ii % w, y = (ii / w) % h, pl = ii / (w * h);
    const float rsy = (float)h / (float)H, rsx = (float)w / (float)W;
    const int Y0 = max(0, (y - 1) * (H / h) - 1), Y1 = min(H - 1, (y + 1) * (H / h) + H / h + 1);
    const int X0 = max(0, (x - 1) * (W / w) - 1), X1 = min(W - 1, (x + 1) * (W / w) + W / w + 1);
    const float* g = dout + (size_t)pl * H * W;
    float acc = 0.f;
    for (int Y = Y0 + part; Y <= Y1; Y += 4) {
        const float wy = bl_weight(Y, y, rsy, h);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = X0; X <= X1; ++X) row += g[(size_t)Y * W + X] * bl_weight(X, x, rsx, w);
        acc += wy * row;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (live && part == 0) din[i] = acc;
}

__global__ void softmax_thresh_pool_kernel(const float* __restrict__ logits, float* __restrict__ mask, int B, int K, int H,
                                           int W, int s, int soft) {
    RPNET_PASS_PRIORITY();
    const int h = H / s, w = W / s;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * h * w) return;
    const int x = i % w, y = (i / w) % h, b = i / (w * h);
    const size_t plane = (size_t)H * W;
    const float* base = logits + (size_t)b * K * plane;
    float acc = 0.f;
    for (int dy = 0; dy < s; ++dy)
        for (int dx = 0; dx < s; ++dx) {
            const size_t o = (size_t)(y * s + dy) * W + (x * s + dx);
            float mx = base[o];
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, base[k * plane + o]);
            float den = 0.f;
            for (int k = 0; k < K; ++k) den += expf(base[k * plane + o] - mx);
            const float p1 = expf(base[plane + o] - mx) / den;
            acc += soft ? p1 : (p1 > 0.5f ? 1.f : 0.f);
        }
    mask[i] = acc / (float)(s * s);
}

// the same for s = 4 (the model's feature scale) and W % 4 == 0: the K x 4 row quads of a thread's window are loaded up
// front as 16-byte vectors (the generic loop above waits for its loads 16 times in a row: 12 us for 8192 threads, 3 us here);
// same arithmetic per pixel, same summation order
template <int KT>
__global__ __launch_bounds__(256) void softmax_thresh_pool4_kernel(const float* __restrict__ logits, float* __restrict__ mask, int B,
                                                                   int K, int H, int W, int soft) {
    RPNET_PASS_PRIORITY();
    const int h = H / 4, w = W / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * h * w) return;
    const int x = i % w, y = (i / w) % h, b = i / (w * h);
    const size_t plane = (size_t)H * W;
    const float* base = logits + (size_t)b * K * plane + (size_t)(y * 4) * W + x * 4;
    f32x4 v[KT][4];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) v[k][dy] = *reinterpret_cast<const f32x4*>(base + k * plane + (size_t)dy * W);
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            float mx = v[0][dy][dx];
#pragma unroll
            for (int k = 1; k < KT; ++k) mx = fmaxf(mx, v[k][dy][dx]);
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) den += expf(v[k][dy][dx] - mx);
            const float p1 = expf(v[1][dy][dx] - mx) / den;
            acc += soft ? p1 : (p1 > 0.5f ? 1.f : 0.f);
        }
    mask[i] = acc / 16.f;
}

// soft-mask gradient pieces (net/rp_net.py:283,308-311 with soft_mask: True)
// dx = g * f(s), ds = sign * <g, x> per pixel, f(s) = s (mode 1) or 1 - s (mode 2)
__global__ __launch_bounds__(256) void rowdot_scale_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ sc, float* __restrict__ dx,
                                                            float* __restrict__ ds, size_t P, int C, int mode, int accumulate_ds) {
    RPNET_PASS_PRIORITY();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int C4 = C / 4;
    for (size_t p = (size_t)blockIdx.x * 4 + wv; p < P; p += (size_t)gridDim.x * 4) {
        const float sv = sc[p];
        const float f = mode == 2 ? 1.f - sv : sv;
        float dot = 0.f;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const f32x4 gv = reinterpret_cast<const f32x4*>(g + p * C)[c4];
            const f32x4 xv = reinterpret_cast<const f32x4*>(x + p * C)[c4];
            dot += gv[0] * xv[0] + gv[1] * xv[1] + gv[2] * xv[2] + gv[3] * xv[3];
            reinterpret_cast<f32x4*>(dx + p * C)[c4] = gv * f;
        }
        dot = wave_sum(dot);
        if (lane == 0) {
            const float v = mode == 2 ? -dot : dot;
            ds[p] = accumulate_ds ? ds[p] + v : v;
        }
    }
}

// d logits of mask = avg_pool(softmax(logits)[:,1], s):  dl_k = p1 * ([k == 1] - p_k) * dmask / s^2
__global__ void softmax_pool_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ dmask,
                                        float* __restrict__ dlogits, int B, int K, int H, int W, int s) {
    RPNET_PASS_PRIORITY();
    const size_t plane = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * plane) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / plane);
    const float* base = logits + (size_t)b * K * plane + (size_t)Y * W + X;
    float mx = base[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, base[k * plane]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(base[k * plane] - mx);
    const float p1 = expf(base[plane] - mx) / den;
    const float gp = dmask[((size_t)b * (H / s) + Y / s) * (W / s) + X / s] / (float)(s * s);
    float* o = dlogits + (size_t)b * K * plane + (size_t)Y * W + X;
    for (int k = 0; k < K; ++k) {
        const float pk = expf(base[k * plane] - mx) / den;
        o[k * plane] = p1 * ((k == 1 ? 1.f : 0.f) - pk) * gp;
    }
}

}  // namespace rpnet

extern "C" int rpnet_rowdot_scale(const float* g, const float* x, const float* scale, float* dx, float* dscale, size_t P,
                                  int C, int mode, int accumulate_dscale, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(g && x && scale && dx && dscale, RPNET_ERR_ARG, "rowdot_scale: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && (mode == 1 || mode == 2), RPNET_ERR_SHAPE, "rowdot_scale: C=%d mode=%d", C, mode);
    size_t nb = (P + 3) / 4; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(rowdot_scale_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, g, x, scale, dx, dscale, P, C, mode, accumulate_dscale);
    return check_launch("rowdot_scale");
}

extern "C" int rpnet_softmax_pool_bwd(const float* logits, const float* dmask, float* dlogits, int B, int K, int H, int W,
                                      int scale, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && dmask && dlogits, RPNET_ERR_ARG, "softmax_pool_bwd: null pointer");
    RPNET_REQUIRE(K >= 2 && H % scale == 0 && W % scale == 0, RPNET_ERR_SHAPE, "softmax_pool_bwd: K=%d H=%d W=%d scale=%d", K, H, W, scale);
    hipLaunchKernelGGL(softmax_pool_bwd_kernel, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, logits, dmask, dlogits, B, K, H, W, scale);
    return check_launch("softmax_pool_bwd");
}

extern "C" int rpnet_mask_adjoint(const float* masks, float* am, float* msum, int B, int nmask, int H, int W, int h, int w,
                                  rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(masks && am && msum, RPNET_ERR_ARG, "mask_adjoint: null pointer");
    RPNET_REQUIRE(H % h == 0 && W % w == 0, RPNET_ERR_SHAPE, "mask_adjoint: %dx%d -> %dx%d", H, W, h, w);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mask_adjoint_kernel, dim3(cdiv((long)B * nmask * h * w, 256)), dim3(256), 0, s, masks, am, B, nmask, H, W, h, w);
    hipLaunchKernelGGL(mask_sum_kernel, dim3(B * nmask), dim3(256), 0, s, masks, msum, B, nmask, H * W);
    return check_launch("mask_adjoint");
}

extern "C" size_t rpnet_masked_pool_workspace_bytes(int B, int nmask, int hw, int C) {
    (void)hw;
    return (size_t)B * rpnet::kPoolSplit * nmask * C * sizeof(float);
}

static bool pow2_le(int v, int hi) { return v >= 1 && v <= hi && (v & (v - 1)) == 0; }

extern "C" int rpnet_masked_pool_fwd(const float* f, const float* am, const float* msum, float* proto, int B, int nmask,
                                     int hw, int C, void* workspace, size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f && am && msum && proto && workspace, RPNET_ERR_ARG, "masked_pool_fwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && pow2_le(C / 4, 256) && nmask >= 1 && nmask <= kMaxMask, RPNET_ERR_SHAPE,
                  "masked_pool_fwd: C=%d nmask=%d", C, nmask);
    RPNET_REQUIRE(workspace_bytes >= rpnet_masked_pool_workspace_bytes(B, nmask, hw, C), RPNET_ERR_WORKSPACE, "masked_pool_fwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(masked_pool_partial, dim3(kPoolSplit, B), dim3(256), 0, s, f, am, (float*)workspace, nmask, hw, C);
    hipLaunchKernelGGL(masked_pool_final, dim3(cdiv(B * nmask * C, 256)), dim3(256), 0, s, (const float*)workspace, msum, proto, B, nmask, C);
    return check_launch("masked_pool_fwd");
}

extern "C" int rpnet_masked_pool_bwd(const float* dproto, const float* am, const float* msum, float* df, int B, int nmask,
                                     int hw, int C, int accumulate, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(dproto && am && msum && df, RPNET_ERR_ARG, "masked_pool_bwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0, RPNET_ERR_SHAPE, "masked_pool_bwd: C=%d", C);
    const size_t total = (size_t)B * hw * (C / 4);
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(masked_pool_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dproto, am, msum, df, B, nmask, hw, C, accumulate);
    return check_launch("masked_pool_bwd");
}

#define RPNET_COS_DISPATCH(L_, ...)                     \
    switch (L_) {                                       \
        case 1: { constexpr int LL = 1; __VA_ARGS__; } break;   \
        case 2: { constexpr int LL = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int LL = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int LL = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int LL = 16; __VA_ARGS__; } break; \
        case 32: { constexpr int LL = 32; __VA_ARGS__; } break; \
        case 64: { constexpr int LL = 64; __VA_ARGS__; } break; \
        default: rpnet::set_error("cosine_match: C=%d must be 4*2^n <= 256", 4 * (L_)); return RPNET_ERR_SHAPE; \
    }

extern "C" int rpnet_cosine_match_fwd(const float* f, const float* proto, float* pred, int B, int K, int hw, int C,
                                      float scaler, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f && proto && pred, RPNET_ERR_ARG, "cosine_match_fwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && K >= 1 && K <= kMaxK, RPNET_ERR_SHAPE, "cosine_match_fwd: C=%d K=%d", C, K);
    RPNET_COS_DISPATCH(C / 4, {
        const int ppb = 256 / LL;
        int nb = cdiv(hw, ppb); if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL((cosine_match_fwd_kernel<LL>), dim3(nb, B), dim3(256), 0, (hipStream_t)stream, f, proto, pred, B, K, hw, scaler);
    });
    return check_launch("cosine_match_fwd");
}

extern "C" size_t rpnet_cosine_match_bwd_workspace_bytes(int B, int K, int hw, int C) {
    (void)hw;
    return (size_t)B * rpnet::cos_blocks(B, hw) * K * C * sizeof(float);
}

extern "C" int rpnet_cosine_match_bwd(const float* f, const float* proto, const float* dpred, float* df, float* dproto,
                                      int B, int K, int hw, int C, float scaler, int accumulate_df, void* workspace,
                                      size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f && proto && dpred && df && dproto && workspace, RPNET_ERR_ARG, "cosine_match_bwd: null pointer");
    RPNET_REQUIRE(C % 4 == 0 && K >= 1 && K <= kMaxK, RPNET_ERR_SHAPE, "cosine_match_bwd: C=%d K=%d", C, K);
    RPNET_REQUIRE(workspace_bytes >= rpnet_cosine_match_bwd_workspace_bytes(B, K, hw, C), RPNET_ERR_WORKSPACE, "cosine_match_bwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = cos_blocks(B, hw);
    RPNET_COS_DISPATCH(C / 4, {
        hipLaunchKernelGGL((cosine_match_bwd_kernel<LL>), dim3(nblk, B), dim3(256), 0, s, f, proto, dpred, df,
                           (float*)workspace, B, K, hw, scaler, accumulate_df);
    });
    hipLaunchKernelGGL(cosine_dproto_final, dim3(B * K), dim3(256), 0, s, (const float*)workspace, dproto, B, nblk, K, C);
    return check_launch("cosine_match_bwd");
}

extern "C" int rpnet_bilinear_up_fwd(const float* in, float* out, int planes, int h, int w, int H, int W, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(in && out, RPNET_ERR_ARG, "bilinear_up_fwd: null pointer");
    const size_t total = (size_t)planes * H * W;
    int nb = (int)((total + 255) / 256); if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bilinear_up_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, in, out, planes, h, w, H, W);
    return check_launch("bilinear_up_fwd");
}

extern "C" int rpnet_bilinear_up_bwd(const float* dout, float* din, int planes, int h, int w, int H, int W, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(dout && din, RPNET_ERR_ARG, "bilinear_up_bwd: null pointer");
    RPNET_REQUIRE(H % h == 0 && W % w == 0, RPNET_ERR_SHAPE, "bilinear_up_bwd: %dx%d -> %dx%d", h, w, H, W);
    hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3(cdiv((long)planes * h * w * 4, 256)), dim3(256), 0, (hipStream_t)stream, dout, din, planes, h, w, H, W);
    return check_launch("bilinear_up_bwd");
}

extern "C" int rpnet_softmax_thresh_pool(const float* logits, float* mask, int B, int K, int H, int W, int scale, int soft,
                                         rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && mask, RPNET_ERR_ARG, "softmax_thresh_pool: null pointer");
    RPNET_REQUIRE(K >= 2 && H % scale == 0 && W % scale == 0, RPNET_ERR_SHAPE, "softmax_thresh_pool: K=%d H=%d W=%d scale=%d", K, H, W, scale);
    const dim3 grid(cdiv((long)B * (H / scale) * (W / scale), 256));
    if (scale == 4 && W % 4 == 0 && (K == 2 || K == 3) && ((size_t)logits & 15) == 0) {
        if (K == 2) hipLaunchKernelGGL(softmax_thresh_pool4_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, logits, mask, B, K, H, W, soft);
        else hipLaunchKernelGGL(softmax_thresh_pool4_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, logits, mask, B, K, H, W, soft);
    } else
        hipLaunchKernelGGL(softmax_thresh_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, mask, B, K, H, W, scale, soft);
    return check_launch("softmax_thresh_pool");
}
