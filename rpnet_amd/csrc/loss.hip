// Losses of the hot path: dice_ce (net/rp_net.py:87-127) and the cross-entropy /
// arg-max pieces of alignLoss (net/rp_net.py:394-440).  Logits are NCHW [B][K][H][W],
// labels int64.  fp64 partial sums, two-stage reductions, no atomics (deterministic).
#include "common.h"

namespace rpnet {

constexpr int kLossBlocks = 64;  // partial blocks per sample
constexpr int kMaxCls = 4;

constexpr int kMaxLogitSets = 16;   // logit tensors per multi-tensor launch (rpnet_dice_ce_multi_*)
struct LogitSet { const float* p[kMaxLogitSets]; };
struct GradSet { float* p[kMaxLogitSets]; };
struct WeightSet { float w[kMaxLogitSets]; };      // multiplicity of each logit tensor in the objective (rpnet_objective_*)

// partial[z][b][blk][2K+2] doubles: inter_k, card_k, ce_sum, count (z = blockIdx.z: which logit tensor of the set)
__global__ __launch_bounds__(256) void dice_ce_partial(const LogitSet set, const int64_t* __restrict__ labels,
                                                        double* __restrict__ partial, int K, int HW, int ignore_index) {
    RPNET_PASS_PRIORITY();
    __shared__ double sm4[4];
    const int b = blockIdx.y;
    const float* lg = set.p[blockIdx.z] + (size_t)b * K * HW;
    const int64_t* lb = labels + (size_t)b * HW;
    double inter[kMaxCls] = {0, 0, 0, 0}, card[kMaxCls] = {0, 0, 0, 0}, ce = 0, cnt = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float v[kMaxCls], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) { v[k] = k < K ? lg[(size_t)k * HW + i] : -INFINITY; mx = fmaxf(mx, v[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) { v[k] = k < K ? expf(v[k] - mx) : 0.f; den += v[k]; }
        const int lab = (int)lb[i];
        const bool valid = lab != ignore_index;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) {
            const float p = v[k] / den;
            const float oh = (lab == k) ? 1.f : 0.f;
            inter[k] += p * oh;
            card[k] += p + oh;
        }
        if (valid) {
            cnt += 1.0;
            if (lab >= 0 && lab < K) ce += (double)((mx + logf(den)) - lg[(size_t)lab * HW + i]);  // -log_softmax[label]
        }
    }
    double* o = partial + (((size_t)blockIdx.z * gridDim.y + b) * gridDim.x + blockIdx.x) * (2 * K + 2);
    for (int k = 0; k < K; ++k) {
        const double a = block_sum256(inter[k], sm4), c = block_sum256(card[k], sm4);
        if (threadIdx.x == 0) { o[k] = a; o[K + k] = c; }
    }
    const double a = block_sum256(ce, sm4), c = block_sum256(cnt, sm4);
    if (threadIdx.x == 0) { o[2 * K] = a; o[2 * K + 1] = c; }
}

// stats: per logit tensor z: [B][2K+2] per sample, then [2K+2] totals.  loss[z] = dice + ce; total[0] (optional) = their sum.
// One 1024-thread block: each of its 16 waves sums the per-block partials of one (sample, slot) pair at a
// time; thread 0 then does the O(B*K) scalar combine.  The n tensors of a set one after the other.
__global__ __launch_bounds__(1024) void dice_ce_final(const double* __restrict__ partial, float* __restrict__ stats,
                                                      float* __restrict__ loss, int B, int K, int nblk, int with_dice,
                                                      int per_sample, const float* __restrict__ sample_weight, int n,
                                                      float* __restrict__ total, const WeightSet wts,
                                                      const float* __restrict__ extra, float extra_scale) {
    RPNET_PASS_PRIORITY();
    extern __shared__ double sums[];  // [B][S]
    const int S = 2 * K + 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nwv = blockDim.x >> 6;
    double all = 0;
    for (int z = 0; z < n; ++z) {
        const double* pz = partial + (size_t)z * B * nblk * S;
        float* sz = stats + (size_t)z * (B + 1) * S;
        for (int idx = wv; idx < B * S; idx += nwv) {
            const int b = idx / S, j = idx - b * S;
            double v = 0;
            for (int blk = lane; blk < nblk; blk += 64) v += pz[((size_t)b * nblk + blk) * S + j];
            v = wave_sum(v);
            if (lane == 0) sums[idx] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot[2 * kMaxCls + 2];
            for (int j = 0; j < S; ++j) tot[j] = 0;
            double ce_ps = 0;
            for (int b = 0; b < B; ++b) {
                for (int j = 0; j < S; ++j) { sz[b * S + j] = (float)sums[b * S + j]; tot[j] += sums[b * S + j]; }
                const double wgt = sample_weight ? (double)sample_weight[b] : 1.0;
                if (wgt != 0.0) ce_ps += wgt * sums[b * S + 2 * K] / sums[b * S + 2 * K + 1];
            }
            for (int j = 0; j < S; ++j) sz[B * S + j] = (float)tot[j];
            double l = per_sample ? ce_ps / (double)B : tot[2 * K] / tot[2 * K + 1];
            if (with_dice) {
                double d = 0;
                for (int k = 0; k < K; ++k) d += 2.0 * tot[k] / (tot[K + k] + 1e-7);
                l += 1.0 - d / (double)K;
            }
            loss[z] = (float)l;
            all += (double)wts.w[z] * (double)(float)l;       // the sum of the fp32 losses, as a chain of fp32 tensors' values would be added in fp64
        }
        __syncthreads();                   // `sums` is rewritten by the next tensor
    }
    if (threadIdx.x == 0 && total) {
        float tt = (float)all;
        if (extra) tt = tt + extra_scale * extra[0];       // fp32 multiply, fp32 add: what `dice_sum + scaler * align_loss` does on tensors
        total[0] = tt;
    }
}

__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(const LogitSet set, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ stats_all, const float* __restrict__ gscale,
                                                           const GradSet gset, int B, int K, int HW, int with_dice,
                                                           int ignore_index, int per_sample,
                                                           const float* __restrict__ sample_weight, int accumulate,
                                                           const WeightSet wts, float* __restrict__ dextra, float extra_scale) {
    RPNET_PASS_PRIORITY();
    const int b = blockIdx.y;
    const int S = 2 * K + 2;
    const float* logits = set.p[blockIdx.z];
    float* dlogits = gset.p[blockIdx.z];
    const float* stats = stats_all + (size_t)blockIdx.z * (B + 1) * S;
    const float g0 = gscale ? gscale[0] : 1.f;
    const float gs = g0 * wts.w[blockIdx.z];
    if (dextra && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) dextra[0] = g0 * extra_scale;
    float ce_coef;
    if (per_sample) {
        const float wgt = sample_weight ? sample_weight[b] : 1.f;
        ce_coef = wgt != 0.f ? wgt / ((float)B * stats[b * S + 2 * K + 1]) : 0.f;
    } else {
        ce_coef = 1.f / stats[B * S + 2 * K + 1];
    }
    float a_on[kMaxCls], a_off[kMaxCls];  // d dice / d p_k for label == k / != k
#pragma unroll
    for (int k = 0; k < kMaxCls; ++k) {
        a_on[k] = 0.f; a_off[k] = 0.f;
        if (with_dice && k < K) {
            const float I = stats[B * S + k], Cc = stats[B * S + K + k] + 1e-7f;
            a_off[k] = (2.f / (float)K) * I / (Cc * Cc);
            a_on[k] = a_off[k] - (2.f / (float)K) / Cc;
        }
    }
    const float* lg = logits + (size_t)b * K * HW;
    const int64_t* lb = labels + (size_t)b * HW;
    float* dl = dlogits + (size_t)b * K * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float p[kMaxCls], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) { p[k] = k < K ? lg[(size_t)k * HW + i] : -INFINITY; mx = fmaxf(mx, p[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) { p[k] = k < K ? expf(p[k] - mx) : 0.f; den += p[k]; }
        const int lab = (int)lb[i];
        const bool valid = lab != ignore_index;
        float a[kMaxCls], ap = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) { p[k] /= den; a[k] = (lab == k) ? a_on[k] : a_off[k]; ap += a[k] * p[k]; }
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k)
            if (k < K) {
                float g = p[k] * (a[k] - ap);
                if (valid) g += ce_coef * (p[k] - (lab == k ? 1.f : 0.f));
                g *= gs;
                float* dst = dl + (size_t)k * HW + i;
                *dst = accumulate ? (*dst + g) : g;
            }
    }
}

// alignLoss pieces (net/rp_net.py:412-417,433-436)
__global__ __launch_bounds__(256) void argmax_masks_kernel(const float* __restrict__ pred, float* __restrict__ masks,
                                                            float* __restrict__ counts, float* __restrict__ keep, int K, int hw) {
    RPNET_PASS_PRIORITY();
    __shared__ double sm4[4];
    const int b = blockIdx.x;
    const float* p = pred + (size_t)b * K * hw;
    double cnt[kMaxCls] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < hw; q += 256) {
        int best = 0; float bv = p[q];
        for (int k = 1; k < K; ++k) { const float v = p[(size_t)k * hw + q]; if (v > bv) { bv = v; best = k; } }
        for (int k = 0; k < K; ++k) masks[((size_t)b * K + k) * hw + q] = (k == best) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < kMaxCls; ++k) cnt[k] += (k == best) ? 1.0 : 0.0;
    }
    for (int k = 0; k < K; ++k) {
        const double c = block_sum256(cnt[k], sm4);
        if (threadIdx.x == 0) {
            counts[b * K + k] = (float)c;
            if (keep) keep[(size_t)k * gridDim.x + b] = c > 0.0 ? 1.f : 0.f;       // [K][B]: a class's row is contiguous
        }
    }
}

__global__ void align_labels_kernel(const float* __restrict__ fore, const float* __restrict__ back, int64_t* __restrict__ lab, size_t n) {
    RPNET_PASS_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int64_t v = 255;
        if (fore[i] == 1.f) v = 1;
        if (back[i] == 1.f) v = 0;
        lab[i] = v;
    }
}

static WeightSet ones_set() {
    WeightSet w{};
    for (int i = 0; i < kMaxLogitSets; ++i) w.w[i] = 1.f;
    return w;
}

}  // namespace rpnet

extern "C" size_t rpnet_loss_workspace_bytes(int B, int K, int H, int W) {
    (void)H; (void)W;
    return (size_t)B * rpnet::kLossBlocks * (2 * K + 2) * sizeof(double);
}

extern "C" int rpnet_dice_ce_fwd(const float* logits, const int64_t* labels, float* loss, float* stats, int B, int K, int H,
                                 int W, int with_dice, int ignore_index, int per_sample, const float* sample_weight,
                                 void* workspace, size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && labels && loss && stats && workspace, RPNET_ERR_ARG, "dice_ce_fwd: null pointer");
    RPNET_REQUIRE(K >= 2 && K <= kMaxCls, RPNET_ERR_SHAPE, "dice_ce_fwd: K=%d", K);
    RPNET_REQUIRE(workspace_bytes >= rpnet_loss_workspace_bytes(B, K, H, W), RPNET_ERR_WORKSPACE, "dice_ce_fwd: workspace");
    hipStream_t s = (hipStream_t)stream;
    LogitSet set{};
    set.p[0] = logits;
    hipLaunchKernelGGL(dice_ce_partial, dim3(kLossBlocks, B), dim3(256), 0, s, set, labels, (double*)workspace, K, H * W, ignore_index);
    hipLaunchKernelGGL(dice_ce_final, dim3(1), dim3(1024), (size_t)B * (2 * K + 2) * sizeof(double), s, (const double*)workspace, stats, loss, B, K, kLossBlocks, with_dice,
                       per_sample, sample_weight, 1, (float*)nullptr, ones_set(), (const float*)nullptr, 0.f);
    return check_launch("dice_ce_fwd");
}

extern "C" int rpnet_dice_ce_multi_fwd(const float* const* logits, int n, const int64_t* labels, float* loss, float* stats, int B,
                                       int K, int H, int W, void* workspace, size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && labels && loss && stats && workspace, RPNET_ERR_ARG, "dice_ce_multi_fwd: null pointer");
    RPNET_REQUIRE(n >= 1 && n <= kMaxLogitSets, RPNET_ERR_ARG, "dice_ce_multi_fwd: %d logit tensors (1..%d per call)", n, kMaxLogitSets);
    RPNET_REQUIRE(K >= 2 && K <= kMaxCls, RPNET_ERR_SHAPE, "dice_ce_multi_fwd: K=%d", K);
    RPNET_REQUIRE(workspace_bytes >= (size_t)n * rpnet_loss_workspace_bytes(B, K, H, W), RPNET_ERR_WORKSPACE, "dice_ce_multi_fwd: workspace");
    LogitSet set{};
    for (int i = 0; i < n; ++i) {
        RPNET_REQUIRE(logits[i], RPNET_ERR_ARG, "dice_ce_multi_fwd: null logit tensor %d", i);
        set.p[i] = logits[i];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_ce_partial, dim3(kLossBlocks, B, n), dim3(256), 0, s, set, labels, (double*)workspace, K, H * W, -1);
    hipLaunchKernelGGL(dice_ce_final, dim3(1), dim3(1024), (size_t)B * (2 * K + 2) * sizeof(double), s, (const double*)workspace, stats, loss, B, K, kLossBlocks, 1,
                       0, (const float*)nullptr, n, loss + n, ones_set(), (const float*)nullptr, 0.f);
    return check_launch("dice_ce_multi_fwd");
}

extern "C" int rpnet_objective_fwd(const float* const* logits, const float* weights, int n, const int64_t* labels, const float* extra,
                                   float extra_scale, float* loss, float* stats, int B, int K, int H, int W, void* workspace,
                                   size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && weights && labels && loss && stats && workspace, RPNET_ERR_ARG, "objective_fwd: null pointer");
    RPNET_REQUIRE(n >= 1 && n <= kMaxLogitSets, RPNET_ERR_ARG, "objective_fwd: %d logit tensors (1..%d per call)", n, kMaxLogitSets);
    RPNET_REQUIRE(K >= 2 && K <= kMaxCls, RPNET_ERR_SHAPE, "objective_fwd: K=%d", K);
    RPNET_REQUIRE(workspace_bytes >= (size_t)n * rpnet_loss_workspace_bytes(B, K, H, W), RPNET_ERR_WORKSPACE, "objective_fwd: workspace");
    LogitSet set{};
    WeightSet wts{};
    for (int i = 0; i < n; ++i) {
        RPNET_REQUIRE(logits[i], RPNET_ERR_ARG, "objective_fwd: null logit tensor %d", i);
        set.p[i] = logits[i];
        wts.w[i] = weights[i];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_ce_partial, dim3(kLossBlocks, B, n), dim3(256), 0, s, set, labels, (double*)workspace, K, H * W, -1);
    hipLaunchKernelGGL(dice_ce_final, dim3(1), dim3(1024), (size_t)B * (2 * K + 2) * sizeof(double), s, (const double*)workspace, stats, loss, B, K, kLossBlocks, 1,
                       0, (const float*)nullptr, n, loss + n, wts, extra, extra_scale);
    return check_launch("objective_fwd");
}

extern "C" int rpnet_dice_ce_bwd(const float* logits, const int64_t* labels, const float* stats, const float* gscale,
                                 float* dlogits, int B, int K, int H, int W, int with_dice, int ignore_index, int per_sample,
                                 const float* sample_weight, int accumulate, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && labels && stats && dlogits, RPNET_ERR_ARG, "dice_ce_bwd: null pointer");
    RPNET_REQUIRE(K >= 2 && K <= kMaxCls, RPNET_ERR_SHAPE, "dice_ce_bwd: K=%d", K);
    int nb = cdiv(H * W, 256); if (nb > 256) nb = 256;
    LogitSet set{};
    GradSet gset{};
    set.p[0] = logits;
    gset.p[0] = dlogits;
    hipLaunchKernelGGL(dice_ce_bwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, set, labels, stats, gscale, gset,
                       B, K, H * W, with_dice, ignore_index, per_sample, sample_weight, accumulate, ones_set(), (float*)nullptr, 0.f);
    return check_launch("dice_ce_bwd");
}

extern "C" int rpnet_dice_ce_multi_bwd(const float* const* logits, float* const* dlogits, int n, const int64_t* labels,
                                       const float* stats, const float* gscale, int B, int K, int H, int W, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && dlogits && labels && stats, RPNET_ERR_ARG, "dice_ce_multi_bwd: null pointer");
    RPNET_REQUIRE(n >= 1 && n <= kMaxLogitSets, RPNET_ERR_ARG, "dice_ce_multi_bwd: %d logit tensors (1..%d per call)", n, kMaxLogitSets);
    RPNET_REQUIRE(K >= 2 && K <= kMaxCls, RPNET_ERR_SHAPE, "dice_ce_multi_bwd: K=%d", K);
    LogitSet set{};
    GradSet gset{};
    for (int i = 0; i < n; ++i) {
        RPNET_REQUIRE(logits[i] && dlogits[i], RPNET_ERR_ARG, "dice_ce_multi_bwd: null tensor %d", i);
        set.p[i] = logits[i];
        gset.p[i] = dlogits[i];
    }
    int nb = cdiv(H * W, 256); if (nb > 256) nb = 256;
    hipLaunchKernelGGL(dice_ce_bwd_kernel, dim3(nb, B, n), dim3(256), 0, (hipStream_t)stream, set, labels, stats, gscale, gset,
                       B, K, H * W, 1, -1, 0, (const float*)nullptr, 0, ones_set(), (float*)nullptr, 0.f);
    return check_launch("dice_ce_multi_bwd");
}

extern "C" int rpnet_objective_bwd(const float* const* logits, float* const* dlogits, const float* weights, int n, const int64_t* labels,
                                   const float* stats, const float* gscale, float* dextra, float extra_scale, int B, int K, int H, int W,
                                   rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(logits && dlogits && weights && labels && stats, RPNET_ERR_ARG, "objective_bwd: null pointer");
    RPNET_REQUIRE(n >= 1 && n <= kMaxLogitSets, RPNET_ERR_ARG, "objective_bwd: %d logit tensors (1..%d per call)", n, kMaxLogitSets);
    RPNET_REQUIRE(K >= 2 && K <= kMaxCls, RPNET_ERR_SHAPE, "objective_bwd: K=%d", K);
    LogitSet set{};
    GradSet gset{};
    WeightSet wts{};
    for (int i = 0; i < n; ++i) {
        RPNET_REQUIRE(logits[i] && dlogits[i], RPNET_ERR_ARG, "objective_bwd: null tensor %d", i);
        set.p[i] = logits[i];
        gset.p[i] = dlogits[i];
        wts.w[i] = weights[i];
    }
    int nb = cdiv(H * W, 256); if (nb > 256) nb = 256;
    hipLaunchKernelGGL(dice_ce_bwd_kernel, dim3(nb, B, n), dim3(256), 0, (hipStream_t)stream, set, labels, stats, gscale, gset,
                       B, K, H * W, 1, -1, 0, (const float*)nullptr, 0, wts, dextra, extra_scale);
    return check_launch("objective_bwd");
}

extern "C" int rpnet_argmax_masks(const float* pred, float* masks, float* counts, float* keep, int B, int K, int hw, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(pred && masks && counts, RPNET_ERR_ARG, "argmax_masks: null pointer");
    RPNET_REQUIRE(K >= 1 && K <= kMaxCls, RPNET_ERR_SHAPE, "argmax_masks: K=%d", K);
    hipLaunchKernelGGL(argmax_masks_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pred, masks, counts, keep, K, hw);
    return check_launch("argmax_masks");
}

extern "C" int rpnet_align_labels(const float* fore, const float* back, int64_t* labels, size_t n, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(fore && back && labels, RPNET_ERR_ARG, "align_labels: null pointer");
    int nb = (int)((n + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(align_labels_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, fore, back, labels, n);
    return check_launch("align_labels");
}
