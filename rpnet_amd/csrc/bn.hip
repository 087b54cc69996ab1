// Train-mode BatchNorm2d + ReLU around the grid-wide reduction, NHWC fp32 (HBM-bound
// kernels: float4 accesses along C, fp64 accumulation for the statistics).
// Replaces nn.BatchNorm2d + nn.ReLU(inplace) of net/modules.py:48-49,51-52,68-69 and
// net/rp_net.py:52-53,57-58,67-68 and their autograd.
#include <stdlib.h>

#include "common.h"
#include "lds_dma.h"
#include "split_bf16.h"

namespace rpnet {

struct BnGeom {
    int C4;        // float4 columns
    int rows_it;   // rows handled per block iteration
    int nblk;      // blocks per group
    int rows_blk;  // rows per block
};

// The pre-BatchNorm tensor as the passes read it (fp32).  `off` = element offset of 4 consecutive channels starting at channel c.
// (Round 4 could also hold it as 2-byte codes decoded here — measured neutral in time and worse in error, removed in round 5;
// the y_dec arguments of rpnet_bn_relu / rpnet_bn_bwd are reserved and must be NULL.)
struct YSrc {
    const float* y;
};
__device__ __forceinline__ f32x4 load_y4(const YSrc s, const size_t off, const int) {
    return *reinterpret_cast<const f32x4*>(s.y + off);
}

// most blocks (= partial-sum rows) per group: the reduction passes keep two 16-byte loads per thread in flight, so the
// 64- and 128-channel levels (0.13 - 0.5 M rows per group) need ~8 blocks per CU to cover the HBM latency (bn_bwd_partial:
// 28.7 -> 23.8 us per launch against 2 per CU); bounded by the workspace, 131072 partial sums per group
static int bn_max_blocks(int C) { const int m = 131072 / C; return m < 256 ? 256 : (m > 1024 ? 1024 : m); }

static BnGeom bn_geom(long R, int C) {
    BnGeom g;
    g.C4 = C / 4;
    g.rows_it = 256 / g.C4;
    long nb = (R + (long)g.rows_it * 4 - 1) / ((long)g.rows_it * 4);
    const int cap = bn_max_blocks(C);
    g.nblk = (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
    g.rows_blk = (int)((R + g.nblk - 1) / g.nblk);
    return g;
}

// Cross-row reduction of a block's per-thread sums (thread t = row tr, float4 column tc; the owners are the threads of
// row 0) in the order rr = 1, 2, ... .  WIN = 256: every thread's sums in LDS at once (16 - 20 KB per block).  WIN = 64:
// through a ONE-wave window, wave after wave (4 - 5 KB): a block then fits into the 8 KB of LDS that a resident block of
// the LDS-DMA convolution kernels leaves free on its CU (155 648 / 147 456 of 163 840 bytes), so the reduction pass of the
// main stream co-runs with the weight-gradient launches of the side stream instead of waiting for whole CUs.  Same order
// of additions either way: bit-identical partial sums.
template <int WIN, bool MAXV>
__device__ __forceinline__ void rows_reduce(double (&a)[4], double (&b)[4], float (&mx)[4], double* red, float* redm, const int t,
                                            const int tc, const int tr, const int C4, const int rows_it) {
    if constexpr (WIN == 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[t * 8 + k] = a[k]; red[t * 8 + 4 + k] = b[k];
            if constexpr (MAXV) redm[t * 4 + k] = mx[k];
        }
        __syncthreads();
        if (tr == 0)
            for (int rr = 1; rr < rows_it; ++rr)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[k] += red[(rr * C4 + tc) * 8 + k];
                    b[k] += red[(rr * C4 + tc) * 8 + 4 + k];
                    if constexpr (MAXV) mx[k] = fmaxf(mx[k], redm[(rr * C4 + tc) * 4 + k]);
                }
    } else {
        const int wave = t >> 6, l = t & 63;
        for (int w = 0; w * 64 < rows_it * C4; ++w) {       // block-uniform trip count
            if (w) __syncthreads();                         // the readers of the previous window are done
            if (wave == w) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    red[l * 8 + k] = a[k]; red[l * 8 + 4 + k] = b[k];
                    if constexpr (MAXV) redm[l * 4 + k] = mx[k];
                }
            }
            __syncthreads();
            if (tr == 0) {
                const int num = 64 * w - tc;                // first row of this window that belongs to column tc
                int rr = num <= 0 ? 1 : (num + C4 - 1) / C4;
                if (rr < 1) rr = 1;
                for (; rr < rows_it && rr * C4 + tc < 64 * w + 64; ++rr) {
                    const int i = rr * C4 + tc - 64 * w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a[k] += red[i * 8 + k];
                        b[k] += red[i * 8 + 4 + k];
                        if constexpr (MAXV) mx[k] = fmaxf(mx[k], redm[i * 4 + k]);
                    }
                }
            }
        }
    }
}

// 64: the one-wave window (default); RPNET_BN_LDS=big: all sums at once (the A/B switch of the co-residency argument above)
static int bn_lds_window() {
    static const int w = [] { const char* e = getenv("RPNET_BN_LDS"); return e && e[0] == 'b' ? 256 : 64; }();
    return w;
}

// partial[(g*nblk + blk)][C][2] doubles: sum, sumsq
template <int WIN>
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ y, double* __restrict__ partial,
                                                         long R, int C, BnGeom gm) {
    RPNET_PASS_PRIORITY();
    __shared__ double red[WIN * 8];
    const int t = threadIdx.x;
    const int tc = t % gm.C4, tr = t / gm.C4;
    const int g = blockIdx.y, blk = blockIdx.x;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (tr < gm.rows_it) {
        const long r0 = (long)blk * gm.rows_blk;
        const long r1 = min(r0 + gm.rows_blk, R);
        const float* base = y + ((size_t)g * R) * C + tc * 4;
        for (long r = r0 + tr; r < r1; r += gm.rows_it) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)r * C);
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += v[k]; q[k] += (double)v[k] * v[k]; }
        }
    }
    float nomax[4] = {0.f, 0.f, 0.f, 0.f};
    rows_reduce<WIN, false>(s, q, nomax, red, nullptr, t, tc, tr, gm.C4, gm.rows_it);
    if (tr == 0) {
        double* o = partial + ((size_t)(g * gm.nblk + blk) * C + tc * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k * 2] = s[k]; o[k * 2 + 1] = q[k]; }
    }
}

// one 256-thread block per channel: threads stride over the per-block partials (up to 2048 rows per group: a single wave
// walked them in 32 dependent rounds, 7 us for a few KB), wave shuffles + one LDS round for the sum
__global__ __launch_bounds__(256) void bn_stats_finalize(const double* __restrict__ partial, int nblk, long R, int C, int G,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  float* running_mean, float* running_var, long long* num_batches_tracked,
                                  float momentum, float eps, float* scale, float* shift, float* mean, float* invstd) {
    RPNET_PASS_PRIORITY();
    __shared__ double red4[8];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (num_batches_tracked && c == 0 && lane == 0) *num_batches_tracked += G;  // one "forward call" per group
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int g = 0; g < G; ++g) {  // sequential: one running-stat update per group, in call order
        double s = 0, q = 0;
        for (int b = lane; b < nblk; b += 256) {
            const double* p = partial + ((size_t)(g * nblk + b) * C + c) * 2;
            s += p[0]; q += p[1];
        }
        s = block_sum256(s, red4); q = block_sum256(q, red4 + 4);
        const double m = s / (double)R;
        double var = q / (double)R - m * m;
        if (var < 0) var = 0;
        const float istd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * istd;
        if (lane == 0) {
            scale[g * C + c] = sc;
            shift[g * C + c] = beta[c] - (float)m * sc;
            mean[g * C + c] = (float)m;
            invstd[g * C + c] = istd;
        }
        const float unb = (float)(R > 1 ? var * (double)R / (double)(R - 1) : var);
        rm = (1.f - momentum) * rm + momentum * (float)m;
        rv = (1.f - momentum) * rv + momentum * unb;
    }
    if (lane == 0) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

__global__ __launch_bounds__(256) void bn_relu_kernel(const YSrc ysrc, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ z,
                                                       size_t total4, int C4, size_t group4, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float sqrt_n,
                                                       float* __restrict__ s_out, const FastDiv fc, const FastDiv fg) {
    RPNET_PASS_PRIORITY();
    if (s_out && blockIdx.x == 0) {      // the fp16 tensor scale of this output (see bn_relu_split_kernel), no planes
        __shared__ float red4s[4];
        float m = 0.f;
        for (int c = threadIdx.x; c < C4 * 4; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_n + fabsf(beta[c]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) red4s[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) *s_out = pow2_scale(fmaxf(fmaxf(red4s[0], red4s[1]), fmaxf(red4s[2], red4s[3])));
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)fc.mod((unsigned)i);
        const int g = (int)fg.div((unsigned)i);
        const f32x4 v = load_y4(ysrc, i * 4, c4 * 4);
        const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[g * C4 + c4];
        const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[g * C4 + c4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
        reinterpret_cast<f32x4*>(z)[i] = o;
    }
}

// same, 8 channels per thread, plus the split-bf16 planes of z (the operand format of the next convolution)
// NP = 2 (fp16 planes): the planes hold z / s with the power-of-two tensor scale s derived from a RIGOROUS bound of the
// output, |z| = |gamma xhat + beta| <= |gamma| sqrt(n) + |beta| (|xhat_i| <= sqrt(n) for any data: (x_i - mean)^2 <= n var):
// s = pow2ceil(max_c bound) 2^-15, so |z / s| <= 2^15 < 65504 whatever the input; typical values (|xhat| ~ 1) land at
// 2^15 / sqrt(n) ~ 8 .. 300.  Every block recomputes s from the C channel parameters; block 0 publishes it in *s_out.
__device__ __forceinline__ float block_max256(float m, float* red4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
}

// the scale alone (see bn_relu_split_kernel): for outputs whose consumers split the fp32 tensor themselves
__global__ __launch_bounds__(256) void bn_act_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const int C, const float sqrt_n, float* __restrict__ s_out) {
    RPNET_PASS_PRIORITY();
    __shared__ float red4[4];
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_n + fabsf(beta[c]));
    m = block_max256(m, red4);
    if (threadIdx.x == 0) *s_out = pow2_scale(m);
}

template <int NP>
__global__ __launch_bounds__(256) void bn_relu_split_kernel(const YSrc ysrc, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ z,
                                                             unsigned short* __restrict__ zs, size_t total8, int C8,
                                                             size_t group8, size_t plane_elems, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float sqrt_n,
                                                             float* __restrict__ s_out, const FastDiv fc, const FastDiv fg) {
    RPNET_PASS_PRIORITY();
    float inv_s = 1.f;
    if (NP <= 2) {
        __shared__ float red4[4];
        float m = 0.f;
        for (int c = threadIdx.x; c < C8 * 8; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_n + fabsf(beta[c]));
        const float sc = pow2_scale(block_max256(m, red4));
        if (blockIdx.x == 0 && threadIdx.x == 0) *s_out = sc;
        inv_s = 1.f / sc;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)fc.mod((unsigned)i);
        const int g = (int)fg.div((unsigned)i);
        const float* sc = scale + (size_t)(g * C8 + c8) * 8;
        const float* sh = shift + (size_t)(g * C8 + c8) * 8;
        float v[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 a = load_y4(ysrc, i * 8 + hh * 4, c8 * 8 + hh * 4);
            const f32x4 s4 = reinterpret_cast<const f32x4*>(sc)[hh], h4 = reinterpret_cast<const f32x4*>(sh)[hh];
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[k] = fmaxf(a[k] * s4[k] + h4[k], 0.f); v[hh * 4 + k] = o[k]; }
            if (z) reinterpret_cast<f32x4*>(z)[i * 2 + hh] = o;      // z == NULL: only the planes are wanted
        }
        if (NP <= 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= inv_s;
        }
        u32x4 pl[NP];
        split8<NP>(v, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(zs + p * plane_elems + i * 8) = pl[p];
    }
}

// partial[(g*nblk + blk)][C][2] doubles: s1 = sum dz*m, s2 = sum dz*m*xhat; pmax[(g*nblk + blk)][C] = max |dz*m| (optional)
template <int WIN>
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ dz, const YSrc ysrc,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       double* __restrict__ partial, float* __restrict__ pmax, long R, int C,
                                                       BnGeom gm) {
    RPNET_PASS_PRIORITY();
    __shared__ double red[WIN * 8];
    __shared__ float redm[WIN * 4];
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x;
    const int tc = t % gm.C4, tr = t / gm.C4;
    const int g = blockIdx.y, blk = blockIdx.x;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (tr < gm.rows_it) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + g * C + tc * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + g * C + tc * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + tc * 4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + g * C + tc * 4);
        const long r0 = (long)blk * gm.rows_blk;
        const long r1 = min(r0 + gm.rows_blk, R);
        const size_t base = ((size_t)g * R) * C + tc * 4;
        auto take = [&](const f32x4 v, const f32x4 d) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dm = (v[k] * sc[k] + sh[k] > 0.f) ? d[k] : 0.f;
                s1[k] += dm;
                s2[k] += (double)dm * ((v[k] - mu[k]) * is[k]);
                mx[k] = fmaxf(mx[k], fabsf(dm));
            }
        };
        // Rows are dealt to the blocks CYCLICALLY in groups of rows_it (block b: groups b, b + nblk, ...): at any moment the
        // blocks of the launch read one contiguous window of both tensors.  (A contiguous range per block put all blocks
        // a fixed 2^k bytes apart — the same HBM channels at the same time: 3.7 - 4.4 TB/s against the 5.4 of the
        // element-wise passes.)  Two groups per pass: four 16-byte loads in flight per thread.
        (void)r0; (void)r1;
        const long stride = (long)gm.nblk * gm.rows_it;
        long r = (long)blk * gm.rows_it + tr;
        for (; r + stride < R; r += 2 * stride) {
            const f32x4 v0 = load_y4(ysrc, base + (size_t)r * C, tc * 4);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dz + base + (size_t)r * C);
            const f32x4 v1 = load_y4(ysrc, base + (size_t)(r + stride) * C, tc * 4);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dz + base + (size_t)(r + stride) * C);
            take(v0, d0);
            take(v1, d1);
        }
        if (r < R)
            take(load_y4(ysrc, base + (size_t)r * C, tc * 4), *reinterpret_cast<const f32x4*>(dz + base + (size_t)r * C));
    }
    rows_reduce<WIN, true>(s1, s2, mx, red, redm, t, tc, tr, gm.C4, gm.rows_it);
    if (tr == 0) {
        double* o = partial + ((size_t)(g * gm.nblk + blk) * C + tc * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k * 2] = s1[k]; o[k * 2 + 1] = s2[k]; }
        if (pmax) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pmax[(size_t)(g * gm.nblk + blk) * C + tc * 4 + k] = mx[k];
        }
    }
}

// coef[g][C][2] floats = (s1/R, s2/R); dgamma/dbeta summed over groups
// bound[c] (optional, with pmax): a rigorous bound of |dy| of channel c over all groups,
//   |dy| = |scale| |dz m - s1/R - xhat s2/R| <= |scale| (max |dz m| + |s1|/R + sqrt(R) |s2|/R)
__global__ __launch_bounds__(256) void bn_bwd_finalize(const double* __restrict__ partial, int nblk, long R, int C, int G,
                                float* coef, float* dgamma, float* dbeta, int accumulate, const float* __restrict__ pmax,
                                const float* __restrict__ scale, float* __restrict__ bound) {
    RPNET_PASS_PRIORITY();
    __shared__ double red4[8];
    __shared__ float redm[4];
    const int c = blockIdx.x, lane = threadIdx.x;
    double tg = 0, tb = 0;
    float bnd = 0.f;
    for (int g = 0; g < G; ++g) {
        double s1 = 0, s2 = 0;
        float mx = 0.f;
        for (int b = lane; b < nblk; b += 256) {
            const double* p = partial + ((size_t)(g * nblk + b) * C + c) * 2;
            s1 += p[0]; s2 += p[1];
            if (pmax) mx = fmaxf(mx, pmax[(size_t)(g * nblk + b) * C + c]);
        }
        s1 = block_sum256(s1, red4); s2 = block_sum256(s2, red4 + 4);
        if (pmax) {
            mx = block_max256(mx, redm);
            bnd = fmaxf(bnd, fabsf(scale[g * C + c]) * (mx + (float)(fabs(s1) / (double)R) + (float)(fabs(s2) / sqrt((double)R))));
        }
        if (lane == 0) {
            coef[(g * C + c) * 2] = (float)(s1 / (double)R);
            coef[(g * C + c) * 2 + 1] = (float)(s2 / (double)R);
        }
        tb += s1; tg += s2;
    }
    if (pmax && lane == 0) bound[c] = bnd * 1.0001f;
    if (lane == 0) {
        if (accumulate) { dgamma[c] += (float)tg; dbeta[c] += (float)tb; }
        else { dgamma[c] = (float)tg; dbeta[c] = (float)tb; }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dz, const YSrc ysrc,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ coef, float* __restrict__ dy,
                                                     size_t total4, int C, size_t group4, const FastDiv fc, const FastDiv fg) {
    RPNET_PASS_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)fc.mod((unsigned)i);
        const int g = (int)fg.div((unsigned)i);
        const int o = g * C + c4 * 4;
        const f32x4 v = load_y4(ysrc, i * 4, c4 * 4);
        const f32x4 d = reinterpret_cast<const f32x4*>(dz)[i];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + o);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + o);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + o);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + o);
        const f32x4 cf0 = *reinterpret_cast<const f32x4*>(coef + (size_t)o * 2), cf1 = *reinterpret_cast<const f32x4*>(coef + (size_t)o * 2 + 4);
        const float c1[4] = {cf0[0], cf0[2], cf1[0], cf1[2]}, c2[4] = {cf0[1], cf0[3], cf1[1], cf1[3]};
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dm = (v[k] * sc[k] + sh[k] > 0.f) ? d[k] : 0.f;
            const float xh = (v[k] - mu[k]) * is[k];
            r[k] = sc[k] * (dm - c1[k] - xh * c2[k]);
        }
        reinterpret_cast<f32x4*>(dy)[i] = r;
    }
}

// same, 8 channels per thread: dy as split-bf16 planes (and, when `dy` is not NULL, also in fp32)
template <int NP>
// amdgpu_waves_per_eu(8, 8): at most 64 registers (62 without spills instead of 68), so that TWO waves of this pass fit into
// the 128 registers per lane a resident LDS-DMA GEMM wave leaves free on its SIMD (one at 68): the pass runs beside the
// weight-gradient launches of the side stream — 18.09 -> 18.05 ms per batch-8 step, configs[4] 33.34 -> 33.11 ms (the
// three-plane form, bf16x3, pays for the cap with two spilled dwords)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void bn_bwd_apply_split(const float* __restrict__ dz, const YSrc ysrc,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, float* __restrict__ dy,
                                                           unsigned short* __restrict__ dys, size_t total8, int C,
                                                           size_t group8, size_t plane_elems, const float* __restrict__ bound,
                                                           float* __restrict__ s_out, const FastDiv fc, const FastDiv fg) {
    RPNET_PASS_PRIORITY();
    float inv_s = 1.f;
    if (NP <= 2) {     // fp16 planes of dy / s, s = pow2ceil(max_c bound[c]) 2^-15 (see bn_bwd_finalize)
        __shared__ float red4[4];
        float m = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, bound[c]);
        const float sc = pow2_scale(block_max256(m, red4));
        if (blockIdx.x == 0 && threadIdx.x == 0) *s_out = sc;
        inv_s = 1.f / sc;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)fc.mod((unsigned)i);
        const int g = (int)fg.div((unsigned)i);
        float r[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int o = g * C + c8 * 8 + hh * 4;
            const f32x4 v = load_y4(ysrc, i * 8 + hh * 4, c8 * 8 + hh * 4);
            const f32x4 d = reinterpret_cast<const f32x4*>(dz)[i * 2 + hh];
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + o);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + o);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + o);
            const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + o);
            const f32x4 cf0 = *reinterpret_cast<const f32x4*>(coef + (size_t)o * 2), cf1 = *reinterpret_cast<const f32x4*>(coef + (size_t)o * 2 + 4);
            const float c1[4] = {cf0[0], cf0[2], cf1[0], cf1[2]}, c2[4] = {cf0[1], cf0[3], cf1[1], cf1[3]};
            f32x4 q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dm = (v[k] * sc[k] + sh[k] > 0.f) ? d[k] : 0.f;
                const float xh = (v[k] - mu[k]) * is[k];
                q[k] = sc[k] * (dm - c1[k] - xh * c2[k]);
                r[hh * 4 + k] = q[k];
            }
            if (dy) reinterpret_cast<f32x4*>(dy)[i * 2 + hh] = q;
        }
        if (NP <= 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] *= inv_s;
        }
        u32x4 pl[NP];
        split8<NP>(r, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dys + p * plane_elems + i * 8) = pl[p];
    }
}


// ---- BatchNorm + ReLU + MaxPool2d(2, 2) in one pass (train mode; net/unet.py:442-448: x1 and x2 feed nothing but their
// pool).  The fp32 z of the full-resolution tensor is never written: max and ReLU commute, so the pooled value is
// relu(max of the four affine values), and the backward finds the window's first maximum again from y.
struct PoolGeom {
    int Ho, Wo, W;          // pooled height / width, input width
    int imgs_per_group;     // N / groups
    size_t img_elems;       // H W C of one input image
    FastDiv fC8, fWo, fHo, fIpg, fHoWo;     // the index arithmetic of the passes (C / 8, Wo, Ho, imgs_per_group, Ho * Wo)
};
static PoolGeom make_pool_geom(int HW, int pool_w, int N, int groups, int C) {
    PoolGeom pg{HW / pool_w / 2, pool_w / 2, pool_w, N / groups, (size_t)HW * C, {}, {}, {}, {}, {}};
    pg.fC8 = FastDiv(C / 8); pg.fWo = FastDiv(pg.Wo); pg.fHo = FastDiv(pg.Ho); pg.fIpg = FastDiv(pg.imgs_per_group);
    pg.fHoWo = FastDiv((unsigned)pg.Ho * pg.Wo);
    return pg;
}

// thread = one pooled pixel x 8 channels; planes [NP][N, Ho, Wo, C] of relu(max) / s (and the fp32 pooled tensor when zp != NULL)
// DRAIN: as bn_bwd_apply_pool_split below — every load of a half waited for with a full `s_waitcnt vmcnt(0)` before any of its
// values is read.  Round 4 found wrong 2 x 2 window decisions in the backward form of this pass when it shared its CUs with other
// kernels of the step; nobody has seen THIS pass fail, but it makes the same decisions from the same loads on the same streams, so
// until the mechanism is known it runs with the same two guards (the wait, and the LDS reservation of the launch).
template <int NP, bool DRAIN = false>
__global__ __launch_bounds__(256) void bn_relu_pool_split_kernel(const YSrc ysrc, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ zp,
                                                                  unsigned short* __restrict__ zs, size_t total8, int C8,
                                                                  PoolGeom pg, size_t plane_elems, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float sqrt_n,
                                                                  float* __restrict__ s_out) {
    float inv_s = 1.f;
    if (NP <= 2) {
        __shared__ float red4[4];
        float m = 0.f;
        for (int c = threadIdx.x; c < C8 * 8; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_n + fabsf(beta[c]));
        const float sc = pow2_scale(block_max256(m, red4));
        if (blockIdx.x == 0 && threadIdx.x == 0) *s_out = sc;
        inv_s = 1.f / sc;
    }
    const size_t rowC = (size_t)pg.W * C8 * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        unsigned pp, py, pn;
        const int c8 = (int)pg.fC8.divmod((unsigned)i, pp);
        const int ox = (int)pg.fWo.divmod(pp, py);
        const int oy = (int)pg.fHo.divmod(py, pn);
        const int n = (int)pn;
        const int g = (int)pg.fIpg.div(pn);
        const float* sc = scale + (size_t)(g * C8 + c8) * 8;
        const float* sh = shift + (size_t)(g * C8 + c8) * 8;
        const size_t src = (size_t)n * pg.img_elems + ((size_t)(2 * oy) * pg.W + 2 * ox) * C8 * 8 + c8 * 8;
        float v[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int cc = c8 * 8 + hh * 4;
            f32x4 a = load_y4(ysrc, src + hh * 4, cc);
            f32x4 b = load_y4(ysrc, src + C8 * 8 + hh * 4, cc);
            f32x4 c = load_y4(ysrc, src + rowC + hh * 4, cc);
            f32x4 d = load_y4(ysrc, src + rowC + C8 * 8 + hh * 4, cc);
            f32x4 s4 = reinterpret_cast<const f32x4*>(sc)[hh], h4 = reinterpret_cast<const f32x4*>(sh)[hh];
            if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(s4), "+v"(h4) : : "memory");
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = fmaxf(fmaxf(a[k] * s4[k] + h4[k], b[k] * s4[k] + h4[k]), fmaxf(c[k] * s4[k] + h4[k], d[k] * s4[k] + h4[k]));
                o[k] = fmaxf(m, 0.f);
                v[hh * 4 + k] = o[k];
            }
            if (zp) reinterpret_cast<f32x4*>(zp)[i * 2 + hh] = o;
        }
        if (NP <= 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= inv_s;
        }
        u32x4 pl[NP];
        split8<NP>(v, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(zs + p * plane_elems + i * 8) = pl[p];
    }
}

// the window's first maximum in (row, column) scan order (maxpool2_bwd_kernel's rule) receives the pooled gradient
__device__ __forceinline__ int pool_argmax(const float a0, const float a1, const float a2, const float a3, float& best) {
    int q = 0; best = a0;
    if (a1 > best) { best = a1; q = 1; }
    if (a2 > best) { best = a2; q = 2; }
    if (a3 > best) { best = a3; q = 3; }
    return q;
}

// reduction pass over POOLED rows (Rp = R / 4 per group): dz is zero off the window maxima, so
// s1 = sum dp [z_max > 0], s2 = sum dp [z_max > 0] xhat(argmax); same partial layout as bn_bwd_partial
template <int WIN>
__global__ __launch_bounds__(256) void bn_bwd_partial_pool(const float* __restrict__ dp, const YSrc ysrc,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            double* __restrict__ partial, float* __restrict__ pmax, long Rp, int C,
                                                            BnGeom gm, PoolGeom pg) {
    __shared__ double red[WIN * 8];
    __shared__ float redm[WIN * 4];
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x;
    const int tc = t % gm.C4, tr = t / gm.C4;
    const int g = blockIdx.y, blk = blockIdx.x;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (tr < gm.rows_it) {
        f32x4 sc = *reinterpret_cast<const f32x4*>(scale + g * C + tc * 4);
        f32x4 sh = *reinterpret_cast<const f32x4*>(shift + g * C + tc * 4);
        f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + tc * 4);
        f32x4 is = *reinterpret_cast<const f32x4*>(invstd + g * C + tc * 4);
        // (window decisions from values behind a FULL wait, the loaded registers tied: see bn_bwd_apply_pool_split)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sc), "+v"(sh), "+v"(mu), "+v"(is) : : "memory");
        const long r0 = (long)blk * gm.rows_blk;
        const long r1 = min(r0 + gm.rows_blk, Rp);
        const size_t gbase = (size_t)g * pg.imgs_per_group * pg.img_elems + tc * 4;
        const size_t rowC = (size_t)pg.W * C;
        for (long r = r0 + tr; r < r1; r += gm.rows_it) {
            unsigned nq, oq;
            const unsigned rem = pg.fHoWo.divmod((unsigned)r, nq);
            const int ox = (int)pg.fWo.divmod(rem, oq);
            const int nl = (int)nq, oy = (int)oq;
            const size_t src = gbase + (size_t)nl * pg.img_elems + ((size_t)(2 * oy) * pg.W + 2 * ox) * C;
            f32x4 v0 = load_y4(ysrc, src, tc * 4), v1 = load_y4(ysrc, src + C, tc * 4);
            f32x4 v2 = load_y4(ysrc, src + rowC, tc * 4), v3 = load_y4(ysrc, src + rowC + C, tc * 4);
            f32x4 d = *reinterpret_cast<const f32x4*>(dp + ((size_t)g * Rp + r) * C + tc * 4);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(d) : : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float best;
                const int q = pool_argmax(v0[k] * sc[k] + sh[k], v1[k] * sc[k] + sh[k], v2[k] * sc[k] + sh[k], v3[k] * sc[k] + sh[k], best);
                const float vq = q == 0 ? v0[k] : q == 1 ? v1[k] : q == 2 ? v2[k] : v3[k];
                const float dm = best > 0.f ? d[k] : 0.f;
                s1[k] += dm;
                s2[k] += (double)dm * ((vq - mu[k]) * is[k]);
                mx[k] = fmaxf(mx[k], fabsf(dm));
            }
        }
    }
    rows_reduce<WIN, true>(s1, s2, mx, red, redm, t, tc, tr, gm.C4, gm.rows_it);
    if (tr == 0) {
        double* o = partial + ((size_t)(g * gm.nblk + blk) * C + tc * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k * 2] = s1[k]; o[k * 2 + 1] = s2[k]; }
        if (pmax) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pmax[(size_t)(g * gm.nblk + blk) * C + tc * 4 + k] = mx[k];
        }
    }
}

// thread = one pooled pixel x 8 channels: the four dy of its window (planes at full resolution, fp32 too when dy != NULL)
//
// (Round 5 on the diagnosis below: the trigger is NOT specific to the LDS-DMA weight-gradient kernel — with both guards off and
// RPNET_BN_LDS=big the two-chain schedule WITHOUT a weight-gradient side stream still differs in 8 of 8 steps
// (profiles/r05_pool_fault_repro.txt); the stand-alone pair of this pass and one weight-gradient launch does not reproduce it
// (0 of 400), and a probe that reads element 1 of a 16-byte load in the instruction behind its counted wait, beside an LDS-DMA
// streaming kernel, sees no stale value in 6.7e9 reads (tools/probe/vmcnt_probe.cpp).  What is established stays: the symptom, that the
// two guards remove it in every schedule tried, and the tests that would show it again.)
// DRAIN (round 4): every load of a half is waited for with `s_waitcnt vmcnt(0)` BEFORE any of its values is read (the asm
// statement ties the loaded registers, so the compiler can make no early copy).  Without it hipcc issues the 18 loads of an
// iteration back to back and reads them behind counted waits (vmcnt(17), vmcnt(16), ...), which is correct while loads
// return in issue order.  Observed on MI355X (profiles/r04_pool_apply_fault.txt): when the blocks of
// this pass share their CUs with a running LDS-DMA weight-gradient kernel (conv_wgrad9_dma_kernel on the side stream — the
// default schedule since round 3), a few dozen of the 4 - 8 M window decisions of a launch come out as if the FIRST value
// compared (element 1 of a float4 of y) had not yet landed: the pooled gradient then lands on the wrong pixel of its 2 x 2
// window, every later value of the thread (read behind later waits) is right.  Inputs and outputs of the launch were pinned
// and compared between runs: identical y, statistics and dz, 2 - 130 differing elements of dy, 25 - 100 % of the steps
// depending on what else is resident; with the full wait: 0 of 16.  The wrong decisions moved the gradients of the
// layers below by 1e-4 ... 5e-2 (relative, max norm) in those steps.
template <int NP, bool DRAIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void bn_bwd_apply_pool_split(const float* __restrict__ dp, const YSrc ysrc,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ coef, float* __restrict__ dy,
                                                                unsigned short* __restrict__ dys, size_t total8, int C, PoolGeom pg,
                                                                size_t plane_elems, const float* __restrict__ bound,
                                                                float* __restrict__ s_out) {
    const int C8 = C / 8;
    float inv_s = 1.f;
    if (NP <= 2) {
        __shared__ float red4[4];
        float m = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, bound[c]);
        const float sc = pow2_scale(block_max256(m, red4));
        if (blockIdx.x == 0 && threadIdx.x == 0) *s_out = sc;
        inv_s = 1.f / sc;
    }
    const size_t rowC = (size_t)pg.W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        unsigned pp, py, pn;
        const int c8 = (int)pg.fC8.divmod((unsigned)i, pp);
        const int ox = (int)pg.fWo.divmod(pp, py);
        const int oy = (int)pg.fHo.divmod(py, pn);
        const int n = (int)pn;
        const int g = (int)pg.fIpg.div(pn);
        const size_t e00 = (size_t)n * pg.img_elems + ((size_t)(2 * oy) * pg.W + 2 * ox) * C + c8 * 8;
        const size_t offs[4] = {e00, e00 + C, e00 + rowC, e00 + rowC + C};
        float r[4][8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int o = g * C + c8 * 8 + hh * 4;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = load_y4(ysrc, offs[q] + hh * 4, c8 * 8 + hh * 4);
            f32x4 d = reinterpret_cast<const f32x4*>(dp)[i * 2 + hh];
            f32x4 sc = *reinterpret_cast<const f32x4*>(scale + o);
            f32x4 sh = *reinterpret_cast<const f32x4*>(shift + o);
            f32x4 mu = *reinterpret_cast<const f32x4*>(mean + o);
            f32x4 is = *reinterpret_cast<const f32x4*>(invstd + o);
            f32x4 cf0 = *reinterpret_cast<const f32x4*>(coef + (size_t)o * 2), cf1 = *reinterpret_cast<const f32x4*>(coef + (size_t)o * 2 + 4);
            if constexpr (DRAIN)      // every load of this half has landed before ANY of its values is read: the loaded registers
                                      // THEMSELVES are tied to the statement (no copy the compiler could make behind a counted wait)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(d), "+v"(sc), "+v"(sh), "+v"(mu), "+v"(is), "+v"(cf0), "+v"(cf1)
                             :
                             : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float best;
                const int qb = pool_argmax(v[0][k] * sc[k] + sh[k], v[1][k] * sc[k] + sh[k], v[2][k] * sc[k] + sh[k],
                                           v[3][k] * sc[k] + sh[k], best);
                const float dm = best > 0.f ? d[k] : 0.f;
                const float c1 = k < 2 ? cf0[2 * k] : cf1[2 * k - 4], c2 = k < 2 ? cf0[2 * k + 1] : cf1[2 * k - 3];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r[q][hh * 4 + k] = sc[k] * ((q == qb ? dm : 0.f) - c1 - (v[q][k] - mu[k]) * is[k] * c2);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (dy) {
                *reinterpret_cast<f32x4*>(dy + offs[q]) = f32x4{r[q][0], r[q][1], r[q][2], r[q][3]};
                *reinterpret_cast<f32x4*>(dy + offs[q] + 4) = f32x4{r[q][4], r[q][5], r[q][6], r[q][7]};
            }
            if (NP <= 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) r[q][k] *= inv_s;
            }
            u32x4 pl[NP];
            split8<NP>(r[q], pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dys + p * plane_elems + offs[q]) = pl[p];
        }
    }
}

// LDS bytes the pooled passes reserve WITHOUT using them (RPNET_BN_POOL_ALONE: unset / 1 = 53248, 0 = none, n > 1 = n bytes), so
// that their blocks never share a CU with a block of an LDS-DMA kernel.  Round 4 reserved 24 KB, enough against the 144 - 156 KB
// blocks of the weight-gradient and 128-wide convolution kernels; the 64-wide convolution forms (116 - 120 KB) and the collapsed
// up_conv weight gradient (112 KB) would still have fitted beside such a block.  52 KB excludes every LDS-DMA kernel the training
// step launches (the smallest, 112 KB: 114688 + 53248 > 163840) and still leaves three blocks = twelve waves of the pass per CU.
static size_t pool_alone_bytes() {
    static const size_t v = [] {
        const char* e = getenv("RPNET_BN_POOL_ALONE");
        if (!e || !e[0]) return (size_t)kGuardedPassLds;
        const long n = atol(e);
        return n <= 0 ? (size_t)0 : (n == 1 ? (size_t)kGuardedPassLds : (size_t)n);
    }();
    return v;
}

static int elt_grid(size_t total4) {
    size_t b = (total4 + 255) / 256;
    return (int)(b > 2048 * 4 ? 2048 * 4 : (b < 1 ? 1 : b));
}

}  // namespace rpnet

extern "C" size_t rpnet_bn_workspace_bytes(int C, int groups) {
    // partial sums (fp64), coefficients, per-block maxima and per-channel bounds (the last two for fp16 split outputs)
    const size_t rows = (size_t)groups * rpnet::bn_max_blocks(C);
    return rows * C * 2 * sizeof(double) + (size_t)groups * C * 2 * sizeof(float) + rows * C * sizeof(float) +
           (size_t)C * sizeof(float);
}

extern "C" size_t rpnet_bn_bwd_coef_offset(int C, int groups) {
    return (size_t)groups * rpnet::bn_max_blocks(C) * C * 2 * sizeof(double);
}

static int bn_check(const char* who, int N, int HW, int C, int groups) {
    using namespace rpnet;
    RPNET_REQUIRE(C % 4 == 0 && C / 4 <= 256, RPNET_ERR_SHAPE, "%s: C=%d must be a multiple of 4 and <= 1024", who, C);
    RPNET_REQUIRE(groups >= 1 && N % groups == 0, RPNET_ERR_SHAPE, "%s: N=%d not divisible by groups=%d", who, N, groups);
    (void)HW;
    return 0;
}

extern "C" int rpnet_bn_stats(const float* y, int N, int HW, int C, int groups, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, long long* num_batches_tracked,
                              float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                              void* workspace, size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(y && gamma && beta && scale && shift && mean && invstd && workspace, RPNET_ERR_ARG, "bn_stats: null pointer");
    if (int rc = bn_check("bn_stats", N, HW, C, groups)) return rc;
    RPNET_REQUIRE(workspace_bytes >= rpnet_bn_workspace_bytes(C, groups), RPNET_ERR_WORKSPACE, "bn_stats: workspace too small");
    const long R = (long)(N / groups) * HW;
    const BnGeom gm = bn_geom(R, C);
    hipStream_t s = (hipStream_t)stream;
    if (bn_lds_window() == 64) hipLaunchKernelGGL(bn_stats_partial<64>, dim3(gm.nblk, groups), dim3(256), 0, s, y, (double*)workspace, R, C, gm);
    else hipLaunchKernelGGL(bn_stats_partial<256>, dim3(gm.nblk, groups), dim3(256), 0, s, y, (double*)workspace, R, C, gm);
    hipLaunchKernelGGL(bn_stats_finalize, dim3(C), dim3(256), 0, s, (const double*)workspace, gm.nblk, R, C,
                       groups, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift,
                       mean, invstd);
    return check_launch("bn_stats");
}

extern "C" int rpnet_bn_stats_from_partial(const double* partial, int nblk, int N, int HW, int C, int groups,
                                           const float* gamma, const float* beta, float* running_mean,
                                           float* running_var, long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift,
                                           float* mean, float* invstd, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(partial && gamma && beta && scale && shift && mean && invstd && nblk > 0, RPNET_ERR_ARG,
                  "bn_stats_from_partial: null pointer");
    RPNET_REQUIRE(groups >= 1 && N % groups == 0, RPNET_ERR_SHAPE, "bn_stats_from_partial: N=%d groups=%d", N, groups);
    const long R = (long)(N / groups) * HW;
    hipLaunchKernelGGL(bn_stats_finalize, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nblk, R, C, groups, gamma,
                       beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, mean, invstd);
    return check_launch("bn_stats_from_partial");
}

extern "C" int rpnet_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* scale, float* shift, int C,
                                    rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, RPNET_ERR_ARG, "bn_eval_affine: null pointer");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, scale, shift, C);
    return check_launch("bn_eval_affine");
}

extern "C" int rpnet_bn_relu(const float* y, const float* scale, const float* shift, float* z, void* z_split, int planes,
                             const float* gamma, const float* beta, float* split_scale, int N, int HW, int C, int groups,
                             int pool_w, const float* y_dec, int y_dec_stride, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(y && scale && shift && (z || z_split), RPNET_ERR_ARG, "bn_relu: null pointer");
    RPNET_REQUIRE(!y_dec && !y_dec_stride, RPNET_ERR_ARG, "bn_relu: y_dec / y_dec_stride are reserved (must be NULL / 0)");
    const YSrc ysrc{y};
    if (int rc = bn_check("bn_relu", N, HW, C, groups)) return rc;
    if (pool_w > 0) {      // + MaxPool2d(2, 2): z / z_split are [N, H/2, W/2, C]
        RPNET_REQUIRE(z_split && planes >= 1 && planes <= 3 && C % 8 == 0, RPNET_ERR_ARG, "bn_relu: the pooled form writes split planes (planes=%d C=%d)", planes, C);
        RPNET_REQUIRE(HW % pool_w == 0 && pool_w % 2 == 0 && (HW / pool_w) % 2 == 0, RPNET_ERR_SHAPE, "bn_relu: pooled image %d x %d", HW / pool_w, pool_w);
        RPNET_REQUIRE(planes == 3 || (gamma && beta && split_scale), RPNET_ERR_ARG, "bn_relu: fp16 planes (1 or 2) need gamma, beta and the scale output");
        const PoolGeom pg = make_pool_geom(HW, pool_w, N, groups, C);
        const size_t total8 = (size_t)N * (HW / 4) * C / 8, pe = (size_t)N * (HW / 4) * C;
        RPNET_REQUIRE(total8 < kIndex32, RPNET_ERR_SHAPE, "bn (pooled): %zu elements do not fit the 32-bit index arithmetic of the passes", total8);
        const float sqrt_n = sqrtf((float)((size_t)(N / groups) * HW)) * 1.0001f;
        // the guards of the pooled passes (see bn_bwd_apply_pool_split): LDS reservation of the launch, full wait in the kernel
        const size_t alone_f = pool_alone_bytes();
        static const bool drain_f = [] { const char* e = getenv("RPNET_BN_POOL_DRAIN"); return !(e && e[0] == '0'); }();
#define RPNET_BN_POOL(NP_)                                                                                                  \
    do { if (drain_f) hipLaunchKernelGGL((bn_relu_pool_split_kernel<NP_, true>), dim3(elt_grid(total8)), dim3(256), alone_f, (hipStream_t)stream, ysrc, scale, shift, z, \
                       (unsigned short*)z_split, total8, C / 8, pg, pe, gamma, beta, sqrt_n, split_scale);                  \
    else hipLaunchKernelGGL((bn_relu_pool_split_kernel<NP_, false>), dim3(elt_grid(total8)), dim3(256), alone_f, (hipStream_t)stream, ysrc, scale, shift, z, \
                       (unsigned short*)z_split, total8, C / 8, pg, pe, gamma, beta, sqrt_n, split_scale); } while (0)
        if (planes == 3) RPNET_BN_POOL(3);
        else if (planes == 2) RPNET_BN_POOL(2);
        else RPNET_BN_POOL(1);
#undef RPNET_BN_POOL
        return check_launch("bn_relu_pool");
    }
    const size_t total4 = (size_t)N * HW * C / 4, group4 = total4 / groups;
    RPNET_REQUIRE(total4 < kIndex32, RPNET_ERR_SHAPE, "bn: %zu 16-byte elements do not fit the 32-bit index arithmetic of the passes", total4);
    if (z_split) {
        RPNET_REQUIRE(planes >= 1 && planes <= 3 && C % 8 == 0, RPNET_ERR_SHAPE, "bn_relu: split planes=%d C=%d", planes, C);
        const size_t total8 = total4 / 2, pe = (size_t)N * HW * C;
        RPNET_REQUIRE(planes == 3 || (gamma && beta && split_scale), RPNET_ERR_ARG,
                      "bn_relu: fp16 planes (1 or 2) need gamma, beta and the scale output");
        const float sqrt_n = sqrtf((float)((size_t)(N / groups) * HW)) * 1.0001f;
        if (planes == 3)
            hipLaunchKernelGGL(bn_relu_split_kernel<3>, dim3(elt_grid(total8)), dim3(256), 0, (hipStream_t)stream, ysrc, scale,
                               shift, z, (unsigned short*)z_split, total8, C / 8, group4 / 2, pe, gamma, beta, sqrt_n, split_scale, FastDiv(C / 8), FastDiv((unsigned)(group4 / 2)));
        else if (planes == 2)
            hipLaunchKernelGGL(bn_relu_split_kernel<2>, dim3(elt_grid(total8)), dim3(256), 0, (hipStream_t)stream, ysrc, scale,
                               shift, z, (unsigned short*)z_split, total8, C / 8, group4 / 2, pe, gamma, beta, sqrt_n, split_scale, FastDiv(C / 8), FastDiv((unsigned)(group4 / 2)));
        else
            hipLaunchKernelGGL(bn_relu_split_kernel<1>, dim3(elt_grid(total8)), dim3(256), 0, (hipStream_t)stream, ysrc, scale,
                               shift, z, (unsigned short*)z_split, total8, C / 8, group4 / 2, pe, gamma, beta, sqrt_n, split_scale, FastDiv(C / 8), FastDiv((unsigned)(group4 / 2)));
        return check_launch("bn_relu_split");
    }
    RPNET_REQUIRE(!split_scale || (gamma && beta), RPNET_ERR_ARG, "bn_relu: the tensor scale needs gamma and beta");
    hipLaunchKernelGGL(bn_relu_kernel, dim3(elt_grid(total4)), dim3(256), 0, (hipStream_t)stream, ysrc, scale, shift, z,
                       total4, C / 4, group4, gamma, beta, sqrtf((float)((size_t)(N / groups) * HW)) * 1.0001f, split_scale, FastDiv(C / 4),
                       FastDiv((unsigned)group4));
    return check_launch("bn_relu");
}

extern "C" int rpnet_bn_act_scale(const float* gamma, const float* beta, float* split_scale, int N, int HW, int C, int groups,
                                  rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(gamma && beta && split_scale && groups >= 1 && C > 0, RPNET_ERR_ARG, "bn_act_scale: bad argument");
    hipLaunchKernelGGL(bn_act_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, C,
                       sqrtf((float)((size_t)(N / groups) * HW)) * 1.0001f, split_scale);
    return check_launch("bn_act_scale");
}

extern "C" int rpnet_bn_bwd(const float* dz, const float* y, const float* gamma, const float* scale, const float* shift,
                            const float* mean, const float* invstd, float* dy, void* dy_split, int planes, float* split_scale,
                            float* dgamma, float* dbeta, int N, int HW, int C, int groups, int accumulate,
                            const double* given_partial, const float* given_pmax, int given_rows, int pool_w, void* workspace,
                            size_t workspace_bytes, const float* y_dec, int y_dec_stride, rpnet_stream_t stream) {
    using namespace rpnet;
    (void)gamma;
    RPNET_REQUIRE(!y_dec && !y_dec_stride, RPNET_ERR_ARG, "bn_bwd: y_dec / y_dec_stride are reserved (must be NULL / 0)");
    const YSrc ysrc{y};
    RPNET_REQUIRE(dz && scale && shift && mean && invstd && dgamma && dbeta && workspace, RPNET_ERR_ARG, "bn_bwd: null pointer");
    // y may be NULL only where nothing reads it: the reduction already ran elsewhere (given_partial) and no dy is asked for
    RPNET_REQUIRE(y || (given_partial && !dy && !dy_split), RPNET_ERR_ARG, "bn_bwd: y is NULL but a pass that reads it was asked for");
    RPNET_REQUIRE(!dy_split || (planes >= 1 && planes <= 3 && C % 8 == 0), RPNET_ERR_SHAPE, "bn_bwd: split planes=%d C=%d",
                  planes, C);
    if (int rc = bn_check("bn_bwd", N, HW, C, groups)) return rc;
    RPNET_REQUIRE(workspace_bytes >= rpnet_bn_workspace_bytes(C, groups), RPNET_ERR_WORKSPACE, "bn_bwd: workspace too small");
    const long R = (long)(N / groups) * HW;
    const BnGeom gm = bn_geom(R, C);
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)workspace;
    float* coef = (float*)((char*)workspace + (size_t)groups * bn_max_blocks(C) * C * 2 * sizeof(double));
    const bool f16 = dy_split && planes <= 2;
    RPNET_REQUIRE(!f16 || split_scale, RPNET_ERR_ARG, "bn_bwd: fp16 planes (1 or 2) need the scale output");
    float* pmax = f16 ? coef + (size_t)groups * C * 2 : nullptr;
    float* bound = f16 ? pmax + (size_t)groups * bn_max_blocks(C) * C : nullptr;
    if (pool_w > 0) {
        // dz is the gradient of the POOLED output [N, H/2, W/2, C] (rpnet_bn_relu(pool_w)): the window maxima are found
        // again from y; dy comes out at full resolution
        RPNET_REQUIRE(dy_split && !given_partial, RPNET_ERR_ARG, "bn_bwd: the pooled form writes split planes and runs its own reduction");
        RPNET_REQUIRE(HW % pool_w == 0 && pool_w % 2 == 0 && (HW / pool_w) % 2 == 0, RPNET_ERR_SHAPE, "bn_bwd: pooled image %d x %d", HW / pool_w, pool_w);
        const PoolGeom pg = make_pool_geom(HW, pool_w, N, groups, C);
        const long Rp = R / 4;
        const BnGeom gp = bn_geom(Rp, C);
        // RPNET_BN_POOL_ALONE (default on): the pooled passes reserve LDS they do not use, so that their blocks never share a CU with
        // a block of an LDS-DMA kernel (pool_alone_bytes) — see bn_bwd_apply_pool_split
        const size_t alone = pool_alone_bytes();
        if (bn_lds_window() == 64)
            hipLaunchKernelGGL(bn_bwd_partial_pool<64>, dim3(gp.nblk, groups), dim3(256), alone, s, dz, ysrc, scale, shift, mean, invstd, partial,
                               pmax, Rp, C, gp, pg);
        else
            hipLaunchKernelGGL(bn_bwd_partial_pool<256>, dim3(gp.nblk, groups), dim3(256), alone, s, dz, ysrc, scale, shift, mean, invstd, partial,
                               pmax, Rp, C, gp, pg);
        hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, s, (const double*)partial, gp.nblk, R, C, groups,
                           coef, dgamma, dbeta, accumulate, (const float*)pmax, scale, bound);
        const size_t total8 = (size_t)N * (HW / 4) * C / 8, pe = (size_t)N * HW * C;
        RPNET_REQUIRE(total8 < kIndex32, RPNET_ERR_SHAPE, "bn (pooled): %zu elements do not fit the 32-bit index arithmetic of the passes", total8);
        // DRAIN (default; RPNET_BN_POOL_DRAIN=0: the A/B switch that brings the fault back): see bn_bwd_apply_pool_split
        static const bool drain = [] { const char* e = getenv("RPNET_BN_POOL_DRAIN"); return !(e && e[0] == '0'); }();
#define RPNET_BN_POOL_BWD(NP_)                                                                                              \
    do { if (drain) hipLaunchKernelGGL((bn_bwd_apply_pool_split<NP_, true>), dim3(elt_grid(total8)), dim3(256), alone, s, dz, ysrc, scale, shift, mean, invstd, \
                       (const float*)coef, dy, (unsigned short*)dy_split, total8, C, pg, pe, (const float*)bound, split_scale); \
    else hipLaunchKernelGGL((bn_bwd_apply_pool_split<NP_, false>), dim3(elt_grid(total8)), dim3(256), alone, s, dz, ysrc, scale, shift, mean, invstd, \
                       (const float*)coef, dy, (unsigned short*)dy_split, total8, C, pg, pe, (const float*)bound, split_scale); } while (0)
        if (planes == 3) RPNET_BN_POOL_BWD(3);
        else if (planes == 2) RPNET_BN_POOL_BWD(2);
        else RPNET_BN_POOL_BWD(1);
#undef RPNET_BN_POOL_BWD
        return check_launch("bn_bwd_pool");
    }
    if (given_partial) {
        // the caller made the sums itself (the first layer without its pre-BatchNorm tensor: rpnet_conv1_bn_bwd_partial):
        // [groups * given_rows][C][2] sums (and [..][C] maxima for fp16 planes)
        RPNET_REQUIRE(given_rows > 0 && (!f16 || given_pmax), RPNET_ERR_ARG, "bn_bwd: given partial sums need their row count (and maxima for fp16 planes)");
        hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, s, given_partial, given_rows, R, C, groups, coef, dgamma, dbeta,
                           accumulate, f16 ? given_pmax : (const float*)nullptr, scale, bound);
    } else {
        if (bn_lds_window() == 64)
            hipLaunchKernelGGL(bn_bwd_partial<64>, dim3(gm.nblk, groups), dim3(256), 0, s, dz, ysrc, scale, shift, mean, invstd, partial,
                               pmax, R, C, gm);
        else
            hipLaunchKernelGGL(bn_bwd_partial<256>, dim3(gm.nblk, groups), dim3(256), 0, s, dz, ysrc, scale, shift, mean, invstd, partial,
                               pmax, R, C, gm);
        hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, s, (const double*)partial, gm.nblk, R, C, groups,
                           coef, dgamma, dbeta, accumulate, (const float*)pmax, scale, bound);
    }
    // dy == NULL and dy_split == NULL: reduction pass only — dgamma, dbeta and the coefficients (workspace +
    // rpnet_bn_bwd_coef_offset) for a consumer that forms dy itself (rpnet_conv1_wgrad_bn)
    if (!dy && !dy_split) return check_launch("bn_bwd");
    const size_t total4 = (size_t)N * HW * C / 4, group4 = total4 / groups;
    RPNET_REQUIRE(total4 < kIndex32, RPNET_ERR_SHAPE, "bn: %zu 16-byte elements do not fit the 32-bit index arithmetic of the passes", total4);
    if (dy_split) {
        const size_t total8 = total4 / 2, pe = (size_t)N * HW * C;
        if (planes == 3)
            hipLaunchKernelGGL(bn_bwd_apply_split<3>, dim3(elt_grid(total8)), dim3(256), 0, s, dz, ysrc, scale, shift, mean, invstd,
                               (const float*)coef, dy, (unsigned short*)dy_split, total8, C, group4 / 2, pe, (const float*)bound,
                               split_scale, FastDiv(C / 8), FastDiv((unsigned)(group4 / 2)));
        else if (planes == 2)
            hipLaunchKernelGGL(bn_bwd_apply_split<2>, dim3(elt_grid(total8)), dim3(256), 0, s, dz, ysrc, scale, shift, mean, invstd,
                               (const float*)coef, dy, (unsigned short*)dy_split, total8, C, group4 / 2, pe, (const float*)bound,
                               split_scale, FastDiv(C / 8), FastDiv((unsigned)(group4 / 2)));
        else
            hipLaunchKernelGGL(bn_bwd_apply_split<1>, dim3(elt_grid(total8)), dim3(256), 0, s, dz, ysrc, scale, shift, mean, invstd,
                               (const float*)coef, dy, (unsigned short*)dy_split, total8, C, group4 / 2, pe, (const float*)bound,
                               split_scale, FastDiv(C / 8), FastDiv((unsigned)(group4 / 2)));
        return check_launch("bn_bwd");
    }
    hipLaunchKernelGGL(bn_bwd_apply, dim3(elt_grid(total4)), dim3(256), 0, s, dz, ysrc, scale, shift, mean, invstd,
                       (const float*)coef, dy, total4, C, group4, FastDiv(C / 4), FastDiv((unsigned)group4));
    return check_launch("bn_bwd");
}
