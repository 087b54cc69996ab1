// Local-window correlation (Correlation(), net/rp_net.py:153-181; definition in corr.hip) and its gradients on
// the bf16 matrix pipe with split-bf16 operands (the arithmetic of conv_split.hip: fp32 value = NP exact bf16
// planes, NP = 3 -> six partial products, fp32 accumulate).
//
// Forward: a block owns an 8x8 pixel tile p and the (8 + 2R)^2 halo q of f2 around it and computes the dense
// 64 x NQ score matrix S[p][q] = <f1[p,:], f2[q,:]> as a GEMM over the channels (K = C, 32 per step) — 2.9x more
// products than the (2R+1)^2 window needs, on a pipe that is 16x the VALU; the window is then gathered out of S
// through LDS and written as one contiguous [cstride] row per pixel.  Both operands are [pixel][channel] with the
// channel contiguous = K-contiguous: the LDS images and fragment reads are those of the convolution kernels.
//
// Backward (one kernel, SIGN = +1 for d f1 from (dcorr, f2), SIGN = -1 for d f2 from (dcorr transposed, f1)):
// df[p][ch] = sum_q G[p][q] fo[q][ch] with G the window gradient scattered onto the halo positions (zero outside
// the window) — a GEMM over K = NQ halo positions.  G's 32-column slabs are built in LDS from the fp32 gradient
// tile and split on the fly; fo's halo is [q][channel], i.e. K is the ROW index, so its fragments come from
// ds_read_b64_tr_b16 as in conv_wgrad_split.hip.  A block computes 64 pixels x 128 channels.
#include <stdlib.h>

#include "common.h"
#include "split_bf16.h"

namespace rpnet {

using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4c;

template <int R, int NP>
__global__ __launch_bounds__(256, 2) void local_corr_mfma_fwd_kernel(const unsigned short* __restrict__ f1s,
                                                                      const unsigned short* __restrict__ f2s,
                                                                      float* __restrict__ corr, const int B, const int h,
                                                                      const int w, const int C, const int cstride,
                                                                      const float inv_sqrt_c_in, const float* __restrict__ s1,
                                                                      const float* __restrict__ s2, float* __restrict__ out_absmax,
                                                                      unsigned short* __restrict__ cplanes,
                                                                      const float* __restrict__ cscale) {
    // NP <= 2: fp16 planes of f / s with the producers' tensor scales (rpnet_bn_relu); the scores are multiplied by s1 s2
    const float inv_sqrt_c = NP <= 2 ? inv_sqrt_c_in * (*s1 * *s2) : inv_sqrt_c_in;
    constexpr int K = 2 * R + 1, KK = K * K, HT = 8 + 2 * R, NQ = HT * HT, NT_N = (NQ + 31) / 32, NQP = NT_N * 32;
    constexpr int A_BYTES = 64 * 64, B_BYTES = NQP * 64;            // one plane: [row][32 channels]
    constexpr int BJ = (NQ * 4 + 255) / 256;                        // halo pieces per thread and plane
    constexpr int JT = (NT_N + 3) / 4;                              // N tiles per wave
    constexpr int SMEM = NP * (A_BYTES + B_BYTES) > 32 * NQP * 4 ? NP * (A_BYTES + B_BYTES) : 32 * NQP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const bsm = smem + NP * A_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, hh = lane >> 5;
    const int tiles_x = (w + 7) / 8;
    // XCD-aware order (common.h): an XCD works on a contiguous run of tiles — one image of eight — so the 5x re-read
    // of the f2 halo by neighbouring tiles hits that XCD's L2 instead of the fabric: 52 -> 44 us at the CRE shape.
    // (The backward kernel is not fetch-bound: the same order cost it 6 %, a second slab of prefetch 2 %: measured.)
    const int lin = xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int b = lin / gridDim.x, tl = lin - b * gridDim.x;
    const int ty0 = (tl / tiles_x) * 8, tx0 = (tl % tiles_x) * 8;
    const size_t plane = (size_t)B * h * w * C, img = (size_t)b * h * w * C;

    __amdgpu_buffer_rsrc_t r1[NP], r2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        r1[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(f1s + p * plane + img), (short)0, h * w * C * 2, 0x00020000);
        r2[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(f2s + p * plane + img), (short)0, h * w * C * 2, 0x00020000);
    }
    const int skg = t & 3;
    // f1 tile piece: row t >> 2 (pixel of the tile), k-group t & 3
    int aoff, adst;
    {
        const int row = t >> 2, y = ty0 + (row >> 3), x = tx0 + (row & 7);
        aoff = (y < h && x < w) ? (y * w + x) : -1;
        adst = row * 64 + 16 * (skg ^ ((row >> 2) & 3));
    }
    int boff[BJ], bdst[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int e = t + 256 * j, q = e >> 2;
        const int y = ty0 + q / HT - R, x = tx0 + q % HT - R;
        boff[j] = (e < NQ * 4 && y >= 0 && y < h && x >= 0 && x < w) ? (y * w + x) : -1;
        bdst[j] = e < NQ * 4 ? q * 64 + 16 * (skg ^ ((q >> 2) & 3)) : -1;
    }
    // rows NQ .. NQP-1 of the halo image are never gathered; keep them finite
    for (int e = t; e < NP * (NQP - NQ) * 4; e += 256) {
        const int p = e / ((NQP - NQ) * 4), r = e - p * ((NQP - NQ) * 4);
        *reinterpret_cast<u32x4*>(bsm + p * B_BYTES + NQ * 64 + r * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    f32x16 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (li >> 2) & 3;
    // the next 32-channel slab travels from global memory to registers while the MFMAs of the current one run
    u32x4 ra[NP], rb[NP][BJ];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ra[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r1[p], aoff * (C * 2) + skg * 16, c0 * 2, 0));
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                rb[p][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r2[p], boff[j] * (C * 2) + skg * 16, c0 * 2, 0));
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<u32x4*>(smem + p * A_BYTES + adst) = ra[p];
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                if (bdst[j] >= 0) *reinterpret_cast<u32x4*>(bsm + p * B_BYTES + bdst[j]) = rb[p][j];
        }
        if (c0 + 32 < C) fetch(c0 + 32);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = 16 * ((2 * s + hh) ^ sw);
            bf16x8 af[NP][2], bfr[NP][JT];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) af[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * A_BYTES + (i * 32 + li) * 64 + koff);
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const int jt = wv + 4 * j;
                    bfr[p][j] = *reinterpret_cast<const bf16x8*>(bsm + p * B_BYTES + ((jt < NT_N ? jt : 0) * 32 + li) * 64 + koff);
                }
            }
            constexpr int NPROD = nprod<NP>();
#pragma unroll
            for (int q = 0; q < NPROD; ++q) {
                const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
                for (int j = 0; j < JT; ++j)
                    if (wv + 4 * j < NT_N) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[i][j] = mma16<NP>(af[pa][i], bfr[pb][j], acc[i][j]);
                    }
            }
        }
    }
    // gather the (2R+1)^2 window of every pixel out of S, half a tile (32 pixels) at a time
    float* S = reinterpret_cast<float*>(smem);          // [32][NQP]
    float* cb = corr + (size_t)b * h * w * cstride;
    // optional (round 6, eval mode): the correlation also as fp16 planes of corr / *cscale — a PREDICTED power-of-two scale (the caller's
    // rpnet_predict_scales history; out_absmax of this launch is the check), what the 1x1 convolution over cat([corr, fm1]) reads:
    // no second pass over the tensor
    unsigned short* cpb = cplanes ? cplanes + (size_t)b * h * w * cstride : nullptr;
    const size_t cpstride = (size_t)B * h * w * cstride;
    const float cinv = cplanes ? 1.f / *cscale : 1.f;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int jt = wv + 4 * j;
            if (jt < NT_N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * hh) * NQP + jt * 32 + li] = acc[i][j][r];
            }
        }
        __syncthreads();
        for (int e = t; e < 32 * cstride; e += 256) {
            const int pl = e / cstride, o = e - pl * cstride;
            const int py = i * 4 + (pl >> 3), px = pl & 7;
            const int y = ty0 + py, x = tx0 + px;
            float v = 0.f;
            if (o < KK) {
                const int a = o / K, c = o - a * K;
                v = S[pl * NQP + (py + c) * HT + px + a] * inv_sqrt_c;
            }
            if (y < h && x < w) {
                const size_t idx = ((size_t)y * w + x) * cstride + o;
                cb[idx] = v;
                amax = fmaxf(amax, fabsf(v));
                if (cpb) {
                    const float vs = v * cinv;
                    const unsigned hb = f16_bits(vs);
                    cpb[idx] = (unsigned short)hb;
                    if (NP == 2) cpb[cpstride + idx] = (unsigned short)f16_bits(vs - f16_val(hb));
                }
            }
        }
    }
    if (out_absmax) {      // max |corr| of the launch (as rpnet_conv_desc.out_absmax): the bound its fp16 planes are scaled by
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0 && amax > __hip_atomic_load(out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned*>(out_absmax), __float_as_uint(amax));
    }
}

// df[b,p,ch] = inv_sqrt_c * sum_o g[b,p,o] * fo[b, p + SIGN*off(o), ch]   (see corr.hip)
// A block = 8 x 8 pixels x 128 NJ channels: building the split G slab (VALU work that does not depend on the channel) is
// what a block spends most of its issue slots on, so NJ = 2 (all 256 channels of the encoder features) halves it per MFMA.
template <int R, int NP, int SIGN, int NJ>
__global__ __launch_bounds__(256, 2) void local_corr_mfma_bwd_kernel(const float* __restrict__ g, const unsigned short* __restrict__ fos,
                                                                      float* __restrict__ df, const int B, const int h,
                                                                      const int w, const int C, const int cstride,
                                                                      const float inv_sqrt_c, const float* __restrict__ s_fo,
                                                                      const float* __restrict__ add, const int xcd_order) {
    constexpr int K = 2 * R + 1, KK = K * K, HT = 8 + 2 * R, NQ = HT * HT, NCH = (NQ + 31) / 32, GS = (KK + 3) & ~3;
    constexpr int BN = 128 * NJ, RSB = BN * 2 + 64;                     // fo row: 128 channels + pad (conflict-free transposing reads)
    constexpr int A_BYTES = 64 * 64, B_BYTES = 32 * RSB;
    constexpr int LOOP_LDS = 64 * GS * 4 + NP * (A_BYTES + B_BYTES), OUT_LDS = 64 * (BN + 4) * 4;     // K loop / output tile (epilogue)
    __shared__ __attribute__((aligned(16))) unsigned char smem[LOOP_LDS > OUT_LDS ? LOOP_LDS : OUT_LDS];
    float* gs = reinterpret_cast<float*>(smem);                     // [64][GS] window gradients of the tile
    unsigned char* const asm_ = smem + 64 * GS * 4;
    unsigned char* const bsm = asm_ + NP * A_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, hh = lane >> 5;
    const int tiles_x = (w + 7) / 8;
    int b = blockIdx.z, tl = blockIdx.x;
    const int n0 = blockIdx.y * BN;
    const int abl = xcd_order >> 4;          // diagnostic (tools/bench_corr_bwd.py, RPNET_CORR_BWD_XCD = 16 * bits + order): 1 = no G build,
                                               // 2 = no fo staging, 4 = no MFMAs, 8 = no epilogue stores; results are then meaningless
    if ((xcd_order & 1) && gridDim.y == 1) {
        // XCD-aware order (common.h): an XCD works on a contiguous run of tiles, so that the halo rows neighbouring tiles share
        // are fetched into ONE private L2 instead of up to five
        const int lin = xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
        b = lin / gridDim.x;
        tl = lin - b * gridDim.x;
    }
    const int ty0 = (tl / tiles_x) * 8, tx0 = (tl % tiles_x) * 8;
    const size_t plane = (size_t)B * h * w * C, img = (size_t)b * h * w * C;
    __amdgpu_buffer_rsrc_t rf[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
        rf[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(fos + p * plane + img), (short)0, h * w * C * 2, 0x00020000);

    // fo slab pieces: rows brow + RPP j, 16-byte piece t % PR of the row's BN channels
    constexpr int PR = 16 * NJ, RPP = 256 / PR, BJ = 32 / RPP;
    const int brow = t / PR, bpc = (t % PR) * 16;
    // the fo rows of the next slab travel from global memory to registers while the MFMAs of the current one run
    u32x4 rfo[NP][BJ];
    auto fetch_fo = [&](int q0) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int q = q0 + brow + RPP * j;
            const int y = ty0 + q / HT - R, x = tx0 + q % HT - R;
            const int voff = (q < NQ && y >= 0 && y < h && x >= 0 && x < w) ? (y * w + x) * (C * 2) + bpc : (int)0x80000000;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                rfo[p][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rf[p], voff, n0 * 2, 0));
        }
    };
    fetch_fo(0);

    const float* gb = g + (size_t)b * h * w * cstride;
    if ((cstride & 3) == 0 && (GS & 3) == 0) {
        // 16-byte loads, all of a thread's in flight at once (round 6: the scalar loop below was a chain of 31 dependent
        // 4-byte load -> LDS store rounds per thread: a fifth of the launch)
        constexpr int G4 = GS / 4, NV = (64 * G4 + 255) / 256;
        f32x4 gv[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e4 = t + 256 * k, p = e4 / G4, o4 = e4 - p * G4;
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            gv[k] = (e4 < 64 * G4 && y < h && x < w) ? *reinterpret_cast<const f32x4*>(gb + ((size_t)y * w + x) * cstride + 4 * o4)
                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e4 = t + 256 * k, p = e4 / G4, o4 = e4 - p * G4;
            if (e4 < 64 * G4) {
                f32x4 v = gv[k] * inv_sqrt_c;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * o4 + q >= KK) v[q] = 0.f;
                *reinterpret_cast<f32x4*>(gs + p * GS + 4 * o4) = v;
            }
        }
    } else {
        for (int e = t; e < 64 * GS; e += 256) {
            const int p = e / GS, o = e - p * GS;
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            gs[e] = (o < KK && y < h && x < w) ? gb[((size_t)y * w + x) * cstride + o] * inv_sqrt_c : 0.f;
        }
    }
    // NP <= 2: the window gradients have no a-priori bound, but the tile is right here: a block-local power-of-two scale
    // from its own maximum (exact), the fo planes carry their producer's tensor scale; the result takes both back
    float g_inv = 1.f, out_scale = 1.f;
    if (NP <= 2) {
        __shared__ float red4[4];
        __syncthreads();
        float m = 0.f;
        for (int e = t; e < 64 * GS; e += 256) m = fmaxf(m, fabsf(gs[e]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) red4[wv] = m;
        __syncthreads();
        const float sg = pow2_scale(fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3])));
        g_inv = 1.f / sg;
        out_scale = sg * *s_fo;
    }
    // G slab piece of this thread: pixel row ap = t >> 2 (py, px), k-group t & 3
    const int ap = t >> 2, apy = ap >> 3, apx = ap & 7, akg = t & 3;
    const int adst = ap * 64 + 16 * (akg ^ ((ap >> 2) & 3));

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int gq = lane >> 4, L = lane & 15;
    const int b_lane = (8 * (gq >> 1) + (L >> 2)) * RSB + (wv * 32 + 16 * (gq & 1) + 4 * (L & 3)) * 2;
    for (int ch = 0; ch < NCH; ++ch) {
        const int q0 = ch * 32;
        __syncthreads();      // previous slab consumed (and gs complete on the first pass)
        if (!(abl & 1)) {   // G[p][q0 + 8 akg .. +7]
            float v[8];
            const int qs = q0 + akg * 8;
            int qy = qs / HT, qx = qs - qy * HT;              // rows beyond NQ fall outside the window by themselves (c >= K)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = SIGN > 0 ? qy - apy : apy + 2 * R - qy;
                const int a = SIGN > 0 ? qx - apx : apx + 2 * R - qx;
                const bool in = (unsigned)c < (unsigned)K && (unsigned)a < (unsigned)K;
                v[i] = in ? gs[ap * GS + a * K + c] * g_inv : 0.f;
                if (++qx == HT) { qx = 0; ++qy; }
            }
            u32x4 o[NP];
            split8<NP>(v, o);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(asm_ + p * A_BYTES + adst) = o[p];
        }
        if (!(abl & 2)) {
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(bsm + p * B_BYTES + (brow + RPP * j) * RSB + bpc) = rfo[p][j];
        if (ch + 1 < NCH) fetch_fo(q0 + 32);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = 16 * ((2 * s + hh) ^ sw);
            bf16x8 af[NP][2], bfr[NP][NJ];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) af[p][i] = *reinterpret_cast<const bf16x8*>(asm_ + p * A_BYTES + (i * 32 + li) * 64 + koff);
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn) {
                    const unsigned char* bp = bsm + p * B_BYTES + b_lane + 16 * s * RSB + jn * 256;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)bp);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(bp + 4 * RSB));
                    bfr[p][jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
            constexpr int NPROD = nprod<NP>();
            if (abl & 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn) acc[i][jn][0] += (float)af[0][i][0] + (float)bfr[0][jn][0] + (float)af[NP - 1][i][1] + (float)bfr[NP - 1][jn][1];
                continue;
            }
#pragma unroll
            for (int q = 0; q < NPROD; ++q) {
                const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = mma16<NP>(af[pa][i], bfr[pb][jn], acc[i][jn]);
            }
        }
    }
    float* dfb = df + (size_t)b * h * w * C;
    const float* addb = add ? add + (size_t)b * h * w * C : nullptr;    // a second gradient of the same tensor, summed here
    const int col = n0 + wv * 32 + li;
    if (abl & 8) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][jn][r];
        if (sacc == 1.2345f) dfb[0] = sacc;
        return;
    }
    // the 64 x BN tile through LDS so that a thread stores (and, for `add`, loads) 16 bytes of one pixel's channels: 16 stores per
    // thread instead of 64 four-byte ones (the epilogue was a fifth of the launch; conv_epilogue.h made the same change in round 2)
    constexpr int TS = BN + 4;                                          // padded row (floats)
    float* ot = reinterpret_cast<float*>(smem);
    __syncthreads();                                                    // the last slab's fragments are read
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) ot[pl * TS + wv * 32 + li + jn * 128] = NP <= 2 ? acc[i][jn][r] * out_scale : acc[i][jn][r];
        }
    __syncthreads();
    constexpr int C4 = BN / 4;                                          // 16-byte pieces per pixel
#pragma unroll
    for (int k = 0; k < 64 * C4 / 256; ++k) {
        const int idx = t + 256 * k, pl = idx / C4, c4 = idx - pl * C4;
        const int y = ty0 + (pl >> 3), x = tx0 + (pl & 7);
        if (y < h && x < w) {
            const size_t o = ((size_t)y * w + x) * C + n0 + 4 * c4;
            f32x4 v = *reinterpret_cast<const f32x4*>(ot + pl * TS + 4 * c4);
            if (addb) v += *reinterpret_cast<const f32x4*>(addb + o);
            *reinterpret_cast<f32x4*>(dfb + o) = v;
        }
    }
    (void)col;
}

int launch_corr_transpose(const float* dcorr, float* dct, int B, int h, int w, int cstride, int r, hipStream_t s);   // corr.hip

}  // namespace rpnet

extern "C" int rpnet_local_corr_split_fwd(const void* f1s, const void* f2s, float* corr, int B, int h, int w, int C, int r,
                                          int cstride, int planes, const float* scale1, const float* scale2, float* out_absmax,
                                          void* corr_planes, const float* corr_plane_scale, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f1s && f2s && corr, RPNET_ERR_ARG, "local_corr_split_fwd: null pointer");
    RPNET_REQUIRE(r == 5 && C % 32 == 0 && cstride >= 121 && cstride <= 160 &&
                      (planes == 3 || ((planes == 2 || planes == 1) && scale1 && scale2)),
                  RPNET_ERR_SHAPE, "local_corr_split_fwd: r=%d (5) C=%d cstride=%d planes=%d (1, 2 = fp16 planes + their scales)", r, C,
                  cstride, planes);
    RPNET_REQUIRE((size_t)h * w * C * 2 < (1UL << 31), RPNET_ERR_SHAPE, "local_corr_split_fwd: image too large");
    const int tiles = cdiv(h, 8) * cdiv(w, 8);
    const float isc = 1.0f / sqrtf((float)C);
    RPNET_REQUIRE(!corr_planes || (planes <= 2 && corr_plane_scale), RPNET_ERR_ARG,
                  "local_corr_split_fwd: output planes are fp16 planes (planes 1 / 2) of corr / *corr_plane_scale");
    const unsigned short* a = (const unsigned short*)f1s;
    const unsigned short* b2 = (const unsigned short*)f2s;
    unsigned short* cp = (unsigned short*)corr_planes;
    if (planes == 3)
        hipLaunchKernelGGL((local_corr_mfma_fwd_kernel<5, 3>), dim3(tiles, B), dim3(256), 0, (hipStream_t)stream, a, b2, corr, B, h, w, C, cstride, isc,
                           (const float*)nullptr, (const float*)nullptr, out_absmax, (unsigned short*)nullptr, (const float*)nullptr);
    else if (planes == 2)
        hipLaunchKernelGGL((local_corr_mfma_fwd_kernel<5, 2>), dim3(tiles, B), dim3(256), 0, (hipStream_t)stream, a, b2, corr, B, h, w, C,
                           cstride, isc, scale1, scale2, out_absmax, cp, corr_plane_scale);
    else
        hipLaunchKernelGGL((local_corr_mfma_fwd_kernel<5, 1>), dim3(tiles, B), dim3(256), 0, (hipStream_t)stream, a, b2, corr, B, h, w, C,
                           cstride, isc, scale1, scale2, out_absmax, cp, corr_plane_scale);
    return check_launch("local_corr_split_fwd");
}

extern "C" int rpnet_local_corr_split_bwd(const void* f1s, const void* f2s, const float* dcorr, float* df1, float* df2, int B,
                                          int h, int w, int C, int r, int cstride, int planes, const float* scale1,
                                          const float* scale2, const float* df1_add, void* workspace, size_t workspace_bytes,
                                          rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(f1s && f2s && dcorr && df1 && df2 && workspace, RPNET_ERR_ARG, "local_corr_split_bwd: null pointer");
    RPNET_REQUIRE(r == 5 && C % 128 == 0 && cstride >= 121 && (planes == 3 || ((planes == 2 || planes == 1) && scale1 && scale2)), RPNET_ERR_SHAPE,
                  "local_corr_split_bwd: r=%d (5) C=%d (multiple of 128) cstride=%d planes=%d", r, C, cstride, planes);
    RPNET_REQUIRE(workspace_bytes >= rpnet_local_corr_bwd_workspace_bytes(B, h, w, cstride), RPNET_ERR_WORKSPACE,
                  "local_corr_split_bwd: workspace too small");
    RPNET_REQUIRE((size_t)h * w * C * 2 < (1UL << 31), RPNET_ERR_SHAPE, "local_corr_split_bwd: image too large");
    hipStream_t s = (hipStream_t)stream;
    float* dct = (float*)workspace;
    const float isc = 1.0f / sqrtf((float)C);
    const int tiles = cdiv(h, 8) * cdiv(w, 8);
    const unsigned short* a = (const unsigned short*)f1s;
    const unsigned short* b2 = (const unsigned short*)f2s;
    // two planes or one: a block covers 256 channels when C allows (the G slab is built once for all of them)
    const bool wide = planes <= 2 && C % 256 == 0;
    static const int xcd_order = getenv("RPNET_CORR_BWD_XCD") ? atoi(getenv("RPNET_CORR_BWD_XCD")) : 0;
    const dim3 grid(tiles, wide ? C / 256 : C / 128, B);
#define RPNET_CORR_BWD(NP_, SIGN_, NJ_, ...) hipLaunchKernelGGL((local_corr_mfma_bwd_kernel<5, NP_, SIGN_, NJ_>), grid, dim3(256), 0, s, __VA_ARGS__)
#define RPNET_CORR_BWD_PASS(SIGN_, G_, FO_, DF_, SFO_, ADD_)                                                       \
    do {                                                                                                            \
        if (planes == 3) RPNET_CORR_BWD(3, SIGN_, 1, G_, FO_, DF_, B, h, w, C, cstride, isc, (const float*)nullptr, ADD_, xcd_order); \
        else if (planes == 2 && wide) RPNET_CORR_BWD(2, SIGN_, 2, G_, FO_, DF_, B, h, w, C, cstride, isc, SFO_, ADD_, xcd_order);     \
        else if (planes == 2) RPNET_CORR_BWD(2, SIGN_, 1, G_, FO_, DF_, B, h, w, C, cstride, isc, SFO_, ADD_, xcd_order);             \
        else if (wide) RPNET_CORR_BWD(1, SIGN_, 2, G_, FO_, DF_, B, h, w, C, cstride, isc, SFO_, ADD_, xcd_order);                    \
        else RPNET_CORR_BWD(1, SIGN_, 1, G_, FO_, DF_, B, h, w, C, cstride, isc, SFO_, ADD_, xcd_order);                              \
    } while (0)
    RPNET_CORR_BWD_PASS(1, dcorr, b2, df1, scale2, df1_add);
    if (int rc = launch_corr_transpose(dcorr, dct, B, h, w, cstride, r, s)) return rc;
    RPNET_CORR_BWD_PASS(-1, (const float*)dct, a, df2, scale1, (const float*)nullptr);
#undef RPNET_CORR_BWD_PASS
#undef RPNET_CORR_BWD
    return check_launch("local_corr_split_bwd");
}
