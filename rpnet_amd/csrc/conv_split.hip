// fp32-accurate 3x3 convolution on the bf16 matrix pipe of gfx950 (v_mfma_f32_32x32x16_bf16,
// 16x the rate of v_mfma_f32_32x32x2_f32).  Every fp32 operand is carried as NP bf16 planes
//   x = h + m (+ l),   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)        (exact for NP = 3)
// and x*y is the sum of the partial products of the planes that matter, accumulated in fp32:
//   NP = 3:  hh + hm + mh + hl + lh + mm      6 MFMAs, dropped terms <= 2^-23 |x*y|  (fp32 round-off level)
//   NP = 2:  FP16 planes of x / s (power-of-two scale from a rigorous bound, split_bf16.h):  hh + hl + lh
//                                             3 MFMAs (v_mfma_f32_32x32x16_f16), dropped term <= 2^-22 |x*y|
// i.e. 16/6 = 2.7x (NP = 3) / 16/3 = 5.3x (NP = 2) the fp32 matrix rate for the same algorithmic FLOPs.
//
// The split is done ONCE by whoever produces a tensor (rpnet_split_bf16; weights by
// rpnet_pack_conv_weight_split), never inside the GEMM: the implicit-GEMM kernel below only moves
// 16-byte groups of 8 bf16 from HBM/L2 to LDS to the MFMA operand registers.
//
// GEMM view as conv_igemm.hip (M = pixels, N = Cout, K = taps*Cin, 32 channels per K-step).
// LDS image of a plane of a tile: [row][64 B = 4 k-groups of 8 bf16], the position of k-group g inside
// the row XOR-swizzled with (row >> 2) & 3: both the 16-byte stores (4 lanes per row, rows consecutive)
// and the b128 fragment reads (16-lane groups = 16 rows, one k-group) are bank-conflict free.
// A fragment of the 32x32x16 MFMA = rows li, k-group 2*s + h (lane = li + 32 h): ONE ds_read_b128.
#include <algorithm>

#include "conv_epilogue.h"
#include "split_bf16.h"

namespace rpnet {

// __builtin_amdgcn_iglp_opt(0) (the compiler's MFMA / DS interleaving pass) on the tap loop of the patch kernel: measured
// +2.8 % with two planes (323-326 -> 334 TF over the step's conv launches), -4 % with three; neutral on the 4-wave
// patch kernel and -3 % on the weight gradient, where it is not applied.
constexpr bool IGLP = true;

template <int NP>
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const int mode, unsigned short* __restrict__ out,
                                                          const size_t n8, const FastDiv fC, const size_t plane_elems) {
    RPNET_PASS_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (mode) {
            float s = scale[fC.div((unsigned)i)];
            if (mode == 2) s = 1.f - s;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= s;
        }
        u32x4 o[NP];
        split8<NP>(v, o);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(out + p * plane_elems + i * 8) = o[p];
    }
}

// fp16 planes of x * f(mask) / s with the tensor scale s = max(*s_a, *s_b) (device scalars; s_b optional): the operand
// of a convolution whose input is a BatchNorm output (scale from rpnet_bn_relu), pooled / masked / concatenated
template <int NP>
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, const float* __restrict__ mask, const int mode,
                                                         const float* __restrict__ s_a, const float* __restrict__ s_b,
                                                         float* __restrict__ s_out, unsigned short* __restrict__ out,
                                                         const size_t n8, const FastDiv fC, const size_t plane_elems,
                                                         const int a_is_bound) {
    RPNET_PASS_PRIORITY();
    // a_is_bound: *s_a is a bound of |x| (rpnet_conv_desc.out_absmax), not yet a scale
    const float sc = a_is_bound ? pow2_scale(*s_a) : fmaxf(*s_a, s_b ? *s_b : 0.f);
    if (s_out && blockIdx.x == 0 && threadIdx.x == 0) *s_out = sc;
    const float inv = 1.f / sc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (mode) {      // x * f(mask) rounded to fp32 as the reference does, then the exact power-of-two scale
            float f = mask[fC.div((unsigned)i)];
            if (mode == 2) f = 1.f - f;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= inv;
        u32x4 o[NP];
        split8<NP>(v, o);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(out + p * plane_elems + i * 8) = o[p];
    }
}

__global__ void pow2_scale_kernel(const float* __restrict__ bound, float* __restrict__ s_out) {
    RPNET_PASS_PRIORITY(); *s_out = pow2_scale(*bound); }

__global__ __launch_bounds__(256) void predict_scales_kernel(const float* __restrict__ measured, float* __restrict__ bound,
                                                              float* __restrict__ scale, const int n, const float safety,
                                                              const int check, int* __restrict__ violations) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float m = measured[i];
    if (check && m > bound[i]) atomicAdd(violations, 1);
    const float s = pow2_scale(m * safety);
    scale[i] = s;
    bound[i] = s * 32768.f;
}

// power-of-two row scales of the fp16 weight planes: t[cout] over (cin, tap), u[gathered cin row] over (cout, tap)
__device__ __forceinline__ void weight_row_scale_block(const int b, const float* __restrict__ w, float* __restrict__ t,
                                                       float* __restrict__ u, int cout, int cin_w, int taps, int off0, int split,
                                                       int off1) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float m = 0.f;
    if (b < cout) {
        const float* row = w + (size_t)b * cin_w * taps;
        for (int e = tid; e < cin_w * taps; e += 256) m = fmaxf(m, fabsf(row[e]));
    } else {
        const int cin = b - cout;
        for (int e = tid; e < cout * taps; e += 256) m = fmaxf(m, fabsf(w[((size_t)(e / taps) * cin_w + cin) * taps + e % taps]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (b < cout) t[b] = pow2_scale(m);
        else {
            const int cin = b - cout;
            u[cin < split ? off0 + cin : off1 + (cin - split)] = pow2_scale(m);
        }
    }
}

// the weights of up to RPNET_PACK_MAX layers in ONE launch (blockIdx.y = layer): a training step repacks every layer
constexpr int kPackMax = 24;
struct PackItems {
    rpnet_pack_item it[kPackMax];
};

__global__ __launch_bounds__(256) void weight_row_scale_kernel(const PackItems items) {
    const rpnet_pack_item& q = items.it[blockIdx.y];
    if ((int)blockIdx.x >= q.cout + q.cin) return;
    weight_row_scale_block(blockIdx.x, q.w, q.row_scale_wp, q.row_scale_wd, q.cout, q.cin, q.taps, q.cin_off0, q.cin_split, q.cin_off1);
}

// w [Cout][cin_w][taps] -> wp [plane][tap][Cin_g/32][Cout][32] (+ wd [plane][tapflip][Cout/32][Cin_g][32])
// GATHER: the block walks 32 PACKED rows and looks their source channel up (rows outside both ranges are zero padding) —
// for layers whose channel ranges are not multiples of 8 (the 1x1 convolution over cat([corr(121 -> 128), fm1]));
// otherwise it walks 32 source channels whose packed rows are contiguous in groups of 8.
template <int NP, bool GATHER>
__device__ __forceinline__ void pack_weight_split_block(float (*tile)[32][33], const int bx, const int by, const float* __restrict__ w,
                                                        unsigned short* __restrict__ wp, unsigned short* __restrict__ wd, int taps,
                                                        int Cin_g, int Cout, int cin_w, int off0, int split, int off1,
                                                        const float* __restrict__ t_row, const float* __restrict__ u_row) {
    // tile [tap][cin_l][cout_l]
    const int t = threadIdx.x;
    const int ci0 = bx * 32, co0 = by * 32;
    const int ncin = GATHER ? 32 : min(32, cin_w - ci0);
    if (GATHER) {
        for (int e = t; e < 32 * 32 * taps; e += 256) {
            const int col = e / (32 * taps), r = e - col * (32 * taps);
            const int c = r / taps, tap = r - c * taps;
            const int R = ci0 + c;
            int src = -1;
            if (R >= off0 && R < off0 + split) src = R - off0;
            else if (R >= off1 && R < off1 + (cin_w - split)) src = split + (R - off1);
            tile[tap][c][col] = src >= 0 ? w[((size_t)(co0 + col) * cin_w + src) * taps + tap] : 0.f;
        }
    } else {
        const int nel = ncin * taps;
        for (int e = t; e < 32 * nel; e += 256) {
            const int col = e / nel, r = e - col * nel;
            const int c = r / taps, tap = r - c * taps;
            tile[tap][c][col] = w[((size_t)(co0 + col) * cin_w + ci0) * taps + r];
        }
    }
    __syncthreads();
    const size_t plane = (size_t)taps * Cin_g * Cout;
    // wp: item (tap, cout_l, g): 8 consecutive input channels of one output channel
    for (int e = t; e < taps * 32 * 4; e += 256) {
        const int g = e & 3, cl = (e >> 2) & 31, tap = e >> 7;
        if (g * 8 < ncin) {
            const int cin = ci0 + g * 8;
            const int row = GATHER ? cin : (cin < split ? off0 + cin : off1 + (cin - split));
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = tile[tap][g * 8 + q][cl];
            if (NP <= 2) {
                const float inv = 1.f / t_row[co0 + cl];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= inv;
            }
            u32x4 o[NP];
            split8<NP>(v, o);
            const size_t dst = (((size_t)tap * (Cin_g >> 5) + (row >> 5)) * Cout + co0 + cl) * 32 + (row & 31);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(wp + p * plane + dst) = o[p];
        }
    }
    if (wd) {
        // wd: GEMM K = cout, N = gathered cin: item (tap, cin_l, g): 8 consecutive output channels
        for (int e = t; e < taps * 32 * 4; e += 256) {
            const int g = e & 3, c = (e >> 2) & 31, tap = e >> 7;
            if (c < ncin) {
                const int cin = ci0 + c;
                const int row = GATHER ? cin : (cin < split ? off0 + cin : off1 + (cin - split));
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = tile[tap][c][g * 8 + q];
                if (NP <= 2) {
                    const float inv = 1.f / u_row[row];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= inv;
                }
                u32x4 o[NP];
                split8<NP>(v, o);
                const int tf = taps - 1 - tap;
                const size_t dst = (((size_t)tf * (Cout >> 5) + (co0 >> 5)) * Cin_g + row) * 32 + g * 8;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(wd + p * plane + dst) = o[p];
            }
        }
    }
}

// channel ranges that the 8-channel groups of the plain walk cannot express
__host__ __device__ static inline bool pack_needs_gather(const rpnet_pack_item& q) {
    return (q.cin % 8) || (q.cin_off0 % 8) || (q.cin_split % 8) || (q.cin_off1 % 8);
}

template <int NP>
__global__ __launch_bounds__(256) void pack_weight_split_kernel(const PackItems items) {
    __shared__ float tile[9][32][33];
    const rpnet_pack_item& q = items.it[blockIdx.y];
    const bool gather = pack_needs_gather(q);
    const int tx = gather ? q.cin_pad / 32 : (q.cin + 31) / 32, ntiles = tx * (q.cout / 32);
    if ((int)blockIdx.x >= ntiles) return;
    if (gather)
        pack_weight_split_block<NP, true>(tile, blockIdx.x % tx, blockIdx.x / tx, q.w, (unsigned short*)q.wp, (unsigned short*)q.wd, q.taps,
                                          q.cin_pad, q.cout, q.cin, q.cin_off0, q.cin_split, q.cin_off1, q.row_scale_wp, q.row_scale_wd);
    else
        pack_weight_split_block<NP, false>(tile, blockIdx.x % tx, blockIdx.x / tx, q.w, (unsigned short*)q.wp, (unsigned short*)q.wd, q.taps,
                                           q.cin_pad, q.cout, q.cin, q.cin_off0, q.cin_split, q.cin_off1, q.row_scale_wp, q.row_scale_wd);
}

// WGM x 2 waves; wave tile (32 WM) x (32 WN); block tile BM = 32 WGM WM, BN = 64 WN.
// DB: two LDS stages and ONE barrier per K-step (tile s+1 is written into the other stage while tile s is
// being multiplied, tile s+2 is in flight in registers); otherwise one stage and two barriers, relying on
// co-resident blocks to cover the staging.
template <int WGM, int WM, int WN, int NP, bool DB>
__global__ __launch_bounds__(WGM * 128, (DB ? 1 : (WGM == 2 ? 2 : 1))) void conv_igemm_split_kernel(
    const rpnet_conv_desc d, const int M, const int Cin, const int Cout, const int tiles_n, const int ntiles) {
    constexpr int NT = WGM * 128;                 // threads
    constexpr int BM = 32 * WGM * WM, BN = 64 * WN;
    constexpr int RP = NT / 4;                    // tile rows staged per pass (4 lanes per 64-byte row)
    constexpr int AJ = BM / RP, BJ = (BN + RP - 1) / RP;
    constexpr bool B_PART = BN < RP;              // fewer B rows than one pass: only the first threads stage B
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;       // one plane of a tile
    constexpr int STAGE = NP * (A_BYTES + B_BYTES);
    __shared__ __attribute__((aligned(16))) unsigned char smem[cmax((DB ? 2 : 1) * STAGE, epilogue_lds_bytes<WN, WGM>())];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    const int tile = xcd_swizzle(blockIdx.x, ntiles);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const int dil = d.dilation > 1 ? d.dilation : 1;

    // staging: thread t moves k-group (t & 3) of tile rows (t >> 2) + RP j — four lanes cover the 64
    // contiguous bytes of one row of one plane
    const int srow = t >> 2, skg = t & 3;
    int rn[AJ], ry[AJ], rx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + srow + RP * j;
        if (m < M) {
            const int n = m / HW, rem = m - n * HW;
            rn[j] = n;
            ry[j] = rem / W;
            rx[j] = rem - ry[j] * W;
        } else {
            rn[j] = -1; ry[j] = 0; rx[j] = 0;
        }
    }
    const int kchunks = Cin >> 5;
    const int nsteps = d.taps * kchunks;
    const int rot = (int)(blockIdx.x % (unsigned)kchunks);
    int l_tap = 0, l_c0 = rot << 5, l_kc = 0;

    const size_t plane0 = (size_t)d.N * Hs * Ws * d.C0, plane1 = (size_t)d.N * Hs * Ws * d.C1;
    const size_t planew = (size_t)d.taps * Cin * Cout;
    const unsigned short* x0 = reinterpret_cast<const unsigned short*>(d.x0);
    const unsigned short* x1 = reinterpret_cast<const unsigned short*>(d.x1 ? d.x1 : d.x0);
    const unsigned short* wq = reinterpret_cast<const unsigned short*>(d.w);
    __amdgpu_buffer_rsrc_t rs0[NP], rs1[NP], rsw[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rs0[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x0 + p * plane0), (short)0,
                                                   (int)(plane0 * 2), 0x00020000);
        rs1[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x1 + p * (d.x1 ? plane1 : plane0)), (short)0,
                                                   (int)((d.x1 ? plane1 : plane0) * 2), 0x00020000);
        rsw[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq + p * planew), (short)0,
                                                   (int)(planew * 2), 0x00020000);
    }
    int roff[AJ];
    auto tap_setup = [&](int tap) {
        int ky = 0, kx = 0;
        if (d.taps == 9) { ky = (tap / 3 - 1) * dil; kx = (tap - (tap / 3) * 3 - 1) * dil; }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int iy = ry[j] + ky, ix = rx[j] + kx;
            const bool inb = rn[j] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            roff[j] = inb ? (rn[j] * Hs + (iy >> ups)) * Ws + (ix >> ups) : -1;
        }
    };
    tap_setup(0);
    const bool stage_b = !B_PART || srow < BN;
    int wvoff[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) wvoff[j] = stage_b ? (srow + RP * j) * 64 + skg * 16 : -1;   // -1: out of range, zeros

    u32x4 ra[NP][AJ], rb[NP][BJ];
    auto load_tile = [&]() {
        const bool first = l_c0 < d.C0;
        const int Cs = first ? d.C0 : d.C1;
        const int cc = first ? l_c0 : l_c0 - d.C0;
        const int soff = cc * 2;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int voff = roff[j] * (Cs * 2) + skg * 16;   // roff == -1 -> beyond num_records -> zeros
#pragma unroll
            for (int p = 0; p < NP; ++p)
                ra[p][j] = first ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0[p], voff, soff, 0))
                                 : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1[p], voff, soff, 0));
        }
        const int wsoff = ((l_tap * kchunks + (l_c0 >> 5)) * Cout + n0) * 64;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                rb[p][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw[p], wvoff[j], wsoff, 0));
        l_c0 += 32;
        if (l_c0 == Cin) l_c0 = 0;
        if (++l_kc == kchunks) {
            l_kc = 0;
            if (++l_tap < d.taps) tap_setup(l_tap);
        }
    };
    // RP j rows further: same swizzle, (RP j >> 2) & 3 == 0
    const int sdst = srow * 64 + 16 * (skg ^ ((srow >> 2) & 3));
    auto store_tile = [&](unsigned char* st) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int j = 0; j < AJ; ++j)
                *reinterpret_cast<u32x4*>(st + p * A_BYTES + j * (RP * 64) + sdst) = ra[p][j];
            if (stage_b) {
#pragma unroll
                for (int j = 0; j < BJ; ++j)
                    *reinterpret_cast<u32x4*>(st + NP * A_BYTES + p * B_BYTES + j * (RP * 64) + sdst) = rb[p][j];
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int a_row = (wm * WM * 32 + li) * 64, b_row = (wn * WN * 32 + li) * 64;
    auto mma_slice = [&](const unsigned char* st, int s) {
        const int koff = 16 * ((2 * s + h) ^ sw);
        bf16x8 af[NP][WM], bfr[NP][WN];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
                af[p][i] = *reinterpret_cast<const bf16x8*>(st + p * A_BYTES + a_row + i * 2048 + koff);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bfr[p][j] = *reinterpret_cast<const bf16x8*>(st + NP * A_BYTES + p * B_BYTES + b_row + j * 2048 + koff);
        }
        // smallest partial products first
        constexpr int NPROD = nprod<NP>();
#pragma unroll
        for (int q = 0; q < NPROD; ++q) {
            const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = mma16<NP>(af[pa][i], bfr[pb][j], acc[i][j]);
        }
    };

    // two sources with DIFFERENT fp16 tensor scales (acc_scale_x1: the 1x1 convolution over cat([corr, fm1])): the
    // accumulator is kept in units of the source of the tile being multiplied — an exact power-of-two rescale whenever
    // the chunk sequence crosses from one source to the other — and ends in units of source 0 (acc_scale_x)
    const bool two_scales = d.acc_scale_x1 != nullptr && d.C1 > 0;
    const float r01 = two_scales ? *d.acc_scale_x / *d.acc_scale_x1 : 1.f;      // units of s0 -> units of s1
    int cur_src = 0;
    auto enter_tile = [&](int ks) {
        if (!two_scales) return;
        int c = rot + ks;
        c -= (c / kchunks) * kchunks;
        const int src = (c << 5) < d.C0 ? 0 : 1;
        if (src != cur_src) {
            const float f = src ? r01 : 1.f / r01;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
            cur_src = src;
        }
    };
    if (DB) {
        load_tile();
        store_tile(smem);
        if (nsteps > 1) load_tile();
        __syncthreads();
        for (int ks = 0; ks < nsteps; ++ks) {
            enter_tile(ks);
            unsigned char* cur = smem + (ks & 1) * STAGE;
            unsigned char* nxt = smem + ((ks + 1) & 1) * STAGE;
            if (ks + 1 < nsteps) {
                store_tile(nxt);                       // tile ks+1 (fetched during step ks-1)
                if (ks + 2 < nsteps) load_tile();      // tile ks+2: a whole K-step to land
            }
            mma_slice(cur, 0);
            mma_slice(cur, 1);
            __syncthreads();
        }
    } else {
        load_tile();
        store_tile(smem);
        __syncthreads();
        for (int ks = 0; ks < nsteps; ++ks) {
            enter_tile(ks);
            const bool more = ks + 1 < nsteps;
            if (more) load_tile();
            mma_slice(smem, 0);
            mma_slice(smem, 1);
            __syncthreads();
            if (more) store_tile(smem);
            __syncthreads();
        }
    }
    if (two_scales && cur_src) {
        const float f = 1.f / r01;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
    }
    conv_epilogue<WM, WN, WGM, LinearRows, NP == 1>(d, acc, LinearRows{m0, M}, M, Cout, HW, n0, tm, wm, wn, li, h, smem);
}

// 256 x 128 tile whose 256 output pixels are a (256 / TW) x TW PATCH of one image, 8 waves (4 x 2), 3x3 taps only.
// The input halo of the patch ((TH + 2) x (TW + 2) pixels x 32 channels x NP planes) is staged in LDS ONCE per
// channel chunk and serves all nine taps — a tap is an address offset of the A-fragment reads — so per K-step
// only the weight slab moves (24 KB instead of 72 KB for this tile): the L1 -> LDS path stops being the limiter.
// Weights are double-buffered (one barrier per K-step); the next chunk's halo waits in registers during the nine
// taps of the current one.  A 32-pixel MFMA tile row is TW = 32 consecutive pixels of a patch row (or two
// 16-pixel rows), so the (row >> 2) & 3 k-group swizzle keeps the fragment reads conflict-free.
template <int TW, int NP, int WN>
__global__ __launch_bounds__(512, 1) void conv_igemm_split_halo_kernel(const rpnet_conv_desc d, const int Cin, const int Cout,
                                                                        const int tiles_n, const int ntiles) {
    constexpr int NT = 512, BM = 256, BN = 64 * WN, TH = BM / TW, PW = TW + 2, HALO = (TH + 2) * PW;
    constexpr int A_BYTES = HALO * 64, B_BYTES = BN * 64;
    constexpr int HJ = (HALO * 4 + NT - 1) / NT;
    constexpr int WM = 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[cmax(NP * A_BYTES + 2 * NP * B_BYTES, epilogue_lds_bytes<WN, 4>())];
    unsigned char* const bsm = smem + NP * A_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    const int tile = xcd_swizzle(blockIdx.x, ntiles);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int n0 = tn * BN;
    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const int pxn = W / TW, ppi = (H / TH) * pxn;
    const int n = tm / ppi, prem = tm - n * ppi;
    const int y0 = (prem / pxn) * TH, x0 = (prem % pxn) * TW;

    // halo pieces this thread stages: halo row hr = e >> 2, k-group e & 3
    int hoff[HJ], hdst[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int e = t + NT * j;
        const int hr = e >> 2, kg = e & 3;
        const int hy = hr / PW, hx = hr - hy * PW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool inb = e < HALO * 4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        hoff[j] = inb ? (n * Hs + (iy >> ups)) * Ws + (ix >> ups) : -1;
        hdst[j] = e < HALO * 4 ? hr * 64 + 16 * (kg ^ ((hr >> 2) & 3)) : -1;
    }
    const int skg16 = (t & 3) * 16;
    const int kchunks = Cin >> 5;
    const int nsteps = 9 * kchunks;
    const int rot = (int)(blockIdx.x % (unsigned)kchunks);
    auto chunk_c0 = [&](int ci) { int c = rot + ci; if (c >= kchunks) c -= kchunks; return c << 5; };

    const size_t plane0 = (size_t)d.N * Hs * Ws * d.C0, plane1 = (size_t)d.N * Hs * Ws * d.C1;
    const size_t planew = (size_t)9 * Cin * Cout;
    const unsigned short* x0p = reinterpret_cast<const unsigned short*>(d.x0);
    const unsigned short* x1p = reinterpret_cast<const unsigned short*>(d.x1 ? d.x1 : d.x0);
    const unsigned short* wq = reinterpret_cast<const unsigned short*>(d.w);
    __amdgpu_buffer_rsrc_t rs0[NP], rs1[NP], rsw[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rs0[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x0p + p * plane0), (short)0,
                                                   (int)(plane0 * 2), 0x00020000);
        rs1[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x1p + p * (d.x1 ? plane1 : plane0)), (short)0,
                                                   (int)((d.x1 ? plane1 : plane0) * 2), 0x00020000);
        rsw[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq + p * planew), (short)0,
                                                   (int)(planew * 2), 0x00020000);
    }
    u32x4 ha[NP][HJ], rb[NP];
    auto load_halo = [&](int c0) {
        const bool first = c0 < d.C0;
        const int Cs = first ? d.C0 : d.C1;
        const int soff = (first ? c0 : c0 - d.C0) * 2;
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int voff = hoff[j] * (Cs * 2) + skg16;     // -1 -> beyond num_records -> zeros (the padding)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                ha[p][j] = first ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0[p], voff, soff, 0))
                                 : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1[p], voff, soff, 0));
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < HJ; ++j)
                if (hdst[j] >= 0) *reinterpret_cast<u32x4*>(smem + p * A_BYTES + hdst[j]) = ha[p][j];
    };
    const int brow = t >> 2;
    const bool stage_b = brow < BN;              // BN = 64: the first four waves stage the weights
    const int wvoff = stage_b ? brow * 64 + skg16 : -1;
    const int bdst = brow * 64 + 16 * ((t & 3) ^ ((brow >> 2) & 3));
    auto load_b = [&](int tap, int c0) {
        const int wsoff = ((tap * kchunks + (c0 >> 5)) * Cout + n0) * 64;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            rb[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw[p], wvoff, wsoff, 0));
    };
    auto store_b = [&](int stage) {
        if (stage_b) {
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(bsm + (stage * NP + p) * B_BYTES + bdst) = rb[p];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int hr00[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mloc = (wm * WM + i) * 32 + li;
        hr00[i] = (mloc / TW) * PW + (mloc % TW);
    }
    const int sw = (li >> 2) & 3;
    const int b_row = (wn * WN * 32 + li) * 64;
    auto mma_tap = [&](int tap, int stage) {
        const int ky = tap / 3, kx = tap - ky * 3;
        int arow[WM], asw[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int hr = hr00[i] + ky * PW + kx;
            arow[i] = hr * 64;
            asw[i] = (hr >> 2) & 3;
        }
        const unsigned char* bst = bsm + stage * NP * B_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kg = 2 * s + h;
            bf16x8 af[NP][WM], bfr[NP][WN];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    af[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * A_BYTES + arow[i] + 16 * (kg ^ asw[i]));
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    bfr[p][j] = *reinterpret_cast<const bf16x8*>(bst + p * B_BYTES + b_row + j * 2048 + 16 * (kg ^ sw));
            }
            constexpr int NPROD = nprod<NP>();
#pragma unroll
            for (int q = 0; q < NPROD; ++q) {
                const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = mma16<NP>(af[pa][i], bfr[pb][j], acc[i][j]);
            }
            if (NP <= 2 && IGLP) __builtin_amdgcn_iglp_opt(0);
        }
    };

    load_halo(chunk_c0(0));
    store_halo();
    load_b(0, chunk_c0(0));
    store_b(0);
    load_b(1, chunk_c0(0));        // nsteps >= 9
    __syncthreads();
    int ci = 0, tap = 0;
    for (int ks = 0; ks < nsteps; ++ks) {
        if (ks + 1 < nsteps) {
            store_b((ks + 1) & 1);                       // weights of step ks + 1 (fetched during step ks - 1)
            if (ks + 2 < nsteps) {
                int t2 = tap + 2, c2 = ci;
                if (t2 >= 9) { t2 -= 9; ++c2; }
                load_b(t2, chunk_c0(c2));                // a whole K-step to land
            }
        }
        const bool next_chunk = ci + 1 < kchunks;
        if (tap == 0 && next_chunk) load_halo(chunk_c0(ci + 1));   // waits in registers for nine taps
        mma_tap(tap, ks & 1);
        __syncthreads();
        if (++tap == 9) {
            tap = 0;
            ++ci;
            if (next_chunk) {
                store_halo();
                __syncthreads();
            }
        }
    }
    conv_epilogue<WM, WN, 4, PatchRows<TW>, NP == 1>(d, acc, PatchRows<TW>{(n * H + y0) * W + x0, W}, d.N * HW, Cout, HW, n0, tm, wm, wn,
                                            li, h, smem);
}

// WMT = 4: the same kernel on 256-pixel patches — a 4-wave block then computes what the 8-wave kernel above does (wave
// tile 128 x 64: 12 fragment reads per 24 MFMAs instead of 8 per 12), two such blocks per CU cover each other's staging,
// barriers, prologue and epilogue.
template <int TW, int NP, int WN, int WMT = 2>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_halo4_kernel(const rpnet_conv_desc d, const int Cin, const int Cout,
                                                                        const int tiles_n, const int ntiles) {
    constexpr int NT = 256, BM = 64 * WMT, BN = 64 * WN, TH = BM / TW, PW = TW + 2, HALO = (TH + 2) * PW;
    constexpr int A_BYTES = HALO * 64, B_BYTES = BN * 64;
    constexpr int HJ = (HALO * 4 + NT - 1) / NT;
    constexpr int WM = WMT;
    constexpr int BJ = (BN * 4 + NT - 1) / NT;                  // weight pieces per thread and plane
    __shared__ __attribute__((aligned(16))) unsigned char smem[cmax(NP * A_BYTES + NP * B_BYTES, epilogue_lds_bytes<WN, 2>())];   // one weight stage
    unsigned char* const bsm = smem + NP * A_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    const int tile = xcd_swizzle(blockIdx.x, ntiles);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int n0 = tn * BN;
    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const int pxn = W / TW, ppi = (H / TH) * pxn;
    const int n = tm / ppi, prem = tm - n * ppi;
    const int y0 = (prem / pxn) * TH, x0 = (prem % pxn) * TW;

    // halo pieces this thread stages: halo row hr = e >> 2, k-group e & 3
    int hoff[HJ], hdst[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int e = t + NT * j;
        const int hr = e >> 2, kg = e & 3;
        const int hy = hr / PW, hx = hr - hy * PW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool inb = e < HALO * 4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        hoff[j] = inb ? (n * Hs + (iy >> ups)) * Ws + (ix >> ups) : -1;
        hdst[j] = e < HALO * 4 ? hr * 64 + 16 * (kg ^ ((hr >> 2) & 3)) : -1;
    }
    const int skg16 = (t & 3) * 16;
    const int kchunks = Cin >> 5;
    const int nsteps = 9 * kchunks;
    const int rot = (int)(blockIdx.x % (unsigned)kchunks);
    auto chunk_c0 = [&](int ci) { int c = rot + ci; if (c >= kchunks) c -= kchunks; return c << 5; };

    const size_t plane0 = (size_t)d.N * Hs * Ws * d.C0, plane1 = (size_t)d.N * Hs * Ws * d.C1;
    const size_t planew = (size_t)9 * Cin * Cout;
    const unsigned short* x0p = reinterpret_cast<const unsigned short*>(d.x0);
    const unsigned short* x1p = reinterpret_cast<const unsigned short*>(d.x1 ? d.x1 : d.x0);
    const unsigned short* wq = reinterpret_cast<const unsigned short*>(d.w);
    __amdgpu_buffer_rsrc_t rs0[NP], rs1[NP], rsw[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rs0[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x0p + p * plane0), (short)0,
                                                   (int)(plane0 * 2), 0x00020000);
        rs1[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x1p + p * (d.x1 ? plane1 : plane0)), (short)0,
                                                   (int)((d.x1 ? plane1 : plane0) * 2), 0x00020000);
        rsw[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq + p * planew), (short)0,
                                                   (int)(planew * 2), 0x00020000);
    }
    u32x4 ha[NP][HJ], rb[NP][BJ];
    auto load_halo = [&](int c0) {
        const bool first = c0 < d.C0;
        const int Cs = first ? d.C0 : d.C1;
        const int soff = (first ? c0 : c0 - d.C0) * 2;
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int voff = hoff[j] * (Cs * 2) + skg16;     // -1 -> beyond num_records -> zeros (the padding)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                ha[p][j] = first ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0[p], voff, soff, 0))
                                 : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1[p], voff, soff, 0));
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < HJ; ++j)
                if (hdst[j] >= 0) *reinterpret_cast<u32x4*>(smem + p * A_BYTES + hdst[j]) = ha[p][j];
    };
    const int brow = t >> 2;                     // + 64 j
    const bool stage_b = brow < BN;
    const int bdst = brow * 64 + 16 * ((t & 3) ^ ((brow >> 2) & 3));
    auto load_b = [&](int tap, int c0) {
        const int wsoff = ((tap * kchunks + (c0 >> 5)) * Cout + n0) * 64;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int wvoff = (brow + 64 * j) < BN ? (brow + 64 * j) * 64 + skg16 : -1;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                rb[p][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw[p], wvoff, wsoff, 0));
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            if (brow + 64 * j < BN) {
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(bsm + p * B_BYTES + j * 4096 + bdst) = rb[p][j];
            }
    };
    (void)stage_b;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int hr00[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mloc = (wm * WM + i) * 32 + li;
        hr00[i] = (mloc / TW) * PW + (mloc % TW);
    }
    const int sw = (li >> 2) & 3;
    const int b_row = (wn * WN * 32 + li) * 64;
    auto mma_tap = [&](int tap, int stage) {
        const int ky = tap / 3, kx = tap - ky * 3;
        int arow[WM], asw[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int hr = hr00[i] + ky * PW + kx;
            arow[i] = hr * 64;
            asw[i] = (hr >> 2) & 3;
        }
        const unsigned char* bst = bsm + stage * NP * B_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kg = 2 * s + h;
            bf16x8 af[NP][WM], bfr[NP][WN];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    af[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * A_BYTES + arow[i] + 16 * (kg ^ asw[i]));
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    bfr[p][j] = *reinterpret_cast<const bf16x8*>(bst + p * B_BYTES + b_row + j * 2048 + 16 * (kg ^ sw));
            }
            constexpr int NPROD = nprod<NP>();
#pragma unroll
            for (int q = 0; q < NPROD; ++q) {
                const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = mma16<NP>(af[pa][i], bfr[pb][j], acc[i][j]);
            }
        }
    };

    // single weight stage, two barriers per K-step: a second resident block of the CU covers them (63 KB of LDS
    // and <= 256 registers per block); the halo rewrite of a chunk switch shares the barrier pair.
    load_halo(chunk_c0(0));
    store_halo();
    load_b(0, chunk_c0(0));
    store_b();
    __syncthreads();
    int ci = 0, tap = 0;
    for (int ks = 0; ks < nsteps; ++ks) {
        const bool more = ks + 1 < nsteps;
        const bool next_chunk = ci + 1 < kchunks;
        if (more) {
            int t1 = tap + 1, c1 = ci;
            if (t1 == 9) { t1 = 0; ++c1; }
            load_b(t1, chunk_c0(c1));
        }
        // WMT = 2: the next chunk's halo waits in registers for nine taps.  WMT = 4 (128 accumulator registers): no room
        // for that — it is fetched at the chunk switch, its latency covered by the CU's other block
        if (WMT == 2 && tap == 0 && next_chunk) load_halo(chunk_c0(ci + 1));
        mma_tap(tap, 0);
        __syncthreads();
        if (more) store_b();
        if (tap == 8 && next_chunk) {
            if (WMT != 2) load_halo(chunk_c0(ci + 1));
            store_halo();
        }
        __syncthreads();
        if (++tap == 9) { tap = 0; ++ci; }
    }
    conv_epilogue<WM, WN, 2, PatchRows<TW>, NP == 1>(d, acc, PatchRows<TW>{(n * H + y0) * W + x0, W}, d.N * HW, Cout, HW, n0, tm, wm, wn,
                                            li, h, smem);
}

template <int TW, int WN>
static int launch_split_halo(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    const int tiles_m = M / 256, tiles_n = Cout / (64 * WN);
    const int ntiles = tiles_m * tiles_n;
    if (d->split_planes == 3)
        hipLaunchKernelGGL((conv_igemm_split_halo_kernel<TW, 3, WN>), dim3(ntiles), dim3(512), 0, s, *d, Cin, Cout, tiles_n, ntiles);
    else if (d->split_planes == 2)
        hipLaunchKernelGGL((conv_igemm_split_halo_kernel<TW, 2, WN>), dim3(ntiles), dim3(512), 0, s, *d, Cin, Cout, tiles_n, ntiles);
    else
        hipLaunchKernelGGL((conv_igemm_split_halo_kernel<TW, 1, WN>), dim3(ntiles), dim3(512), 0, s, *d, Cin, Cout, tiles_n, ntiles);
    return check_launch("conv_igemm_split_halo");
}

template <int TW, int WN, int WMT = 2>
static int launch_split_halo4(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    const int tiles_m = M / (64 * WMT), tiles_n = Cout / (64 * WN);
    const int ntiles = tiles_m * tiles_n;
    if (d->split_planes == 3) {
        if constexpr (WMT == 2)
            hipLaunchKernelGGL((conv_igemm_split_halo4_kernel<TW, 3, WN, WMT>), dim3(ntiles), dim3(256), 0, s, *d, Cin, Cout, tiles_n, ntiles);
        else {      // three planes need 89 KB of LDS per block: no second block on the CU, nothing to gain
            set_error("conv_igemm_split_halo4: the 4-wave 256-pixel form exists for fp16 planes only");
            return RPNET_ERR_ARG;
        }
    } else if (d->split_planes == 2)
        hipLaunchKernelGGL((conv_igemm_split_halo4_kernel<TW, 2, WN, WMT>), dim3(ntiles), dim3(256), 0, s, *d, Cin, Cout, tiles_n, ntiles);
    else
        hipLaunchKernelGGL((conv_igemm_split_halo4_kernel<TW, 1, WN, WMT>), dim3(ntiles), dim3(256), 0, s, *d, Cin, Cout, tiles_n, ntiles);
    return check_launch("conv_igemm_split_halo4");
}

// tile width of the halo variant: 128 output channels per block when they split that way, else 64
static int halo_bn(const rpnet_conv_desc* d, int Cout) {
    return ((Cout % 128 == 0) && (d->Co1 == 0 || d->Co0 % 128 == 0)) ? 128 : 64;
}

// halo variant usable: dense 3x3, image made of whole (256 / TW) x TW patches
static int halo_tw(const rpnet_conv_desc* d, int Cout) {
    if (d->taps != 9 || d->dilation > 1) return 0;
    if (d->W % 32 == 0 && d->H % 8 == 0) return 32;
    if (d->W % 16 == 0 && d->H % 16 == 0) return 16;
    return 0;
}

// tile variants: {WGM, WM, WN, double-buffered}
struct SplitVariant { int wgm, wm, wn, db, slots; };
static const SplitVariant kSplitVariants[] = {
    {2, 2, 2, 0, 768},    // 0: 128 x 128, 4 waves, 3 blocks per CU
    {2, 2, 1, 0, 1024},   // 1: 128 x 64
    {2, 1, 2, 0, 1024},   // 2:  64 x 128
    {2, 1, 1, 0, 1536},   // 3:  64 x 64
    {4, 2, 2, 1, 256},    // 4-6: 8-wave / two-stage forms of this kernel (213 / 205 / 168 TF on 512 -> 512 at M = 16384
    {4, 2, 2, 0, 256},    //      against 187 for variant 0): superseded by the patch kernels below, not instantiated
    {2, 2, 2, 1, 256},
    {4, 2, 2, 1, 256},    // 7: 256 x 128 on an image patch, input halo resident in LDS (conv_igemm_split_halo_kernel)
    {2, 2, 2, 0, 512},    // 8: 128 x 128 on an image patch, 4 waves, two blocks per CU (conv_igemm_split_halo4_kernel)
    {2, 2, 1, 0, 512},    // 9: 128 x 64 of the same kernel: twice the blocks for the smallest grids
    {2, 4, 2, 0, 512},    // 10: (round 2's 256 x 128 four-wave, two-blocks-per-CU form of the patch kernel: 431 spilled registers; removed in round 5)
    {2, 4, 2, 0, 256},    // 11: 256 x 128 on an image patch, four waves (one per SIMD), operands by LDS-DMA (conv_split_dma.hip)
    {2, 4, 1, 0, 256},    // 12: 256 x 64 of the same kernel (wave tile 128 x 32)
    {2, 4, 2, 0, 256},    // 13: 256 x 128 of the same kernel on ONE fp16 plane, 64 channels per K-step
    {2, 2, 1, 0, 256},    // 14: 128 x 64 of the same kernel (wave tile 64 x 32): grids that 256 x 64 tiles leave half empty
};
constexpr int kNumSplitVariants = sizeof(kSplitVariants) / sizeof(kSplitVariants[0]);

template <int WGM, int WM, int WN, bool DB>
static int launch_split(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    constexpr int BM = 32 * WGM * WM, BN = 64 * WN;
    const int tiles_m = cdiv(M, BM), tiles_n = Cout / BN;
    const int ntiles = tiles_m * tiles_n;
    if (d->split_planes == 3)
        hipLaunchKernelGGL((conv_igemm_split_kernel<WGM, WM, WN, 3, DB>), dim3(ntiles), dim3(WGM * 128), 0, s, *d, M, Cin,
                           Cout, tiles_n, ntiles);
    else if (d->split_planes == 2)
        hipLaunchKernelGGL((conv_igemm_split_kernel<WGM, WM, WN, 2, DB>), dim3(ntiles), dim3(WGM * 128), 0, s, *d, M, Cin,
                           Cout, tiles_n, ntiles);
    else
        hipLaunchKernelGGL((conv_igemm_split_kernel<WGM, WM, WN, 1, DB>), dim3(ntiles), dim3(WGM * 128), 0, s, *d, M, Cin,
                           Cout, tiles_n, ntiles);
    return check_launch("conv_igemm_split");
}

// 128-pixel patch variant: (128 / TW) x TW patches
static int halo4_tw(const rpnet_conv_desc* d) {
    if (d->taps != 9 || d->dilation > 1) return 0;
    if (d->W % 32 == 0 && d->H % 4 == 0) return 32;
    if (d->W % 16 == 0 && d->H % 8 == 0) return 16;
    return 0;
}

// conv_split_dma.hip
int conv_fwd_split_dma(const rpnet_conv_desc* d, int M, int Cin, int Cout, int tw, int wn, hipStream_t s, int parts = 1, int bm = 256);
int conv_splitk_parts(const rpnet_conv_desc* d, int M, int Cin, int Cout);

// The LDS-DMA kernel (conv_split_dma.hip) reaches every operand through ONE buffer descriptor over all its planes, with 32-bit
// byte offsets and num_records: a source's planes together, and the weight pack's, must stay below 2 GiB (64 channels at 512^2
// from 32 images of two planes on would wrap, and the buffer rule would then return zeros instead of faulting).  Operands
// beyond that take the register-staged kernels (variants 7 / 10 and below), whose addressing is 64-bit.
static bool dma_addressable(const rpnet_conv_desc* d, int Cout) {
    const size_t msrc = (size_t)d->N * (d->H >> d->upsample) * (d->W >> d->upsample);
    const size_t np = d->split_planes > 0 ? d->split_planes : 1, lim = (size_t)1 << 31;
    return np * msrc * d->C0 * 2 < lim && np * msrc * d->C1 * 2 < lim && np * 9 * (size_t)(d->C0 + d->C1) * Cout * 2 < lim;
}

// same rule as conv_igemm.hip: fewest idle block slots
int choose_tile_split(const rpnet_conv_desc* d, int M, int Cout) {
    const bool n128 = (Cout % 128 == 0) && (d->Co1 == 0 || d->Co0 % 128 == 0);
    const bool addr = dma_addressable(d, Cout);
    if (d->tune > 0) {      // tuning / test override carried by the descriptor: tile variant d->tune - 1
        const int v = (d->tune & 0xff) - 1;      // bits 8..: ablation switches of conv_split_dma.hip
        if ((v == 12 || v == 14) && d->split_planes == 2 && halo_tw(d, Cout) && addr) return v;
        if (v == 13 && d->split_planes == 1 && halo_tw(d, Cout) && halo_bn(d, Cout) == 128 && (d->C0 + d->C1) % 64 == 0 && d->C0 % 64 == 0 && addr)
            return v;
        if (((v == 7 || (v == 11 && d->split_planes == 2 && addr)) && halo_tw(d, Cout) &&
             halo_bn(d, Cout) == 128) ||
            ((v == 8 || v == 9) && halo4_tw(d)))
            return v;
        if (v >= 0 && v < 4 && (kSplitVariants[v].wn == 1 || n128)) return v;
    }
    // the halo-resident 256 x 128 kernel wins whenever its grid fills the machine (one block per CU)
    // halo-resident patch kernels: 256 x 128 with one 8-wave block per CU when that grid fills the machine, else (and
    // for 64-wide output tiles) 128-pixel patches with two 4-wave blocks per CU
    const int hbn = halo_bn(d, Cout);
    // one fp16 plane: a third of the MFMAs per staged byte, so the LDS fragment reads limit the 8-wave patch kernel; the
    // 4-wave form of it (wave tile 128 x 64, two blocks per CU) reads 25 % less per MFMA: 598 vs 531 TF on 128 -> 128 at
    // M = 262144, 767 vs 671 TF on 256 -> 256 at M = 65536 — once its grid fills both block slots of every CU
    // (round 3: where channel counts come in multiples of 64 the LDS-DMA kernel takes these layers — variant 13)
    if (d->split_planes == 1 && !(d->tune & 0x10000) && addr && hbn == 128 && halo_tw(d, Cout) && (d->C0 + d->C1) % 64 == 0 && d->C0 % 64 == 0 &&
        (long)(M / 256) * (Cout / 128) >= 192)
        return 13;
    // 256 x 128 patches, one block per CU: two fp16 planes -> the LDS-DMA kernel (conv_split_dma.hip: +14 ... 20 % over the
    // register-staged 8-wave kernel on every such layer, same bits); tune bit 16 = the round-2 policy (A/B switch)
    const bool dma = d->split_planes == 2 && !(d->tune & 0x10000) && addr;
    if (hbn == 128 && halo_tw(d, Cout) && (long)(M / 256) * (Cout / 128) >= (dma ? 192 : 224)) return dma ? 11 : 7;
    // grids too small for that: 256 x 64 tiles of the DMA kernel (twice the blocks; 419 vs 317 TF on 1024 -> 1024 at
    // M = 4096) from half a machine of blocks upwards
    if (dma && halo_tw(d, Cout) && d->C0 + d->C1 >= 128 && (long)(M / 256) * (Cout / 64) >= 128) return 12;
    // (variant 14, 128 x 64 tiles of the same kernel, is NOT chosen for exactly half a machine of those — the CRE convolutions
    // of an eval-mode call at batch 2, 256 -> 256 at M = 8192: alone it is faster (29 vs 40 us back to back, 49 vs 57 us
    // in the call's kernel trace), but the call replayed from its HIP graph got slower, 3.02 -> 3.17 ms on one box, three
    // alternations — 256 busy CUs at a lower MFMA rate per CU against 128; reachable through `tune` and tested)
    if (halo4_tw(d)) {
        // 64-wide tiles (twice the blocks) win on every grid this kernel sees — 169 vs 107 TF at M = 4096, 1024 -> 512;
        // 182 vs 160 TF at M = 16384, 256 -> 256 — until the 128-wide grid alone is two full rounds of the machine
        const long t8 = (long)(M / 128) * (Cout / hbn);
        if (hbn == 128 && t8 >= 1024) return 8;
        if ((long)(M / 128) * (Cout / 64) >= 64) return 9;
    }
    int best = -1;
    double best_fill = -1.0;
    for (int c = 0; c < 4; ++c) {
        const SplitVariant& v = kSplitVariants[c];
        if (v.wn == 2 && !n128) continue;
        const long tiles = (long)cdiv(M, 32 * v.wgm * v.wm) * (Cout / (64 * v.wn));
        const long waves = (tiles + v.slots - 1) / v.slots;
        const double fill = (double)tiles / (double)(waves * v.slots);
        if (fill > best_fill + 0.02) { best_fill = fill; best = c; }
    }
    return best;
}

int split_tile_rows(int variant, int* wave_rows) {
    const SplitVariant& v = kSplitVariants[variant];
    *wave_rows = v.wgm;
    return 32 * v.wgm * v.wm;
}

// split K (conv_split_dma.hip) when the caller lent a workspace and the launch is one of the half-empty ones
static int splitk_parts_for(const rpnet_conv_desc* d, int M, int Cin, int Cout) {
    if (!halo_tw(d, Cout) || !dma_addressable(d, Cout)) return 1;
    return conv_splitk_parts(d, M, Cin, Cout);
}

size_t conv_splitk_bytes(const rpnet_conv_desc* d, int M, int Cin, int Cout) {
    const int parts = splitk_parts_for(d, M, Cin, Cout);
    return parts > 1 ? (size_t)parts * M * Cout * sizeof(float) : 0;
}

int conv_fwd_split(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    if (d->splitk_ws) {
        const int parts = splitk_parts_for(d, M, Cin, Cout);
        if (parts > 1 && d->splitk_ws_bytes >= (size_t)parts * M * Cout * sizeof(float))
            return conv_fwd_split_dma(d, M, Cin, Cout, halo_tw(d, Cout), 1, s, parts);
    }
    switch (choose_tile_split(d, M, Cout)) {
        case 0: return launch_split<2, 2, 2, false>(d, M, Cin, Cout, s);
        case 1: return launch_split<2, 2, 1, false>(d, M, Cin, Cout, s);
        case 2: return launch_split<2, 1, 2, false>(d, M, Cin, Cout, s);
        case 3: return launch_split<2, 1, 1, false>(d, M, Cin, Cout, s);
        case 11: return conv_fwd_split_dma(d, M, Cin, Cout, halo_tw(d, Cout), 2, s);
        case 12: return conv_fwd_split_dma(d, M, Cin, Cout, halo_tw(d, Cout), 1, s);
        case 13: return conv_fwd_split_dma(d, M, Cin, Cout, halo_tw(d, Cout), 2, s);
        case 14: return conv_fwd_split_dma(d, M, Cin, Cout, halo_tw(d, Cout), 1, s, 1, 128);
        case 9:
            return halo4_tw(d) == 32 ? launch_split_halo4<32, 1>(d, M, Cin, Cout, s) : launch_split_halo4<16, 1>(d, M, Cin, Cout, s);
        case 8:
            if (halo_bn(d, Cout) == 128)
                return halo4_tw(d) == 32 ? launch_split_halo4<32, 2>(d, M, Cin, Cout, s) : launch_split_halo4<16, 2>(d, M, Cin, Cout, s);
            return halo4_tw(d) == 32 ? launch_split_halo4<32, 1>(d, M, Cin, Cout, s) : launch_split_halo4<16, 1>(d, M, Cin, Cout, s);
        default:
            if (halo_bn(d, Cout) == 128)
                return halo_tw(d, Cout) == 32 ? launch_split_halo<32, 2>(d, M, Cin, Cout, s) : launch_split_halo<16, 2>(d, M, Cin, Cout, s);
            return halo_tw(d, Cout) == 32 ? launch_split_halo<32, 1>(d, M, Cin, Cout, s) : launch_split_halo<16, 1>(d, M, Cin, Cout, s);
    }
}

}  // namespace rpnet

extern "C" int rpnet_split_bf16(const float* x, const float* scale, int scale_mode, void* out, size_t rows, int C,
                                int planes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && out && (scale_mode == 0 || scale), RPNET_ERR_ARG, "split_bf16: null pointer");
    RPNET_REQUIRE(C > 0 && C % 8 == 0 && planes == 3 && scale_mode >= 0 && scale_mode <= 2, RPNET_ERR_SHAPE,
                  "split_bf16: C=%d planes=%d (3; two planes are fp16 with a tensor scale: rpnet_split_f16) mode=%d", C, planes, scale_mode);
    if (rows == 0) return RPNET_OK;
    const size_t n8 = rows * (size_t)(C / 8);
    RPNET_REQUIRE(n8 < kIndex32, RPNET_ERR_SHAPE, "split: %zu elements do not fit the 32-bit index arithmetic", n8);
    const int grid = (int)(n8 / 256 + 1 < 16384 ? n8 / 256 + 1 : 16384);
    hipLaunchKernelGGL(split_bf16_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, scale, scale_mode,
                       (unsigned short*)out, n8, FastDiv(C / 8), rows * (size_t)C);
    return check_launch("split_bf16");
}

extern "C" int rpnet_predict_scales(const float* measured, float* bound, float* scale, int n, float safety, int check, int* violations,
                                    rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(measured && bound && scale && (!check || violations), RPNET_ERR_ARG, "predict_scales: null pointer");
    RPNET_REQUIRE(n >= 0 && safety >= 1.f, RPNET_ERR_ARG, "predict_scales: n %d safety %g", n, (double)safety);
    if (n == 0) return RPNET_OK;
    hipLaunchKernelGGL(predict_scales_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, measured, bound, scale, n, safety,
                       check, violations);
    return check_launch("predict_scales");
}

extern "C" int rpnet_pow2_scale(const float* bound, float* s_out, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(bound && s_out, RPNET_ERR_ARG, "pow2_scale: null pointer");
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, bound, s_out);
    return check_launch("pow2_scale");
}

extern "C" int rpnet_split_f16(const float* x, const float* mask, int mask_mode, const float* s_a, const float* s_b, float* s_out,
                               void* out, size_t rows, int C, int planes, int a_is_bound, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && out && s_a && (mask_mode == 0 || mask), RPNET_ERR_ARG, "split_f16: null pointer");
    RPNET_REQUIRE(C > 0 && C % 8 == 0 && mask_mode >= 0 && mask_mode <= 2 && (planes == 1 || planes == 2), RPNET_ERR_SHAPE,
                  "split_f16: C=%d mode=%d planes=%d", C, mask_mode, planes);
    if (rows == 0) return RPNET_OK;
    const size_t n8 = rows * (size_t)(C / 8);
    RPNET_REQUIRE(n8 < kIndex32, RPNET_ERR_SHAPE, "split: %zu elements do not fit the 32-bit index arithmetic", n8);
    const int grid = (int)(n8 / 256 + 1 < 16384 ? n8 / 256 + 1 : 16384);
    if (planes == 2)
        hipLaunchKernelGGL(split_f16_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mask, mask_mode, s_a, s_b, s_out,
                           (unsigned short*)out, n8, FastDiv(C / 8), rows * (size_t)C, a_is_bound);
    else
        hipLaunchKernelGGL(split_f16_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mask, mask_mode, s_a, s_b, s_out,
                           (unsigned short*)out, n8, FastDiv(C / 8), rows * (size_t)C, a_is_bound);
    return check_launch("split_f16");
}

extern "C" int rpnet_pack_conv_weights_split(const rpnet_pack_item* items, int n, int planes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(items && n >= 1 && n <= kPackMax, RPNET_ERR_ARG, "pack_conv_weights_split: %d items (1..%d)", n, kPackMax);
    RPNET_REQUIRE(planes >= 1 && planes <= 3, RPNET_ERR_SHAPE, "pack_conv_weights_split: planes %d", planes);
    PackItems pk;
    int max_rows = 0, max_tiles = 0;
    for (int i = 0; i < n; ++i) {
        const rpnet_pack_item& q = items[i];
        RPNET_REQUIRE(q.w && q.wp, RPNET_ERR_ARG, "pack_conv_weights_split: null pointer (item %d)", i);
        RPNET_REQUIRE(planes == 3 || (q.row_scale_wp && q.row_scale_wd), RPNET_ERR_ARG,
                      "pack_conv_weights_split: fp16 planes (1 or 2) need the row scale outputs (item %d)", i);
        RPNET_REQUIRE(q.cout % 32 == 0 && q.cin_pad % 32 == 0 && (q.taps == 9 || q.taps == 1 || q.taps == 4), RPNET_ERR_SHAPE,
                      "pack_conv_weights_split: cout %d cin_pad %d taps %d (item %d)", q.cout, q.cin_pad, q.taps, i);
        RPNET_REQUIRE(q.cin_off0 + q.cin_split <= q.cin_pad && q.cin_off1 + (q.cin - q.cin_split) <= q.cin_pad, RPNET_ERR_SHAPE,
                      "pack_conv_weights_split: channel ranges exceed cin_pad (item %d)", i);
        pk.it[i] = q;
        max_rows = std::max(max_rows, q.cout + q.cin);
        max_tiles = std::max(max_tiles, (pack_needs_gather(q) ? q.cin_pad / 32 : cdiv(q.cin, 32)) * (q.cout / 32));
    }
    for (int i = n; i < kPackMax; ++i) pk.it[i] = items[0];
    hipStream_t s = (hipStream_t)stream;
    if (planes <= 2)    // row_scale_wd covers the cin_pad gathered rows; the caller presets the padding rows (any non-zero value)
        hipLaunchKernelGGL(weight_row_scale_kernel, dim3(max_rows, n), dim3(256), 0, s, pk);
    if (planes == 3) hipLaunchKernelGGL(pack_weight_split_kernel<3>, dim3(max_tiles, n), dim3(256), 0, s, pk);
    else if (planes == 2) hipLaunchKernelGGL(pack_weight_split_kernel<2>, dim3(max_tiles, n), dim3(256), 0, s, pk);
    else hipLaunchKernelGGL(pack_weight_split_kernel<1>, dim3(max_tiles, n), dim3(256), 0, s, pk);
    return check_launch("pack_conv_weights_split");
}

extern "C" int rpnet_pack_conv_weight_split(const float* w, void* wp, void* wd, int cout, int cin, int taps, int cin_off0,
                                            int cin_split, int cin_off1, int cin_pad, int planes, float* row_scale_wp,
                                            float* row_scale_wd, rpnet_stream_t stream) {
    rpnet_pack_item q;
    q.w = w; q.wp = wp; q.wd = wd; q.row_scale_wp = row_scale_wp; q.row_scale_wd = row_scale_wd;
    q.cout = cout; q.cin = cin; q.taps = taps; q.cin_off0 = cin_off0; q.cin_split = cin_split; q.cin_off1 = cin_off1; q.cin_pad = cin_pad;
    return rpnet_pack_conv_weights_split(&q, 1, planes, stream);
}
