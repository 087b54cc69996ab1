// Weight gradient of the 3x3 convolutions on the bf16 matrix pipe with split-bf16 operands (see
// conv_split.hip for the arithmetic): all nine taps of a 64 (cin) x 64 (cout) tile in one block, as in
// conv_wgrad9_kernel, dWp[tap][cin][cout] = sum_p x[p + tap][cin] * dy[p][cout], K = pixels.
//
// Both operands are K-major in memory (NHWC: a pixel's channels are contiguous) while a bf16 MFMA operand
// register holds 8 consecutive k of ONE row/column.  gfx950's transposing LDS read does that turn for
// free: the LDS images stay [pixel][channel] (straight 16-byte row copies from HBM) and
// ds_read_b64_tr_b16 hands lane c of a 16-lane group the 4 pixels x channel c column of a
// [4 pixel][16 channel] block; every lane passes the address of its own (pixel, 4 channels) piece, so
//   * the kx = -1 / 0 / +1 taps are the same read one pixel row up or down (an immediate offset), and
//   * the image-border masks of the kx = -1 / +1 taps (dy rows whose left / right neighbour is outside
//     the image) are the same read redirected to a row of zeros — no masked copies of dy.
// LDS: per plane an x strip [3 ky][34 pixel][64 cin] and a dy tile [32 pixel + zero row][64 cout], rows
// padded to 192 bytes (four consecutive rows then cover all 64 banks: conflict-free transposing reads).
// One block per CU (9 x 16 accumulator registers per lane); 108 (NP = 3) MFMAs per wave per 32-pixel
// K-step keep the matrix pipe busy across the two barriers of the single-stage pipeline.
#include <algorithm>

#include "common.h"

namespace rpnet {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

void wgrad9_plan(int M, int Cin, int Cout, int* ksplit, int* steps_per_split);   // conv_wgrad.hip

template <int NP, bool POW2>
__global__ __launch_bounds__(256, 1) void conv_wgrad9_split_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                     float* __restrict__ partial, const int M, const int Cin,
                                                                     const int Cout, const int tiles, const int tiles_n,
                                                                     const int ksplit, const int steps_per_split,
                                                                     const int lw, const int lh) {
    constexpr int BM = 64, BK = 32, SJ = BK + 2, RS = 192;
    constexpr int A_PLANE = 3 * SJ * RS, B_PLANE = (BK + 1) * RS;
    constexpr int A_ITEMS = 3 * SJ * 8, A_IT = (A_ITEMS + 255) / 256;   // 16-byte pieces of one plane of the strip
    __shared__ __attribute__((aligned(16))) unsigned char smem[NP * (A_PLANE + B_PLANE)];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    int tile, z;
    if ((ksplit & 7) == 0) {       // the blocks of one pixel chunk share an XCD (and its L2)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        z = (j / tiles) * 8 + xcd;
        tile = j - (j / tiles) * tiles;
    } else {
        z = blockIdx.x / tiles;
        tile = blockIdx.x - z * tiles;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int cm0 = tm * BM, n0 = tn * 64;

    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const unsigned short* src; int Cs, cc;
    if (cm0 < d.C0) { src = reinterpret_cast<const unsigned short*>(d.x0); Cs = d.C0; cc = cm0; }
    else { src = reinterpret_cast<const unsigned short*>(d.x1); Cs = d.C1; cc = cm0 - d.C0; }
    const size_t planex = (size_t)d.N * Hs * Ws * Cs, planey = (size_t)M * Cout;

    const int total_steps = (M + BK - 1) / BK;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    __amdgpu_buffer_rsrc_t rsx[NP], rsy[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rsx[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src + p * planex), (short)0, (int)(planex * 2), 0x00020000);
        rsy[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dy + p * planey), (short)0, (int)(planey * 2), 0x00020000);
    }
    // strip pieces this thread stages: row r = e >> 3 of [3 ky][34], 16-byte column e & 7; source q = p0 + qoff
    int qoff[A_IT], kyv[A_IT], adst[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int e = min(t + 256 * i, A_ITEMS - 1);
        const int r = e >> 3;
        const int kyi = r / SJ, j = r - kyi * SJ;
        kyv[i] = kyi - 1;
        qoff[i] = j - 1 + (kyi - 1) * W;
        adst[i] = r * RS + (e & 7) * 16;
    }
    const int brow = t >> 3, bdst = NP * A_PLANE + brow * RS + (t & 7) * 16;
    u32x4 ra[NP][A_IT], rb[NP];
    auto load_tile = [&](int st) {
        const int p0 = st * BK;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int q = p0 + qoff[i];
            int pix, yq;
            if (POW2) {
                yq = (q >> lw) & (H - 1);
                pix = ups ? (((q >> (lw + lh)) * Hs + (yq >> 1)) * Ws + ((q & (W - 1)) >> 1)) : q;
            } else {
                const int n = q / HW, rem = q - n * HW;
                yq = rem / W;
                const int xq = rem - yq * W;
                pix = (n * Hs + (yq >> ups)) * Ws + (xq >> ups);
            }
            const int yp = yq - kyv[i];                     // row of the output pixel this source serves
            const bool ok = (unsigned)q < (unsigned)M && (unsigned)yp < (unsigned)H;
            const int voff = ok ? pix * (Cs * 2) + (adst[i] % RS) : (int)0x80000000;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                ra[p][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx[p], voff, cc * 2, 0));
        }
        const int yoff = (p0 + brow) * (Cout * 2) + (t & 7) * 16;   // past M*Cout: beyond num_records, zeros
#pragma unroll
        for (int p = 0; p < NP; ++p)
            rb[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy[p], yoff, n0 * 2, 0));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                if (t + 256 * i < A_ITEMS) *reinterpret_cast<u32x4*>(smem + p * A_PLANE + adst[i]) = ra[p][i];
            *reinterpret_cast<u32x4*>(smem + p * B_PLANE + bdst) = rb[p];
        }
    };
    if (t < 8 * NP) {   // the zero row (row 32) of every dy plane
        const u32x4 zero = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + NP * A_PLANE + (t >> 3) * B_PLANE + BK * RS + (t & 7) * 16) = zero;
    }

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // transposing-read geometry: 16-lane group g = lane >> 4 covers channels 16 (g & 1).. of k rows 8 (g >> 1)..;
    // lane L of the group addresses row (L >> 2), channels 4 (L & 3)..
    const int g = lane >> 4, L = lane & 15;
    const int krow = 8 * (g >> 1) + (L >> 2);
    const int a_base = krow * RS + (wm * 32 + 16 * (g & 1) + 4 * (L & 3)) * 2;
    const int b_col = NP * A_PLANE + (wn * 32 + 16 * (g & 1) + 4 * (L & 3)) * 2;
    auto tr = [&](int byte_off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + byte_off));
    };

    if (s_begin < s_end) {
        load_tile(s_begin);
        store_tile();
        __syncthreads();
        for (int st = s_begin; st < s_end; ++st) {
            const bool more = st + 1 < s_end;
            if (more) load_tile(st + 1);
            // dy rows this lane addresses in the 4 (slice, half) reads, and their kx = -1 / +1 border redirects
            int b0[4], bm_[4], bp_[4];
#pragma unroll
            for (int se = 0; se < 4; ++se) {
                const int row = 16 * (se >> 1) + 4 * (se & 1) + krow;
                const int p = st * BK + row;
                const int ox = POW2 ? (p & (W - 1)) : (p % W);
                b0[se] = b_col + row * RS;
                bm_[se] = ox >= 1 ? b0[se] : b_col + BK * RS;
                bp_[se] = ox <= W - 2 ? b0[se] : b_col + BK * RS;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 bf[3][NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    bf[0][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(bm_[2 * s] + p * B_PLANE), tr(bm_[2 * s + 1] + p * B_PLANE), 0, 1, 2, 3, 4, 5, 6, 7));
                    bf[1][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(b0[2 * s] + p * B_PLANE), tr(b0[2 * s + 1] + p * B_PLANE), 0, 1, 2, 3, 4, 5, 6, 7));
                    bf[2][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(bp_[2 * s] + p * B_PLANE), tr(bp_[2 * s + 1] + p * B_PLANE), 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        bf16x8 af[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const int o = a_base + p * A_PLANE + (ky * SJ + 16 * s + kx) * RS;   // strip row = pixel + kx (kx - 1 + 1)
                            af[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(o), tr(o + 4 * RS), 0, 1, 2, 3, 4, 5, 6, 7));
                        }
                        constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0};
                        constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
                        constexpr int NPROD = NP == 3 ? 6 : 3;
#pragma unroll
                        for (int q = 0; q < NPROD; ++q) {
                            const int pa = NP == 3 ? PA3[q] : PA2[q], pb = NP == 3 ? PB3[q] : PB2[q];
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa], bf[kx][pb], acc[ky * 3 + kx], 0, 0, 0);
                        }
                    }
            }
            __syncthreads();
            if (more) store_tile();
            __syncthreads();
        }
    }
    if (z >= ksplit) return;
    const int li = lane & 31, h = lane >> 5;
    const int col = n0 + wn * 32 + li;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* out = partial + ((size_t)(z * 9 + tap) * Cin) * Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = cm0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(size_t)row * Cout + col] = acc[tap][r];
        }
    }
}

static int ilog2x(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// launches the nine-tap split kernel into `part9` ([ksplit][9][Cin][Cout] fp32, plan = wgrad9_plan)
int conv_wgrad9_split(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9,
                      hipStream_t s) {
    const int tiles_n9 = Cout / 64, tiles9 = (Cin / 64) * tiles_n9;
    const int lw = ilog2x(d->W), lh = ilog2x(d->H);
    const bool p2 = lw >= 0 && lh >= 0;
    const unsigned short* dys = (const unsigned short*)dy;
#define RPNET_W9S(NPL, P2)                                                                                                \
    hipLaunchKernelGGL((conv_wgrad9_split_kernel<NPL, P2>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout, \
                       tiles9, tiles_n9, ks9, sps9, P2 ? lw : 0, P2 ? lh : 0)
    if (d->split_planes == 3) { if (p2) RPNET_W9S(3, true); else RPNET_W9S(3, false); }
    else { if (p2) RPNET_W9S(2, true); else RPNET_W9S(2, false); }
#undef RPNET_W9S
    return check_launch("conv_wgrad9_split");
}

}  // namespace rpnet
