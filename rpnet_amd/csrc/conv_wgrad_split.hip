// Weight gradient of the 3x3 convolutions on the bf16 matrix pipe with split-bf16 operands (see
// conv_split.hip for the arithmetic): all nine taps of a 64 (cin) x 64 (cout) tile in one block, as in
// conv_wgrad9_kernel, dWp[tap][cin][cout] = sum_p x[p + tap][cin] * dy[p][cout], K = pixels.
//
// Both operands are K-major in memory (NHWC: a pixel's channels are contiguous) while a bf16 MFMA operand
// register holds 8 consecutive k of ONE row/column.  gfx950's transposing LDS read does that turn for
// free: the LDS images stay [pixel][channel] (straight 16-byte row copies from HBM) and
// ds_read_b64_tr_b16 hands lane c of a 16-lane group the 4 pixels x channel c column of a
// [4 pixel][16 channel] block; every lane passes the address of its own (pixel, 4 channels) piece, so
//   * the kx = -1 / 0 / +1 taps are the same dy read one pixel row up or down (the sum runs over the x pixel q,
//     x[q + ky W] * dy[q - kx]: the shift sits on the dy side, so an x fragment serves three taps), and
//   * the image-border masks of the kx = -1 / +1 taps (q and q - kx in different image rows) are the same read
//     redirected to a row of zeros — no masked copies of dy.
// LDS: per plane an x strip [3 ky][32 pixel][64 cin] and a dy tile [34 pixel + zero row][64 cout], rows
// padded to 192 bytes (four consecutive rows then cover all 64 banks: conflict-free transposing reads).
// One block per CU (9 x 16 accumulator registers per lane, two LDS stages): 108 (NP = 3) MFMAs per wave per
// 32-pixel K-step, one barrier per step, the next tile's LDS writes interleaved with the MFMAs.
#include <algorithm>

#include "common.h"
#include "split_bf16.h"

namespace rpnet {

using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;


// KYW = false: 4 waves, each a 32x32 quadrant of the tile for all nine taps (9 x 16 accumulators, one wave per SIMD).
// KYW = true: 12 waves = 4 quadrants x 3 tap rows (ky): 3 x 16 accumulators per wave, three waves per SIMD, so one
//             wave's LDS reads / staging issue under the other waves' MFMAs; the dy fragments are read by three
//             waves instead of one (LDS reads 0.67 -> 1.33 per MFMA, still a third of the LDS bandwidth).
template <int NP, bool POW2, bool KYW>
__global__ __launch_bounds__(KYW ? 768 : 256, 1) void conv_wgrad9_split_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                     float* __restrict__ partial, const int M, const int Cin,
                                                                     const int Cout, const int tiles, const int tiles_n,
                                                                     const int ksplit, const int steps_per_split,
                                                                     const int lw, const int lh) {
    constexpr int BM = 64, BK = 32, RS = 192;
    constexpr int A_PLANE = 3 * BK * RS, B_ROWS = BK + 2, B_PLANE = (B_ROWS + 1) * RS;
    constexpr int NT = KYW ? 768 : 256;
    constexpr int A_IT = 3 * BK * 8 / NT;                     // 16-byte pieces of one plane of the x strip per thread
    constexpr int NACC = KYW ? 3 : 9;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];   // two stages: 147 KB of the CU's 160 KB

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int wm = (wv & 3) >> 1, wn = wv & 1, wky = wv >> 2;
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
    // x strip pieces this thread stages: row r = e >> 3 of [3 ky][32], 16-byte column e & 7; source q = p0 + qoff
    int qoff[A_IT], kyv[A_IT], adst[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int e = t + NT * i;
        const int r = e >> 3;
        const int kyi = r / BK, j = r - kyi * BK;
        kyv[i] = kyi - 1;
        qoff[i] = j + (kyi - 1) * W;
        adst[i] = r * RS + (e & 7) * 16;
    }
    // dy pieces: rows p0 - 1 .. p0 + 32 (34 rows x 8 pieces = 272: a second, partial pass for t < 16)
    const int c16 = (t & 7) * 16;
    const int brow0 = t >> 3, brow1 = 32 + (t >> 3);
    const bool b1 = brow0 < B_ROWS, b2 = !KYW && t < (B_ROWS - 32) * 8;   // 34 rows: a second partial pass with 256 threads
    u32x4 ra[NP][A_IT], rb[NP][2];
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
            const int yp = yq - kyv[i];                     // image row of the x pixel this source row pairs with
            const bool ok = (unsigned)q < (unsigned)M && (unsigned)yp < (unsigned)H;
            const int voff = ok ? pix * (Cs * 2) + c16 : (int)0x80000000;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                ra[p][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx[p], voff, cc * 2, 0));
        }
        // pixels before 0 (negative offset) or past M (beyond num_records) read as zeros
        const int y0 = b1 ? (p0 - 1 + brow0) * (Cout * 2) + c16 : (int)0x80000000;
        const int y1 = b2 ? (p0 - 1 + brow1) * (Cout * 2) + c16 : (int)0x80000000;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            rb[p][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy[p], y0, n0 * 2, 0));
            if (!KYW) rb[p][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy[p], y1, n0 * 2, 0));
        }
    };
    auto store_tile = [&](int stage) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                *reinterpret_cast<u32x4*>(smem + stage + p * A_PLANE + adst[i]) = ra[p][i];
            unsigned char* bpl = smem + stage + NP * A_PLANE + p * B_PLANE;
            if (b1) *reinterpret_cast<u32x4*>(bpl + brow0 * RS + c16) = rb[p][0];
            if (b2) *reinterpret_cast<u32x4*>(bpl + brow1 * RS + c16) = rb[p][1];
        }
    };
    if (t < 16 * NP) {   // the zero row (row 34) of every dy plane of both stages
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int pl = t >> 3;
        *reinterpret_cast<u32x4*>(smem + (pl / NP) * STAGE + NP * A_PLANE + (pl % NP) * B_PLANE + B_ROWS * RS + c16) = zero;
    }

    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // transposing-read geometry: 16-lane group g = lane >> 4 covers channels 16 (g & 1).. of k rows 8 (g >> 1)..;
    // lane L of the group addresses row (L >> 2), channels 4 (L & 3)..
    const int g = lane >> 4, L = lane & 15;
    const int krow = 8 * (g >> 1) + (L >> 2);
    const int a_base0 = krow * RS + (wm * 32 + 16 * (g & 1) + 4 * (L & 3)) * 2;
    const int b_col0 = NP * A_PLANE + (wn * 32 + 16 * (g & 1) + 4 * (L & 3)) * 2;
    auto tr = [&](int byte_off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + byte_off));
    };

    // Two LDS stages, one barrier per K-step: tile st + 1 (fetched during step st - 1) is written into the other stage
    // between the MFMA groups of step st, tile st + 2 is then put in flight.
    if (s_begin < s_end) {
        load_tile(s_begin);
        store_tile(0);
        if (s_begin + 1 < s_end) load_tile(s_begin + 1);
        __syncthreads();
        for (int st = s_begin; st < s_end; ++st) {
            const bool more = st + 1 < s_end;
            const int cur = ((st - s_begin) & 1) * STAGE;
            const int a_base = a_base0 + cur, b_col = b_col0 + cur;
            // x pixel rows this lane addresses in the 4 (slice, half) reads; the dy row paired with x pixel q for tap
            // kx is q - (kx - 1) = tile row (q - p0) + 2 - kx, redirected to the zero row when q and it are not in
            // the same image row
            int bk0[4], bk1[4], bk2[4];
#pragma unroll
            for (int se = 0; se < 4; ++se) {
                const int row = 16 * (se >> 1) + 4 * (se & 1) + krow;
                const int q = st * BK + row;
                const int ox = POW2 ? (q & (W - 1)) : (q % W);
                const int zr = b_col + B_ROWS * RS;
                bk0[se] = ox <= W - 2 ? b_col + (row + 2) * RS : zr;     // kx = -1: dy[q + 1]
                bk1[se] = b_col + (row + 1) * RS;                        // kx =  0: dy[q]
                bk2[se] = ox >= 1 ? b_col + row * RS : zr;               // kx = +1: dy[q - 1]
            }
            auto load_bf = [&](int s, bf16x8 (&bf)[3][NP]) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    bf[0][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(bk0[2 * s] + p * B_PLANE), tr(bk0[2 * s + 1] + p * B_PLANE), 0, 1, 2, 3, 4, 5, 6, 7));
                    bf[1][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(bk1[2 * s] + p * B_PLANE), tr(bk1[2 * s + 1] + p * B_PLANE), 0, 1, 2, 3, 4, 5, 6, 7));
                    bf[2][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(bk2[2 * s] + p * B_PLANE), tr(bk2[2 * s + 1] + p * B_PLANE), 0, 1, 2, 3, 4, 5, 6, 7));
                }
            };
            auto load_af = [&](int s, int ky, bf16x8 (&af)[NP]) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int o = a_base + p * A_PLANE + (ky * BK + 16 * s) * RS;
                    af[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(o), tr(o + 4 * RS), 0, 1, 2, 3, 4, 5, 6, 7));
                }
            };
            constexpr int NPROD = nprod<NP>();
            if constexpr (KYW) {
                // this wave: one tap row (ky = wky), three kx taps of its 32x32 quadrant
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 af[NP], bf[3][NP];
                    load_af(s, wky, af);
                    load_bf(s, bf);
                    if (s == 1 && more) {
                        store_tile(STAGE - cur);
                        if (st + 2 < s_end) load_tile(st + 2);
                    }
#pragma unroll
                    for (int q = 0; q < NPROD; ++q) {
                        const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[kx] = mma16<NP>(af[pa], bf[kx][pb], acc[kx]);
                    }
                }
            } else {
                // All fragments of a 16-pixel slice (3 ky strips of x, 3 kx variants of dy, NP planes each) feed its
                // 9 x NPROD MFMAs; the two slices are double-buffered in registers and the taps advance together so
                // that consecutive MFMAs never share an accumulator.
                auto mma_slice = [&](const bf16x8 (&af)[3][NP], const bf16x8 (&bf)[3][NP]) {
#pragma unroll
                    for (int q = 0; q < NPROD; ++q) {
                        const int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap)
                            acc[tap % NACC] = mma16<NP>(af[tap / 3][pa], bf[tap % 3][pb], acc[tap % NACC]);
                    }
                };
                bf16x8 afA[3][NP], bfA[3][NP], afB[3][NP], bfB[3][NP];
                load_bf(0, bfA);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) load_af(0, ky, afA[ky]);
                load_bf(1, bfB);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) load_af(1, ky, afB[ky]);
                __builtin_amdgcn_sched_barrier(0);
                mma_slice(afA, bfA);
                if (more) {
                    store_tile(STAGE - cur);
                    if (st + 2 < s_end) load_tile(st + 2);
                }
                mma_slice(afB, bfB);
            }
            __syncthreads();
        }
    }
    if (z >= ksplit) return;
    const int li = lane & 31, h = lane >> 5;
    const int col = n0 + wn * 32 + li;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int tap = KYW ? wky * 3 + a : a;
        float* out = partial + ((size_t)(z * 9 + tap) * Cin) * Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = cm0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(size_t)row * Cout + col] = acc[a][r];
        }
    }
}

// Single-tap form (the 1x1 convolution): dW[cin][cout] = sum_p x[p][cin] dy[p][cout] — the centre tap of the kernel above
// without the strips, shifts and border rows.  64 x 64 tile, 4 waves (32 x 32 quadrants), two LDS stages of a 32-pixel
// K-step; the work is 9x smaller than a 3x3 layer's, so the kernel is bound by streaming x and dy once (HBM / L2), not by
// the matrix pipe.  Sources x0 | x1 as in the forward gather (a tile lies in one source).
template <int NP>
__global__ __launch_bounds__(256, 2) void conv_wgrad1_split_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                   float* __restrict__ partial, const int M, const int Cin,
                                                                   const int Cout, const int tiles, const int tiles_n,
                                                                   const int steps_per_split) {
    constexpr int BK = 32, RS = 192;
    constexpr int A_PLANE = BK * RS, B_PLANE = BK * RS;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int z = blockIdx.x / tiles, tile = blockIdx.x - z * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int cm0 = tm * 64, n0 = tn * 64;

    const unsigned short* src; int Cs, cc;
    if (cm0 < d.C0) { src = reinterpret_cast<const unsigned short*>(d.x0); Cs = d.C0; cc = cm0; }
    else { src = reinterpret_cast<const unsigned short*>(d.x1); Cs = d.C1; cc = cm0 - d.C0; }
    const size_t planex = (size_t)M * Cs, planey = (size_t)M * Cout;
    const int total_steps = (M + BK - 1) / BK;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    __amdgpu_buffer_rsrc_t rsx[NP], rsy[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rsx[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src + p * planex), (short)0, (int)(planex * 2), 0x00020000);
        rsy[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dy + p * planey), (short)0, (int)(planey * 2), 0x00020000);
    }
    // one 16-byte piece of the x tile and one of the dy tile per thread and plane: row t >> 3, column piece t & 7
    const int row = t >> 3, c16 = (t & 7) * 16;
    const int dst = row * RS + c16;
    u32x4 ra[NP], rb[NP];
    auto load_tile = [&](int st) {
        const int pix = st * BK + row;             // rows past M lie beyond num_records: zeros
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ra[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx[p], pix * (Cs * 2) + c16, cc * 2, 0));
            rb[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy[p], pix * (Cout * 2) + c16, n0 * 2, 0));
        }
    };
    auto store_tile = [&](int stage) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<u32x4*>(smem + stage + p * A_PLANE + dst) = ra[p];
            *reinterpret_cast<u32x4*>(smem + stage + NP * A_PLANE + p * B_PLANE + dst) = rb[p];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // transposing-read geometry as in the nine-tap kernel
    const int g = lane >> 4, L = lane & 15;
    const int krow = 8 * (g >> 1) + (L >> 2);
    const int a_base0 = krow * RS + (wm * 32 + 16 * (g & 1) + 4 * (L & 3)) * 2;
    const int b_base0 = NP * A_PLANE + krow * RS + (wn * 32 + 16 * (g & 1) + 4 * (L & 3)) * 2;
    auto tr = [&](int byte_off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + byte_off));
    };
    if (s_begin < s_end) {
        load_tile(s_begin);
        store_tile(0);
        if (s_begin + 1 < s_end) load_tile(s_begin + 1);
        __syncthreads();
        for (int st = s_begin; st < s_end; ++st) {
            const bool more = st + 1 < s_end;
            const int cur = ((st - s_begin) & 1) * STAGE;
            if (more) {
                store_tile(STAGE - cur);
                if (st + 2 < s_end) load_tile(st + 2);
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                bf16x8 af[NP], bf[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int oa = cur + a_base0 + p * A_PLANE + 16 * sl * RS;
                    const int ob = cur + b_base0 + p * B_PLANE + 16 * sl * RS;
                    af[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(oa), tr(oa + 4 * RS), 0, 1, 2, 3, 4, 5, 6, 7));
                    bf[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr(ob), tr(ob + 4 * RS), 0, 1, 2, 3, 4, 5, 6, 7));
                }
                constexpr int NPROD = nprod<NP>();
#pragma unroll
                for (int q = 0; q < NPROD; ++q) acc = mma16<NP>(af[prod_a<NP>(q)], bf[prod_b<NP>(q)], acc);
            }
            __syncthreads();
        }
    }
    const int li = lane & 31, h = lane >> 5;
    const int col = n0 + wn * 32 + li;
    float* out = partial + (size_t)z * Cin * Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rw = cm0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(size_t)rw * Cout + col] = acc[r];
    }
}

// split-K plan of the single-tap kernel: never more splits than rpnet_conv_wgrad_workspace_bytes (wgrad_plan) provides for
void wgrad1_split_plan(int M, int Cin, int Cout, int* ksplit, int* steps_per_split) {
    const int tiles = (Cin / 64) * (Cout / 64);
    const int total_steps = (M + 31) / 32;
    int ks = (512 + tiles - 1) / tiles;                    // two blocks per CU
    ks = std::max(1, std::min(ks, std::max(1, total_steps / 8)));
    ks = std::min(ks, kWgrad1MaxSplits);
    *steps_per_split = (total_steps + ks - 1) / ks;
    *ksplit = (total_steps + *steps_per_split - 1) / *steps_per_split;
}

int conv_wgrad1_split(const rpnet_conv_desc* d, const void* dy, float* part, int M, int Cin, int Cout, int ks, int sps, hipStream_t s) {
    const int tiles_n = Cout / 64, tiles = (Cin / 64) * tiles_n;
    const unsigned short* dys = (const unsigned short*)dy;
    if (d->split_planes == 3)
        hipLaunchKernelGGL((conv_wgrad1_split_kernel<3>), dim3(tiles * ks), dim3(256), 0, s, *d, dys, part, M, Cin, Cout, tiles, tiles_n, sps);
    else if (d->split_planes == 2)
        hipLaunchKernelGGL((conv_wgrad1_split_kernel<2>), dim3(tiles * ks), dim3(256), 0, s, *d, dys, part, M, Cin, Cout, tiles, tiles_n, sps);
    else
        hipLaunchKernelGGL((conv_wgrad1_split_kernel<1>), dim3(tiles * ks), dim3(256), 0, s, *d, dys, part, M, Cin, Cout, tiles, tiles_n, sps);
    return check_launch("conv_wgrad1_split");
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
    const bool kyw = d->tune != 4;       // tuning override carried by the descriptor: 4 = one wave per quadrant, all nine taps
#define RPNET_W9S(NPL, P2)                                                                                                     \
    do {                                                                                                                       \
        if (kyw)                                                                                                               \
            hipLaunchKernelGGL((conv_wgrad9_split_kernel<NPL, P2, true>), dim3(tiles9 * ks9), dim3(768), 0, s, *d, dys, part9, \
                               M, Cin, Cout, tiles9, tiles_n9, ks9, sps9, P2 ? lw : 0, P2 ? lh : 0);                           \
        else                                                                                                                   \
            hipLaunchKernelGGL((conv_wgrad9_split_kernel<NPL, P2, false>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, \
                               M, Cin, Cout, tiles9, tiles_n9, ks9, sps9, P2 ? lw : 0, P2 ? lh : 0);                           \
    } while (0)
    if (d->split_planes == 3) { if (p2) RPNET_W9S(3, true); else RPNET_W9S(3, false); }
    else if (d->split_planes == 2) { if (p2) RPNET_W9S(2, true); else RPNET_W9S(2, false); }
    else { if (p2) RPNET_W9S(1, true); else RPNET_W9S(1, false); }
#undef RPNET_W9S
    return check_launch("conv_wgrad9_split");
}

}  // namespace rpnet
