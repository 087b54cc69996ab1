// Weight gradient of the 3x3 convolutions on fp16 split planes — conv_wgrad9_split_kernel (conv_wgrad_split.hip) rebuilt
// around the LDS-DMA path with ONE wave per SIMD, as conv_split_dma.hip did for the forward / input-gradient kernel:
//   dWp[z][tap][cin][cout] = sum over the block's pixels p of x[p + tap][cin] * dy[p][cout]
// (GEMM M = cin, N = cout, K = pixels; all nine taps of a 64 x 64 tile in one block; split-K over pixel ranges z).
//   * four waves, each a 32 x 32 quadrant of the tile for ALL nine taps (9 x 16 accumulator registers): an x fragment of
//     tap row ky serves the three kx taps and a dy fragment the three ky taps — 24 transposing reads per 27 MFMAs;
//   * both operands stay [pixel][channel] in LDS (what NHWC memory gives) and reach the MFMA's k-contiguous operand
//     registers through ds_read_b64_tr_b16.  Rows are 128 bytes, UNPADDED (a DMA instruction writes 1 KB = 8 whole rows,
//     lane-linear): the four consecutive pixel rows a 32-lane group reads are kept on four different bank quarters by
//     swapping the 64-byte halves of a row with bit 1 of its row index — applied to the per-lane SOURCE address of the
//     DMA and to the read address (the round-2 kernel padded rows to 192 bytes instead, which a DMA cannot write);
//   * per 32-pixel K-step the DMA engine writes three x strips (tap rows ky = -1 / 0 / +1: pixels p + ky W, rows outside
//     the image or the tensor read as zeros through the descriptor's bounds check) and the dy tile (pixels p0 - 1 .. p0 + 38;
//     the kx = -1 / +1 taps are the same reads one row up / down, redirected to a row of zeros at the image's left / right
//     border) into a ring of FOUR stages; one raw s_barrier per K-step behind a counted s_waitcnt vmcnt: the loads of the
//     next two steps stay in flight, the data of step k + 1 are visible one barrier early and its first fragments are read
//     under the MFMAs of step k;
//   * the issue order is pinned (one fragment read behind each MFMA, a DMA behind every second one);
//   * the partial sums leave through a per-wave LDS transpose as 16-byte stores (36 instead of 144 per thread).
// Same tiles, same split-K plan and the same order of accumulation as conv_wgrad9_split_kernel: bit-identical partial sums.
#include <type_traits>

#include "common.h"
#include "lds_dma.h"
#include "split_bf16.h"

namespace rpnet {

using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for_w(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_w<N, I + 1>(f);
    }
}

template <int NP, bool POW2>
__global__ __launch_bounds__(256, 1) void conv_wgrad9_dma_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                  float* __restrict__ partial, const int M, const int Cin,
                                                                  const int Cout, const int tiles, const int tiles_n,
                                                                  const int ksplit, const int steps_per_split, const int lw,
                                                                  const int lh) {
    constexpr int BK = 32, RB = 128;                               // pixels per K-step, bytes per LDS row (64 channels)
    constexpr int A_PLANE = 3 * BK * RB;                           // [3 ky][32 pixel] rows: 12 KB
    constexpr int B_ROWS = 40, ZROW = 40, B_PLANE = 48 * RB;       // 5 pieces of 8 rows + the zero row: 6 KB
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);                // 36 KB (NP = 2)
    constexpr int NS = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
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
    const int cm0 = tm * 64, n0 = tn * 64;

    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const unsigned short* src; int Cs, cc;
    if (cm0 < d.C0) { src = reinterpret_cast<const unsigned short*>(d.x0); Cs = d.C0; cc = cm0; }
    else { src = reinterpret_cast<const unsigned short*>(d.x1); Cs = d.C1; cc = cm0 - d.C0; }
    const size_t planex = (size_t)d.N * Hs * Ws * Cs, planey = (size_t)M * Cout;
    const int pbx = (int)(planex * 2), pby = (int)(planey * 2);

    const int total_steps = (M + BK - 1) / BK;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    // one descriptor per tensor over all its planes (the plane is part of the scalar offset)
    const srd_t rsx = make_srd(src, NP * pbx), rsy = make_srd(dy, NP * pby);
    const unsigned lds0 = lds_addr(smem);

    // DMA lane geometry: lane l of a 1 KB piece writes 16 bytes at piece + 16 l = row (l >> 3), 16-byte slot (l & 7); the
    // slot belongs to the 64-byte half (l >> 2) & 1, which holds the SOURCE half ^ bit 1 of the row (pieces start at
    // multiples of 8 rows)
    const int drow = lane >> 3;
    const int dcol = ((((lane >> 2) & 1) ^ ((lane >> 4) & 1)) << 6) | ((lane & 3) << 4);       // source byte offset inside the row
    // x strips: this wave moves rows 8 wv .. 8 wv + 7 of each of the three strips (every plane); dy: piece wv, wave 0 also
    // the fifth piece (rows 32 .. 39)
    auto dma_x = [&](auto kyc, const int st, const int stage) {
        constexpr int kyi = decltype(kyc)::value;
        const int q = st * BK + 8 * wv + drow + (kyi - 1) * W;
        int pix, yq;
        if (POW2) {      // (n Hs + (y >> ups)) Ws + (x >> ups): the pixel itself without up-sampling
            yq = (q >> lw) & (H - 1);
            pix = ((q >> (lw + lh)) * Hs + (yq >> ups)) * Ws + ((q & (W - 1)) >> ups);
        } else {
            const int n = q / HW, rem = q - n * HW;
            yq = rem / W;
            const int xq = rem - yq * W;
            pix = (n * Hs + (yq >> ups)) * Ws + (xq >> ups);
        }
        const int yp = yq - (kyi - 1);                     // image row of the dy pixel this source row pairs with
        const bool ok = (unsigned)q < (unsigned)M && (unsigned)yp < (unsigned)H;
        const int voff = ok ? pix * (Cs * 2) + dcol : (int)0x80000000;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            lds_dma16(rsx, lds0 + stage * STAGE + p * A_PLANE + (kyi * BK + 8 * wv) * RB, voff, cc * 2 + p * pbx);
        }
    };
    auto dma_y = [&](const int piece, const int st, const int stage) {
        // pixels before 0 or past M read as zeros (an offset beyond num_records; the planes share one descriptor, so the
        // bound is checked here)
        const int pixrow = st * BK - 1 + 8 * piece + drow;
        const int voff = (unsigned)pixrow < (unsigned)M ? pixrow * (Cout * 2) + dcol : (int)0x80000000;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            lds_dma16(rsy, lds0 + stage * STAGE + NP * A_PLANE + p * B_PLANE + 8 * piece * RB, voff, n0 * 2 + p * pby);
        }
    };
    auto dma_step = [&](const int st, const int stage) {
        dma_x(std::integral_constant<int, 0>{}, st, stage);
        dma_x(std::integral_constant<int, 1>{}, st, stage);
        dma_x(std::integral_constant<int, 2>{}, st, stage);
        dma_y(wv, st, stage);
        if (wv == 0) dma_y(4, st, stage);
    };
    // wave 0 issues (3 + 2) NP DMAs per step, the others (3 + 1) NP: "everything but the last step's" as a wait count
    auto wait_all_but_one_step = [&]() {
        if (wv == 0) {
            if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    };
    static_assert(NP == 2, "two fp16 planes (one plane has 9 MFMAs per slice for 12 fragment reads: another schedule)");

    // the zero row of every dy plane of every stage (the DMA never writes it)
    if (t < NS * NP * 8) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int pl = t >> 3;
        *reinterpret_cast<u32x4*>(smem + (pl / NP) * STAGE + NP * A_PLANE + (pl % NP) * B_PLANE + ZROW * RB + (t & 7) * 16) = zero;
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
    const int cbyte = (16 * (g & 1) + 4 * (L & 3)) * 2;               // inside the wave's 64-byte half
    // x: strip rows ky 32 + 16 s + krow (+ 4): bit 1 of the row index is that of krow
    const int a_off = krow * RB + (((wm ^ ((krow >> 1) & 1)) << 6) | cbyte);
    // dy: tile row of pixel q for tap kx is (q - p0) + 2 - kx: bit 1 of the row index depends on the shift
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int r = krow + 2 - kx;
        b_off[kx] = r * RB + (((wn ^ ((r >> 1) & 1)) << 6) | cbyte);
    }
    const int b_zero = ZROW * RB + ((wn << 6) | cbyte);
    auto tr = [&](int byte_off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(smem + byte_off));
    };

    // fragment double buffer [slice parity]: af[ky][plane], bf[kx][plane]; a fragment = two transposing reads (rows krow.. and
    // krow + 4..).  Read k of a slice in order of first use (products l*h, h*l, h*h: the l plane of x and the h plane of dy
    // first): per plane pair x(ky 0), dy(kx 0..2), x(ky 1), x(ky 2) — two reads each.
    s16x4 afr[2][3][NP][2], bfr[2][3][NP][2];
    int bsel[2][3][2];          // [slice][kx][row half]: byte offset of the dy read inside a plane (tile row or the zero row)
    auto b_addr = [&](auto sc, const int st) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = 16 * s + 4 * e + krow;
            const int q = st * BK + row;
            const int ox = POW2 ? (q & (W - 1)) : (q % W);
            bsel[s][0][e] = ox <= W - 2 ? b_off[0] + (16 * s + 4 * e) * RB : b_zero;     // kx = -1: dy[q + 1]
            bsel[s][1][e] = b_off[1] + (16 * s + 4 * e) * RB;                            // kx =  0: dy[q]
            bsel[s][2][e] = ox >= 1 ? b_off[2] + (16 * s + 4 * e) * RB : b_zero;         // kx = +1: dy[q - 1]
        }
    };
    constexpr int NR = NP * 12, NMMA = nprod<NP>() * 9;
    auto read_frag = [&](auto sc, auto kc, const int sbase) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value;
        constexpr int grp = k / 12, r = (k - grp * 12) >> 1, e = k & 1;      // plane pair, fragment of the pair, row half
        constexpr int pa = NP - 1 - grp, pb = grp;
        if constexpr (r == 0 || r >= 4) {
            constexpr int ky = r == 0 ? 0 : r - 3;
            afr[s][ky][pa][e] = tr(sbase + pa * A_PLANE + (ky * BK + 16 * s + 4 * e) * RB + a_off);
        } else {
            constexpr int kx = r - 1;
            bfr[s][kx][pb][e] = tr(sbase + NP * A_PLANE + pb * B_PLANE + bsel[s][kx][e]);
        }
    };
    auto frag = [](const s16x4 lo, const s16x4 hi) {
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma_one = [&](auto sc, auto mc) {
        constexpr int s = decltype(sc)::value, m = decltype(mc)::value;
        constexpr int q = m / 9, tap = m - q * 9, ky = tap / 3, kx = tap - ky * 3;
        constexpr int pa = prod_a<NP>(q), pb = prod_b<NP>(q);
        acc[tap] = mma16<NP>(frag(afr[s][ky][pa][0], afr[s][ky][pa][1]), frag(bfr[s][kx][pb][0], bfr[s][kx][pb][1]), acc[tap]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if (s_begin < s_end) {
        // clamp: the tail re-fetches the last step into stages nobody reads again (uniform DMA counts per wave)
        auto stc = [&](int st) { return st < s_end ? st : s_end - 1; };
        dma_step(s_begin, 0);
        dma_step(stc(s_begin + 1), 1);
        dma_step(stc(s_begin + 2), 2);
        wait_all_but_one_step();                       // steps 0 and 1 have landed (this wave's part); zero rows written
        __builtin_amdgcn_s_barrier();
        b_addr(I0{}, s_begin);
        static_for_w<NR>([&](auto kc) { read_frag(I0{}, kc, 0); });
        for (int st = s_begin; st < s_end; ++st) {
            const int rel = st - s_begin;
            const int sbase = (rel & 3) * STAGE, sbase_n = ((rel + 1) & 3) * STAGE;
            const int dstage = (rel + 3) & 3, dst_step = stc(st + 3);
            // first half: MFMAs of slice 0 | reads of slice 1 | the x strips of step st + 3
            b_addr(I1{}, st);
            __builtin_amdgcn_sched_barrier(0);
            static_for_w<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I0{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR) {
                    read_frag(I1{}, mc, sbase);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 3 || m == 9 || m == 15) {
                    dma_x(std::integral_constant<int, (m - 3) / 6>{}, dst_step, dstage);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // second half: MFMAs of slice 1 | reads of step st + 1 / slice 0 (visible since the previous barrier) | dy of st + 3
            b_addr(I0{}, st + 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for_w<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I1{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR) {
                    read_frag(I0{}, mc, sbase_n);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 5) {
                    dma_y(wv, dst_step, dstage);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 11) {
                    if (wv == 0) dma_y(4, dst_step, dstage);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            wait_all_but_one_step();                   // this wave's part of step st + 2 has landed
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's DMAs: the epilogue reuses the memory
        __builtin_amdgcn_s_barrier();
    }
    if (z >= ksplit) return;
    // partial sums: each wave turns its 32 x 32 tap tiles through a private LDS slab so that a lane holds 4 consecutive
    // output channels of one cin row: 16-byte stores
    const int li = lane & 31, h = lane >> 5;
    constexpr int SW = 36;
    float* slab = reinterpret_cast<float*>(smem) + wv * 32 * SW;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * h) * SW + li] = acc[tap][r];
        __builtin_amdgcn_wave_barrier();
        float* out = partial + ((size_t)(z * 9 + tap) * Cin + cm0 + wm * 32) * Cout + n0 + wn * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gidx = q * 64 + lane, rr = gidx >> 3, c4 = gidx & 7;
            *reinterpret_cast<f32x4*>(out + (size_t)rr * Cout + c4 * 4) = *reinterpret_cast<const f32x4*>(&slab[rr * SW + c4 * 4]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static int ilog2d(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// launches the nine-tap DMA kernel into `part9` ([ksplit][9][Cin][Cout] fp32, plan = wgrad9_plan); two fp16 planes only
int conv_wgrad9_split_dma(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9,
                          hipStream_t s) {
    const int tiles_n9 = Cout / 64, tiles9 = (Cin / 64) * tiles_n9;
    const int lw = ilog2d(d->W), lh = ilog2d(d->H);
    const bool p2 = lw >= 0 && lh >= 0;
    const unsigned short* dys = (const unsigned short*)dy;
#define RPNET_W9D(NPL, P2)                                                                                                 \
    hipLaunchKernelGGL((conv_wgrad9_dma_kernel<NPL, P2>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout, \
                       tiles9, tiles_n9, ks9, sps9, P2 ? lw : 0, P2 ? lh : 0)
    if (d->split_planes == 2) { if (p2) RPNET_W9D(2, true); else RPNET_W9D(2, false); }
    else {
        set_error("conv_wgrad9_split_dma: two fp16 planes only");
        return RPNET_ERR_ARG;
    }
#undef RPNET_W9D
    return check_launch("conv_wgrad9_split_dma");
}

}  // namespace rpnet
