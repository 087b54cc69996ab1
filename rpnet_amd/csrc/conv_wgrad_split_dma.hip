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

// FAST: power-of-two images at least a K-step wide, no up-sampling (25 of the 27 weight-gradient launches of the step): the
// addressing of a step collapses to a handful of scalar instructions (below).  POW2 (without FAST): shifts instead of divisions.
// K64: ONE fp16 plane (the f16 arithmetic of BASELINE configs[4]) in the same LDS geometry, as in conv_split_dma.hip: a
// K-step is 64 pixels of one image row, whose two 32-pixel halves take the places of the two planes; the products are the
// diagonal ones (half p of x with half p of dy): 18 MFMAs per wave and slice for the same 24 fragment reads.
// ABL (tools/wgrad_anatomy.sh; results are then meaningless): 1 = only the centre x strip is fetched (the DMA count of a strip
// ring), 2 = no DMA at all, 3 = no fragment reads, 4 = neither (MFMAs, barriers and waits only)
template <int NP, bool POW2, bool FAST, bool K64 = false, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_wgrad9_dma_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                  float* __restrict__ partial, const int M, const int Cin,
                                                                  const int Cout, const int tiles, const int tiles_n,
                                                                  const int ksplit, const int steps_per_split, const int lw,
                                                                  const int lh) {
    constexpr int BK = 32, RB = 128;                               // pixels per K-step, bytes per LDS row (64 channels)
    constexpr int A_PLANE = 3 * BK * RB, A_STAGE = NP * A_PLANE;   // [3 ky][32 pixel] rows: 12 KB per plane
    constexpr int ZROW = 40, B_PLANE = 48 * RB, B_STAGE = NP * B_PLANE;   // 5 pieces of 8 rows + the zero row: 6 KB per plane
    constexpr int NS = 4;
    // LDS: [4 stages of x strips: 96 KB][4 stages of dy tiles: 48 KB].  The stage of a K-step is a COMPILE-TIME constant (the
    // loop is unrolled by four), so every fragment read is a lane-constant address register + an immediate offset: the x
    // reads go through two base registers (stages 0-1 / 2-3: immediates stay below 64 KB), the dy reads through the twelve
    // (slice, tap column, row half) registers that also carry the image-border redirects.
    constexpr int BOFF = NS * A_STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BOFF + NS * B_STAGE];
    RPNET_ASSERT_NO_CORESIDENCE(sizeof(smem));
    static_assert(NP == 2, "two plane slots: two fp16 planes, or the two halves of a 64-pixel step of one plane (K64)");
    static_assert(!K64 || FAST, "the one-plane form needs a whole 64-pixel step in one image row");
    constexpr int PXS = K64 ? 64 : BK;                              // pixels per K-step

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    int tile, z;
    if ((ksplit & 7) == 0) {       // the blocks of one pixel chunk share an XCD (and its L2)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int jt = uni(j / tiles);
        z = jt * 8 + xcd;
        tile = j - jt * tiles;
    } else {
        z = uni(blockIdx.x / tiles);
        tile = blockIdx.x - z * tiles;
    }
    const int tm = uni(tile / tiles_n), tn = tile - tm * tiles_n;
    const int cm0 = tm * 64, n0 = tn * 64;

    const int H = d.H, W = d.W, HW = H * W;
    const int ups = FAST ? 0 : d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const unsigned short* src; int Cs, cc;
    if (cm0 < d.C0) { src = reinterpret_cast<const unsigned short*>(d.x0); Cs = d.C0; cc = cm0; }
    else { src = reinterpret_cast<const unsigned short*>(d.x1); Cs = d.C1; cc = cm0 - d.C0; }
    const size_t planex = (size_t)d.N * Hs * Ws * Cs, planey = (size_t)M * Cout;
    const int pbx = (int)(planex * 2), pby = (int)(planey * 2);

    const int total_steps = (M + PXS - 1) / PXS;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    // one descriptor per tensor over all its planes (the plane is part of the scalar offset)
    const srd_t rsx = make_srd(src, (K64 ? 1 : NP) * pbx), rsy = make_srd(dy, (K64 ? 1 : NP) * pby);
    // the second plane slot: the other plane, or (K64) the next 32 pixels of the one plane
    const int x_slot1 = K64 ? BK * (Cs * 2) : pbx, y_slot1 = K64 ? BK * (Cout * 2) : pby;
    const unsigned lds0 = lds_addr(smem);
    const unsigned ldsw = lds0 + 8 * wv * RB;                       // this wave's 8 rows inside a strip / dy piece wv
    const unsigned lds4 = lds0 + 32 * RB;                           // dy piece 4 (wave 0)

    // DMA lane geometry: lane l of a 1 KB piece writes 16 bytes at piece + 16 l = row (l >> 3), 16-byte slot (l & 7); the
    // slot belongs to the 64-byte half (l >> 2) & 1, which holds the SOURCE half ^ bit 1 of the row (pieces start at
    // multiples of 8 rows)
    const int drow = lane >> 3;
    const int dcol = ((((lane >> 2) & 1) ^ ((lane >> 4) & 1)) << 6) | ((lane & 3) << 4);       // source byte offset inside the row
    const int xlane = (drow >> ups) * (Cs * 2) + dcol;              // per-lane part of an x source address (POW2)
    const int Cs2 = Cs * 2, wcs = W * Cs2;
    // x strips: this wave moves rows 8 wv .. 8 wv + 7 of each of the three strips (every plane).  POW2: the 8 pixels of a
    // piece lie in one image row, so validity and source pixel are wave-uniform (scalar unit; the lane part is a constant)
    auto dma_x = [&](auto kyc, auto stagec, const int st) {
        constexpr int kyi = decltype(kyc)::value, stage = decltype(stagec)::value;
        constexpr int DST = stage * A_STAGE + kyi * BK * RB;
        if constexpr (ABL == 2 || ABL == 4 || (ABL == 1 && kyi != 1)) return;
        int voff, soff;
        if constexpr (FAST) {
            const int qb0 = st * PXS + 8 * wv;                        // centre-row pixel of the piece; source = qb0 + (ky - 1) W
            const int yc = (qb0 >> lw) & (H - 1);
            const bool ok = kyi == 1 || (kyi == 0 ? yc >= 1 : yc <= H - 2);
            soff = qb0 * Cs2 + cc * 2 + (kyi - 1) * wcs;
            soff = ok ? soff : 0;
            voff = ok ? xlane : (int)0x80000000;
        } else if constexpr (POW2) {
            const int qb = st * BK + 8 * wv + (kyi - 1) * W;
            const int yq = (qb >> lw) & (H - 1);
            const bool ok = (unsigned)qb < (unsigned)M && (unsigned)(yq - (kyi - 1)) < (unsigned)H;
            const int pixb = ((qb >> (lw + lh)) * Hs + (yq >> ups)) * Ws + ((qb & (W - 1)) >> ups);
            soff = ok ? pixb * Cs2 + cc * 2 : 0;
            voff = ok ? xlane : (int)0x80000000;
        } else {
            const int q = st * BK + 8 * wv + drow + (kyi - 1) * W;
            const int n = q / HW, rem = q - n * HW;
            const int yq = rem / W;
            const int xq = rem - yq * W;
            const int pix = (n * Hs + (yq >> ups)) * Ws + (xq >> ups);
            const bool ok = (unsigned)q < (unsigned)M && (unsigned)(yq - (kyi - 1)) < (unsigned)H;
            voff = ok ? pix * Cs2 + dcol : (int)0x80000000;
            soff = cc * 2;
        }
        lds_dma16_at<DST>(rsx, ldsw, voff, soff);
        lds_dma16_at<DST + A_PLANE>(rsx, ldsw, voff, soff + x_slot1);      // (an invalid row stays invalid: voff is out of range)
    };
    // dy: piece wv (rows 8 wv ..), wave 0 also the fifth piece (rows 32 .. 39); pixels before 0 or past M read as zeros (an
    // offset beyond num_records; the planes share one descriptor, so the bound is checked here)
    const int ylane = drow * (Cout * 2) + dcol, ylane_m1 = ylane - Cout * 2;       // (lane 0 of ylane_m1 is negative: out of range)
    auto dma_y = [&](auto stagec, const bool fifth, const int st) {
        constexpr int stage = decltype(stagec)::value;
        constexpr int DST = BOFF + stage * B_STAGE;
        if constexpr (ABL == 2 || ABL == 4) return;
        const int rowb = st * PXS - 1 + (fifth ? 32 : 8 * wv);      // first pixel of the piece (uniform): -1 for step 0, piece 0
        const bool neg = rowb < 0;
        const int soff = (neg ? 0 : rowb) * (Cout * 2) + n0 * 2;
        int voff = neg ? ylane_m1 : ylane;
        voff = rowb + drow < M ? voff : (int)0x80000000;
        // second slot: the other plane at the same pixels, or (K64) the piece 32 pixels on (never before pixel 0)
        const int soff1 = K64 ? (rowb + BK) * (Cout * 2) + n0 * 2 : soff + y_slot1;
        const int voff1 = K64 ? (rowb + BK + drow < M ? ylane : (int)0x80000000) : voff;
        if (fifth) {
            lds_dma16_at<DST>(rsy, lds4, voff, soff);
            lds_dma16_at<DST + B_PLANE>(rsy, lds4, voff1, soff1);
        } else {
            lds_dma16_at<DST>(rsy, ldsw, voff, soff);
            lds_dma16_at<DST + B_PLANE>(rsy, ldsw, voff1, soff1);
        }
    };
    auto dma_step = [&](auto stagec, const int st) {
        dma_x(std::integral_constant<int, 0>{}, stagec, st);
        dma_x(std::integral_constant<int, 1>{}, stagec, st);
        dma_x(std::integral_constant<int, 2>{}, stagec, st);
        dma_y(stagec, false, st);
        if (wv == 0) dma_y(stagec, true, st);
    };
    // wave 0 issues (3 + 2) NP DMAs per step, the others (3 + 1) NP: "everything but the last step's" as a wait count
    auto wait_all_but_one_step = [&]() {
        if constexpr (ABL == 2 || ABL == 4) return;
        if constexpr (ABL == 1) {
            if (wv == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            return;
        }
        if (wv == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };

    // the zero row of every dy plane of every stage (the DMA never writes it)
    if (t < NS * NP * 8) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int pl = t >> 3;
        *reinterpret_cast<u32x4*>(smem + BOFF + (pl / NP) * B_STAGE + (pl % NP) * B_PLANE + ZROW * RB + (t & 7) * 16) = zero;
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
    const unsigned char* const aptr0 = smem + a_off;
    const unsigned char* const aptr1 = smem + a_off + 2 * A_STAGE;
    // dy: tile row of pixel q for tap kx is (q - p0) + 2 - kx: bit 1 of the row index depends on the shift
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int r = krow + 2 - kx;
        b_off[kx] = BOFF + r * RB + (((wn ^ ((r >> 1) & 1)) << 6) | cbyte);
    }
    const int b_zero = BOFF + ZROW * RB + ((wn << 6) | cbyte);
    auto tr = [&](const unsigned char* p) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
    };

    // fragment double buffer [slice parity]: af[ky][plane], bf[kx][plane]; a fragment = two transposing reads (rows krow.. and
    // krow + 4..).  Read k of a slice in order of first use (products l*h, h*l, h*h: the l plane of x and the h plane of dy
    // first): per plane pair x(ky 0), dy(kx 0..2), x(ky 1), x(ky 2) — two reads each.
    s16x4 afr[2][3][NP][2], bfr[2][3][NP][2];
    if constexpr (ABL >= 3) {       // (the fragments are never read: give them defined, varying contents)
#pragma unroll
        for (int i = 0; i < 2 * 3 * NP * 2; ++i) {
            (&afr[0][0][0][0])[i] = s16x4{(short)(lane + i), (short)(15360 + i), (short)lane, (short)i};
            (&bfr[0][0][0][0])[i] = s16x4{(short)(lane * 3 + i), (short)(15361 + i), (short)(lane + 7), (short)(i * 5)};
        }
    }
    // [slice][kx][row half]: address of the dy read inside plane 0 of stage 0 (tile row or the zero row)
    const unsigned char* bsel[2][3][2];
    const unsigned char* bconst[2][3][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int e = 0; e < 2; ++e) bsel[s2][kx][e] = bconst[s2][kx][e] = smem + b_off[kx] + (16 * s2 + 4 * e) * RB;
    const unsigned char* const bzero = smem + b_zero;
    const bool lane_first = krow == 0, lane_last = krow == 11;       // tile rows 0 (slice 0, half 0) and 31 (slice 1, half 1)
    auto b_addr = [&](auto sc, const int st) {
        constexpr int s = decltype(sc)::value;
        if constexpr (FAST) {
            // a step's 32 pixels lie in one image row: only its first pixel can lack a left neighbour, only its last a right one
            // (K64: the first pixel of the step is in slot 0, the last in slot 1: read_frag takes bsel for that slot only)
            const int x0 = (st * PXS) & (W - 1);
            if constexpr (s == 0) bsel[0][2][0] = (x0 == 0 && lane_first) ? bzero : bconst[0][2][0];           // kx = +1: dy[q - 1]
            else bsel[1][0][1] = (x0 == W - PXS && lane_last) ? bzero : bconst[1][0][1];                       // kx = -1: dy[q + 1]
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int row = 16 * s + 4 * e + krow;
                const int q = st * PXS + row;
                const int ox = POW2 ? (q & (W - 1)) : (q % W);
                bsel[s][0][e] = ox <= W - 2 ? bconst[s][0][e] : bzero;     // kx = -1: dy[q + 1]
                bsel[s][2][e] = ox >= 1 ? bconst[s][2][e] : bzero;         // kx = +1: dy[q - 1]
            }
        }
    };
    constexpr int NR = NP * 12, NMMA = (K64 ? 2 : nprod<NP>()) * 9;
    auto read_frag = [&](auto sc, auto kc, auto stagec) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value, stage = decltype(stagec)::value;
        constexpr int grp = k / 12, r = (k - grp * 12) >> 1, e = k & 1;      // plane pair, fragment of the pair, row half
        constexpr int pa = K64 ? grp : NP - 1 - grp, pb = grp;
        if constexpr (ABL >= 3) return;
        if constexpr (r == 0 || r >= 4) {
            constexpr int ky = r == 0 ? 0 : r - 3;
            constexpr int off = (stage & 1) * A_STAGE + pa * A_PLANE + (ky * BK + 16 * s + 4 * e) * RB;
            afr[s][ky][pa][e] = tr((stage >> 1 ? aptr1 : aptr0) + off);
        } else {
            constexpr int kx = r - 1;
            // K64: the left-border redirect belongs to slot 0 (first pixel of the step), the right-border one to slot 1
            constexpr bool sel = !K64 || (pb == 0 ? (s == 0 && kx == 2 && e == 0) : (s == 1 && kx == 0 && e == 1));
            bfr[s][kx][pb][e] = tr((sel ? bsel[s][kx][e] : bconst[s][kx][e]) + (stage * B_STAGE + pb * B_PLANE));
        }
    };
    auto frag = [](const s16x4 lo, const s16x4 hi) {
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma_one = [&](auto sc, auto mc) {
        constexpr int s = decltype(sc)::value, m = decltype(mc)::value;
        constexpr int q = m / 9, tap = m - q * 9, ky = tap / 3, kx = tap - ky * 3;
        constexpr int pa = K64 ? q : prod_a<NP>(q), pb = K64 ? q : prod_b<NP>(q);
        acc[tap] = mma16<NP>(frag(afr[s][ky][pa][0], afr[s][ky][pa][1]), frag(bfr[s][kx][pb][0], bfr[s][kx][pb][1]), acc[tap]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if (s_begin < s_end) {
        // clamp: the tail re-fetches the last step into stages nobody reads again (uniform DMA counts per wave)
        auto stc = [&](int st) { return st < s_end ? st : s_end - 1; };
        dma_step(std::integral_constant<int, 0>{}, s_begin);
        dma_step(std::integral_constant<int, 1>{}, stc(s_begin + 1));
        dma_step(std::integral_constant<int, 2>{}, stc(s_begin + 2));
        wait_all_but_one_step();                       // steps 0 and 1 have landed (this wave's part); zero rows written
        __builtin_amdgcn_s_barrier();
        b_addr(I0{}, s_begin);
        static_for_w<NR>([&](auto kc) { read_frag(I0{}, kc, I0{}); });
        // one K-step on stage K (compile time): Ring of four stages: the DMAs of step st + 3 go to stage K + 3
        auto step = [&](auto kc4, const int st) {
            constexpr int K = decltype(kc4)::value;
            using SK = std::integral_constant<int, K>;
            using SN = std::integral_constant<int, (K + 1) & 3>;
            using SD = std::integral_constant<int, (K + 3) & 3>;
            const int dst_step = stc(st + 3);
            // first half: MFMAs of slice 0 | reads of slice 1 | the x strips of step st + 3
            b_addr(I1{}, st);
            __builtin_amdgcn_sched_barrier(0);
            static_for_w<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I0{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR) {
                    read_frag(I1{}, mc, SK{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NR > NMMA && m < NR - NMMA) {      // K64: 24 reads behind 18 MFMAs
                    read_frag(I1{}, std::integral_constant<int, NMMA + m>{}, SK{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 3 || m == 9 || m == 15) {
                    dma_x(std::integral_constant<int, (m - 3) / 6>{}, SD{}, dst_step);
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
                    read_frag(I0{}, mc, SN{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NR > NMMA && m < NR - NMMA) {
                    read_frag(I0{}, std::integral_constant<int, NMMA + m>{}, SN{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 5) {
                    dma_y(SD{}, false, dst_step);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 11) {
                    if (wv == 0) dma_y(SD{}, true, dst_step);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            wait_all_but_one_step();                   // this wave's part of step st + 2 has landed
            __builtin_amdgcn_s_barrier();
        };
        for (int st = s_begin; st < s_end; st += 4) {
            step(std::integral_constant<int, 0>{}, st);
            if (st + 1 >= s_end) break;
            step(std::integral_constant<int, 1>{}, st + 1);
            if (st + 2 >= s_end) break;
            step(std::integral_constant<int, 2>{}, st + 2);
            if (st + 3 >= s_end) break;
            step(std::integral_constant<int, 3>{}, st + 3);
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

// one fp16 plane through the DMA kernel (K64): power-of-two images at least 64 pixels wide, no up-sampling, whole 64-pixel
// steps per split
bool conv_wgrad9_dma_one_plane_ok(const rpnet_conv_desc* d, int M, int sps9) {
    return d->split_planes == 1 && !d->upsample && d->W >= 64 && ilog2d(d->W) >= 0 && ilog2d(d->H) >= 0 && M % 64 == 0 && sps9 % 2 == 0;
}

// launches the nine-tap DMA kernel into `part9` ([ksplit][9][Cin][Cout] fp32, plan = wgrad9_plan); two fp16 planes, or one
// where conv_wgrad9_dma_one_plane_ok
int conv_wgrad9_split_dma(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9,
                          hipStream_t s) {
    const int tiles_n9 = Cout / 64, tiles9 = (Cin / 64) * tiles_n9;
    const int lw = ilog2d(d->W), lh = ilog2d(d->H);
    // POW2: the scalar addressing needs the 8 pixels of a DMA piece in ONE image row (W a power of two >= 8); narrower images
    // (the 4 x 4 level of a 64 x 64 episode) take the per-lane path
    const bool p2 = lw >= 3 && lh >= 0;
    const unsigned short* dys = (const unsigned short*)dy;
    const bool fast = p2 && d->W >= 32 && !d->upsample;
#define RPNET_W9D(NPL, P2, FA)                                                                                                 \
    hipLaunchKernelGGL((conv_wgrad9_dma_kernel<NPL, P2, FA>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout, \
                       tiles9, tiles_n9, ks9, sps9, P2 ? lw : 0, P2 ? lh : 0)
    const int abl = (d->tune >> 8) & 7;
    if (d->split_planes == 2 && fast && abl) {
#define RPNET_W9A(A)                                                                                                           \
    hipLaunchKernelGGL((conv_wgrad9_dma_kernel<2, true, true, false, A>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, \
                       Cout, tiles9, tiles_n9, ks9, sps9, lw, lh)
        if (abl == 1) RPNET_W9A(1); else if (abl == 2) RPNET_W9A(2); else if (abl == 3) RPNET_W9A(3); else RPNET_W9A(4);
#undef RPNET_W9A
        return check_launch("conv_wgrad9_split_dma (ablation)");
    }
    if (d->split_planes == 2) { if (fast) RPNET_W9D(2, true, true); else if (p2) RPNET_W9D(2, true, false); else RPNET_W9D(2, false, false); }
    else if (conv_wgrad9_dma_one_plane_ok(d, M, sps9)) {
        hipLaunchKernelGGL((conv_wgrad9_dma_kernel<2, true, true, true>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout,
                           tiles9, tiles_n9, ks9, sps9 / 2, lw, lh);
    } else {
        set_error("conv_wgrad9_split_dma: two fp16 planes, or one on images of >= 64 pixels per row");
        return RPNET_ERR_ARG;
    }
#undef RPNET_W9D
    return check_launch("conv_wgrad9_split_dma");
}

}  // namespace rpnet
