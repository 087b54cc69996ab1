// Weight gradient of the 3x3 convolutions on fp16 planes, round 6: conv_wgrad9_dma_kernel (conv_wgrad_split_dma.hip) with its K
// dimension walked DOWN THE IMAGE COLUMNS, so that every x strip is fetched once instead of three times:
//   dWp[z][tap][cin][cout] = sum over the block's pixels q of x[q + (ky - 1) W][cin] * dy[q - (kx - 1)][cout]
// (autograd of nn.Conv2d wrt its weight, net/modules.py:47-54,66-69; GEMM M = cin, N = cout, K = pixels; all nine taps of a
// 64 x 64 tile in one block; split-K over ranges of K-steps z).
//   * A K-step is 32 pixels of ONE image row (K64: 64 pixels, the two halves in the two plane slots).  Steps are numbered
//     column-major: step s = (column, y), column = (image, 32-pixel block of the row), y = image row.  The tap rows ky = -1 / 0 / +1
//     of step (c, y) are the strips of image rows y - 1 / y / y + 1 of the same column — the centre strips of steps s - 1, s, s + 1.
//     A RING of eight strip slots (slot = s & 7) therefore receives ONE new strip per step (that of step s + 4) where the
//     row-major kernel fetched three: 18 instead of 34 LDS-DMA pieces per block and step (a piece costs 60 - 180 issue cycles
//     beside the MFMAs, MI355X_MICROARCH.md "LDS-DMA piece issue cost": the row-major kernel spent 8 - 10 of them per wave on
//     54 MFMAs), and the L2 -> LDS traffic of the x operand is a third.  Rows above / below the image read a strip of zeros
//     (a ninth slot, written once); a split range that begins or ends inside a column fetches the neighbouring row like any other.
//   * Everything else is the row-major kernel's: four waves = quadrants x all nine taps (9 x 16 accumulator registers), both
//     operands [pixel][channel] in LDS rows of 128 bytes whose 64-byte halves are swapped by bit 1 of the row (conflict-free
//     ds_read_b64_tr_b16 AND lane-linear DMA), the dy tile (pixels q0 - 1 .. q0 + 38; the kx taps are the same reads one row up /
//     down, redirected to a row of zeros at the image's left / right border) in a ring of four stages whose index is a compile-
//     time constant, one raw s_barrier per K-step behind a counted s_waitcnt vmcnt, fragments double-buffered, issue order pinned,
//     partial sums through a per-wave LDS transpose as 16-byte stores, wgrad_reduce_kernel behind it.
// Power-of-two images at least one K-step wide, no up-sampling (rpnet_conv_wgrad falls back to conv_wgrad9_dma_kernel otherwise).
// The pixels of a split are summed in another order than the row-major kernel's: equal to rounding, not bit for bit.
#include <type_traits>

#include "common.h"
#include "lds_dma.h"
#include "split_bf16.h"

namespace rpnet {

using s16x4r = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4r lds_s16x4r_t;

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for_r(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_r<N, I + 1>(f);
    }
}

// ABL (tools/wgrad_anatomy.sh; results are then meaningless): 2 = no DMA at all, 3 = no fragment reads, 4 = neither, 5 = DMAs without
// the counted waits, 6 = without waits and barriers
// LA: how many K-steps ahead of their use the DMAs are issued (the group "of step st" = the strip that step st + 1 brings in and the dy
// tile of step st; a group is waited for two steps before its step, so LA - 2 groups stay in flight across every barrier); NSY: dy stages
// (LA mod NSY must not be 0 or 1; the strip ring's eight slots allow LA = 2 .. 5)
// BI (diagnostic, rpnet_conv_desc.tune 18; profiles/r06_pool_fault.txt): the DMAs through __builtin_amdgcn_raw_ptr_buffer_load_lds — as the LDS-DMA
// CONVOLUTION kernels issue them — instead of the inline-asm statement of lds_dma.h.  hipcc then drains vmcnt in front of every transposing
// read (slow): only there to tell whether the pooled-pass fault needs the inline-asm form.
template <bool K64, int ABL = 0, int LA = 5, int NSY = 6, bool BI = false>
__global__ __launch_bounds__(256, 1) void conv_wgrad9_ring_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                   float* __restrict__ partial, const int M, const int Cin,
                                                                   const int Cout, const int tiles, const int tiles_n,
                                                                   const int ksplit, const int steps_per_split, const int lw,
                                                                   const int lh) {
    constexpr int NP = 2, BK = 32, RB = 128;                       // plane slots, pixels per slot and step, bytes per LDS row
    constexpr int A_PLANE = BK * RB, A_STRIP = NP * A_PLANE;       // one strip: [2 slots][32 pixel] rows = 8 KB
    constexpr int NSX = 8, ZOFF = NSX * A_STRIP;                   // ring of eight strips + the strip of zeros: 72 KB
    constexpr int ZROW = 40, B_PLANE = 48 * RB, B_STAGE = NP * B_PLANE;   // dy: 5 pieces of 8 rows + the zero row: 6 KB per slot
    constexpr int NS = NSY, BOFF = ZOFF + A_STRIP;                 // NSY dy stages of 12 KB
    static_assert(LA >= 2 && LA <= 5 && LA % NSY >= 2, "the DMAs of step st + LA must not land in a slot / stage that step st or st + 1 reads");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BOFF + NS * B_STAGE];      // 72 + 72 KB (NSY = 6)
    RPNET_ASSERT_NO_CORESIDENCE(sizeof(smem));
    constexpr int PXS = K64 ? 64 : BK;                             // pixels per K-step

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    // block -> (tile, pixel chunk z) so that the blocks of ONE pixel chunk — which read the same x strips (tiles of one cin row) and dy
    // tiles (one cout column) — sit on as few XCDs (private L2s; the dispatcher places block b on XCD b % 8) as possible:
    // ksplit a multiple of 8: one XCD per chunk (8 chunks at a time); ksplit = 1, 2 or 4 with enough tiles: 8 / ksplit XCDs per chunk,
    // each with a contiguous share of the tiles (round 5 spread such a chunk over all 8: its operands were fetched up to 8 times,
    // 180 MB per launch at 512 channels against 134 MB once)
    int tile, z;
    const int xg = ksplit < 8 ? 8 / max(ksplit, 1) : 0;          // XCDs per chunk
    if ((ksplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int jt = uni(j / tiles);
        z = jt * 8 + xcd;
        tile = j - jt * tiles;
    } else if (xg * ksplit == 8 && tiles % xg == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;     // j = 0 .. tiles / xg - 1
        z = uni(xcd / xg);
        tile = (xcd - z * xg) * (tiles / xg) + j;
    } else {
        z = uni(blockIdx.x / tiles);
        tile = blockIdx.x - z * tiles;
    }
    const int tm = uni(tile / tiles_n), tn = tile - tm * tiles_n;
    const int cm0 = tm * 64, n0 = tn * 64;

    const int H = d.H, W = d.W;
    const unsigned short* src; int Cs, cc;
    if (cm0 < d.C0) { src = reinterpret_cast<const unsigned short*>(d.x0); Cs = d.C0; cc = cm0; }
    else { src = reinterpret_cast<const unsigned short*>(d.x1); Cs = d.C1; cc = cm0 - d.C0; }
    const size_t planex = (size_t)d.N * H * W * Cs, planey = (size_t)M * Cout;
    const int pbx = (int)(planex * 2), pby = (int)(planey * 2);

    const int total_steps = M / PXS;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);
    // step s -> first pixel of its row segment
    const int lxb = lw - (K64 ? 6 : 5), xbm = (1 << lxb) - 1;
    auto q0_of = [&](const int s) {
        const int col = s >> lh, y = s & (H - 1);
        return ((((col >> lxb) << lh) + y) << lw) + (col & xbm) * PXS;
    };

    const srd_t rsx = make_srd(src, (K64 ? 1 : NP) * pbx), rsy = make_srd(dy, (K64 ? 1 : NP) * pby);
    const int x_slot1 = K64 ? BK * (Cs * 2) : pbx, y_slot1 = K64 ? BK * (Cout * 2) : pby;
    const __amdgpu_buffer_rsrc_t bix = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), (short)0, (K64 ? 1 : NP) * pbx, 0x00020000);
    const __amdgpu_buffer_rsrc_t biy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dy), (short)0, (K64 ? 1 : NP) * pby, 0x00020000);
    // one 1 KB DMA piece to LDS byte address `dst` (scalar): the inline-asm statement, or (BI) the builtin
    auto dma16 = [&](const bool is_x, const unsigned base, auto offc, const int voff, const int soff) {
        constexpr int OFF = decltype(offc)::value;                  // (compile-time part of the LDS address: one s_add into M0)
        if constexpr (BI) {
            auto* lp = (__attribute__((address_space(3))) void*)(smem + (base - lds_addr(smem)) + OFF);
            if (is_x) __builtin_amdgcn_raw_ptr_buffer_load_lds(bix, lp, 16, voff, soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(biy, lp, 16, voff, soff, 0, 0);
        } else if constexpr (OFF == 0) {
            lds_dma16(is_x ? rsx : rsy, base, voff, soff);
        } else {
            lds_dma16_at<OFF>(is_x ? rsx : rsy, base, voff, soff);
        }
    };
    const unsigned lds0 = lds_addr(smem);
    const unsigned ldsw = lds0 + 8 * wv * RB;                       // this wave's 8 rows inside a strip / dy piece wv
    const unsigned lds4 = lds0 + 32 * RB + (wv & 1) * B_PLANE;      // dy piece 4: slot 0 by wave 0, slot 1 by wave 1

    // DMA lane geometry (as conv_wgrad9_dma_kernel): lane l of a 1 KB piece writes 16 bytes at piece + 16 l = row (l >> 3),
    // 16-byte slot (l & 7); the slot belongs to the 64-byte half (l >> 2) & 1, which holds the SOURCE half ^ bit 1 of the row
    const int drow = lane >> 3;
    const int dcol = ((((lane >> 2) & 1) ^ ((lane >> 4) & 1)) << 6) | ((lane & 3) << 4);
    const int Cs2 = Cs * 2;
    const int xlane = drow * Cs2 + dcol;
    // the centre strip of step `rs` (clamped to the tensor by the caller) into ring slot `slot`: this wave's rows 8 wv .. 8 wv + 7
    auto dma_x = [&](const int slot, const int rs) {
        if constexpr (ABL == 2 || ABL == 4) return;
        const int soff = (q0_of(rs) + 8 * wv) * Cs2 + cc * 2;
        const unsigned dst = ldsw + (slot & (NSX - 1)) * A_STRIP;
        dma16(true, dst, std::integral_constant<int, 0>{}, xlane, soff);
        dma16(true, dst, std::integral_constant<int, A_PLANE>{}, xlane, soff + x_slot1);
    };
    // dy: piece wv (rows 8 wv ..) of both slots; waves 0 / 1 also the fifth piece (rows 32 .. 39) of slot 0 / 1; pixels before 0
    // or past M read as zeros (an offset beyond num_records; the planes share one descriptor, so the bound is checked here)
    const int ylane = drow * (Cout * 2) + dcol, ylane_m1 = ylane - Cout * 2;       // (lane 0 of ylane_m1 is negative: out of range)
    auto dma_y = [&](auto stagec, const int st) {
        constexpr int stage = decltype(stagec)::value;
        constexpr int DST = BOFF + stage * B_STAGE;
        if constexpr (ABL == 2 || ABL == 4) return;
        const int rowb = q0_of(st) - 1 + 8 * wv;                    // first pixel of the piece (uniform): -1 for step 0, piece 0
        const bool neg = rowb < 0;
        const int soff = (neg ? 0 : rowb) * (Cout * 2) + n0 * 2;
        int voff = neg ? ylane_m1 : ylane;
        voff = rowb + drow < M ? voff : (int)0x80000000;
        const int soff1 = K64 ? (rowb + BK) * (Cout * 2) + n0 * 2 : soff + y_slot1;
        const int voff1 = K64 ? (rowb + BK + drow < M ? ylane : (int)0x80000000) : voff;
        dma16(false, ldsw, std::integral_constant<int, DST>{}, voff, soff);
        dma16(false, ldsw, std::integral_constant<int, DST + B_PLANE>{}, voff1, soff1);
    };
    auto dma_y5 = [&](auto stagec, const int st) {                  // waves 0 and 1 only
        constexpr int stage = decltype(stagec)::value;
        constexpr int DST = BOFF + stage * B_STAGE;
        if constexpr (ABL == 2 || ABL == 4) return;
        const int rowb = q0_of(st) + 31 + (K64 ? wv * BK : 0);      // (never negative)
        const int soff = rowb * (Cout * 2) + n0 * 2 + (K64 ? 0 : wv * y_slot1);
        const int voff = rowb + drow < M ? ylane : (int)0x80000000;
        dma16(false, lds4, std::integral_constant<int, DST>{}, voff, soff);
    };
    // the DMAs "of step st": the strip step st + 1 brings in (its own centre row) and the dy tile of step st
    const int xlast = total_steps - 1;
    auto stc = [&](const int st) { return st < s_end ? st : s_end - 1; };       // dy: the tail re-fetches the last step
    auto xsc = [&](const int r) { return r < xlast ? r : xlast; };
    // waves 0 / 1 issue 2 + 2 + 1 DMAs per step, waves 2 / 3 2 + 2.  ONE wait count for all (a wave-dependent immediate is a
    // branch): "at most 4 (LA - 2) outstanding" leaves the newest LA - 2 groups of waves 2 / 3 in flight and makes waves 0 / 1
    // wait for LA - 2 DMAs more of the oldest of them — issued LA - 2 steps earlier, long landed
    auto wait_all_but_one_step = [&]() {
        if constexpr (ABL == 2 || ABL == 4 || ABL >= 5) return;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 2) * 4) : "memory");
    };

    // the strip of zeros and the zero row of every dy slot of every stage (the DMA never writes them)
    {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + ZOFF + t * 16) = zero;
        *reinterpret_cast<u32x4*>(smem + ZOFF + 4096 + t * 16) = zero;
        if (t < NS * NP * 8) {
            const int pl = t >> 3;
            *reinterpret_cast<u32x4*>(smem + BOFF + (pl / NP) * B_STAGE + (pl % NP) * B_PLANE + ZROW * RB + (t & 7) * 16) = zero;
        }
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
    const int a_off = krow * RB + (((wm ^ ((krow >> 1) & 1)) << 6) | cbyte);
    const unsigned char* const aptr = smem + a_off;
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int r = krow + 2 - kx;
        b_off[kx] = BOFF + r * RB + (((wn ^ ((r >> 1) & 1)) << 6) | cbyte);
    }
    const int b_zero = BOFF + ZROW * RB + ((wn << 6) | cbyte);
    auto tr = [&](const unsigned char* p) -> s16x4r {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4r_t*)p);
    };

    // x: the three strips of a step as lane pointers (ring slot or the strip of zeros: a scalar select per step)
    struct ABase { const unsigned char* p[3]; };
    auto a_bases = [&](const int st) {
        const int y = st & (H - 1);
        const int o0 = y > 0 ? ((st - 1) & (NSX - 1)) * A_STRIP : ZOFF;
        const int o2 = y < H - 1 ? ((st + 1) & (NSX - 1)) * A_STRIP : ZOFF;
        ABase b;
        b.p[0] = aptr + o0;
        b.p[1] = aptr + (st & (NSX - 1)) * A_STRIP;
        b.p[2] = aptr + o2;
        return b;
    };

    s16x4r afr[2][3][NP][2], bfr[2][3][NP][2];
    if constexpr (ABL == 3 || ABL == 4) {       // (the fragments are never read: give them defined, varying contents)
#pragma unroll
        for (int i = 0; i < 2 * 3 * NP * 2; ++i) {
            (&afr[0][0][0][0])[i] = s16x4r{(short)(lane + i), (short)(15360 + i), (short)lane, (short)i};
            (&bfr[0][0][0][0])[i] = s16x4r{(short)(lane * 3 + i), (short)(15361 + i), (short)(lane + 7), (short)(i * 5)};
        }
    }
    // [slice][kx][row half]: address of the dy read inside slot 0 of stage 0 (tile row or the zero row)
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
    // a step's pixels lie in one image row: only its first pixel can lack a left neighbour (first block of the row), only its last
    // a right one (last block) — (K64: the first pixel of the step is in slot 0, the last in slot 1: read_frag takes bsel there only)
    auto b_addr = [&](auto sc, const int st) {
        constexpr int s = decltype(sc)::value;
        const int xb = (st >> lh) & xbm;
        if constexpr (s == 0) bsel[0][2][0] = (xb == 0 && lane_first) ? bzero : bconst[0][2][0];           // kx = +1: dy[q - 1]
        else bsel[1][0][1] = (xb == xbm && lane_last) ? bzero : bconst[1][0][1];                            // kx = -1: dy[q + 1]
    };
    constexpr int NR = NP * 12, NMMA = (K64 ? 2 : nprod<NP>()) * 9;
    auto read_frag = [&](auto sc, auto kc, auto stagec, const ABase& ab) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value, stage = decltype(stagec)::value;
        constexpr int grp = k / 12, r = (k - grp * 12) >> 1, e = k & 1;      // plane pair, fragment of the pair, row half
        constexpr int pa = K64 ? grp : NP - 1 - grp, pb = grp;
        if constexpr (ABL == 3 || ABL == 4) return;
        if constexpr (r == 0 || r >= 4) {
            constexpr int ky = r == 0 ? 0 : r - 3;
            afr[s][ky][pa][e] = tr(ab.p[ky] + (pa * A_PLANE + (16 * s + 4 * e) * RB));
        } else {
            constexpr int kx = r - 1;
            constexpr bool sel = !K64 || (pb == 0 ? (s == 0 && kx == 2 && e == 0) : (s == 1 && kx == 0 && e == 1));
            bfr[s][kx][pb][e] = tr((sel ? bsel[s][kx][e] : bconst[s][kx][e]) + (stage * B_STAGE + pb * B_PLANE));
        }
    };
    auto frag = [](const s16x4r lo, const s16x4r hi) {
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
        // strips s_begin - 1 (the row above the range's first, when the range begins inside a column), s_begin, then the DMA groups of
        // steps s_begin .. s_begin + LA - 1
        dma_x(s_begin - 1, s_begin > 0 ? s_begin - 1 : 0);
        dma_x(s_begin, s_begin);
        static_for_r<LA>([&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            dma_x(s_begin + gi + 1, xsc(s_begin + gi + 1));
            dma_y(std::integral_constant<int, gi % NSY>{}, stc(s_begin + gi));
            if (wv < 2) dma_y5(std::integral_constant<int, gi % NSY>{}, stc(s_begin + gi));
        });
        wait_all_but_one_step();                       // steps 0 and 1 have landed (this wave's part); zero rows written
        __builtin_amdgcn_s_barrier();
        ABase ab = a_bases(s_begin);
        b_addr(I0{}, s_begin);
        b_addr(I1{}, s_begin);
        static_for_r<NR>([&](auto kc) { read_frag(I0{}, kc, I0{}, ab); });

        // ---- the K loop.  Its scalar bookkeeping is INCREMENTAL and spread over the gaps between MFMAs, at most four single-issue
        // instructions per gap beside the gap's fragment read (one wave per SIMD hides about five per 32-cycle MFMA,
        // MI355X_MICROARCH.md "single-issue instructions HIDDEN per v_mfma gap"; the first form of this kernel computed every
        // address from the step number where it was needed — 17 to 23 scalar instructions in ONE gap, three times per step — and
        // was no faster than the row-major kernel with twice its DMAs: profiles/r06_wgrad_anatomy.txt).
        // Prefetch state: group g = st + LA = the dy tile of step g (first pixel qa) and the centre strip of step g + 1 (first
        // pixel qb, in image row yb and row block xbb, going to ring offset xslot).  Groups past the range's (or the tensor's) end are
        // fetched like any other and never read; offsets past the tensor read zeros (the descriptor's bound).
        const int g0 = s_begin + LA;
        int qa = q0_of(g0), qb = q0_of(g0 + 1);
        int yb = (g0 + 1) & (H - 1), xbb = ((g0 + 1) >> lh) & xbm;
        int xslot = ((g0 + 1) & (NSX - 1)) * A_STRIP;
        int cslot = (s_begin & (NSX - 1)) * A_STRIP;               // ring offset of the centre strip of the step being computed
        const int cx = 8 * wv * Cs2 + cc * 2;
        const int Cout2 = Cout * 2, c8 = 8 * wv - 1, cy = n0 * 2;
        const int c5 = 31 + (K64 ? wv * BK : 0), cy5 = cy + (K64 ? 0 : wv * y_slot1);
        const int col_next = PXS - W - (H - 1) * W, col_wrap = PXS - W;       // after q += W: to the next block of the row / the next image
        int sx = 0, sy = 0, sy5 = 0, rowb = 0, vy = ylane, y1 = 0, o0 = 0, o1 = 0, o2 = 0;
        unsigned dstx = 0;
        constexpr bool DMA_ON = !(ABL == 2 || ABL == 4);
        auto xa = [&]() { sx = qb * Cs2 + cx; dstx = ldsw + xslot; };
        auto xd0 = [&]() { if constexpr (DMA_ON) dma16(true, dstx, std::integral_constant<int, 0>{}, xlane, sx); };
        auto xd1 = [&]() { if constexpr (DMA_ON) dma16(true, dstx, std::integral_constant<int, A_PLANE>{}, xlane, sx + x_slot1); };
        auto ya1 = [&]() {                                            // first pixel of this wave's piece: -1 for step 0, piece 0
            rowb = qa + c8;
            vy = rowb < 0 ? ylane_m1 : ylane;
        };
        auto ya2 = [&]() { sy = max(rowb, 0) * Cout2 + cy; };
        auto yd0 = [&](auto stagec) {
            if constexpr (DMA_ON) dma16(false, ldsw, std::integral_constant<int, BOFF + decltype(stagec)::value * B_STAGE>{}, vy, sy);
        };
        auto yd1 = [&](auto stagec) {
            constexpr int DST = BOFF + decltype(stagec)::value * B_STAGE + B_PLANE;
            if constexpr (!DMA_ON) return;
            if constexpr (K64) dma16(false, ldsw, std::integral_constant<int, DST>{}, ylane, (rowb + BK) * Cout2 + cy);
            else dma16(false, ldsw, std::integral_constant<int, DST>{}, vy, sy + y_slot1);
        };
        auto y5a = [&]() { sy5 = (qa + c5) * Cout2 + cy5; };
        auto y5d = [&](auto stagec) {
            if constexpr (DMA_ON)
                if (wv < 2) dma16(false, lds4, std::integral_constant<int, BOFF + decltype(stagec)::value * B_STAGE>{}, ylane, sy5);
        };
        auto qs = [&]() { qa = qb; xslot = (xslot + A_STRIP) & (NSX * A_STRIP - 1); };
        auto adv = [&]() {
            qb += W;
            yb += 1;
            if (yb == H) {           // (uniform: once per image column)
                const bool last = xbb == xbm;
                yb = 0;
                qb += last ? col_wrap : col_next;
                xbb = last ? 0 : xbb + 1;
            }
        };
        // the strips of step st + 1 as lane pointers
        auto ab1 = [&](const int st) { y1 = (st + 1) & (H - 1); o1 = (cslot + A_STRIP) & (NSX * A_STRIP - 1); };
        auto ab2 = [&]() { o0 = y1 > 0 ? cslot : ZOFF; o2 = y1 < H - 1 ? ((o1 + A_STRIP) & (NSX * A_STRIP - 1)) : ZOFF; };
        auto ab3 = [&]() { ab.p[0] = aptr + o0; ab.p[1] = aptr + o1; ab.p[2] = aptr + o2; cslot = o1; };
        constexpr int M0 = K64 ? 6 : 1;                // (K64: the first six gaps carry two fragment reads each)
        // one K-step on dy stage K (compile time); the DMAs of step st + LA go to dy stage K + LA and strip slot (st + LA + 1) & 7
        auto step = [&](auto kc4, const int st) {
            constexpr int K = decltype(kc4)::value;
            using SK = std::integral_constant<int, K>;
            using SN = std::integral_constant<int, (K + 1) % NSY>;
            using SD = std::integral_constant<int, (K + LA) % NSY>;
            // first half: MFMAs of slice 0 | reads of slice 1 | the strip of step st + LA + 1 | the strip pointers of step st + 1
            static_for_r<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I0{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR) {
                    read_frag(I1{}, mc, SK{}, ab);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NR > NMMA && m < NR - NMMA) {      // K64: 24 reads behind 18 MFMAs
                    read_frag(I1{}, std::integral_constant<int, NMMA + m>{}, SK{}, ab);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == M0) xa();
                if constexpr (m == M0 + 2) xd0();
                if constexpr (m == M0 + 4) xd1();
                if constexpr (m == NMMA - 3) ab1(st);
                if constexpr (m == NMMA - 2) ab2();
                if constexpr (m == NMMA - 1) { ab3(); b_addr(I0{}, st + 1); }
                if constexpr (m == M0 || m == M0 + 2 || m == M0 + 4 || m >= NMMA - 3) __builtin_amdgcn_sched_barrier(0);
            });
            // second half: MFMAs of slice 1 | reads of step st + 1 / slice 0 (visible since the previous barrier) | dy of st + LA
            static_for_r<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I1{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR) {
                    read_frag(I0{}, mc, SN{}, ab);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NR > NMMA && m < NR - NMMA) {
                    read_frag(I0{}, std::integral_constant<int, NMMA + m>{}, SN{}, ab);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == M0) ya1();
                if constexpr (m == M0 + 1) ya2();
                if constexpr (m == M0 + 2) yd0(SD{});
                if constexpr (m == M0 + 4) yd1(SD{});
                if constexpr (m == M0 + 5) y5a();
                if constexpr (m == M0 + 6) y5d(SD{});
                if constexpr (m == M0 + 8) qs();
                if constexpr (m == M0 + 9) adv();
                if constexpr (m == NMMA - 1) b_addr(I1{}, st + 1);
                if constexpr ((m >= M0 && m <= M0 + 9) || m == NMMA - 1) __builtin_amdgcn_sched_barrier(0);
            });
            wait_all_but_one_step();                   // this wave's part of step st + 2 has landed (LA - 2 groups stay in flight)
            if constexpr (ABL != 6) __builtin_amdgcn_s_barrier();
        };
        for (int st = s_begin; st < s_end; st += NSY) {
            bool done = false;
            static_for_r<NSY>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (!done) {
                    step(kc, st + k);
                    done = st + k + 1 >= s_end;
                }
            });
            if (done) break;
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

static int ilog2r(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// the column-major kernel takes: power-of-two images at least a K-step wide (32 pixels on two planes, 64 on one), no up-sampling,
// whole K-steps (on one plane: an even number of 32-pixel steps per split, as conv_wgrad9_dma_one_plane_ok)
bool conv_wgrad9_ring_ok(const rpnet_conv_desc* d, int M, int sps9) {
    const int lw = ilog2r(d->W), lh = ilog2r(d->H);
    if (lw < 0 || lh < 0 || d->upsample) return false;
    if (d->split_planes == 2) return d->W >= 32;
    return d->split_planes == 1 && d->W >= 64 && M % 64 == 0 && sps9 % 2 == 0;
}

// launches the column-major nine-tap kernel into `part9` ([ksplit][9][Cin][Cout] fp32, plan = wgrad9_plan)
int conv_wgrad9_ring(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9,
                     hipStream_t s) {
    const int tiles_n9 = Cout / 64, tiles9 = (Cin / 64) * tiles_n9;
    const int lw = ilog2r(d->W), lh = ilog2r(d->H);
    const unsigned short* dys = (const unsigned short*)dy;
    const int abl = (d->tune >> 8) & 7;
#define RPNET_W9R(K6, A, SPS)                                                                                                   \
    hipLaunchKernelGGL((conv_wgrad9_ring_kernel<K6, A>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout, tiles9, \
                       tiles_n9, ks9, SPS, lw, lh)
    if (d->split_planes == 2 && (d->tune & 255) == 18) {        // diagnostic: DMAs through the builtin (profiles/r06_pool_fault.txt)
        hipLaunchKernelGGL((conv_wgrad9_ring_kernel<false, 0, 5, 6, true>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout, tiles9,
                           tiles_n9, ks9, sps9, lw, lh);
    } else if (d->split_planes == 2 && (d->tune & 255) == 17) {       // A/B: the shallow pipeline of the row-major kernel (three steps ahead, four dy stages)
        hipLaunchKernelGGL((conv_wgrad9_ring_kernel<false, 0, 3, 4>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dys, part9, M, Cin, Cout, tiles9,
                           tiles_n9, ks9, sps9, lw, lh);
    } else if (d->split_planes == 2) {
        if (abl == 2) RPNET_W9R(false, 2, sps9);
        else if (abl == 3) RPNET_W9R(false, 3, sps9);
        else if (abl == 4) RPNET_W9R(false, 4, sps9);
        else if (abl == 5) RPNET_W9R(false, 5, sps9);
        else if (abl == 6) RPNET_W9R(false, 6, sps9);
        else RPNET_W9R(false, 0, sps9);
    } else {
        RPNET_W9R(true, 0, sps9 / 2);
    }
#undef RPNET_W9R
    return check_launch("conv_wgrad9_ring");
}

}  // namespace rpnet
