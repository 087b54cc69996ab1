// Weight gradient of up_conv (nn.Upsample(scale_factor=2) -> Conv2d 3x3, net/modules.py:61-75) on the collapsed form of
// conv_up4_dma.hip: per output phase (py, px) = (Y & 1, X & 1) the layer is a 2 x 2-tap convolution of the LOW-resolution input,
//     dWc[(py, px)][r][c][ci][co] = sum over low-resolution pixels q = (n, Y, X) of x[n, Y + py + r - 1, X + px + c - 1][ci] * dy[n, 2Y + py, 2X + px][co],
// and the gradient of an original tap is the sum of the four collapsed taps it was added into:
//     dW[ky][kx] = sum_{py, px} dWc[(py, px)][r(py, ky)][c(px, kx)],   r(0, k) = (k >= 1), r(1, k) = (k >= 2)
// — 16 tap products per low-resolution pixel instead of 36 (= 9 per high-resolution pixel): 4 / 9 of the multiply-adds.
//
// The GEMM kernel is conv_wgrad9_dma_kernel (conv_wgrad_split_dma.hip: 64 x 64 tile, four waves = quadrants, operands
// [pixel][channel] in LDS through the DMA engine, transposing fragment reads, four-stage ring, counted vmcnt + one barrier per
// 32-pixel K-step) on the low-resolution pixel grid with TWO x strips (3x3 tap rows py, py + 1) and TWO dy shifts (tap columns
// px, px + 1) — four accumulators per quadrant — and the dy tile gathered from the high-resolution gradient at stride 2
// (per-lane DMA source addresses).  One phase per block (grid = tiles x K splits x 4 phases); wgrad_up4_reduce_kernel sums the K
// splits and the phases into the state_dict layout.  Power-of-two images at least 8 low-resolution pixels wide (every level of
// the U-Net); anything else stays on the nine-product form (rpnet_conv_wgrad).
#include <type_traits>

#include "common.h"
#include "lds_dma.h"
#include "split_bf16.h"

namespace rpnet {

using s16x4u = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4u lds_s16x4u_t;

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for_wu(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_wu<N, I + 1>(f);
    }
}

// d.N / d.H / d.W: the HIGH-resolution tensor (dy [2 planes][N][H][W][Cout]); x = d.x0 [2 planes][N][H/2][W/2][Cin].
// Ml: low-resolution pixels; lw / lh: log2 of the low-resolution width / height.
// K64: ONE fp16 plane (the f16 arithmetic of BASELINE configs[4]): a K-step is 64 low-resolution pixels whose two 32-pixel halves take
// the places of the two planes; the products are the diagonal ones (half p of x with half p of dy): 8 MFMAs per wave and slice.
template <bool K64>
__global__ __launch_bounds__(256, 1) void conv_wgrad_up4_kernel(const rpnet_conv_desc d, const unsigned short* __restrict__ dy,
                                                                 float* __restrict__ partial, const int Ml, const int Cin, const int Cout,
                                                                 const int tiles, const int tiles_n, const int ksplit,
                                                                 const int steps_per_split, const int lw, const int lh) {
    constexpr int NP = 2, BK = 32, RB = 128;
    constexpr int A_PLANE = 2 * BK * RB, A_STAGE = NP * A_PLANE;      // [2 tap rows][32 pixel] rows: 8 KB per plane
    constexpr int ZROW = 40, B_PLANE = 48 * RB, B_STAGE = NP * B_PLANE;
    constexpr int NS = 4;
    constexpr int BOFF = NS * A_STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BOFF + NS * B_STAGE];
    RPNET_ASSERT_NO_CORESIDENCE(sizeof(smem));

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    // block -> (phase, K split, tile): the blocks of one pixel chunk and phase share an XCD where the split count allows
    const int per_phase = tiles * ksplit;
    const int phase = uni(blockIdx.x / per_phase);
    const int bx = blockIdx.x - phase * per_phase;
    int tile, z;
    if ((ksplit & 7) == 0) {
        const int xcd = bx & 7, j = bx >> 3;
        const int jt = uni(j / tiles);
        z = jt * 8 + xcd;
        tile = j - jt * tiles;
    } else {
        z = uni(bx / tiles);
        tile = bx - z * tiles;
    }
    const int py = phase >> 1, px = phase & 1;
    const int tm = uni(tile / tiles_n), tn = tile - tm * tiles_n;
    const int cm0 = tm * 64, n0 = tn * 64;

    const int W = d.W, Wl = W >> 1, Hl = d.H >> 1;
    const int Cs = d.C0;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(d.x0);
    const size_t planex = (size_t)Ml * Cs, planey = (size_t)d.N * d.H * d.W * Cout;
    const int pbx = (int)(planex * 2), pby = (int)(planey * 2);

    constexpr int PXS = K64 ? 64 : BK;                               // low-resolution pixels per K-step
    constexpr int NPM = K64 ? 1 : NP;                                // planes in memory
    const int total_steps = (Ml + PXS - 1) / PXS;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    const srd_t rsx = make_srd(src, NPM * pbx), rsy = make_srd(dy, NPM * pby);
    const unsigned lds0 = lds_addr(smem);
    const unsigned ldsw = lds0 + 8 * wv * RB;
    const unsigned lds4 = lds0 + 32 * RB;

    const int drow = lane >> 3;
    const int dcol = ((((lane >> 2) & 1) ^ ((lane >> 4) & 1)) << 6) | ((lane & 3) << 4);
    const int Cs2 = Cs * 2;
    const int xlane = drow * Cs2 + dcol;
    // x strips: local tap row kyl = 0, 1 -> 3x3 tap row py + kyl: source pixel q + (py + kyl - 1) Wl.  The 8 pixels of a piece lie
    // in one image row (Wl >= 8, pieces start at multiples of 8): validity and source pixel are wave-uniform
    auto dma_x = [&](auto kyc, auto stagec, const int st) {
        constexpr int kyl = decltype(kyc)::value, stage = decltype(stagec)::value;
        constexpr int DST = stage * A_STAGE + kyl * BK * RB;
        const int dyr = py + kyl - 1;                              // -1, 0, +1
        const int qb = st * PXS + 8 * wv + dyr * Wl;
        const int yq = (qb >> lw) & (Hl - 1);
        const bool ok = (unsigned)qb < (unsigned)Ml && (unsigned)(yq - dyr) < (unsigned)Hl;
        const int soff = ok ? qb * Cs2 + cm0 * 2 : 0;
        const int voff = ok ? xlane : (int)0x80000000;
        lds_dma16_at<DST>(rsx, ldsw, voff, soff);
        if constexpr (K64) {      // slot 1: the same piece 32 pixels on (its own row / image checks)
            const int qb1 = qb + BK, yq1 = (qb1 >> lw) & (Hl - 1);
            const bool ok1 = (unsigned)qb1 < (unsigned)Ml && (unsigned)(yq1 - dyr) < (unsigned)Hl;
            lds_dma16_at<DST + A_PLANE>(rsx, ldsw, ok1 ? xlane : (int)0x80000000, ok1 ? qb1 * Cs2 + cm0 * 2 : 0);
        } else
            lds_dma16_at<DST + A_PLANE>(rsx, ldsw, voff, soff + pbx);
    };
    // dy tile: rows r = 0 .. 39 hold the low-resolution pixels st BK - 1 + r of this phase = dy[n, 2Y + py, 2X + px]: per-lane source
    // addresses (a piece may straddle an image row, whose pixels are not equidistant in the high-resolution tensor)
    const int Co2 = Cout * 2;
    auto dma_y = [&](auto stagec, const bool fifth, const int st) {
        constexpr int stage = decltype(stagec)::value;
        constexpr int DST = BOFF + stage * B_STAGE;
        const int q = st * PXS - 1 + (fifth ? 32 : 8 * wv) + drow;
        auto addr = [&](const int qq) {
            const int row = qq >> lw;                              // n Hl + Y
            const int hi = (2 * row + py) * W + 2 * (qq & (Wl - 1)) + px;
            return (unsigned)qq < (unsigned)Ml ? hi * Co2 + dcol : (int)0x80000000;
        };
        const int voff = addr(q);
        const int voff1 = K64 ? addr(q + BK) : voff;              // slot 1: the other plane, or (K64) the piece 32 pixels on
        const int soff = n0 * 2, soff1 = K64 ? soff : soff + pby;
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
        dma_y(stagec, false, st);
        if (wv == 0) dma_y(stagec, true, st);
    };
    // wave 0 issues (2 + 2) NP DMAs per step, the others (2 + 1) NP
    auto wait_all_but_one_step = [&]() {
        if (wv == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };

    if (t < NS * NP * 8) {      // the zero row of every dy plane of every stage (the DMA never writes it)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int pl = t >> 3;
        *reinterpret_cast<u32x4*>(smem + BOFF + (pl / NP) * B_STAGE + (pl % NP) * B_PLANE + ZROW * RB + (t & 7) * 16) = zero;
    }

    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int g = lane >> 4, L = lane & 15;
    const int krow = 8 * (g >> 1) + (L >> 2);
    const int cbyte = (16 * (g & 1) + 4 * (L & 3)) * 2;
    const int a_off = krow * RB + (((wm ^ ((krow >> 1) & 1)) << 6) | cbyte);
    const unsigned char* const aptr = smem + a_off;
    // dy: tile row of pixel q for 3x3 tap column kx is (q - p0) + 2 - kx; local column kxl -> kx = px + kxl
    int b_off[2];
#pragma unroll
    for (int kxl = 0; kxl < 2; ++kxl) {
        const int r = krow + 2 - (px + kxl);
        b_off[kxl] = BOFF + r * RB + (((wn ^ ((r >> 1) & 1)) << 6) | cbyte);
    }
    const int b_zero = BOFF + ZROW * RB + ((wn << 6) | cbyte);
    auto tr = [&](const unsigned char* p) -> s16x4u { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4u_t*)p); };

    s16x4u afr[2][2][NP][2], bfr[2][2][NP][2];       // [slice][local tap row / column][plane slot][row half]
    // [slice][local tap column][plane slot (K64: pixel half)][row half]: address of the dy read inside stage 0 (tile row or the zero row)
    constexpr int NSL = K64 ? 2 : 1;
    const unsigned char* bsel[2][2][NSL][2];
    const unsigned char* bconst[2][2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kxl = 0; kxl < 2; ++kxl)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bconst[s2][kxl][e] = smem + b_off[kxl] + (16 * s2 + 4 * e) * RB;
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) bsel[s2][kxl][sl][e] = bconst[s2][kxl][e];
            }
    const unsigned char* const bzero = smem + b_zero;
    // the shifted pixel must lie in the same image row: kx = 0 reads dy[q + 1] (not past the right border), kx = 2 reads dy[q - 1]
    auto b_addr = [&](auto sc, const int st) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = st * PXS + BK * sl + 16 * s + 4 * e + krow;
                const int ox = q & (Wl - 1);
                // (both columns written with selects: a conditional store to one of them becomes a runtime-indexed array = scratch)
                bsel[s][0][sl][e] = (px != 0 || ox <= Wl - 2) ? bconst[s][0][e] : bzero;      // px = 0: column 0 is kx = 0
                bsel[s][1][sl][e] = (px == 0 || ox >= 1) ? bconst[s][1][e] : bzero;           // px = 1: column 1 is kx = 2
            }
    };
    constexpr int NR = NP * 8, NMMA = (K64 ? 2 : nprod<NP>()) * 4;      // 16 reads, 12 (K64: 8) MFMAs per slice
    static_assert(NR <= 2 * NMMA, "at most two fragment reads behind each MFMA");
    // read k of a slice, in order of first use (products l*h, h*l, h*h): per plane pair x(row 0), dy(col 0), dy(col 1), x(row 1), two
    // row halves each
    auto read_frag = [&](auto sc, auto kc, auto stagec) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value, stage = decltype(stagec)::value;
        constexpr int grp = k / 8, r = (k - grp * 8) >> 1, e = k & 1;
        constexpr int pa = K64 ? grp : NP - 1 - grp, pb = grp;
        if constexpr (r == 0 || r == 3) {
            constexpr int kyl = r == 0 ? 0 : 1;
            constexpr int off = stage * A_STAGE + pa * A_PLANE + (kyl * BK + 16 * s + 4 * e) * RB;
            afr[s][kyl][pa][e] = tr(aptr + off);
        } else {
            constexpr int kxl = r - 1;
            bfr[s][kxl][pb][e] = tr(bsel[s][kxl][K64 ? pb : 0][e] + (stage * B_STAGE + pb * B_PLANE));
        }
    };
    auto frag = [](const s16x4u lo, const s16x4u hi) {
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma_one = [&](auto sc, auto mc) {
        constexpr int s = decltype(sc)::value, m = decltype(mc)::value;
        constexpr int q = m / 4, tap = m - q * 4, kyl = tap >> 1, kxl = tap & 1;
        constexpr int pa = K64 ? q : prod_a<NP>(q), pb = K64 ? q : prod_b<NP>(q);
        acc[tap] = mma16<NP>(frag(afr[s][kyl][pa][0], afr[s][kyl][pa][1]), frag(bfr[s][kxl][pb][0], bfr[s][kxl][pb][1]), acc[tap]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if (s_begin < s_end) {
        auto stc = [&](int st) { return st < s_end ? st : s_end - 1; };
        dma_step(std::integral_constant<int, 0>{}, s_begin);
        dma_step(std::integral_constant<int, 1>{}, stc(s_begin + 1));
        dma_step(std::integral_constant<int, 2>{}, stc(s_begin + 2));
        wait_all_but_one_step();
        __builtin_amdgcn_s_barrier();
        b_addr(I0{}, s_begin);
        static_for_wu<NR>([&](auto kc) { read_frag(I0{}, kc, I0{}); });
        auto step = [&](auto kc4, const int st) {
            constexpr int K = decltype(kc4)::value;
            using SK = std::integral_constant<int, K>;
            using SN = std::integral_constant<int, (K + 1) & 3>;
            using SD = std::integral_constant<int, (K + 3) & 3>;
            const int dst_step = stc(st + 3);
            // first half: MFMAs of slice 0 | reads of slice 1 | the x strips of step st + 3
            b_addr(I1{}, st);
            __builtin_amdgcn_sched_barrier(0);
            static_for_wu<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I0{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                read_frag(I1{}, mc, SK{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR - NMMA) {
                    read_frag(I1{}, std::integral_constant<int, NMMA + m>{}, SK{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == NMMA / 2 - 1 || m == NMMA - 3) {
                    dma_x(std::integral_constant<int, (m == NMMA / 2 - 1 ? 0 : 1)>{}, SD{}, dst_step);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // second half: MFMAs of slice 1 | reads of step st + 1 / slice 0 | dy of step st + 3
            b_addr(I0{}, st + 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for_wu<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma_one(I1{}, mc);
                __builtin_amdgcn_sched_barrier(0);
                read_frag(I0{}, mc, SN{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < NR - NMMA) {
                    read_frag(I0{}, std::integral_constant<int, NMMA + m>{}, SN{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == NMMA / 2 - 1) {
                    dma_y(SD{}, false, dst_step);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == NMMA - 3) {
                    if (wv == 0) dma_y(SD{}, true, dst_step);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            wait_all_but_one_step();
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (z >= ksplit) return;
    // partial [phase][z][tap 0..3][Cin][Cout]: each wave turns its 32 x 32 tap tiles through a private LDS slab (16-byte stores)
    const int li = lane & 31, h = lane >> 5;
    constexpr int SW = 36;
    float* slab = reinterpret_cast<float*>(smem) + wv * 32 * SW;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * h) * SW + li] = acc[tap][r];
        __builtin_amdgcn_wave_barrier();
        float* out = partial + ((size_t)((phase * ksplit + z) * 4 + tap) * Cin + cm0 + wm * 32) * Cout + n0 + wn * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gidx = q * 64 + lane, rr = gidx >> 3, c4 = gidx & 7;
            *reinterpret_cast<f32x4*>(out + (size_t)rr * Cout + c4 * 4) = *reinterpret_cast<const f32x4*>(&slab[rr * SW + c4 * 4]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// partial [4 phases][ksplit][4][Cin][Cout] -> dw [Cout][Cin][3][3]: a block = a 32 (cin) x 32 (cout) tile of all nine taps; thread =
// (cin row, four output channels); the K splits of each (phase, tap slot) are added in index order, the four phases of a tap in
// phase order, the result times the two power-of-two operand scales
__global__ __launch_bounds__(256) void wgrad_up4_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, const int ksplit,
                                                                const int Cin, const int Cout, const int accumulate,
                                                                const float* __restrict__ sx, const float* __restrict__ sdy) {
    __shared__ float tile[9][32][33];
    const int t = threadIdx.x;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int rr = t >> 3, c4 = (t & 7) * 4;
    const size_t tstride = (size_t)Cin * Cout, zstride = 4 * tstride, pstride = (size_t)ksplit * zstride;
    const float scl = *sx * *sdy;
    f32x4 v[4][4];      // [phase][tap slot]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const float* p = partial + ph * pstride + sl * tstride + (size_t)(ci0 + rr) * Cout + co0 + c4;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            int z = 0;
            for (; z + 2 <= ksplit; z += 2) {
                s0 += *reinterpret_cast<const f32x4*>(p + (size_t)z * zstride);
                s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(z + 1) * zstride);
            }
            if (z < ksplit) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)z * zstride);
            v[ph][sl] = s0 + s1;
        }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                const int r = py == 0 ? (ky >= 1) : (ky >= 2), c = px == 0 ? (kx >= 1) : (kx >= 2);
                s += v[ph][r * 2 + c];
            }
            s *= scl;
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[ky * 3 + kx][rr][c4 + k] = s[k];
        }
    __syncthreads();
    const int rowlen = 32 * 9;
    for (int e = t; e < 32 * rowlen; e += 256) {
        const int col = e / rowlen, r = e - col * rowlen;
        const int c = r / 9, tap = r - c * 9;
        float* o = dw + ((size_t)(co0 + col) * Cin + ci0) * 9 + r;
        *o = accumulate ? *o + tile[tap][c][col] : tile[tap][c][col];
    }
}

static int ilog2u(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// split-K plan: tiles x ksplit x 4 phases ~ 256 blocks (one per CU), at least 8 K-steps per block
static void up4_wgrad_plan(int Ml, int Cin, int Cout, int pxs, int* ksplit, int* sps) {
    const int tiles = (Cin / 64) * (Cout / 64) * 4;
    const int total_steps = (Ml + pxs - 1) / pxs;
    long ks = tiles >= 256 ? 1 : (256 + tiles - 1) / tiles;
    ks = std::min<long>(ks, std::max(1, total_steps / 8));
    if (ks >= 8) ks = (ks / 8) * 8;
    *sps = (int)((total_steps + ks - 1) / ks);
    int k2 = (total_steps + *sps - 1) / *sps;
    if (ks >= 8) k2 = ((k2 + 7) / 8) * 8;
    *ksplit = k2;
}

static bool up4_wgrad_ok(const rpnet_conv_desc* d) {
    if (!d || (d->split_planes != 2 && d->split_planes != 1) || d->C1 || d->Co1 || d->x1 || d->H % 2 || d->W % 2) return false;
    const int Hl = d->H / 2, Wl = d->W / 2;
    if (ilog2u(Wl) < 3 || ilog2u(Hl) < 0 || d->C0 % 64 || d->Co0 % 64) return false;
    const size_t lim = (size_t)1 << 31;
    return (size_t)d->N * Hl * Wl * d->C0 * 4 < lim && (size_t)d->N * d->H * d->W * d->Co0 * 4 < lim &&
           ((size_t)d->N * Hl * Wl) % (d->split_planes == 1 ? 64 : 32) == 0;
}

}  // namespace rpnet

extern "C" int rpnet_conv_wgrad_up4_supported(const rpnet_conv_desc* d) { return rpnet::up4_wgrad_ok(d) ? 1 : 0; }

extern "C" size_t rpnet_conv_wgrad_up4_workspace_bytes(int N, int H, int W, int cin, int cout) {
    int ks, sps, ks1, sps1;      // the larger of the two plans (two planes: 32-pixel steps; one plane: 64-pixel steps)
    rpnet::up4_wgrad_plan(N * (H / 2) * (W / 2), cin, cout, 32, &ks, &sps);
    rpnet::up4_wgrad_plan(N * (H / 2) * (W / 2), cin, cout, 64, &ks1, &sps1);
    return (size_t)4 * (ks > ks1 ? ks : ks1) * 4 * cin * cout * sizeof(float);
}

extern "C" int rpnet_conv_wgrad_up4(const rpnet_conv_desc* d, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                    rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(d && d->x0 && workspace && (dy || dw), RPNET_ERR_ARG, "conv_wgrad_up4: null pointer");
    RPNET_REQUIRE(up4_wgrad_ok(d), RPNET_ERR_SHAPE, "conv_wgrad_up4: N=%d H=%d W=%d C0=%d Co0=%d planes=%d do not fit (rpnet_conv_wgrad_up4_supported)",
                  d->N, d->H, d->W, d->C0, d->Co0, d->split_planes);
    const int Hl = d->H / 2, Wl = d->W / 2, Ml = d->N * Hl * Wl, Cin = d->C0, Cout = d->Co0;
    int ks, sps;
    const bool one = d->split_planes == 1;
    up4_wgrad_plan(Ml, Cin, Cout, one ? 64 : 32, &ks, &sps);
    RPNET_REQUIRE(workspace_bytes >= rpnet_conv_wgrad_up4_workspace_bytes(d->N, d->H, d->W, Cin, Cout), RPNET_ERR_WORKSPACE, "conv_wgrad_up4: workspace");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    if (dy) {       // dy == NULL: reduce phase only (the two-phase form of rpnet_conv_wgrad)
        const int tiles_n = Cout / 64, tiles = (Cin / 64) * tiles_n;
        if (one)
            hipLaunchKernelGGL(conv_wgrad_up4_kernel<true>, dim3(tiles * ks * 4), dim3(256), 0, s, *d, (const unsigned short*)dy, part, Ml, Cin, Cout,
                               tiles, tiles_n, ks, sps, ilog2u(Wl), ilog2u(Hl));
        else
            hipLaunchKernelGGL(conv_wgrad_up4_kernel<false>, dim3(tiles * ks * 4), dim3(256), 0, s, *d, (const unsigned short*)dy, part, Ml, Cin, Cout,
                               tiles, tiles_n, ks, sps, ilog2u(Wl), ilog2u(Hl));
        if (int rc = check_launch("conv_wgrad_up4")) return rc;
    }
    if (!dw) return RPNET_OK;
    RPNET_REQUIRE(d->acc_scale_x && d->acc_scale_dy, RPNET_ERR_ARG, "conv_wgrad_up4: fp16 planes need acc_scale_x and acc_scale_dy");
    hipLaunchKernelGGL(wgrad_up4_reduce_kernel, dim3(Cin / 32, Cout / 32), dim3(256), 0, s, (const float*)part, dw, ks, Cin, Cout, d->accumulate,
                       d->acc_scale_x, d->acc_scale_dy);
    return check_launch("wgrad_up4_reduce");
}
