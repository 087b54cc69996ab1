// up_conv (nn.Upsample(scale_factor=2) -> Conv2d 3x3, net/modules.py:61-75) without its redundant multiplications.
//
// A 3x3 convolution over a nearest-x2 up-sampled image reads, for an output pixel (2Y + py, 2X + px), only a 2 x 2 block of
// SOURCE pixels: rows (2Y + py + ky - 1) >> 1 for ky = 0, 1, 2 are {Y - 1, Y, Y} for py = 0 and {Y, Y, Y + 1} for py = 1 (the same in
// x).  The nine products per (cin, cout) collapse to four with the weights of coinciding taps added up front:
//     y[2Y + py, 2X + px, co] = sum_{r, c in {0, 1}} sum_ci Wc[(py, px)][r][c][co][ci] * x[Y + py + r - 1, X + px + c - 1, ci],
//     Wc[(py, px)][r][c] = sum_{ky in S(py, r)} sum_{kx in S(px, c)} w[ky][kx],   S(0, 0) = {0}, S(0, 1) = {1, 2}, S(1, 0) = {0, 1}, S(1, 1) = {2}
// (rpnet_upconv_collapse_weights) — 4 / 9 of the multiply-adds of the layer, forward, input gradient and (conv_wgrad_up4.hip)
// weight gradient; Up5 and Up4 are 17 % of the training step's FLOPs as the reference writes them.  The result differs from
// the nine-product form by the rounding of the weight sums (2^-24 relative), far inside the 1e-3 bar; rpnet_conv_fwd with
// `upsample` stays the bit-exact nine-product form (RPNET_UPCONV_COLLAPSE=0).
//
// As a GEMM this is an ordinary 3x3 convolution on the LOW-resolution grid with 4 x Cout output columns (phase-major) whose
// weight tensor is block sparse: phase (py, px) only has the taps {py, py + 1} x {px, px + 1}.  The kernel is the LDS-DMA patch
// kernel of conv_split_dma.hip (256 low-resolution pixels x 128 / 64 columns, four waves, halo double buffer, four-stage
// weight ring, counted vmcnt + one raw barrier per K-step, pinned issue order) with FOUR K-steps per channel chunk instead of
// nine:
//   forward (DG = false): a block's column tile lies in one phase -> its four taps; the epilogue scatters the patch to the
//     high-resolution pixels (2Y + py, 2X + px) (PatchRowsUp) — bias, fused BatchNorm statistics (one partial row per block,
//     phases as extra row tiles), operand scales as in every other convolution launch;
//   input gradient (DG = true): dx[Y, X, ci] = sum_{phase, r, c, co} Wc[phase][r][c][co][ci] dy[2 (Y - (py + r - 1)) + py, ..., co]: the
//     K dimension runs over (phase, co) — a chunk of 32 K-channels lies in one phase, its halo is gathered from the
//     high-resolution dy at stride 2 (per-lane DMA source addresses), its taps are {1 - py + r', 1 - px + c'} of the flipped pack;
//     the output IS the low-resolution gradient (no separate 2 x 2 sum over a high-resolution tensor, rpnet_upsample2_bwd).
// The halo of a chunk (up to 6 piece positions per wave) has to arrive within four K-steps instead of nine: positions go out in
// steps 0 and 1 of the previous chunk, step 2's wait covers them all.
#include <type_traits>

#include "conv_epilogue.h"
#include "lds_dma.h"
#include "split_bf16.h"

namespace rpnet {

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for_u(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_u<N, I + 1>(f);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TW>
struct PatchRowsUp {   // pixel (ty, tx) of a low-resolution patch -> high-resolution pixel (2 ty + py, 2 tx + px)
    static constexpr bool kAlwaysValid = true;
    int base, W4;      // base: linear index of (2 y0 + py, 2 x0 + px); W4 = 2 x the high-resolution width
    __device__ __forceinline__ int operator()(int local) const { return base + (local / TW) * W4 + (local % TW) * 2; }
};

// Kc: GEMM K channels (forward: Cin; input gradient: 4 x the layer's Cout), Ncols: GEMM columns (forward: 4 x Cout; input
// gradient: the layer's Cin).  d.N, d.H, d.W: the HIGH-resolution tensor (the layer's output / its gradient).
// K64: ONE fp16 plane (the f16 arithmetic of BASELINE configs[4]) in the same LDS geometry, as in conv_split_dma.hip: a K-step is one tap
// of 64 channels whose two 32-channel halves take the places of the two planes; the products are the diagonal ones.
template <int TW, int WN, bool DG, bool K64 = false>
__global__ __launch_bounds__(256, 1) void conv_up4_dma_kernel(const rpnet_conv_desc d, const int Kc, const int Ncols, const int tiles_n,
                                                               const int ntiles) {
    constexpr int NP = 2, WM = 4;
    constexpr int BM = 64 * WM, BN = 64 * WN, TH = BM / TW, PW = TW + 2, HALO = (TH + 2) * PW;
    constexpr int HP = (HALO + 15) / 16;
    constexpr int HPW = (HP + 3) / 4;
    constexpr int A_BYTES = HP * 1024, HBUF = NP * A_BYTES;
    constexpr int B_BYTES = BN * 64, STAGE = NP * B_BYTES;
    constexpr int NS = 4;
    constexpr int NW = WN * NP;
    constexpr int NPOS0 = (HPW + 1) / 2, NPOS1 = HPW - NPOS0;      // halo piece positions issued in steps 0 / 1 of a chunk
    static_assert(HPW <= 6, "halo piece positions per wave");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[cmax(2 * HBUF + NS * STAGE, epilogue_lds_bytes<WN, 2>())];
    RPNET_ASSERT_NO_CORESIDENCE(sizeof(smem));
    constexpr int WOFF = 2 * HBUF;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    const int tile = xcd_swizzle(blockIdx.x, ntiles);
    const int tm = uni(tile / tiles_n), tn = tile - tm * tiles_n;
    const int n0 = tn * BN;
    const int H = d.H, W = d.W, Hl = H >> 1, Wl = W >> 1;
    const int pxn = Wl / TW, ppi = (Hl / TH) * pxn;
    const int n = uni(tm / ppi), prem = tm - n * ppi;
    const int y0 = uni(prem / pxn) * TH, x0 = uni(prem % pxn) * TW;
    const int C0 = d.C0;
    // forward: the phase of this block's columns
    const int Cout_out = Ncols >> 2;
    const int fphase = DG ? 0 : uni(n0 / Cout_out);
    const int fpy = fphase >> 1, fpx = fphase & 1;

    constexpr int CSH = K64 ? 6 : 5;               // channels per K-step: 64 (one plane, two halves) or 32 (per plane)
    const int kchunks = Kc >> CSH;
    const int rot = uni((int)(blockIdx.x % (unsigned)kchunks));
    auto chunk_c0 = [&](int ci) { int c = rot + ci; if (c >= kchunks) c -= kchunks; return c << CSH; };
    // tap base (Py, Px) of a chunk: the 3x3 index of its tap slot (r, c) is (Py + r, Px + c)
    auto chunk_py = [&](int c0) { return DG ? 1 - (uni(c0 / C0) >> 1) : fpy; };
    auto chunk_px = [&](int c0) { return DG ? 1 - (uni(c0 / C0) & 1) : fpx; };

    const size_t plane0 = DG ? (size_t)d.N * H * W * C0 : (size_t)d.N * Hl * Wl * C0;
    const size_t planew = (size_t)4 * Kc * Ncols;
    const unsigned short* x0p = reinterpret_cast<const unsigned short*>(d.x0);
    const unsigned short* wq = reinterpret_cast<const unsigned short*>(d.w);
    const int pb0 = (int)(plane0 * 2), pbw = (int)(planew * 2);
    constexpr int NPM = K64 ? 1 : NP;              // planes in memory
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x0p), (short)0, NPM * pb0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq), (short)0, NPM * pbw, 0x00020000);
    // offset of plane / half p inside the source and inside the weights (K64: the next 32 channels of a pixel / the next weight slab)
    const int ps0 = K64 ? 64 : pb0, psw = K64 ? Ncols * 64 : pbw;

    const int drow = lane >> 2;
    const int dkg16 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    int hpix[HPW];              // source pixel of this lane's halo row (forward: low resolution; input gradient: phase (0, 0) of dy)
    bool hok[HPW];
    int hpos[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        int pos = wv + 4 * i;
        if (pos >= HP) pos -= 4;
        hpos[i] = pos;
        const int hr = pos * 16 + drow;
        const int hy = hr / PW, hx = hr - hy * PW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        hok[i] = hr < HALO && iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
        hpix[i] = DG ? (n * H + 2 * iy) * W + 2 * ix : (n * Hl + iy) * Wl + ix;
    }
    auto dma_halo = [&](auto ic, int c0, int buf) {
        constexpr int i = decltype(ic)::value;
        int pix = hpix[i], cs = c0;
        if constexpr (DG) {
            const int ph = uni(c0 / C0);
            cs = c0 - ph * C0;
            pix += (ph >> 1) * W + (ph & 1);
        }
        const int voff = hok[i] ? pix * (C0 * 2) + dkg16 : (int)0x80000000;      // out of range -> zeros (the padding)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            auto* dst = (__attribute__((address_space(3))) void*)(smem + buf * HBUF + p * A_BYTES + hpos[i] * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, uni(cs * 2) + p * ps0, 0, 0);
        }
    };
    const int wvoff = (16 * WN * wv + drow) * 64 + dkg16;
    auto w_soff = [&](int idx, int c0) { return uni(((idx * (Kc >> 5) + (c0 >> 5)) * Ncols + n0) * 64); };
    auto dma_w = [&](int idx, int c0, int stage) {
        const int wsoff = w_soff(idx, c0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < WN; ++q) {
                auto* dst = (__attribute__((address_space(3))) void*)(smem + WOFF + stage * STAGE + p * B_BYTES + (16 * WN * wv + 16 * q) * 64);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, wvoff + q * 1024, wsoff + p * psw, 0, 0);
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
    const int b_off = WOFF + (wn * WN * 32 + li) * 64 + 16 * (h ^ ((li >> 2) & 3));

    bf16x8 af[2][NP][WM], bfr[2][NP][WN];
    int aaddr[WM];
    auto a_addr = [&](const int ky, const int kx) {      // 3x3 tap index (ky, kx) -> the A fragment addresses inside a halo buffer
        const int off = ky * PW + kx;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int hr = hr00[i] + off;
            aaddr[i] = hr * 64 + 16 * (h ^ ((hr >> 2) & 3));
        }
    };
    constexpr int NPROD = K64 ? 2 : nprod<NP>();
    constexpr int NR = NP * (WM + WN), NMMA = NPROD * WM * WN;
    static_assert(NR <= NMMA, "one fragment read of the next slice behind each MFMA of this one");
    auto read_frag = [&](auto sc, auto kc, const int abase, const int bbase) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value;
        constexpr int grp = k / (WM + WN), r = k - grp * (WM + WN);
        constexpr int pa = K64 ? grp : NP - 1 - grp, pb = grp;
        if constexpr (r == 0 || r > WN) {
            constexpr int i = r == 0 ? 0 : r - WN;
            af[s][pa][i] = *reinterpret_cast<const bf16x8*>(smem + abase + (aaddr[i] ^ (32 * s)) + pa * A_BYTES);
        } else {
            constexpr int j = r - 1;
            bfr[s][pb][j] = *reinterpret_cast<const bf16x8*>(smem + bbase + (b_off ^ (32 * s)) + pb * B_BYTES + j * 2048);
        }
    };
    auto mma_one = [&](auto sc, auto mc) {
        constexpr int s = decltype(sc)::value, m = decltype(mc)::value;
        constexpr int q = m / (WM * WN), ij = m - q * (WM * WN), i = ij / WN, j = ij - i * WN;
        constexpr int pa = K64 ? q : prod_a<NP>(q), pb = K64 ? q : prod_b<NP>(q);
        acc[i][j] = mma16<NP>(af[s][pa][i], bfr[s][pb][j], acc[i][j]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto dma_w_one = [&](auto ec, const int wsoff, const int stage) {
        constexpr int e = decltype(ec)::value, p = e / WN, q = e - p * WN;
        auto* dst = (__attribute__((address_space(3))) void*)(smem + WOFF + stage * STAGE + p * B_BYTES + (16 * WN * wv + 16 * q) * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, wvoff + q * 1024, wsoff + p * psw, 0, 0);
    };

    // ---- prologue: halo of chunk 0, weight slabs of steps 0, 1, 2
    {
        const int c0 = chunk_c0(0);
        static_for_u<HPW>([&](auto ic) { dma_halo(ic, c0, 0); });
        dma_w(0, c0, 0);
        dma_w(1, c0, 1);
        dma_w(2, c0, 2);
        wait_vmcnt<NW>();
        __builtin_amdgcn_s_barrier();
        a_addr(chunk_py(c0), chunk_px(c0));
    }
    static_for_u<NR>([&](auto kc) { read_frag(I0{}, kc, 0, 0); });

    // One K-step: tap slot IDX = (r, c) = (IDX >> 1, IDX & 1) of chunk ci, step ks, weight stage ks & 3 (see conv_split_dma.hip for
    // the two half steps and what each barrier makes visible).
    auto step = [&](auto idxc, const int ci, const int ks) {
        constexpr int IDX = decltype(idxc)::value;
        const int hb = ci & 1;
        const bool last_chunk = ci + 1 >= kchunks;
        constexpr int I3 = (IDX + 3) & 3;
        const int c3 = chunk_c0((IDX + 3 < 4 || last_chunk) ? ci : ci + 1);      // (the tail re-fetches valid data nobody reads)
        const int wsoff = w_soff(I3, c3);
        const int wstage = (ks + 3) & 3;
        const int abase = hb * HBUF, bbase = (ks & 3) * STAGE;
        static_for_u<NMMA>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            mma_one(I0{}, mc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < NR) {
                read_frag(I1{}, mc, abase, bbase);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (m >= NMMA - 2 * NW && (NMMA - 1 - m) % 2 == 0) {
                dma_w_one(std::integral_constant<int, NW - 1 - (NMMA - 1 - m) / 2>{}, wsoff, wstage);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // the next step: slot (IDX + 1) & 3 of this chunk or slot 0 of the next one
        constexpr int IN = (IDX + 1) & 3;
        const int cn = chunk_c0((IDX < 3 || last_chunk) ? ci : ci + 1);
        a_addr(chunk_py(cn) + (IN >> 1), chunk_px(cn) + (IN & 1));
        const int abase_n = (IDX < 3 ? hb : hb ^ 1) * HBUF, bbase_n = ((ks + 1) & 3) * STAGE;
        const int ch = chunk_c0(last_chunk ? ci : ci + 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for_u<NMMA>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            mma_one(I1{}, mc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < NR) {
                read_frag(I0{}, mc, abase_n, bbase_n);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the halo of chunk ci + 1 into the other buffer: piece positions 0 .. NPOS0 - 1 in step 0, the rest in step 1, one
            // position behind every third MFMA of the tail
            if constexpr (IDX == 0 && m >= NMMA - 3 * NPOS0 && (NMMA - 1 - m) % 3 == 0) {
                dma_halo(std::integral_constant<int, NPOS0 - 1 - (NMMA - 1 - m) / 3>{}, ch, hb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (IDX == 1 && NPOS1 > 0 && m >= NMMA - 3 * NPOS1 && (NMMA - 1 - m) % 3 == 0) {
                dma_halo(std::integral_constant<int, NPOS0 + NPOS1 - 1 - (NMMA - 1 - m) / 3>{}, ch, hb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // slab ks + 2 (and every older DMA) has landed; step 2 also waits for the halo pieces of step 1: they are read after
        // ITS barrier (the fragment prefetch of step 3's second half)
        constexpr int N = IDX == 0 ? NW + NPOS0 * NP : (IDX == 1 ? NPOS0 * NP + NW + NPOS1 * NP : NW);
        wait_vmcnt<N>();
        __builtin_amdgcn_s_barrier();
    };

    int ks = 0;
    for (int ci = 0; ci < kchunks; ++ci) {
        step(std::integral_constant<int, 0>{}, ci, ks + 0);
        step(std::integral_constant<int, 1>{}, ci, ks + 1);
        step(std::integral_constant<int, 2>{}, ci, ks + 2);
        step(std::integral_constant<int, 3>{}, ci, ks + 3);
        ks += 4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if constexpr (DG) {
        conv_epilogue<WM, WN, 2, PatchRows<TW>, false>(d, acc, PatchRows<TW>{(n * Hl + y0) * Wl + x0, Wl}, d.N * Hl * Wl, Ncols, Hl * Wl, n0, tm,
                                                        wm, wn, li, h, smem);
    } else {
        // (last argument: the row scales of the weights are indexed by the packed (phase, cout) row)
        conv_epilogue<WM, WN, 2, PatchRowsUp<TW>, false>(d, acc, PatchRowsUp<TW>{(n * H + 2 * y0 + fpy) * W + 2 * x0 + fpx, 2 * W}, d.N * H * W,
                                                          Cout_out, H * W, n0 - fphase * Cout_out, tm * 4 + fphase, wm, wn, li, h, smem,
                                                          fphase * Cout_out);
    }
}

// Wc [4 Cout][Cin][2][2] from w [Cout][Cin][3][3] (see the header): row (py * 2 + px) * Cout + co
__global__ __launch_bounds__(256) void upconv_collapse_kernel(const float* __restrict__ w, float* __restrict__ wc, const int Cout, const int Cin) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;         // (phase, co, ci)
    if (i >= (size_t)4 * Cout * Cin) return;
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout), ph = (int)(i / ((size_t)Cin * Cout));
    const int py = ph >> 1, px = ph & 1;
    float k[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) k[q] = w[((size_t)co * Cin + ci) * 9 + q];
    float o[4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // S(p, 0) = {0} or {0, 1}; S(p, 1) = {1, 2} or {2}
            const int ya = r == 0 ? 0 : (py ? 2 : 1), yb = r == 0 ? (py ? 1 : 0) : 2;
            const int xa = c == 0 ? 0 : (px ? 2 : 1), xb = c == 0 ? (px ? 1 : 0) : 2;
            float s = 0.f;
            for (int ky = ya; ky <= yb; ++ky)
                for (int kx = xa; kx <= xb; ++kx) s += k[ky * 3 + kx];
            o[r * 2 + c] = s;
        }
    *reinterpret_cast<f32x4*>(wc + i * 4) = f32x4{o[0], o[1], o[2], o[3]};
}

static int up4_tw(int Hl, int Wl) {
    if (Wl % 32 == 0 && Hl % 8 == 0) return 32;
    if (Wl % 16 == 0 && Hl % 16 == 0) return 16;
    return 0;
}

// wn (1 / 2) of the launch, or 0 when the shapes do not fit (mode 1: forward, 2: input gradient)
static int up4_plan(const rpnet_conv_desc* d, int mode, int* tw, int* Kc, int* Ncols) {
    if (!d || (d->split_planes != 2 && d->split_planes != 1) || d->H % 2 || d->W % 2 || d->C1 || d->Co1 || d->x1 || d->y1) return 0;
    const int Hl = d->H / 2, Wl = d->W / 2;
    const bool one = d->split_planes == 1;              // one fp16 plane: 64-channel K-steps, 128-wide tiles only
    *tw = up4_tw(Hl, Wl);
    if (!*tw || d->C0 % (one ? 64 : 32) || d->Co0 % (one ? 128 : 64)) return 0;
    *Kc = mode == 1 ? d->C0 : 4 * d->C0;
    *Ncols = mode == 1 ? 4 * d->Co0 : d->Co0;
    const size_t lim = (size_t)1 << 31;
    const size_t src = (size_t)d->N * (mode == 1 ? Hl * Wl : d->H * d->W) * d->C0 * 2 * d->split_planes;
    if (src >= lim || (size_t)2 * 4 * *Kc * *Ncols * 2 >= lim || (size_t)d->N * d->H * d->W >= lim) return 0;
    const long tiles_m = (long)d->N * Hl * Wl / 256;
    if (mode == 1) return (d->Co0 % 128 == 0) ? 2 : 1;       // a column tile never straddles two phases
    if (one) return 2;
    return (*Ncols % 128 == 0 && tiles_m * (*Ncols / 128) >= 192) ? 2 : 1;
}

}  // namespace rpnet

extern "C" int rpnet_conv_up4_supported(const rpnet_conv_desc* d, int mode) {
    int tw, Kc, Nc;
    return (mode == 1 || mode == 2) && rpnet::up4_plan(d, mode, &tw, &Kc, &Nc) > 0;
}

extern "C" int rpnet_conv_up4_stats_blocks(const rpnet_conv_desc* d) {
    int tw, Kc, Nc;
    if (!rpnet::up4_plan(d, 1, &tw, &Kc, &Nc) || d->groups < 1 || d->N % d->groups) return 0;
    return (int)((long)(d->N / d->groups) * d->H * d->W / 256);      // 256 output pixels of one phase per block
}

extern "C" int rpnet_upconv_collapse_weights(const float* w, float* wc, int Cout, int Cin, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(w && wc && Cout > 0 && Cin > 0, RPNET_ERR_ARG, "upconv_collapse_weights: bad argument");
    hipLaunchKernelGGL(upconv_collapse_kernel, dim3(cdiv((long)4 * Cout * Cin, 256)), dim3(256), 0, (hipStream_t)stream, w, wc, Cout, Cin);
    return check_launch("upconv_collapse_weights");
}

extern "C" int rpnet_conv_up4(const rpnet_conv_desc* d, int mode, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(d && d->x0 && d->w && d->y0 && (mode == 1 || mode == 2), RPNET_ERR_ARG, "conv_up4: null pointer / mode");
    int tw = 0, Kc = 0, Nc = 0;
    const int wn = up4_plan(d, mode, &tw, &Kc, &Nc);
    RPNET_REQUIRE(wn > 0, RPNET_ERR_SHAPE, "conv_up4: N=%d H=%d W=%d C0=%d Co0=%d planes=%d do not fit (rpnet_conv_up4_supported)", d->N, d->H, d->W,
                  d->C0, d->Co0, d->split_planes);
    RPNET_REQUIRE(d->acc_scale_col && d->acc_scale_x, RPNET_ERR_ARG, "conv_up4: fp16 planes need acc_scale_col and acc_scale_x");
    RPNET_REQUIRE(mode == 1 || !d->stats_partial, RPNET_ERR_ARG, "conv_up4: statistics belong to the forward launch");
    RPNET_REQUIRE(!d->stats_partial || rpnet_conv_up4_stats_blocks(d) > 0, RPNET_ERR_SHAPE, "conv_up4: statistic groups do not split into whole tiles");
    RPNET_REQUIRE(!d->out_scale_mode && !d->y_split && !d->bnb_y && !d->tile_skip && !d->splitk_ws && !d->y_enc, RPNET_ERR_ARG,
                  "conv_up4: plain epilogue only (bias, eval affine, statistics, accumulate, out_absmax)");
    const int tiles_m = d->N * (d->H / 2) * (d->W / 2) / 256, tiles_n = Nc / (64 * wn), ntiles = tiles_m * tiles_n;
    hipStream_t s = (hipStream_t)stream;
#define RPNET_UP4(TWV, WNV, DGV) hipLaunchKernelGGL((conv_up4_dma_kernel<TWV, WNV, DGV>), dim3(ntiles), dim3(256), 0, s, *d, Kc, Nc, tiles_n, ntiles)
#define RPNET_UP4K(TWV, DGV) hipLaunchKernelGGL((conv_up4_dma_kernel<TWV, 2, DGV, true>), dim3(ntiles), dim3(256), 0, s, *d, Kc, Nc, tiles_n, ntiles)
    if (d->split_planes == 1) {
        if (mode == 1) { if (tw == 32) RPNET_UP4K(32, false); else RPNET_UP4K(16, false); }
        else { if (tw == 32) RPNET_UP4K(32, true); else RPNET_UP4K(16, true); }
    } else if (mode == 1) {
        if (tw == 32) { if (wn == 2) RPNET_UP4(32, 2, false); else RPNET_UP4(32, 1, false); }
        else { if (wn == 2) RPNET_UP4(16, 2, false); else RPNET_UP4(16, 1, false); }
    } else {
        if (tw == 32) { if (wn == 2) RPNET_UP4(32, 2, true); else RPNET_UP4(32, 1, true); }
        else { if (wn == 2) RPNET_UP4(16, 2, true); else RPNET_UP4(16, 1, true); }
    }
#undef RPNET_UP4
#undef RPNET_UP4K
    return check_launch("conv_up4");
}
