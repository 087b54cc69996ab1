// 3x3 convolution (forward / input gradient) on fp16 split planes — the patch kernel of conv_split.hip rebuilt around the
// LDS-DMA path of gfx950 (buffer_load_dwordx4 ... lds) with ONE wave per SIMD.
//
// What the 8-wave patch kernel (conv_igemm_split_halo_kernel) spends outside its MFMAs (profiles/r02_pmc_sq_wave_cycles.txt:
// 36 % of the wave cycles parked on s_waitcnt / the barrier of every K-step): operand tiles travel HBM -> VGPR -> LDS, so
// every K-step ends in a barrier BEHIND which the fragment reads of the next step start from cold, and the staging
// registers (halo waiting for nine taps, weight slab) cap the wave tile at 64 x 64.  Here:
//   * block = 256 output pixels (a (256 / TW) x TW patch of one image) x 128 (WN = 2; 64 with WN = 1) output channels, FOUR
//     waves (2 x 2), wave tile 128 x 64: 12 ds_read_b128 per 24 MFMAs instead of 8 per 12, one wave per SIMD with the whole 512-register file;
//   * no staging registers and no ds_write: the input halo of the patch (two buffers: the next channel chunk lands while
//     the nine taps of the current one are multiplied) and the weight slabs (ring of four K-steps) are written by the
//     DMA engine.  The LDS image is lane-linear per DMA instruction, so the XOR swizzle of the k-groups that keeps the
//     b128 fragment reads conflict-free is applied to the per-lane SOURCE address (cdna_hip_programming.md rule 21);
//   * one raw s_barrier per K-step with a COUNTED s_waitcnt vmcnt(N) in front of it: the DMAs of the next two steps stay
//     in flight across the barrier; the weights of step k+1 are visible one barrier EARLY, so the fragments of step
//     k+1 / slice 0 are read during the MFMAs of step k / slice 1 — the matrix pipe does not wait behind a barrier;
//   * fragment registers are double-buffered by hand (slice s+1 is read while slice s is multiplied).
// Zero padding, the nearest x2 up-sampling and the two concatenated sources are resolved by the per-lane source offset
// of the halo DMA (out-of-image -> beyond num_records -> the DMA writes zeros), as in the register-staged kernel.
#include <type_traits>

#include "conv_epilogue.h"
#include "lds_dma.h"
#include "split_bf16.h"

namespace rpnet {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// K64: ONE fp16 plane (the f16 arithmetic of BASELINE configs[4]) in the same LDS geometry — a K-step is one tap of 64
// channels, whose two 32-channel halves take the places of the two planes (NP stays 2 for every byte count): half p of the
// halo / weight slab comes from channels +32 p of the single plane, and the products are the diagonal ones (half p of the
// pixels x half p of the weights): 32 MFMAs per wave and step for the same 24 fragment reads.
// WM = 2: 128-pixel patches (wave tile 64 x 32 WN) for grids that 256-pixel patches leave half empty — the CRE convolutions of
// an eval-mode call at batch 2 (M = 8192: 128 tiles of 256 x 64) — at six MFMAs per six fragment reads and half step.
template <int TW, int NP, int WN, bool K64 = false, int WM = 4>
__global__ __launch_bounds__(256, 1) void conv_igemm_split_dma_kernel(const rpnet_conv_desc d, const int Cin, const int Cout,
                                                                       const int tiles_n, const int ntiles, const int kshift) {
    constexpr int BM = 64 * WM, BN = 64 * WN, TH = BM / TW, PW = TW + 2, HALO = (TH + 2) * PW;
    constexpr int HP = (HALO + 15) / 16;           // 1 KB DMA pieces (16 halo rows of 64 B) per plane
    constexpr int HPW = (HP + 3) / 4;              // piece positions per wave and channel chunk (the last ones may repeat)
    constexpr int A_BYTES = HP * 1024, HBUF = NP * A_BYTES;
    constexpr int B_BYTES = BN * 64, STAGE = NP * B_BYTES;
    constexpr int NS = 4;                          // weight ring: K-steps k .. k+3
    constexpr int NW = WN * NP;                    // weight DMAs per wave and K-step (16 WN rows x NP planes)
    static_assert(HPW <= 6, "halo pieces are issued during taps 0 .. HPW-1; the wait counts below assume HPW <= 6");
    // (the smallest form — 128-pixel patches x 64 columns, variant 14 — needs less than a CU's LDS minus the guarded passes' reservation:
    // its allocation is padded up to that line, so that NO form of this kernel fits beside a guarded pooled BatchNorm block; lds_dma.h)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[cmax(cmax(2 * HBUF + NS * STAGE, epilogue_lds_bytes<WN, 2>()),
                                                                       kLdsPerCu - kGuardedPassLds + 1024)];
    RPNET_ASSERT_NO_CORESIDENCE(sizeof(smem));
    constexpr int WOFF = 2 * HBUF;                 // weight ring behind the two halo buffers

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    const int tile = xcd_swizzle(blockIdx.x, ntiles);
    const int tm = uni(tile / tiles_n), tn = tile - tm * tiles_n;
    const int n0 = tn * BN;
    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const int pxn = W / TW, ppi = (H / TH) * pxn;
    const int n = uni(tm / ppi), prem = tm - n * ppi;
    const int y0 = uni(prem / pxn) * TH, x0 = uni(prem % pxn) * TW;

    constexpr int CSH = K64 ? 6 : 5;               // channels per K-step: 64 (one plane, two halves) or 32 (per plane)
    // split K (kshift > 0: grid.y = 2^kshift parts, conv_fwd_split_dma below): part blockIdx.y multiplies the channel chunks
    // [cbase, cbase + kchunks) and leaves its tile as rows + blockIdx.y * M of the fp32 workspace d.y0 points to
    const int kchunks = (Cin >> CSH) >> kshift;
    const int cbase = (int)blockIdx.y * kchunks;
    const int nsteps = 9 * kchunks;
    const int rot = uni((int)(blockIdx.x % (unsigned)kchunks));
    auto chunk_c0 = [&](int ci) { int c = rot + ci; if (c >= kchunks) c -= kchunks; return (cbase + c) << CSH; };

    const size_t plane0 = (size_t)d.N * Hs * Ws * d.C0, plane1 = (size_t)d.N * Hs * Ws * d.C1;
    const size_t planew = (size_t)9 * Cin * Cout;
    const unsigned short* x0p = reinterpret_cast<const unsigned short*>(d.x0);
    const unsigned short* x1p = reinterpret_cast<const unsigned short*>(d.x1 ? d.x1 : d.x0);
    const unsigned short* wq = reinterpret_cast<const unsigned short*>(d.w);
    // one descriptor per tensor over all its planes (the plane is part of the scalar offset): 12 SGPRs instead of 36
    const int pb0 = (int)(plane0 * 2), pb1 = (int)((d.x1 ? plane1 : plane0) * 2), pbw = (int)(planew * 2);
    constexpr int NPM = K64 ? 1 : NP;              // planes in memory
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x0p), (short)0, NPM * pb0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x1p), (short)0, NPM * pb1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq), (short)0, NPM * pbw, 0x00020000);
    // scalar offset of plane / half p inside a source and inside the weights (K64: the next 32 channels = 64 bytes further in a
    // pixel's row; the next 32-channel weight slab Cout rows of 64 bytes further)
    const int ps0 = K64 ? 64 : pb0, ps1 = K64 ? 64 : pb1, psw = K64 ? Cout * 64 : pbw;

    // DMA lane geometry: lane l of a 1 KB piece writes the 16 bytes at piece + 16 l = row (l >> 2), slot (l & 3); the slot
    // holds k-group (slot ^ ((row >> 2) & 3)), and pieces start at multiples of 16 rows, so the k-group is a lane constant
    const int drow = lane >> 2;
    const int dkg16 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    // halo: this wave's piece positions wv + 4 i (positions past the last piece repeat the wave's previous one: every wave
    // issues the same number of DMAs, which is what the counted waits below rely on)
    int hoff[HPW];
    int hpos[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        int pos = wv + 4 * i;
        if (pos >= HP) pos -= 4;
        hpos[i] = pos;
        const int hr = pos * 16 + drow;
        const int hy = hr / PW, hx = hr - hy * PW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool inb = hr < HALO && iy >= 0 && iy < H && ix >= 0 && ix < W;
        hoff[i] = inb ? (n * Hs + (iy >> ups)) * Ws + (ix >> ups) : -1;      // -1 -> beyond num_records -> zeros (the padding)
    }
    auto dma_halo = [&](auto ic, int c0, int buf) {
        constexpr int i = decltype(ic)::value;
        const bool first = c0 < d.C0;
        const int Cs = first ? d.C0 : d.C1;
        const int soff = uni((first ? c0 : c0 - d.C0) * 2);
        const int voff = hoff[i] * (Cs * 2) + dkg16;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            auto* dst = (__attribute__((address_space(3))) void*)(smem + buf * HBUF + p * A_BYTES + hpos[i] * 1024);
            if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, soff + p * ps0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, soff + p * ps1, 0, 0);
        }
    };
    // weights: this wave moves rows 16 WN wv .. of every plane of a slab (WN pieces each)
    const int wvoff = (16 * WN * wv + drow) * 64 + dkg16;
    auto dma_w = [&](int tap, int c0, int stage) {
        const int wsoff = uni(((tap * (Cin >> 5) + (c0 >> 5)) * Cout + n0) * 64);
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

    // fragment addresses.  A: MFMA row li of tile i = pixel (wm WM + i) 32 + li of the patch -> halo row + tap offset;
    // slice s reads k-group 2 s + h: the address of slice 1 is that of slice 0 with bit 5 flipped
    int hr00[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mloc = (wm * WM + i) * 32 + li;
        hr00[i] = (mloc / TW) * PW + (mloc % TW);
    }
    const int b_off = WOFF + (wn * WN * 32 + li) * 64 + 16 * (h ^ ((li >> 2) & 3));

    // Register double buffer of the fragments of one 16-deep slice: af / bfr [slice parity][plane][tile].  The issue order of
    // a half K-step is PINNED (sched_barrier between the pieces; left to itself hipcc moves every read next to its
    // consumer, which with one wave per SIMD exposes the LDS latency a dozen times per step):
    //   MFMA m of slice s, then fragment read m of the NEXT slice (m < NR = 12 reads; they are consumed at least 12 MFMAs =
    //   384 cycles later), a DMA behind every second one of the following MFMAs (~60 cycles of issue each).
    // Reads are ordered by first use: the products run l*h, h*l, h*h (smallest first), so the l plane of A and the h plane
    // of B go first.
    bf16x8 af[2][NP][WM], bfr[2][NP][WN];
    int aaddr[WM];                 // per tap: byte address of tile i's A fragment (slice 0, plane 0) inside a halo buffer
    auto a_addr_tap = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int hr = hr00[i] + ky * PW + kx;
            aaddr[i] = hr * 64 + 16 * (h ^ ((hr >> 2) & 3));
        }
    };
    constexpr int NPROD = K64 ? 2 : nprod<NP>();
    constexpr int NR = NP * (WM + WN), NMMA = NPROD * WM * WN;
    static_assert(NR <= NMMA, "one fragment read of the next slice behind each MFMA of this one");
    // read k of a slice, in order of first use: plane order of A = (NP-1 .. 0), of B = (0 .. NP-1); per plane pair: A tile 0,
    // the B tiles, the other A tiles
    auto read_frag = [&](auto sc, auto kc, const int abase, const int bbase) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value;
        constexpr int grp = k / (WM + WN), r = k - grp * (WM + WN);      // grp-th plane pair
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

    // a tile whose input (forward) / output factor (input gradient) is zero everywhere (rpnet_conv_desc.skip_*): no K loop,
    // the epilogue runs on the zero accumulators (bias, statistics, factor) as it would after multiplying zeros
    const bool skip_tile = d.tile_skip != nullptr && d.tile_skip[tm] == 0;
    if (!skip_tile) {
    // ---- prologue: halo of chunk 0, weight slabs of steps 0, 1, 2
    {
        const int c0 = chunk_c0(0);
        dma_halo(std::integral_constant<int, 0>{}, c0, 0);
        if constexpr (HPW > 1) dma_halo(std::integral_constant<int, 1>{}, c0, 0);
        if constexpr (HPW > 2) dma_halo(std::integral_constant<int, 2>{}, c0, 0);
        if constexpr (HPW > 3) dma_halo(std::integral_constant<int, 3>{}, c0, 0);
        if constexpr (HPW > 4) dma_halo(std::integral_constant<int, 4>{}, c0, 0);
        if constexpr (HPW > 5) dma_halo(std::integral_constant<int, 5>{}, c0, 0);
        dma_w(0, c0, 0);
        dma_w(1, c0, 1);
        dma_w(2, c0, 2);
    }
    // the halo and the slabs of steps 0 and 1 have landed (slab 2 may still fly): visible to every wave behind the barrier
    static_assert(NW == 1 || NW == 2 || NW == 4, "weight DMAs per wave and step");
    if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (NW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (NW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    a_addr_tap(I0{});
    static_for<NR>([&](auto kc) { read_frag(I0{}, kc, 0, 0); });

    // One K-step (tap TAP of chunk ci, step ks, weight stage ks & 3):
    //   first half:  MFMAs of slice 0 | reads of slice 1 | DMA of the weight slab of step ks + 3 into the stage that step
    //                ks - 1 used;
    //   second half: MFMAs of slice 1 | reads of step ks + 1 / slice 0 (its slab was waited for BEFORE the previous barrier) |
    //                during taps 0 .. HPW-1 the DMA of one halo piece position of chunk ci + 1 into the other halo buffer
    //                (last read during chunk ci - 1);
    //   before the barrier each wave waits until its part of slab ks + 2 (and every older DMA) has landed — N = the DMAs
    //   it issued since: halo piece of the previous step + slab ks + 3 + halo piece of this step.
    auto step = [&](auto tapc, const int ci, const int ks) {
        constexpr int TAP = decltype(tapc)::value;
        const int hb = ci & 1;
        const bool last_chunk = ci + 1 >= kchunks;
        // the tail of the last chunk re-fetches valid data into buffers nobody reads again (uniform DMA counts)
        constexpr int t3 = TAP + 3 < 9 ? TAP + 3 : TAP - 6;
        const int c3 = chunk_c0((TAP + 3 < 9 || last_chunk) ? ci : ci + 1);
        const int wsoff = uni(((t3 * (Cin >> 5) + (c3 >> 5)) * Cout + n0) * 64);
        const int wstage = (ks + 3) & 3;
        const int abase = hb * HBUF, bbase = (ks & 3) * STAGE;
        static_for<NMMA>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            mma_one(I0{}, mc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < NR) {
                read_frag(I1{}, mc, abase, bbase);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the DMAs of the weight slab behind every second MFMA of the tail of the half step
            if constexpr (m >= NMMA - 2 * NW && (NMMA - 1 - m) % 2 == 0) {
                dma_w_one(std::integral_constant<int, NW - 1 - (NMMA - 1 - m) / 2>{}, wsoff, wstage);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        constexpr int TN = TAP < 8 ? TAP + 1 : 0;
        a_addr_tap(std::integral_constant<int, TN>{});
        const int abase_n = (TAP < 8 ? hb : hb ^ 1) * HBUF, bbase_n = ((ks + 1) & 3) * STAGE;
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMMA>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            mma_one(I1{}, mc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < NR) {
                read_frag(I0{}, mc, abase_n, bbase_n);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (m == NMMA - 1 && TAP < HPW) {
                dma_halo(std::integral_constant<int, (TAP < HPW ? TAP : 0)>{}, chunk_c0(last_chunk ? ci : ci + 1), hb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        constexpr int HPREV = (TAP >= 1 && TAP - 1 < HPW) ? NP : 0;
        constexpr int HCUR = TAP < HPW ? NP : 0;
        constexpr int N = HPREV + NW + HCUR;
        static_assert(N == 1 || N == 2 || N == 3 || N == 4 || N == 6 || N == 8, "wait count");
        if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ablation switches for tools/bench_conv_split.py (rpnet_conv_desc.tune bits 8..: 1 = only the first channel chunk,
    // 2 = no epilogue): where a launch's time goes besides its K-steps
    const int nchunks_run = ((d.tune >> 8) & 1) ? 1 : kchunks;
    int ks = 0;
    for (int ci = 0; ci < nchunks_run; ++ci) {
        step(std::integral_constant<int, 0>{}, ci, ks + 0);
        step(std::integral_constant<int, 1>{}, ci, ks + 1);
        step(std::integral_constant<int, 2>{}, ci, ks + 2);
        step(std::integral_constant<int, 3>{}, ci, ks + 3);
        step(std::integral_constant<int, 4>{}, ci, ks + 4);
        step(std::integral_constant<int, 5>{}, ci, ks + 5);
        step(std::integral_constant<int, 6>{}, ci, ks + 6);
        step(std::integral_constant<int, 7>{}, ci, ks + 7);
        step(std::integral_constant<int, 8>{}, ci, ks + 8);
        ks += 9;
    }
    (void)nsteps;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's DMAs: the epilogue reuses the memory
    __builtin_amdgcn_s_barrier();
    }   // !skip_tile
    const int dbg = d.tune >> 8;
    if (dbg & 2) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 123.456f) d.y0[t] = sacc;
        return;
    }
    conv_epilogue<WM, WN, 2, PatchRows<TW>, false>(d, acc, PatchRows<TW>{(n * H + y0) * W + x0 + (int)blockIdx.y * d.N * HW, W},
                                                    d.N * HW, Cout, HW, n0, tm, wm, wn, li, h, smem);
}

// Sum of the split-K parts and the epilogue the parts left out: y = relu((sum_s part_s + bias) * ep_scale + ep_shift), max |y|,
// y as fp16 / bf16 planes.  A thread owns 8 consecutive channels of a pixel; the parts are added in index order (the result
// does not depend on the grid).  HBM-bound: (S + 1) x 4 bytes per element in, 4 + 2 planes out.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ ws, const int S, const size_t MC, const int Cout,
                                                                 const float* __restrict__ bias, const float* __restrict__ ep_scale,
                                                                 const float* __restrict__ ep_shift, const int relu, float* __restrict__ y,
                                                                 unsigned short* __restrict__ ys, const int planes,
                                                                 const float* __restrict__ ysc, float* __restrict__ out_absmax) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float amax = 0.f;
    if (i * 8 < MC) {
        const int col = (int)((i * 8) % (size_t)Cout);
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + i * 8), b = *reinterpret_cast<const f32x4*>(ws + i * 8 + 4);
        for (int s = 1; s < S; ++s) {
            a += *reinterpret_cast<const f32x4*>(ws + (size_t)s * MC + i * 8);
            b += *reinterpret_cast<const f32x4*>(ws + (size_t)s * MC + i * 8 + 4);
        }
        float v8[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (bias) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] += bias[col + k];
        }
        if (ep_scale) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v8[k] = v8[k] * ep_scale[col + k] + ep_shift[col + k];
                if (relu) v8[k] = fmaxf(v8[k], 0.f);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 8) = f32x4{v8[0], v8[1], v8[2], v8[3]};
        *reinterpret_cast<f32x4*>(y + i * 8 + 4) = f32x4{v8[4], v8[5], v8[6], v8[7]};
#pragma unroll
        for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(v8[k]));
        if (ys) {
            if (planes == 3) {
                u32x4 pl[3];
                split8<3>(v8, pl);
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(ys + p * MC + i * 8) = pl[p];
            } else {
                const float ysinv = 1.f / *ysc;
#pragma unroll
                for (int k = 0; k < 8; ++k) v8[k] *= ysinv;
                if (planes == 2) {
                    u32x4 pl[2];
                    split8<2>(v8, pl);
#pragma unroll
                    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(ys + p * MC + i * 8) = pl[p];
                } else {
                    u32x4 pl[1];
                    split8<1>(v8, pl);
                    *reinterpret_cast<u32x4*>(ys + i * 8) = pl[0];
                }
            }
        }
    }
    if (out_absmax) {
        __shared__ float wave_max[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
            if (amax > __hip_atomic_load(out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned*>(out_absmax), __float_as_uint(amax));
        }
    }
}

// flags[tile] = 1 when the factor f(mask) (mode 1: mask, 2: 1 - mask) is non-zero somewhere on the tile's TH x TW patch
// (+ a one-pixel halo when `halo`): one wave per tile
__global__ __launch_bounds__(64) void mask_tile_flags_kernel(const float* __restrict__ mask, const int mode, const int halo,
                                                              const int H, const int W, const int TW, const int TH,
                                                              unsigned char* __restrict__ flags) {
    const int tm = blockIdx.x, lane = threadIdx.x;
    const int pxn = W / TW, ppi = (H / TH) * pxn;
    const int n = tm / ppi, prem = tm - n * ppi;
    const int y0 = (prem / pxn) * TH - halo, x0 = (prem % pxn) * TW - halo;
    const int ph = TH + 2 * halo, pw = TW + 2 * halo;
    bool any = false;
    for (int i = lane; i < ph * pw; i += 64) {
        const int yy = y0 + i / pw, xx = x0 + i % pw;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float m = mask[((size_t)n * H + yy) * W + xx];
            any = any || (mode == 2 ? (1.f - m) != 0.f : m != 0.f);
        }
    }
    const bool r = __any(any);
    if (lane == 0) flags[tm] = r ? 1 : 0;
}

// Parts (a power of two, 1 = no split) into which conv_fwd_split_dma cuts the K range of this launch, given a workspace
// (rpnet_conv_desc.splitk_ws): launches whose 256 x 64 tiles cover half of the CUs or fewer — the eval-mode calls at batch 2:
// M = 8192 (the 22 CRE convolutions of a T = 10 call), 4096, 1024 — when the epilogue is one the reduce launch has.
int conv_splitk_parts(const rpnet_conv_desc* d, int M, int Cin, int Cout) {
    if (d->split_planes != 2 || d->tune != 0 || d->taps != 9 || d->dilation > 1 || Cin < 128 || M % 256) return 1;
    if (d->stats_partial || d->bnb_partial || d->bnb_y || d->accumulate || d->out_scale_mode || d->Co1 || d->y1 || d->acc_scale_x1 ||
        d->groups > 1)
        return 1;
    const long tiles = (long)(M / 256) * (Cout / 64);
    const int kchunks = Cin >> 5;
    // measured (tools/eval_layers.py, batch 2): a quarter of a machine of tiles or fewer always gains (1024 -> 1024 at M = 1024:
    // 205 -> 70 us in four parts; 512 -> 1024: 103 -> 58 us); half a machine gains only where K is long (1024 -> 512 at
    // M = 4096: 136 -> 117 us in two parts) and loses where the second launch and the partial tiles outweigh half of a short
    // K loop (256 -> 256 at M = 8192: 50 -> 60 us; 512 -> 512 at M = 4096: 75 -> 79 us)
    if (tiles > 128 || (tiles > 64 && kchunks < 32)) return 1;
    int parts = 1;
    while (parts < 8 && tiles * parts * 2 <= 256 && kchunks % (parts * 2) == 0 && kchunks / (parts * 2) >= 2) parts *= 2;
    return parts;
}

// launcher for conv_split.hip: two fp16 planes in 128- (wn = 2) or 64-wide (wn = 1) output tiles, or one fp16 plane in 64-channel
// K-steps (128-wide tiles); whole (256 / TW) x TW patches.  parts > 1 (two planes, wn = 1): split K, see above
int conv_fwd_split_dma(const rpnet_conv_desc* d, int M, int Cin, int Cout, int tw, int wn, hipStream_t s, int parts, int bm) {
    const int tiles_m = M / bm, tiles_n = Cout / (64 * wn);
    const int ntiles = tiles_m * tiles_n;
    rpnet_conv_desc dp = *d;
    dp.tile_skip = nullptr;
    if (d->skip_mask && d->skip_ws && (d->skip_mode == 1 || d->skip_mode == 2) && !d->upsample && d->H % (bm / tw) == 0 && d->W % tw == 0) {
        hipLaunchKernelGGL(mask_tile_flags_kernel, dim3(tiles_m), dim3(64), 0, s, d->skip_mask, d->skip_mode, d->skip_halo ? 1 : 0, d->H,
                           d->W, tw, bm / tw, d->skip_ws);
        dp.tile_skip = d->skip_ws;
    }
    int kshift = 0;
    if (parts > 1) {
        if (d->split_planes != 2 || wn != 1 || (size_t)parts * M * Cout * sizeof(float) > d->splitk_ws_bytes || !d->splitk_ws) {
            set_error("conv_igemm_split_dma: split K needs two planes, 64-wide tiles and the workspace");
            return RPNET_ERR_WORKSPACE;
        }
        while ((1 << kshift) < parts) ++kshift;
        // the parts: scaled accumulators only (the fp16 scales are powers of two, so scaling each part is exact)
        dp.y0 = (float*)d->splitk_ws;
        dp.bias = nullptr; dp.ep_scale = nullptr; dp.ep_shift = nullptr; dp.ep_relu = 0;
        dp.out_absmax = nullptr; dp.y_split = nullptr; dp.split_out_planes = 0; dp.y_split_scale = nullptr;
    }
#define RPNET_DMA(TWV, WNV, K64V) \
    hipLaunchKernelGGL((conv_igemm_split_dma_kernel<TWV, 2, WNV, K64V>), dim3(ntiles, parts), dim3(256), 0, s, dp, Cin, Cout, tiles_n, ntiles, kshift)
    if (d->split_planes == 1) {
        if (wn != 2 || Cin % 64 || d->C0 % 64) {
            set_error("conv_igemm_split_dma: one plane needs 128-wide tiles and channel counts in multiples of 64 per source");
            return RPNET_ERR_ARG;
        }
        if (tw == 32) RPNET_DMA(32, 2, true); else RPNET_DMA(16, 2, true);
    } else if (d->split_planes == 2 && bm == 128) {
        if (wn != 1 || parts != 1) {
            set_error("conv_igemm_split_dma: 128-pixel patches come in 64-wide tiles, unsplit");
            return RPNET_ERR_ARG;
        }
        if (tw == 32)
            hipLaunchKernelGGL((conv_igemm_split_dma_kernel<32, 2, 1, false, 2>), dim3(ntiles, 1), dim3(256), 0, s, dp, Cin, Cout, tiles_n, ntiles, 0);
        else
            hipLaunchKernelGGL((conv_igemm_split_dma_kernel<16, 2, 1, false, 2>), dim3(ntiles, 1), dim3(256), 0, s, dp, Cin, Cout, tiles_n, ntiles, 0);
    } else if (d->split_planes == 2) {
        if (wn == 2) { if (tw == 32) RPNET_DMA(32, 2, false); else RPNET_DMA(16, 2, false); }
        else { if (tw == 32) RPNET_DMA(32, 1, false); else RPNET_DMA(16, 1, false); }
    } else {
        set_error("conv_igemm_split_dma: fp16 planes only");
        return RPNET_ERR_ARG;
    }
#undef RPNET_DMA
    if (parts > 1) {
        const size_t MC = (size_t)M * Cout;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((MC / 8 + 255) / 256)), dim3(256), 0, s, (const float*)d->splitk_ws, parts, MC,
                           Cout, d->bias, d->ep_scale, d->ep_shift, d->ep_relu, d->y0, (unsigned short*)d->y_split, d->split_out_planes,
                           d->y_split_scale, d->out_absmax);
    }
    return check_launch("conv_igemm_split_dma");
}

}  // namespace rpnet
