// 3x3 / 1x1 convolution as an implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).  Forward and dgrad share this
// kernel; see include/rpnet_abi.h (rpnet_conv_desc) for what it replaces in the reference.
//
// GEMM view:  M = N*H*W output pixels, N = Cout, K = taps * Cin.
//   A[m][k]  gathered on the fly from the NHWC source(s): for tap (ky,kx) and channel
//            chunk c0..c0+31 the 32 channels of input pixel (y+ky-1, x+kx-1) are one
//            128-byte line -> 8 lanes x float4, fully coalesced; zero outside the image.
//   B[k][n]  pre-packed weights [tap][Cin/4][Cout][4]: a 32 x BN slab is one linear copy
//            into LDS and a lane's four consecutive k for one n are a single ds_read_b128.
// Block = 256 threads = 4 waves (2 x 2); each wave owns WM x WN tiles of 32x32, so the block
// tile is (64*WM) x (64*WN).  K advances 32 channels per step: 4 groups of 8 k; inside a
// group lane-half h feeds k = 8*g + 4*h + q to MFMA q (A and B use the same permutation of
// k, which a GEMM is free to choose) so that both fragments come from one b128 read.
// LDS: A as [BM][32+4] (the +4 pad makes the 16-lane b128 read groups conflict-free),
// B as [8][BN][4].  Register-prefetch pipeline: global loads for step s+1 are issued before
// the 16*WM*WN MFMAs of step s and written to LDS after them; >= 2 blocks per CU overlap one
// block's staging with the other's matrix work.
#include "conv_epilogue.h"

namespace rpnet {

template <int WM, int WN, bool INSCALE>
__global__ __launch_bounds__(256, (WM * WN >= 8 ? 2 : (WM * WN == 4 ? 3 : 4))) void conv_igemm_kernel(const rpnet_conv_desc d, const int M,
                                                          const int Cin, const int Cout,
                                                          const int tiles_n, const int ntiles) {
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
    constexpr int ASTR = BK + 4;
    constexpr int A_F4 = BM / 32;  // float4 per thread per K-step
    constexpr int B_F4 = BN / 32;
    // single LDS stage (34 KB at 128x128): three blocks per CU.  A double-buffered variant (one
    // barrier per K-step, 70 KB, two blocks per CU) measured 15 % SLOWER: occupancy hides the
    // barrier better than removing it does.
    __shared__ __attribute__((aligned(16))) float smem[cmax(BM * ASTR + BK * BN, epilogue_lds_bytes<WN, 2>() / 4)];

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

    // the A rows this thread stages: r = (t >> 3) + 32 * j
    int rn[A_F4], ry[A_F4], rx[A_F4];
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
        const int m = m0 + (t >> 3) + 32 * j;
        if (m < M) {
            const int n = m / HW, rem = m - n * HW;
            rn[j] = n;
            ry[j] = rem / W;
            rx[j] = rem - ry[j] * W;
        } else {
            rn[j] = -1; ry[j] = 0; rx[j] = 0;
        }
    }
    const int acol = (t & 7) * 4;
    const int kchunks = Cin >> 5;
    const int nsteps = d.taps * kchunks;
    const int Cin4 = Cin >> 2;

    // Load stream state: (tap, channel chunk) of the NEXT tile to fetch.  Everything that depends
    // only on the tap (source pixel of each staged row, border validity, the x*mask factor) is
    // computed once per tap, not once per K-step.
    // The channel-chunk loop of every block starts at a different chunk (rot) so that co-resident
    // blocks do not walk the same 128-byte column of their pixel rows in lockstep.  A GEMM may sum
    // K in any order; the order is fixed per tile, so results stay deterministic.
    const int rot = (int)(blockIdx.x % (unsigned)kchunks);
    int l_tap = 0, l_c0 = rot << 5, l_kc = 0;
    // Tile loads are buffer loads (SRD in SGPRs + 32-bit per-lane offset + scalar offset): no 64-bit
    // address arithmetic per K-step, and a row that falls outside the image simply gets an offset
    // beyond num_records — the hardware bounds check returns zeros, no branch, no select.
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.x0), (short)0, (int)((size_t)d.N * Hs * Ws * d.C0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.x1 ? d.x1 : d.x0), (short)0, (int)((size_t)d.N * Hs * Ws * (d.x1 ? d.C1 : d.C0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.w), (short)0, (int)((size_t)d.taps * Cin * Cout * 4), 0x00020000);
    int roff[A_F4];      // source pixel of each staged row, -1 when the tap falls outside the image
    float rsc[A_F4];     // x*mask factor of that pixel (INSCALE only)
    auto tap_setup = [&](int tap) {
        int ky = 0, kx = 0;
        if (d.taps == 9) { ky = (tap / 3 - 1) * dil; kx = (tap - (tap / 3) * 3 - 1) * dil; }
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int iy = ry[j] + ky, ix = rx[j] + kx;
            const bool inb = rn[j] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int pix = (rn[j] * Hs + (iy >> ups)) * Ws + (ix >> ups);
            roff[j] = inb ? pix : -1;
            if (INSCALE) {
                const float sv = d.in_scale[inb ? pix : 0];
                rsc[j] = d.in_scale_mode == 2 ? 1.f - sv : sv;
            }
        }
    };
    tap_setup(0);
    int wvoff[B_F4];     // per-lane byte offset inside the 8 x BN x 4 weight slab
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
        const int idx = t + 256 * j;
        const int k4 = idx / BN, nn = idx - k4 * BN;
        wvoff[j] = (k4 * Cout + nn) * 16;
    }

    f32x4 ra[A_F4], rb[B_F4];
    float rsc_st[A_F4];  // factor of the tile held in ra (tap_setup may already have moved on)
    auto load_tile = [&]() {
        const bool first = l_c0 < d.C0;
        const int Cs = first ? d.C0 : d.C1;
        const int cc = first ? l_c0 : l_c0 - d.C0;
        const int soff = cc * 4;
        const int cs4 = Cs * 4, ac4 = acol * 4;
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int voff = roff[j] * cs4 + ac4;   // roff == -1 -> beyond num_records -> zeros
            ra[j] = first ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, soff, 0))
                          : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, soff, 0));
            if (INSCALE) rsc_st[j] = rsc[j];
        }
        const int wsoff = ((l_tap * Cin4 + (l_c0 >> 2)) * Cout + n0) * 16;
#pragma unroll
        for (int j = 0; j < B_F4; ++j)
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[j], wsoff, 0));
        l_c0 += 32;
        if (l_c0 == Cin) l_c0 = 0;
        if (++l_kc == kchunks) {
            l_kc = 0;
            if (++l_tap < d.taps) tap_setup(l_tap);
        }
    };
    auto store_tile = [&]() {
        float* As = smem;
        float* Bs = As + BM * ASTR;
#pragma unroll
        for (int j = 0; j < A_F4; ++j)
            *reinterpret_cast<f32x4*>(&As[((t >> 3) + 32 * j) * ASTR + acol]) = INSCALE ? ra[j] * rsc_st[j] : ra[j];
#pragma unroll
        for (int j = 0; j < B_F4; ++j) *reinterpret_cast<f32x4*>(&Bs[(t + 256 * j) * 4]) = rb[j];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma_group = [&](const float* As, const float* Bs, int g) {
        f32x4 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
            af[i] = *reinterpret_cast<const f32x4*>(&As[(wm * WM * 32 + i * 32 + li) * ASTR + g * 8 + h * 4]);
#pragma unroll
        for (int j = 0; j < WN; ++j)
            bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((g * 2 + h) * BN + wn * WN * 32 + j * 32 + li) * 4]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
    };

    // Pipeline: tile s+1 is fetched into registers while the MFMAs of tile s run from LDS, and is
    // written to LDS between the two barriers that close the step.
    load_tile();
    store_tile();
    __syncthreads();
    const float* As = smem;
    const float* Bs = As + BM * ASTR;
    for (int ks = 0; ks < nsteps; ++ks) {
        const bool more = ks + 1 < nsteps;
        if (more) load_tile();
        mma_group(As, Bs, 0);
        mma_group(As, Bs, 1);
        mma_group(As, Bs, 2);
        mma_group(As, Bs, 3);
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }

    conv_epilogue<WM, WN>(d, acc, LinearRows{m0, M}, M, Cout, HW, n0, tm, wm, wn, li, h, smem);
}

template <int WM, int WN, bool INSCALE>
static int launch_igemm_t(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tiles_m = cdiv(M, BM), tiles_n = Cout / BN;
    const int ntiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, INSCALE>), dim3(ntiles), dim3(256), 0, s, *d, M, Cin, Cout, tiles_n, ntiles);
    return check_launch("conv_igemm");
}

template <int WM, int WN>
static int launch_igemm(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    if (d->in_scale_mode) return launch_igemm_t<WM, WN, true>(d, M, Cin, Cout, s);
    return launch_igemm_t<WM, WN, false>(d, M, Cin, Cout, s);
}

// split-bf16 operand variant (conv_split.hip)
int choose_tile_split(const rpnet_conv_desc* d, int M, int Cout);
int split_tile_rows(int variant, int* wave_rows);
int conv_fwd_split(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s);
size_t conv_splitk_bytes(const rpnet_conv_desc* d, int M, int Cin, int Cout);

}  // namespace rpnet


// Tile choice = fewest idle block slots.  Resident blocks per CU follow from the VGPR budget of each
// variant (128x128: 3, 128x64 / 64x128: 4, 64x64: 6); a grid that is not close to a whole number
// of machine waves leaves CUs idle for the whole tail (measured 100 -> 131 TF on the wgrad grid),
// so the variant with the best fill wins, ties go to the larger tile.
static int choose_tile(const rpnet_conv_desc* d, int M, int Cout) {
    const bool n128 = (Cout % 128 == 0) && (d->Co1 == 0 || d->Co0 % 128 == 0);
    struct Cand { int wm, wn, slots; } cands[4] = {{2, 2, 768}, {2, 1, 1024}, {1, 2, 1024}, {1, 1, 1536}};
    int best = -1;
    double best_fill = -1.0;
    for (int c = 0; c < 4; ++c) {
        if (cands[c].wn == 2 && !n128) continue;
        const long tiles = (long)rpnet::cdiv(M, 64 * cands[c].wm) * (Cout / (64 * cands[c].wn));
        const long waves = (tiles + cands[c].slots - 1) / cands[c].slots;
        const double fill = (double)tiles / (double)(waves * cands[c].slots);
        if (fill > best_fill + 0.02) { best_fill = fill; best = c; }
    }
    return best;
}

extern "C" int rpnet_conv_stats_blocks(const rpnet_conv_desc* d) {
    if (!d || d->groups < 1 || d->N % d->groups) return 0;
    const int M = d->N * d->H * d->W, Cout = d->Co0 + d->Co1;
    int bm, wave_rows = 2;
    if (d->split_planes) {
        bm = rpnet::split_tile_rows(rpnet::choose_tile_split(d, M, Cout), &wave_rows);
    } else {
        const int best = choose_tile(d, M, Cout);
        bm = (best == 0 || best == 1) ? 128 : 64;
    }
    const long per_group = (long)(d->N / d->groups) * d->H * d->W;
    if (per_group % bm) return 0;
    return (int)(per_group / bm);        // one partial row per block tile (the wave rows are summed in the epilogue)
}

extern "C" int rpnet_conv_tile_variant(const rpnet_conv_desc* d) {
    if (!d || !d->split_planes) return -1;
    return rpnet::choose_tile_split(d, d->N * d->H * d->W, d->Co0 + d->Co1);
}

extern "C" size_t rpnet_conv_splitk_workspace_bytes(const rpnet_conv_desc* d) {
    if (!d || !d->split_planes || d->taps != 9) return 0;
    return rpnet::conv_splitk_bytes(d, d->N * d->H * d->W, d->C0 + d->C1, d->Co0 + d->Co1);
}

extern "C" int rpnet_conv_fwd(const rpnet_conv_desc* d, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(d && d->x0 && d->w && d->y0, RPNET_ERR_ARG, "conv_fwd: null pointer");
    const int Cin = d->C0 + d->C1, Cout = d->Co0 + d->Co1;
    RPNET_REQUIRE(d->taps == 9 || d->taps == 1, RPNET_ERR_ARG, "conv_fwd: taps must be 9 or 1");
    RPNET_REQUIRE(Cin % 32 == 0 && d->C0 % 32 == 0 && (d->C1 == 0 || d->x1), RPNET_ERR_SHAPE,
                  "conv_fwd: Cin (%d+%d) must be a multiple of 32 per source", d->C0, d->C1);
    RPNET_REQUIRE(Cout % 64 == 0 && (d->Co1 == 0 || (d->y1 && d->Co0 % 64 == 0)), RPNET_ERR_SHAPE,
                  "conv_fwd: Cout (%d+%d) must be a multiple of 64 per destination", d->Co0, d->Co1);
    RPNET_REQUIRE(!d->stats_partial || rpnet_conv_stats_blocks(d) > 0, RPNET_ERR_SHAPE,
                  "conv_fwd: statistic groups do not split into whole tiles; use rpnet_bn_stats");
    RPNET_REQUIRE(!d->upsample || (d->H % 2 == 0 && d->W % 2 == 0), RPNET_ERR_SHAPE, "conv_fwd: odd size with upsample");
    RPNET_REQUIRE(!d->y_split || (d->Co1 == 0 && d->split_out_planes >= 1 && d->split_out_planes <= 3), RPNET_ERR_ARG,
                  "conv_fwd: y_split needs a single destination and 1 to 3 planes");
    RPNET_REQUIRE(!d->y_split || d->split_out_planes == 3 || d->y_split_scale, RPNET_ERR_ARG,
                  "conv_fwd: fp16 output planes (split_out_planes 1 / 2) need y_split_scale");
    RPNET_REQUIRE(!d->y_enc && !d->y_enc_stride, RPNET_ERR_ARG, "conv_fwd: y_enc / y_enc_stride are reserved (must be NULL / 0)");
    RPNET_REQUIRE(!d->bnb_y && !d->bnb_stats && !d->bnb_partial && !d->bnb_pmax, RPNET_ERR_ARG, "conv_fwd: the bnb_* fields are reserved (must be NULL)");
    RPNET_REQUIRE((long)d->N * d->H * d->W < (1L << 31), RPNET_ERR_SHAPE, "conv_fwd: too many pixels");
    RPNET_REQUIRE((size_t)d->N * d->H * d->W * (d->C0 > d->C1 ? d->C0 : d->C1) * 4 < (1UL << 31) &&
                      (size_t)d->taps * Cin * Cout * 4 < (1UL << 31),
                  RPNET_ERR_SHAPE, "conv_fwd: a source tensor exceeds the 2 GiB buffer-descriptor range");
    const int M = d->N * d->H * d->W;
    hipStream_t s = (hipStream_t)stream;
    if (d->split_planes) {
        RPNET_REQUIRE(d->split_planes >= 1 && d->split_planes <= 3 && d->in_scale_mode == 0, RPNET_ERR_ARG,
                      "conv_fwd: split operands take 1 to 3 planes and no in_scale (fold it into rpnet_split_bf16 / _f16)");
        RPNET_REQUIRE(d->split_planes == 3 || (d->acc_scale_col && d->acc_scale_x), RPNET_ERR_ARG,
                      "conv_fwd: one / two planes are fp16 planes of operand / scale: acc_scale_col and acc_scale_x are required");
        return conv_fwd_split(d, M, Cin, Cout, s);
    }
    const int best = choose_tile(d, M, Cout);
    switch (best) {
        case 0: return launch_igemm<2, 2>(d, M, Cin, Cout, s);
        case 1: return launch_igemm<2, 1>(d, M, Cin, Cout, s);
        case 2: return launch_igemm<1, 2>(d, M, Cin, Cout, s);
        default: return launch_igemm<1, 1>(d, M, Cin, Cout, s);
    }
}
