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
#include "common.h"

namespace rpnet {

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const rpnet_conv_desc d, const int M,
                                                          const int Cin, const int Cout,
                                                          const int tiles_n, const int ntiles) {
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
    constexpr int ASTR = BK + 4;
    constexpr int A_F4 = BM / 32;  // float4 per thread per K-step
    constexpr int B_F4 = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[BM * ASTR + BK * BN];
    float* As = smem;
    float* Bs = smem + BM * ASTR;

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

    f32x4 ra[A_F4], rb[B_F4];
    auto load_tile = [&](int ks) {
        const int tap = ks / kchunks;
        const int c0 = (ks - tap * kchunks) << 5;
        int ky = 0, kx = 0;
        if (d.taps == 9) { ky = tap / 3 - 1; kx = tap - (tap / 3) * 3 - 1; }
        const float* src; int Cs, cc;
        if (c0 < d.C0) { src = d.x0; Cs = d.C0; cc = c0; } else { src = d.x1; Cs = d.C1; cc = c0 - d.C0; }
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int iy = ry[j] + ky, ix = rx[j] + kx;
            const bool inb = rn[j] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inb) {
                const size_t pix = ((size_t)rn[j] * Hs + (iy >> ups)) * Ws + (ix >> ups);
                v = *reinterpret_cast<const f32x4*>(src + pix * Cs + cc + acol);
                if (d.in_scale_mode) {
                    float s = d.in_scale[pix];
                    if (d.in_scale_mode == 2) s = 1.f - s;
                    v *= s;
                }
            }
            ra[j] = v;
        }
        const float* wbase = d.w + ((size_t)(tap * Cin4 + (c0 >> 2)) * Cout + n0) * 4;
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int idx = t + 256 * j;
            const int k4 = idx / BN, nn = idx - k4 * BN;
            rb[j] = *reinterpret_cast<const f32x4*>(wbase + ((size_t)k4 * Cout + nn) * 4);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_F4; ++j)
            *reinterpret_cast<f32x4*>(&As[((t >> 3) + 32 * j) * ASTR + acol]) = ra[j];
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

    load_tile(0);
    store_tile();
    __syncthreads();
    for (int ks = 0; ks < nsteps; ++ks) {
        const bool more = ks + 1 < nsteps;
        if (more) load_tile(ks + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int per_group = d.groups > 0 ? (d.N / d.groups) * HW : M;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn * WN * 32 + j * 32 + li;
        const float bv = d.bias ? d.bias[col] : 0.f;
        float* dst; int Cd, cd;
        if (col < d.Co0) { dst = d.y0; Cd = d.Co0; cd = col; } else { dst = d.y1; Cd = d.Co1; cd = col - d.Co0; }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) {
                    float v = acc[i][j][r] + bv;
                    if (d.ep_scale) {
                        const int g = row / per_group;
                        v = v * d.ep_scale[g * Cout + col] + d.ep_shift[g * Cout + col];
                    }
                    if (d.ep_relu) v = fmaxf(v, 0.f);
                    if (d.out_scale_mode) {
                        float s = d.out_scale[row];
                        if (d.out_scale_mode == 2) s = 1.f - s;
                        v *= s;
                    }
                    float* p = dst + (size_t)row * Cd + cd;
                    if (d.accumulate) v += *p;
                    *p = v;
                }
            }
        }
    }
}

template <int WM, int WN>
static int launch_igemm(const rpnet_conv_desc* d, int M, int Cin, int Cout, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tiles_m = cdiv(M, BM), tiles_n = Cout / BN;
    const int ntiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN>), dim3(ntiles), dim3(256), 0, s, *d, M, Cin, Cout, tiles_n, ntiles);
    return check_launch("conv_igemm");
}

}  // namespace rpnet

extern "C" int rpnet_conv_fwd(const rpnet_conv_desc* d, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(d && d->x0 && d->w && d->y0, RPNET_ERR_ARG, "conv_fwd: null pointer");
    const int Cin = d->C0 + d->C1, Cout = d->Co0 + d->Co1;
    RPNET_REQUIRE(d->taps == 9 || d->taps == 1, RPNET_ERR_ARG, "conv_fwd: taps must be 9 or 1");
    RPNET_REQUIRE(Cin % 32 == 0 && d->C0 % 32 == 0 && (d->C1 == 0 || d->x1), RPNET_ERR_SHAPE,
                  "conv_fwd: Cin (%d+%d) must be a multiple of 32 per source", d->C0, d->C1);
    RPNET_REQUIRE(Cout % 64 == 0 && (d->Co1 == 0 || (d->y1 && d->Co0 % 64 == 0)), RPNET_ERR_SHAPE,
                  "conv_fwd: Cout (%d+%d) must be a multiple of 64 per destination", d->Co0, d->Co1);
    RPNET_REQUIRE(!d->upsample || (d->H % 2 == 0 && d->W % 2 == 0), RPNET_ERR_SHAPE, "conv_fwd: odd size with upsample");
    RPNET_REQUIRE((long)d->N * d->H * d->W < (1L << 31), RPNET_ERR_SHAPE, "conv_fwd: too many pixels");
    const int M = d->N * d->H * d->W;
    hipStream_t s = (hipStream_t)stream;
    const bool n128 = (Cout % 128 == 0) && (d->Co1 == 0 || d->Co0 % 128 == 0);
    // pick the largest tile that still gives >= ~2 blocks per CU (256 CUs)
    const long t128 = (long)cdiv(M, 128) * (Cout / 128);
    if (n128 && t128 >= 512) return launch_igemm<2, 2>(d, M, Cin, Cout, s);
    if (n128 && (long)cdiv(M, 64) * (Cout / 128) >= 512) return launch_igemm<1, 2>(d, M, Cin, Cout, s);
    if ((long)cdiv(M, 128) * (Cout / 64) >= 512) return launch_igemm<2, 1>(d, M, Cin, Cout, s);
    return launch_igemm<1, 1>(d, M, Cin, Cout, s);
}
