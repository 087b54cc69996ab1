// Weight gradient of the 3x3 / 1x1 convolutions on the fp32 matrix cores, plus the weight
// (re)packing kernels.  See include/rpnet_abi.h for the reference operators replaced.
//
// wgrad GEMM view, one per filter tap:  dWp[cin][cout] = sum_p A[p + tap][cin] * dy[p][cout]
//   M = Cin (gathered, incl. padding rows), N = Cout, K = N*H*W pixels (split across blocks).
// With NHWC activations both operands are K-major in memory (a pixel's channels are
// contiguous), which is exactly what the MFMA fragments want from LDS ([k][m] / [k][n] rows,
// conflict-free ds_read_b32), so tiles are staged by straight float4 row copies.
// grid = (tiles_m * tiles_n, taps, ksplit); each block writes its partial tile to the
// workspace [ksplit][taps][Cin][Cout]; rpnet_wgrad_reduce sums the splits and transposes
// into the nn.Conv2d state_dict layout [Cout][Cin][kh][kw] through LDS so that both the
// reads (along cout) and the writes (along cin,tap) are coalesced.
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace rpnet {

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const rpnet_conv_desc d, const float* __restrict__ dy,
                                                          float* __restrict__ partial, const int M, const int Cin,
                                                          const int Cout, const int tiles_n, const int steps_per_split) {
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
    constexpr int A_F4 = BM / 32, B_F4 = BN / 32;
    constexpr int A_RW = BM / 4, B_RW = BN / 4;  // float4 per pixel row
    __shared__ __attribute__((aligned(16))) float smem[BK * BM + BK * BN];
    float* As = smem;
    float* Bs = smem + BK * BM;

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int cm0 = tm * BM, n0 = tn * BN;
    const int tap = blockIdx.y;
    int ky = 0, kx = 0;
    const int dil = d.dilation > 1 ? d.dilation : 1;
    if (d.taps == 9) { ky = (tap / 3 - 1) * dil; kx = (tap - (tap / 3) * 3 - 1) * dil; }

    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const float* src; int Cs, cc;
    if (cm0 < d.C0) { src = d.x0; Cs = d.C0; cc = cm0; } else { src = d.x1; Cs = d.C1; cc = cm0 - d.C0; }

    const int total_steps = (M + BK - 1) / BK;
    const int s_begin = blockIdx.z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    f32x4 ra[A_F4], rb[B_F4];
    auto load_tile = [&](int st) {
        const int p0 = st * BK;
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int idx = t + 256 * j;
            const int prow = idx / A_RW, c4 = idx - prow * A_RW;
            const int p = min(p0 + prow, M - 1);   // clamped: loads stay unconditional (no branch, no per-load wait)
            const int n = p / HW, rem = p - n * HW;
            const int oy = rem / W, ox = rem - oy * W;
            const int iy = oy + ky, ix = ox + kx;
            const bool ok = p0 + prow < M && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const size_t pix = ok ? ((size_t)n * Hs + (iy >> ups)) * Ws + (ix >> ups) : 0;
            float sc = ok ? 1.f : 0.f;
            if (d.in_scale_mode) {
                const float sv = d.in_scale[pix];
                sc *= d.in_scale_mode == 2 ? 1.f - sv : sv;
            }
            ra[j] = *reinterpret_cast<const f32x4*>(src + pix * Cs + cc + c4 * 4) * sc;
        }
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int idx = t + 256 * j;
            const int prow = idx / B_RW, c4 = idx - prow * B_RW;
            const int p = p0 + prow;
            rb[j] = *reinterpret_cast<const f32x4*>(dy + (size_t)min(p, M - 1) * Cout + n0 + c4 * 4) * (p < M ? 1.f : 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_F4; ++j) *reinterpret_cast<f32x4*>(&As[(t + 256 * j) * 4]) = ra[j];
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

    if (s_begin < s_end) {
        load_tile(s_begin);
        store_tile();
        __syncthreads();
        for (int st = s_begin; st < s_end; ++st) {
            const bool more = st + 1 < s_end;
            if (more) load_tile(st + 1);
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp) {
                const int k = kp * 2 + h;
                float af[WM], bf[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) af[i] = As[k * BM + wm * WM * 32 + i * 32 + li];
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[j] = Bs[k * BN + wn * WN * 32 + j * 32 + li];
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) store_tile();
            __syncthreads();
        }
    }

    float* out = partial + ((size_t)(blockIdx.z * d.taps + tap) * Cin) * Cout;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn * WN * 32 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cm0 + wm * WM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)row * Cout + col] = acc[i][j][r];
            }
        }
}


// ---------------------------------------------------------------------------------------------
// 3x3 weight gradient, all nine taps in one block.  The single-tap kernel above streams the x
// and dy pixel rows once per tap (arithmetic intensity 16-32 FLOP/B at the L2: bandwidth-bound,
// 32-80 TF measured).  Here a block owns a 64(cin) x 64(cout) tile for ALL taps: per K-step of
// 32 pixels it stages three x strips S_ky[j] = x[p0 - 1 + j + ky*W] (j = 0..33; rows whose
// source image row is outside the image are zero) and the dy rows once, and runs 9 x 16 MFMAs
// per wave against nine accumulators (144 registers): A(tap ky,kx)[k] = S_ky[k + kx + 1].  The
// horizontal border (ox + kx outside the row) cannot be folded into the strips because one
// strip row serves three taps; it is folded into the OTHER operand instead: dy is staged three
// times (as is, zeroed where ox = 0, zeroed where ox = W-1) and tap kx reads its own copy, so the
// MFMA loop is LDS reads + matrix instructions only.  69 FLOP per staged byte, two blocks per CU.
template <bool POW2, bool INSCALE>
__global__ __launch_bounds__(256, 2) void conv_wgrad9_kernel(const rpnet_conv_desc d, const float* __restrict__ dy,
                                                              float* __restrict__ partial, const int M, const int Cin,
                                                              const int Cout, const int tiles, const int tiles_n,
                                                              const int ksplit, const int steps_per_split,
                                                              const int lw, const int lh) {
    constexpr int BM = 64, BN = 64, BK = 32, SJ = BK + 2;
    constexpr int A_IT = (3 * SJ * (BM / 4) + 255) / 256;  // 7
    constexpr int A_SZ = 3 * SJ * BM, B_SZ = BK * BN;
    __shared__ __attribute__((aligned(16))) float smem[A_SZ + 3 * B_SZ];   // As [3][34][64], Bs [3 variants][32][64]

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    // XCD-aware order: the blocks of one pixel chunk (all tiles) run on the same XCD, so the
    // x / dy rows they share are fetched into that XCD's L2 once.
    int tile, z;
    if ((ksplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        z = (j / tiles) * 8 + xcd;
        tile = j - (j / tiles) * tiles;
    } else {
        z = blockIdx.x / tiles;
        tile = blockIdx.x - z * tiles;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int cm0 = tm * BM, n0 = tn * BN;

    const int H = d.H, W = d.W, HW = H * W;
    const int ups = d.upsample;
    const int Hs = H >> ups, Ws = W >> ups;
    const float* src; int Cs, cc;
    if (cm0 < d.C0) { src = d.x0; Cs = d.C0; cc = cm0; } else { src = d.x1; Cs = d.C1; cc = cm0 - d.C0; }

    const int total_steps = (M + BK - 1) / BK;
    const int s_begin = z * steps_per_split;
    const int s_end = min(s_begin + steps_per_split, total_steps);

    // buffer loads: SRD + 32-bit lane offset + scalar offset; anything outside the tensor (pixels
    // past M, rows outside the image) gets an offset beyond num_records and reads as zero.
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src), (short)0, (int)((size_t)d.N * Hs * Ws * Cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dy), (short)0, (int)((size_t)M * Cout * 4), 0x00020000);
    const int cs4 = Cs * 4;
    // per-thread constants of the seven strip elements it stages: q = p0 + qoff[i]
    int qoff[A_IT], kyv[A_IT], c16[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int e = t + 256 * i;
        const int r = e >> 4;
        const int kyi = r / SJ, j = r - kyi * SJ;
        kyv[i] = kyi - 1;
        qoff[i] = j - 1 + (kyi - 1) * W;
        c16[i] = (e & 15) * 16;
    }
    f32x4 ra[A_IT], rb[2];
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
            const int voff = ok ? pix * cs4 + c16[i] : (int)0x80000000;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, voff, cc * 4, 0));
            if (INSCALE) {
                const float sv = d.in_scale[ok ? pix : 0];
                ra[i] *= d.in_scale_mode == 2 ? 1.f - sv : sv;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = t + 256 * i;
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, (p0 + (e >> 4)) * Cout * 4 + (e & 15) * 16, n0 * 4, 0));
        }
    };
    auto store_tile = [&](int st) {
        float* As = smem;
        float* Bs = smem + A_SZ;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int e = t + 256 * i;
            if (e < 3 * SJ * (BM / 4)) *reinterpret_cast<f32x4*>(&As[e * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = t + 256 * i;
            const int p = st * BK + (e >> 4);
            const int ox = POW2 ? (p & (W - 1)) : (p % W);
            *reinterpret_cast<f32x4*>(&Bs[e * 4]) = rb[i] * (ox >= 1 ? 1.f : 0.f);               // kx = -1
            *reinterpret_cast<f32x4*>(&Bs[B_SZ + e * 4]) = rb[i];                                 // kx =  0
            *reinterpret_cast<f32x4*>(&Bs[2 * B_SZ + e * 4]) = rb[i] * (ox <= W - 2 ? 1.f : 0.f);  // kx = +1
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    if (s_begin < s_end) {
        load_tile(s_begin);
        store_tile(s_begin);
        __syncthreads();
        const float* ap = smem + wm * 32 + li + h * BM;          // row k = 2*kp + h
        const float* bp = smem + A_SZ + wn * 32 + li + h * BN;
        for (int st = s_begin; st < s_end; ++st) {
            const bool more = st + 1 < s_end;
            if (more) load_tile(st + 1);
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp) {
                const float bm = bp[kp * 2 * BN], b0 = bp[B_SZ + kp * 2 * BN], b1 = bp[2 * B_SZ + kp * 2 * BN];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float am = ap[(ky * SJ + kp * 2) * BM];
                    const float a0 = ap[(ky * SJ + kp * 2 + 1) * BM];
                    const float a1 = ap[(ky * SJ + kp * 2 + 2) * BM];
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bm, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[ky * 3 + 2], 0, 0, 0);
                }
            }
            __syncthreads();
            if (more) store_tile(st + 1);
            __syncthreads();
        }
    }
    if (z >= ksplit) return;
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

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// split-K plan of the nine-tap kernels: tiles * ksplit = target_blocks (512 = two resident blocks on each of
// the 256 CUs for the fp32 kernel, 256 = one per CU for the split-bf16 kernel: one full wave of the machine
// either way) whenever the K extent allows >= 8 steps per block;
// the partial buffer is then 512 * 9 * 64 * 64 * 4 B = 75 MB whatever the layer.
void wgrad9_plan(int M, int Cin, int Cout, int* ksplit, int* steps_per_split, int target_blocks) {
    const int tiles = (Cin / 64) * (Cout / 64);
    const int total_steps = (M + 31) / 32;
    long ks = tiles >= target_blocks ? 1 : (target_blocks + tiles - 1) / tiles;
    ks = std::min<long>(ks, std::max(1, total_steps / 8));
    if (ks >= 8) ks = (ks / 8) * 8;
    *steps_per_split = (int)((total_steps + ks - 1) / ks);
    int k2 = (total_steps + *steps_per_split - 1) / *steps_per_split;
    if (ks >= 8) k2 = ((k2 + 7) / 8) * 8;  // keep the XCD mapping; surplus chunks are empty
    *ksplit = k2;
}

// partial [ksplit][taps][Cin_g][Cout]  ->  dw [Cout][cin_w][taps]
// grid (cin/32, cout/32[, taps]): a block sums one 32x32 tile (of one tap, or of all taps) over the K splits
// (8 row groups x 32 lanes read 128-byte rows of every split) and writes it transposed.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                            int ksplit, int taps, int Cin_g, int Cout, int cin_w,
                                                            int off0, int split, int off1, int accumulate,
                                                            const float* __restrict__ sx, const float* __restrict__ sdy,
                                                            const float* __restrict__ sx1 = nullptr, const int c0_rows = 0) {
    // one block = a 32 (cin) x 32 (cout) tile of ALL taps: the sums over the K splits land in LDS as
    // [tap][cin][cout] and leave as rows of 32 cin x taps contiguous floats per output channel — the
    // state_dict layout [Cout][Cin][taps] written with full lines instead of 4-byte pieces 36 bytes apart
    __shared__ float tile[9][32][33];
    const int t = threadIdx.x;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const size_t zstride = (size_t)taps * Cin_g * Cout;
    // gridDim.z == taps: one tap per block (small layers: more blocks matter more than full-line writes)
    const int tap_lo = gridDim.z > 1 ? blockIdx.z : 0, tap_hi = gridDim.z > 1 ? blockIdx.z + 1 : taps;
    // thread = (cin row t >> 3, four output channels 4 (t & 7) ..): 16-byte loads, a tap's 32 x 32 tile per pass of the
    // block, three taps (twelve loads per thread) in flight — 4-byte loads, four in flight, ran at 1.6 TB/s
    const int rr = t >> 3, c4 = (t & 7) * 4;
    const int cin = ci0 + rr;
    const bool ok = cin < cin_w;
    const int row = ok ? (cin < split ? off0 + cin : off1 + (cin - split)) : 0;
    // fp16 split operands: the two power-of-two tensor scales (gathered rows of the second source: its own scale)
    const float scl = sx ? ((sx1 && row >= c0_rows) ? *sx1 : *sx) * *sdy : 1.f;
    auto sum_tap = [&](int tap) -> f32x4 {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        if (ok) {
            const float* p = partial + ((size_t)tap * Cin_g + row) * Cout + co0 + c4;
            int z = 0;
            for (; z + 4 <= ksplit; z += 4) {
                s0 += *reinterpret_cast<const f32x4*>(p + (size_t)z * zstride);
                s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(z + 1) * zstride);
                s2 += *reinterpret_cast<const f32x4*>(p + (size_t)(z + 2) * zstride);
                s3 += *reinterpret_cast<const f32x4*>(p + (size_t)(z + 3) * zstride);
            }
            for (; z < ksplit; ++z) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)z * zstride);
        }
        f32x4 s = (s0 + s1) + (s2 + s3);
        if (sx) s *= scl;
        return s;
    };
    auto put = [&](int tap, const f32x4 s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[tap][rr][c4 + k] = s[k];
    };
    int tap = tap_lo;
    for (; tap + 3 <= tap_hi; tap += 3) {
        const f32x4 a0 = sum_tap(tap), a1 = sum_tap(tap + 1), a2 = sum_tap(tap + 2);
        put(tap, a0); put(tap + 1, a1); put(tap + 2, a2);
    }
    for (; tap < tap_hi; ++tap) put(tap, sum_tap(tap));
    __syncthreads();
    const int ncin = min(32, cin_w - ci0);
    if (gridDim.z > 1) {
        for (int e = t; e < 32 * ncin; e += 256) {
            const int col = e / ncin, c = e - col * ncin;
            float* o = dw + ((size_t)(co0 + col) * cin_w + ci0 + c) * taps + tap_lo;
            *o = accumulate ? *o + tile[tap_lo][c][col] : tile[tap_lo][c][col];
        }
        return;
    }
    const int rowlen = ncin * taps;               // contiguous floats per output channel
    for (int e = t; e < 32 * rowlen; e += 256) {
        const int col = e / rowlen, r = e - col * rowlen;
        const int c = r / taps, tap = r - c * taps;
        float* o = dw + ((size_t)(co0 + col) * cin_w + ci0) * taps + r;
        *o = accumulate ? *o + tile[tap][c][col] : tile[tap][c][col];
    }
}

// grid of the reduce: all taps per block (full-line writes) once there are enough 32 x 32 tiles to fill the machine
static dim3 reduce_grid(int cin_w, int Cout, int taps) {
    const int tx = cdiv(cin_w, 32), ty = Cout / 32;
    return dim3(tx, ty, (long)tx * ty >= 256 ? 1 : taps);
}

// w [Cout][cin_w][taps] -> wp [taps][Cin_g/4][Cout][4] (+ wd [taps][Cout/4][Cin_g][4], taps flipped)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                           float* __restrict__ wd, int taps, int Cin_g, int Cout,
                                                           int cin_w, int off0, int split, int off1) {
    __shared__ float tile[9][32][33];  // [tap][cin_l][cout_l]
    const int t = threadIdx.x;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int ncin = min(32, cin_w - ci0);
    const int nel = ncin * taps;
    // 32 cout rows of `nel` contiguous floats each: one flat, fully independent load loop
    for (int e = t; e < 32 * nel; e += 256) {
        const int col = e / nel, r = e - col * nel;
        const int c = r / taps, tap = r - c * taps;
        tile[tap][c][col] = w[((size_t)(co0 + col) * cin_w + ci0) * taps + r];
    }
    __syncthreads();
    // wp: for (tap, cin_l): 32 consecutive cout, element [row/4][cout][row%4]
    for (int e = t; e < taps * 32 * 32; e += 256) {
        const int cl = e & 31, c = (e >> 5) & 31, tap = e >> 10;
        if (c < ncin) {
            const int cin = ci0 + c;
            const int row = cin < split ? off0 + cin : off1 + (cin - split);
            wp[(((size_t)tap * (Cin_g >> 2) + (row >> 2)) * Cout + co0 + cl) * 4 + (row & 3)] = tile[tap][c][cl];
        }
    }
    if (wd) {
        // wd: GEMM K = cout, N = gathered cin: element [tapflip][cout/4][row][cout%4]
        for (int e = t; e < taps * 32 * 32; e += 256) {
            const int q = e & 3, c = (e >> 2) & 31, cq = (e >> 7) & 7, tap = e >> 10;
            if (c < ncin) {
                const int cin = ci0 + c;
                const int row = cin < split ? off0 + cin : off1 + (cin - split);
                const int cout = co0 + cq * 4 + q;
                const int tf = taps - 1 - tap;  // (2-ky, 2-kx) for 3x3; 0 for 1x1
                wd[(((size_t)tf * (Cout >> 2) + (cout >> 2)) * Cin_g + row) * 4 + q] = tile[tap][c][cq * 4 + q];
            }
        }
    }
}

static void wgrad_plan(int M, int Cin, int Cout, int taps, int* bm, int* bn, int* ksplit, int* steps_per_split) {
    *bm = (Cin % 128 == 0 && taps > 1) ? 128 : 64;
    *bn = (Cout % 128 == 0) ? 128 : 64;
    long tiles = (long)(Cin / *bm) * (Cout / *bn) * taps;
    if (tiles < 1) tiles = 1;  // unsupported shape: rpnet_conv_wgrad rejects it, keep the plan finite
    const int total_steps = (M + 31) / 32;
    int ks = (int)((768 + tiles - 1) / tiles);          // aim at >= ~3 blocks per CU
    ks = max(1, min(ks, max(1, total_steps / 8)));       // at least 8 K-steps per block
    // a 1x1 conv has few output tiles (384 x 64 here): the reduce kernel walks the splits serially with only a
    // handful of blocks, so deep splits cost more there than they win in the GEMM (measured 73 -> 45 us)
    if (taps == 1) ks = min(ks, kWgrad1MaxSplits);
    *steps_per_split = (total_steps + ks - 1) / ks;
    if (taps == 9) *steps_per_split += *steps_per_split & 1;      // even: the one-plane DMA kernel takes 64-pixel steps
    *ksplit = (total_steps + *steps_per_split - 1) / *steps_per_split;
}

// split-bf16 operands (conv_wgrad_split.hip)
int conv_wgrad9_split(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9,
                      hipStream_t s);
bool conv_wgrad9_dma_one_plane_ok(const rpnet_conv_desc* d, int M, int sps9);
int conv_wgrad9_split_dma(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9,
                          hipStream_t s);
bool conv_wgrad9_ring_ok(const rpnet_conv_desc* d, int M, int sps9);
int conv_wgrad9_ring(const rpnet_conv_desc* d, const void* dy, float* part9, int M, int Cin, int Cout, int ks9, int sps9, hipStream_t s);
void wgrad1_split_plan(int M, int Cin, int Cout, int* ksplit, int* steps_per_split);
int conv_wgrad1_split(const rpnet_conv_desc* d, const void* dy, float* part, int M, int Cin, int Cout, int ks, int sps, hipStream_t s);

}  // namespace rpnet

extern "C" size_t rpnet_conv_wgrad_workspace_bytes(int N, int H, int W, int cin_gathered, int cout, int taps) {
    int bm, bn, ks, sps;
    rpnet::wgrad_plan(N * H * W, cin_gathered, cout, taps, &bm, &bn, &ks, &sps);
    size_t need = (size_t)ks * taps * cin_gathered * cout * sizeof(float);
    if (taps == 9 && cin_gathered % 64 == 0 && cout % 64 == 0) {   // nine-tap kernel (dense) or single-tap (dilated): max
        int ks9, sps9;
        rpnet::wgrad9_plan(N * H * W, cin_gathered, cout, &ks9, &sps9, 512);   // the larger of the two plans
        need = std::max(need, (size_t)ks9 * 9 * cin_gathered * cout * sizeof(float));
    }
    return need;
}

extern "C" int rpnet_conv_wgrad(const rpnet_conv_desc* d, const float* dy, float* dw, int cin_w, int cin_off0,
                                int cin_split, int cin_off1, void* workspace, size_t workspace_bytes,
                                rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(d && d->x0 && workspace && (dy || dw), RPNET_ERR_ARG, "conv_wgrad: null pointer");
    RPNET_REQUIRE((dy && dw) || (d->split_planes && d->dilation <= 1), RPNET_ERR_ARG,
                  "conv_wgrad: the two-phase form (dy or dw NULL) exists for the split kernels only");
    const int Cin = d->C0 + d->C1, Cout = d->Co0 + d->Co1;
    RPNET_REQUIRE(d->taps == 9 || d->taps == 1, RPNET_ERR_ARG, "conv_wgrad: taps must be 9 or 1");
    RPNET_REQUIRE(!d->split_planes || d->dilation <= 1, RPNET_ERR_ARG,
                  "conv_wgrad: split operands are implemented for dense 3x3 and 1x1 taps only");
    RPNET_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, RPNET_ERR_SHAPE, "conv_wgrad: Cin %d / Cout %d not multiples of 64", Cin, Cout);
    RPNET_REQUIRE((long)d->N * d->H * d->W < (1L << 31), RPNET_ERR_SHAPE, "conv_wgrad: too many pixels");
    const int M = d->N * d->H * d->W;
    RPNET_REQUIRE((size_t)M * (d->C0 > d->C1 ? d->C0 : d->C1) * 4 < (1UL << 31) && (size_t)M * Cout * 4 < (1UL << 31),
                  RPNET_ERR_SHAPE, "conv_wgrad: an operand exceeds the 2 GiB buffer-descriptor range (split the batch)");
    hipStream_t s = (hipStream_t)stream;
    if (d->taps == 9 && d->dilation <= 1) {
        RPNET_REQUIRE(d->C1 == 0 || d->C0 % 64 == 0, RPNET_ERR_SHAPE, "conv_wgrad: source split %d not aligned to 64", d->C0);
        int ks9, sps9;
        wgrad9_plan(M, Cin, Cout, &ks9, &sps9, d->split_planes ? 256 : 512);
        const size_t need9 = (size_t)ks9 * 9 * Cin * Cout * sizeof(float);
        RPNET_REQUIRE(workspace_bytes >= need9, RPNET_ERR_WORKSPACE, "conv_wgrad: workspace %zu < %zu", workspace_bytes, need9);
        const int tiles_n9 = Cout / 64, tiles9 = (Cin / 64) * tiles_n9;
        const int lw = ilog2_exact(d->W), lh = ilog2_exact(d->H);
        float* part9 = (float*)workspace;
        if (d->split_planes) {   // x0/x1 and dy are split-bf16 planes
            RPNET_REQUIRE(d->split_planes >= 1 && d->split_planes <= 3 && d->in_scale_mode == 0, RPNET_ERR_ARG,
                          "conv_wgrad: split operands take 1 to 3 planes and no in_scale");
            if (dy) {    // dy == NULL: reduce phase only (rpnet_conv_wgrad_reduce)
                // two fp16 planes: the LDS-DMA kernel (conv_wgrad_split_dma.hip; same partial sums, bit for bit); tune 8 =
                // the register-staged 12-wave kernel of round 2 (A/B switch), 4 = its 4-wave layout
                // (one fp16 plane: the same kernel with the two halves of a 64-pixel step in the two plane slots, where the
                // image rows are at least that long — conv_wgrad9_dma_one_plane_ok)
                // round 6: where the images are power-of-two and at least a K-step wide, the same kernel with its K-steps walked
                // down the image columns (conv_wgrad_ring.hip: every x strip fetched once; equal to rounding, not bit for bit);
                // tune 16 keeps the row-major kernel (A/B switch)
                const int tv = d->tune & 255;
                const bool dma = (d->split_planes == 2 || conv_wgrad9_dma_one_plane_ok(d, M, sps9)) && tv != 8 && tv != 4;
                const bool ring = dma && tv != 16 && conv_wgrad9_ring_ok(d, M, sps9);
                if (int rc = ring ? conv_wgrad9_ring(d, dy, part9, M, Cin, Cout, ks9, sps9, s)
                             : dma ? conv_wgrad9_split_dma(d, dy, part9, M, Cin, Cout, ks9, sps9, s)
                                   : conv_wgrad9_split(d, dy, part9, M, Cin, Cout, ks9, sps9, s))
                    return rc;
            }
            if (!dw) return RPNET_OK;       // GEMM phase only: the partial sums stay in the workspace
            RPNET_REQUIRE(d->split_planes == 3 || (d->acc_scale_x && d->acc_scale_dy), RPNET_ERR_ARG,
                          "conv_wgrad: fp16 planes need acc_scale_x and acc_scale_dy");
            hipLaunchKernelGGL(wgrad_reduce_kernel, reduce_grid(cin_w, Cout, 9), dim3(256), 0, s, part9, dw, ks9, 9, Cin,
                               Cout, cin_w, cin_off0, cin_split, cin_off1, d->accumulate,
                               d->split_planes <= 2 ? d->acc_scale_x : nullptr, d->acc_scale_dy);
            return check_launch("wgrad_reduce");
        }
#define RPNET_W9(P2, IS, LW, LH)                                                                                   \
    hipLaunchKernelGGL((conv_wgrad9_kernel<P2, IS>), dim3(tiles9 * ks9), dim3(256), 0, s, *d, dy, part9, M, Cin, Cout, \
                       tiles9, tiles_n9, ks9, sps9, LW, LH)
        const bool p2 = lw >= 0 && lh >= 0, is = d->in_scale_mode != 0;
        if (p2 && is) RPNET_W9(true, true, lw, lh);
        else if (p2) RPNET_W9(true, false, lw, lh);
        else if (is) RPNET_W9(false, true, 0, 0);
        else RPNET_W9(false, false, 0, 0);
#undef RPNET_W9
        int rc9 = check_launch("conv_wgrad9");
        if (rc9) return rc9;
        hipLaunchKernelGGL(wgrad_reduce_kernel, reduce_grid(cin_w, Cout, 9), dim3(256), 0, s, part9, dw, ks9, 9, Cin,
                           Cout, cin_w, cin_off0, cin_split, cin_off1, d->accumulate, (const float*)nullptr, (const float*)nullptr);
        return check_launch("wgrad_reduce");
    }
    if (d->split_planes && d->taps == 1) {      // x0/x1 and dy are split planes: the single-tap split kernel
        RPNET_REQUIRE(d->split_planes >= 1 && d->split_planes <= 3 && d->in_scale_mode == 0 && d->upsample == 0, RPNET_ERR_ARG,
                      "conv_wgrad: split 1x1 operands take 1 to 3 planes, no in_scale, no upsampling");
        RPNET_REQUIRE(d->C1 == 0 || (d->C0 % 64 == 0 && d->x1), RPNET_ERR_SHAPE, "conv_wgrad: source split %d not aligned to 64", d->C0);
        int ks1, sps1;
        wgrad1_split_plan(M, Cin, Cout, &ks1, &sps1);
        const size_t need1 = (size_t)ks1 * Cin * Cout * sizeof(float);
        RPNET_REQUIRE(workspace_bytes >= need1, RPNET_ERR_WORKSPACE, "conv_wgrad: workspace %zu < %zu", workspace_bytes, need1);
        float* part1 = (float*)workspace;
        if (dy)
            if (int rc = conv_wgrad1_split(d, dy, part1, M, Cin, Cout, ks1, sps1, s)) return rc;
        if (!dw) return RPNET_OK;
        RPNET_REQUIRE(d->split_planes == 3 || (d->acc_scale_x && d->acc_scale_dy), RPNET_ERR_ARG,
                      "conv_wgrad: fp16 planes need acc_scale_x and acc_scale_dy");
        hipLaunchKernelGGL(wgrad_reduce_kernel, reduce_grid(cin_w, Cout, 1), dim3(256), 0, s, part1, dw, ks1, 1, Cin, Cout, cin_w,
                           cin_off0, cin_split, cin_off1, d->accumulate, d->split_planes <= 2 ? d->acc_scale_x : nullptr,
                           d->acc_scale_dy, d->split_planes <= 2 ? d->acc_scale_x1 : nullptr, d->C0);
        return check_launch("wgrad_reduce");
    }
    int bm, bn, ks, sps;
    wgrad_plan(M, Cin, Cout, d->taps, &bm, &bn, &ks, &sps);
    RPNET_REQUIRE(d->C1 == 0 || d->C0 % bm == 0, RPNET_ERR_SHAPE, "conv_wgrad: source split %d not aligned to tile %d", d->C0, bm);
    const size_t need = (size_t)ks * d->taps * Cin * Cout * sizeof(float);
    RPNET_REQUIRE(workspace_bytes >= need, RPNET_ERR_WORKSPACE, "conv_wgrad: workspace %zu < %zu", workspace_bytes, need);
    const int tiles_n = Cout / bn, tiles = (Cin / bm) * tiles_n;
    dim3 grid(tiles, d->taps, ks);
    float* part = (float*)workspace;
    if (bm == 128 && bn == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, s, *d, dy, part, M, Cin, Cout, tiles_n, sps);
    else if (bm == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, *d, dy, part, M, Cin, Cout, tiles_n, sps);
    else if (bn == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 2>), grid, dim3(256), 0, s, *d, dy, part, M, Cin, Cout, tiles_n, sps);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, *d, dy, part, M, Cin, Cout, tiles_n, sps);
    int rc = check_launch("conv_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, reduce_grid(cin_w, Cout, d->taps), dim3(256), 0, s, part, dw, ks, d->taps,
                       Cin, Cout, cin_w, cin_off0, cin_split, cin_off1, d->accumulate, (const float*)nullptr, (const float*)nullptr);
    return check_launch("wgrad_reduce");
}

extern "C" int rpnet_pack_conv_weight(const float* w, float* wp, float* wd, int cout, int cin, int taps, int cin_off0,
                                      int cin_split, int cin_off1, int cin_pad, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(w && wp, RPNET_ERR_ARG, "pack_conv_weight: null pointer");
    RPNET_REQUIRE(cout % 32 == 0 && cin_pad % 4 == 0 && (taps == 9 || taps == 1), RPNET_ERR_SHAPE,
                  "pack_conv_weight: cout %d cin_pad %d taps %d", cout, cin_pad, taps);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cdiv(cin, 32), cout / 32), dim3(256), 0, (hipStream_t)stream, w, wp, wd,
                       taps, cin_pad, cout, cin, cin_off0, cin_split, cin_off1);
    return check_launch("pack_conv_weight");
}
