// Epilogue shared by the implicit-GEMM convolution kernels (fp32-MFMA and split-plane variants):
// the accumulators of a (32*WGM*WM) x (64*WN) block tile held by WGM x 2 waves, each wave WM x WN
// MFMA tiles of 32x32 in the C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
//
// Round 3: rebuilt around what an epilogue costs on this machine.  Measured on the 256 x 128 patch kernels
// (tools/bench_conv_split.py, DBG=2 ablation of tile variant 11): the round-2 epilogue took 32 us of a 194 us launch
// (512 -> 512, one tile per CU) and 128 of the 306 us of the four-tiles-per-CU 128 -> 128 layer — a lane owns ONE column
// of the MFMA tile, so the output left as 4-byte stores (128 per thread for a 128 x 64 wave tile: the store tail is bound
// by instruction issue, cdna_hip_programming.md T21), every element walked a chain of data-independent branches (row
// validity, eval affine, row factor, accumulate) with its group index from an integer division, and writing the final
// value back into the accumulator array moved whole 16-register tiles between the accumulator and vector files.
// Now: column-wise work (power-of-two scales, bias, eval affine, ReLU, BatchNorm statistics) happens in the MFMA layout
// with per-column constants in registers; each wave then turns its 32-row tiles through a private LDS slab so that a
// lane holds 4 consecutive channels of ONE pixel: row-wise work (mask factor, accumulate, max |value|) happens there and
// the output leaves as 16-byte stores — a quarter of the store instructions, no per-element branches (the feature set is
// resolved once per launch into one of four instantiations).
#pragma once
#include "common.h"
#include "split_bf16.h"

namespace rpnet {

// block-local accumulator row -> output pixel (linear NHW index), or -1 when the row is past the tensor
struct LinearRows {   // the block's rows are consecutive pixels m0, m0 + 1, ...
    static constexpr bool kAlwaysValid = false;
    int m0, M;
    __device__ __forceinline__ int operator()(int local) const { const int r = m0 + local; return r < M ? r : -1; }
};
template <int TW>
struct PatchRows {    // the block's rows walk a (rows / TW) x TW patch of one image, row-major
    static constexpr bool kAlwaysValid = true;
    int base, W;      // linear index of the patch's top-left pixel
    __device__ __forceinline__ int operator()(int local) const { return base + (local / TW) * W + (local % TW); }
};

// LDS bytes the epilogue may use (statistics reduction; per-wave 32-row slabs of the output)
template <int WN, int WGM>
constexpr int epilogue_lds_bytes() {
    constexpr int stats = WGM * 64 * WN * (2 * 8 + 4), slabs = WGM * 2 * 32 * (WN * 32 + 4) * 4;   // (sum, sumsq) doubles + a max
    return stats > slabs ? stats : slabs;
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Output of one wave's tiles.  AFF: eval-mode affine (+ ReLU) per (statistic group, column) — the group of a row is looked
// up per row only when the block's rows may straddle groups (LinearRows); ROWOPS: any of row factor / accumulate /
// max |value| / planes of the output.
template <int WM, int WN, typename RowMap, bool AFF, bool ROWOPS>
__device__ __forceinline__ float conv_epilogue_store(const rpnet_conv_desc& d, f32x16 (&acc)[WM][WN], const RowMap rows, const int M,
                                                     const int Cout, const int per_group, const int n0, const int wm,
                                                     const int wn, const int li, const int h, float* slab, const int cso = 0) {
    constexpr int SW = WN * 32 + 4;
    const int lane = li + 32 * h;
    const int colbase = n0 + wn * WN * 32;
    const float xs = d.acc_scale_x ? *d.acc_scale_x : 1.f;     // fp16 split operands: power-of-two tensor / row scales
    float* dstb;
    int Cd, cd0;
    if (colbase < d.Co0) { dstb = d.y0; Cd = d.Co0; cd0 = colbase; } else { dstb = d.y1; Cd = d.Co1; cd0 = colbase - d.Co0; }
    float asv[WN], bvv[WN], es[WN], eh[WN];
    // one statistic group for the whole block tile (always so for image patches; for consecutive pixels when the tile does
    // not cross a group boundary): the affine of that group sits in registers
    const int r_first = rows(wm * WM * 32), r_last = rows(wm * WM * 32 + WM * 32 - 1);
    const int g0 = (r_first >= 0 ? r_first : 0) / per_group;
    const bool one_group = !AFF || !d.ep_scale || RowMap::kAlwaysValid || (r_last >= 0 && r_last / per_group == g0);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = colbase + j * 32 + li;
        bvv[j] = d.bias ? d.bias[col] : 0.f;
        asv[j] = d.acc_scale_col ? d.acc_scale_col[col + cso] * xs : xs;
        es[j] = 1.f;
        eh[j] = 0.f;
        if (AFF && d.ep_scale) {
            es[j] = d.ep_scale[g0 * Cout + col];
            eh[j] = d.ep_shift[g0 * Cout + col];
        }
    }
    float amax = 0.f;
    // the per-row factors of ALL the wave's tiles are fetched up front (they do not depend on the accumulators): one exposed
    // load latency per launch instead of one per 32-row tile
    constexpr int NQA = WN * 4;
    float sc_all[ROWOPS ? WM : 1][NQA];
    if (ROWOPS && d.out_scale_mode) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int q = 0; q < NQA; ++q) {
                const int row = rows(wm * WM * 32 + i * 32 + (q * 64 + lane) / (WN * 8));
                const float f = (RowMap::kAlwaysValid || row >= 0) ? d.out_scale[row] : 0.f;
                sc_all[i][q] = d.out_scale_mode == 2 ? 1.f - f : f;
            }
    }
    unsigned short* ys = reinterpret_cast<unsigned short*>(d.y_split);
    const size_t yplane = (size_t)M * Cout;
    const float ysinv = (ROWOPS && d.y_split && d.y_split_scale) ? 1.f / *d.y_split_scale : 1.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        // ---- MFMA layout -> slab: column-wise work
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                float v = acc[i][j][r] * asv[j] + bvv[j];
                if (AFF) {
                    if (one_group) v = v * es[j] + eh[j];
                    else {
                        const int row = rows(wm * WM * 32 + i * 32 + rl);
                        const int g = (row >= 0 ? row : 0) / per_group;
                        const int col = colbase + j * 32 + li;
                        v = v * d.ep_scale[g * Cout + col] + d.ep_shift[g * Cout + col];
                    }
                    if (d.ep_relu) v = fmaxf(v, 0.f);
                }
                slab[rl * SW + j * 32 + li] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();               // the slab is private to the wave: LDS ops of one wave complete in order
        // ---- slab -> memory: a lane holds 4 consecutive channels of one pixel
        constexpr int NQ = WN * 4;
        int orow[NQ];
        f32x4 v4[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int gidx = q * 64 + lane;            // (row, 4-channel group) of the 32 x (WN * 32) tile
            const int rr = gidx / (WN * 8), c4 = gidx - rr * (WN * 8);
            orow[q] = rows(wm * WM * 32 + i * 32 + rr);
            v4[q] = *reinterpret_cast<const f32x4*>(&slab[rr * SW + c4 * 4]);
        }
        if (ROWOPS) {
            // the loads of a tile's row factors / previous values go out together, in front of the arithmetic (the branches
            // are uniform and sit outside the loops: one wave per SIMD has nothing else to cover a load's latency with)
            if (d.out_scale_mode) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) v4[q] *= sc_all[ROWOPS ? i : 0][q];
            }
            if (d.accumulate) {
                f32x4 prev[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c4 = (q * 64 + lane) % (WN * 8);
                    prev[q] = (RowMap::kAlwaysValid || orow[q] >= 0) ? *reinterpret_cast<const f32x4*>(dstb + (size_t)orow[q] * Cd + cd0 + c4 * 4)
                                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) v4[q] += prev[q];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (RowMap::kAlwaysValid || orow[q] >= 0)
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v4[q][0]), fabsf(v4[q][1]))), fmaxf(fabsf(v4[q][2]), fabsf(v4[q][3])));
            if (d.y_split) {                           // the planes below take the final values
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int gidx = q * 64 + lane;
                    const int rr = gidx / (WN * 8), c4 = gidx - rr * (WN * 8);
                    *reinterpret_cast<f32x4*>(&slab[rr * SW + c4 * 4]) = v4[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c4 = (q * 64 + lane) % (WN * 8);
            if (RowMap::kAlwaysValid || orow[q] >= 0)
                *reinterpret_cast<f32x4*>(dstb + (size_t)orow[q] * Cd + cd0 + c4 * 4) = v4[q];
        }
        __builtin_amdgcn_wave_barrier();
        if (ROWOPS && d.y_split) {
            // the output also as split planes (the operand format of the next convolution; eval mode): a lane holds 8
            // consecutive channels of a pixel = one 16-byte store per plane
#pragma unroll
            for (int q = 0; q < 2 * WN; ++q) {
                const int gidx = q * 64 + lane;        // (row, 8-channel group) of the 32 x (WN * 32) tile
                const int rr = gidx / (WN * 4), cg = gidx - rr * (WN * 4);
                const int row = rows(wm * WM * 32 + i * 32 + rr);
                if (RowMap::kAlwaysValid || row >= 0) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&slab[rr * SW + cg * 8]);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&slab[rr * SW + cg * 8 + 4]);
                    float v8[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                    const size_t o = (size_t)row * Cout + colbase + cg * 8;
                    if (d.split_out_planes == 3) {
                        u32x4 pl[3];
                        split8<3>(v8, pl);
#pragma unroll
                        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(ys + p * yplane + o) = pl[p];
                    } else {
                        // fp16 planes of output / scale (a predicted power-of-two tensor scale: exact)
#pragma unroll
                        for (int k = 0; k < 8; ++k) v8[k] *= ysinv;
                        if (d.split_out_planes == 2) {
                            u32x4 pl[2];
                            split8<2>(v8, pl);
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(ys + p * yplane + o) = pl[p];
                        } else {
                            u32x4 pl[1];
                            split8<1>(v8, pl);
                            *reinterpret_cast<u32x4*>(ys + o) = pl[0];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    return amax;
}

// WIDE (template argument kept for the call sites of round 2): every instantiation stores 16 bytes per lane now.
template <int WM, int WN, int WGM = 2, typename RowMap = LinearRows, bool WIDE = false>
__device__ __forceinline__ void conv_epilogue(const rpnet_conv_desc& d, f32x16 (&acc)[WM][WN], const RowMap rows,
                                              const int M, const int Cout, const int HW, const int n0, const int tm,
                                              const int wm, const int wn, const int li, const int h, void* lds, const int cso = 0) {
    // cso: offset of this tile's columns inside acc_scale_col beyond their output column (conv_up4_dma.hip: the packed weight
    // rows are (phase, cout), the output columns cout)
    const int per_group = d.groups > 0 ? (d.N / d.groups) * HW : M;
    const float xs = d.acc_scale_x ? *d.acc_scale_x : 1.f;     // fp16 split operands: power-of-two tensor / row scales
    if (d.stats_partial) {
        // train-mode BatchNorm statistics of y = acc + bias, fused: this wave's 32*WM rows of each of
        // its columns -> one (sum, sum of squares) pair per column (the two lane halves hold the same
        // columns); the WGM wave rows of the block are then added up through LDS (`lds`: the kernel's
        // staging memory, free by now) so that ONE partial row per block tile goes to memory.
        // Row blocks never straddle a statistic group (host-checked).
        constexpr int BNC = 64 * WN;
        double* red = reinterpret_cast<double*>(lds);     // [WGM][BNC][2]
        __syncthreads();                                   // every wave is done with the staging memory
        bool valid[WM][16];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                valid[i][r] = RowMap::kAlwaysValid || rows(wm * WM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) >= 0;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn * WN * 32 + j * 32 + li;
            const float bv = d.bias ? d.bias[col] : 0.f;
            const float as = d.acc_scale_col ? d.acc_scale_col[col + cso] * xs : xs;
            // fp64 from the first add on: the variance is formed as E[y^2] - mean^2, and an fp32 running sum of squares
            // would carry ~1e-7 mean^2 of error into it (channels with |mean| >> std)
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double v = valid[i][r] ? (double)(acc[i][j][r] * as + bv) : 0.0;
                    sm += v;
                    sq += v * v;
                }
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (h == 0) {
                double* o = red + ((size_t)wm * BNC + wn * WN * 32 + j * 32 + li) * 2;
                o[0] = sm;
                o[1] = sq;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < BNC * 2; e += WGM * 128) {
            double sacc = 0.0;
#pragma unroll
            for (int r = 0; r < WGM; ++r) sacc += red[(size_t)r * BNC * 2 + e];
            d.stats_partial[((size_t)tm * Cout + n0) * 2 + e] = sacc;
        }
    }
    __syncthreads();                                       // staging memory / statistics reduction: every wave is done with it
    float* slab = reinterpret_cast<float*>(lds) + (size_t)(wm * 2 + wn) * 32 * (WN * 32 + 4);
    const bool aff = d.ep_scale != nullptr || d.ep_relu != 0;
    const bool rowops = d.out_scale_mode != 0 || d.accumulate != 0 || d.out_absmax != nullptr || d.y_split != nullptr;
    float amax = 0.f;
    if (!aff && !rowops) conv_epilogue_store<WM, WN, RowMap, false, false>(d, acc, rows, M, Cout, per_group, n0, wm, wn, li, h, slab, cso);
    else if (!aff) amax = conv_epilogue_store<WM, WN, RowMap, false, true>(d, acc, rows, M, Cout, per_group, n0, wm, wn, li, h, slab, cso);
    else if (!rowops) conv_epilogue_store<WM, WN, RowMap, true, false>(d, acc, rows, M, Cout, per_group, n0, wm, wn, li, h, slab, cso);
    else amax = conv_epilogue_store<WM, WN, RowMap, true, true>(d, acc, rows, M, Cout, per_group, n0, wm, wn, li, h, slab, cso);
    if (d.out_absmax) {      // max |output| of the launch: one order-independent atomic per wave (values >= 0: uint order = float order)
        // and only where it would raise the stored value (it only grows: a stale read costs a redundant atomic, never a
        // maximum) — same-address atomics serialise in L2 at ~11 ns each, 16 K waves of a 256^2-level launch = 0.17 ms
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((threadIdx.x & 63) == 0 && amax > __hip_atomic_load(d.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned*>(d.out_absmax), __float_as_uint(amax));
    }
}

}  // namespace rpnet
