// Shared helpers for the rpnet HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rpnet_abi.h"

namespace rpnet {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define RPNET_REQUIRE(cond, code, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            rpnet::set_error(__VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wavefront reductions (CDNA: wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ double block_sum256(double v, double* smem4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem4[wv] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// First statement of every HBM-bound pass of the main chain.  A SIMD issues vector instructions of all resident waves through one
// port: beside a resident GEMM wave — the default schedule runs the weight gradients of a layer on a side stream beside the passes
// of the layer below — a pass instruction waits for a gap in that wave's MFMA stream, and the pass takes 2.7 x (BatchNorm backward)
// to 4.9 x (glue backward) its time alone (tools/corun_probe.py; the denser the aggressor's MFMA stream the worse: 3.8 x beside the
// weight-gradient kernel with its loads and LDS reads ablated, 1.7 x beside the register-staged kernel).  At priority 3 the pass's
// instruction goes first whenever both are ready; the GEMM waves are off the critical path.
#define RPNET_PASS_PRIORITY() __builtin_amdgcn_s_setprio(3)

// n / d and n % d of a 32-bit index by a launch constant in 5 (7) vector instructions: Granlund & Montgomery 1994, figure 4.1 —
// exact for every 32-bit n and every d >= 1; the multiplier is made on the host.  hipcc's own `%` / `/` of a size_t loop index by
// a kernel argument is a ~130-instruction sequence PER ITERATION, more than the arithmetic of the element-wise passes themselves
// (the BatchNorm apply passes: 280 instructions of index arithmetic in front of 140 of work) — and beside a GEMM wave these passes
// are bound by instruction issue (RPNET_PASS_PRIORITY above).  Launchers check that their element counts fit 32 bits.
struct FastDiv {
    unsigned m, d, sh1, sh2;
    FastDiv() = default;
    explicit FastDiv(unsigned d_) : d(d_) {
        unsigned l = 0;
        while (l < 32 && (1ull << l) < (unsigned long long)d_) ++l;           // ceil(log2 d)
        m = (unsigned)(((((1ull << l) - d_) << 32) / d_) + 1ull);
        sh1 = l < 1 ? l : 1;
        sh2 = l ? l - 1 : 0;
    }
    __host__ __device__ __forceinline__ unsigned div(const unsigned n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned t = __umulhi(m, n);
#else
        const unsigned t = (unsigned)(((unsigned long long)m * n) >> 32);
#endif
        return (t + ((n - t) >> sh1)) >> sh2;
    }
    __host__ __device__ __forceinline__ unsigned mod(const unsigned n) const { return n - div(n) * d; }
    // q = n / d, returns n % d
    __host__ __device__ __forceinline__ unsigned divmod(const unsigned n, unsigned& q) const { q = div(n); return n - q * d; }
};
constexpr size_t kIndex32 = (size_t)1 << 32;

// most K splits of the single-tap (1x1) weight gradient: its GEMM is a latency chain of 32-pixel steps over six output
// tiles, so more and shorter blocks win until the serial walk of the reduce launch takes the gain back
constexpr int kWgrad1MaxSplits = 128;

// XCD-aware tile order: the dispatcher places block b on XCD b % 8 (each XCD has a private
// L2); give every XCD a contiguous run of logical tiles so neighbouring tiles (which share
// an operand panel) hit the same L2.  Bijective for any tile count.
// Block-uniform integers that come out of a division live in VGPRs (hipcc divides in the vector unit and does not prove the
// quotient uniform): every address built from them becomes vector arithmetic and every scalar operand of a buffer instruction a
// readfirstlane loop.  uni() pins such a value to an SGPR.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int xcd_swizzle(int bid, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace rpnet
