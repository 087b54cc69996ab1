// Diagnostic entry points (not on the hot path): probes for the pooled-pass fault (lds_dma.h, profiles/r06_pool_fault.txt).
#include "common.h"
#include "split_bf16.h"

namespace rpnet {

// LDS canary: every block fills `lds_bytes` of dynamic LDS with a pattern of its own, then re-reads all of it again and again for
// about `spin_cycles` shader clocks and counts words that changed.  Launched beside the LDS-DMA kernels with an allocation small enough
// to share their CUs, it answers one question: does a `buffer_load ... lds` of a NEIGHBOURING workgroup ever write outside that
// workgroup's own LDS allocation?  (mismatches[0] = changed words, mismatches[1] = blocks that ran, mismatches[2] = verification sweeps.)
__global__ __launch_bounds__(256) void lds_canary_kernel(const int words, const long long spin_cycles, unsigned* __restrict__ mismatches) {
    extern __shared__ unsigned canary[];
    const unsigned seed = 0x9e3779b9u * (blockIdx.x + 1);
    for (int i = threadIdx.x; i < words; i += 256) canary[i] = seed ^ (unsigned)(i * 2654435761u);
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned bad = 0, sweeps = 0;
    // (wall_clock64 ticks at 100 MHz on gfx950: spin_cycles is given in those ticks)
    while (wall_clock64() - t0 < spin_cycles) {
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned v = canary[i];
            if (v != (seed ^ (unsigned)(i * 2654435761u))) {
                ++bad;
                canary[i] = seed ^ (unsigned)(i * 2654435761u);
            }
        }
        ++sweeps;
    }
    if (bad) atomicAdd(&mismatches[0], bad);
    if (threadIdx.x == 0) {
        atomicAdd(&mismatches[1], 1u);
        atomicAdd(&mismatches[2], sweeps);
    }
}

// MFMA spinner: every block (4 waves, one per SIMD) issues v_mfma_f32_32x32x16_f16 back to back for `spin_ticks` ticks of the 100 MHz
// wall clock — a GEMM's matrix-core load and power draw without its memory traffic, on as many CUs as the caller launches blocks for
// (`lds_bytes` of dynamic LDS keep it at one block per CU).  out[0] = shader cycles, out[1] = wall ticks (their ratio is the shader
// clock under this load), out[2] = MFMAs one wave issued.  tools/corun_probe.py (AGG=spin:<blocks>) runs the passes of the main chain
// beside it: is their slowdown beside a GEMM local to the CUs the GEMM occupies, or the whole chip's (clock / power)?
// PAD: `s_nop 7` statements (8 cycles each) behind every MFMA — does a wave whose next MFMA is not yet at the issue stage leave the
// SIMD's vector issue port to the other waves?
template <int PAD>
__global__ __launch_bounds__(256) void mfma_spin_kernel(const long long spin_ticks, unsigned long long* __restrict__ out) {
    extern __shared__ unsigned canary[];
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (float)((threadIdx.x + k) & 15)); b[k] = (_Float16)(0.002f * (float)((threadIdx.x * 3 + k) & 7)); }
    const long long t0 = wall_clock64(), c0 = clock64();
    unsigned long long n = 0;
    do {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < PAD; ++q) asm volatile("s_nop 7");
                if (PAD) __builtin_amdgcn_sched_barrier(0);
            }
        n += 64;
    } while (wall_clock64() - t0 < spin_ticks);
    const long long c1 = clock64(), t1 = wall_clock64();
    float sink = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc[j][r];
    if (sink == 12345.678f) canary[threadIdx.x] = 1u;          // keeps the accumulators alive
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (unsigned long long)(c1 - c0); out[1] = (unsigned long long)(t1 - t0); out[2] = n; }
}

}  // namespace rpnet

extern "C" int rpnet_debug_mfma_spin(int blocks, int lds_bytes, long long spin_ticks, unsigned long long* out, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(out && blocks > 0 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && spin_ticks > 0 && spin_ticks <= 100000000ll, RPNET_ERR_ARG,
                  "debug_mfma_spin: blocks %d lds_bytes %d spin_ticks %lld (<= 1 s)", blocks, lds_bytes, spin_ticks);
    // blocks = count + 65536 * pad (pad 0 .. 3: `s_nop 7` statements behind every MFMA)
    const int pad = blocks >> 16;
    blocks &= 0xffff;
    RPNET_REQUIRE(blocks > 0 && pad <= 3, RPNET_ERR_ARG, "debug_mfma_spin: blocks %d pad %d", blocks, pad);
#define RPNET_SPIN(P_)                                                                                                              \
    do {                                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_spin_kernel<P_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
        hipLaunchKernelGGL(mfma_spin_kernel<P_>, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, spin_ticks, out);   \
    } while (0)
    if (pad == 0) RPNET_SPIN(0);
    else if (pad == 1) RPNET_SPIN(1);
    else if (pad == 2) RPNET_SPIN(2);
    else RPNET_SPIN(3);
#undef RPNET_SPIN
    return check_launch("debug_mfma_spin");
}

extern "C" int rpnet_debug_lds_canary(int blocks, int lds_bytes, long long spin_ticks, unsigned* mismatches, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(mismatches && blocks > 0 && lds_bytes >= 1024 && lds_bytes <= 64 * 1024 && lds_bytes % 4 == 0, RPNET_ERR_ARG,
                  "debug_lds_canary: blocks %d lds_bytes %d (1 KB .. 64 KB)", blocks, lds_bytes);
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, lds_bytes / 4, spin_ticks, mismatches);
    return check_launch("debug_lds_canary");
}

// Host-only self-test of rpnet::FastDiv (common.h) against the C operators: edge values around multiples of d and 2^k, and a random
// sweep, for small, power-of-two, large and model-sized divisors.  Returns the number of mismatches (0 = exact).
extern "C" long long rpnet_debug_fastdiv_selftest(int random_per_divisor) {
    using rpnet::FastDiv;
    static const unsigned ds[] = {1u, 2u, 3u, 4u, 5u, 6u, 7u, 8u, 9u, 10u, 12u, 15u, 16u, 17u, 20u, 24u, 31u, 32u, 33u, 40u, 48u, 63u, 64u, 65u,
                                  96u, 121u, 128u, 255u, 256u, 257u, 1000u, 1023u, 1024u, 1025u, 4095u, 4096u, 65535u, 65536u, 65537u,
                                  1048576u, 1048577u, 16777215u, 16777216u, 100000007u, 2147483647u, 2147483648u, 2147483649u,
                                  4294967294u, 4294967295u};
    long long bad = 0;
    unsigned long long lcg = 0x9e3779b97f4a7c15ull;
    for (unsigned d : ds) {
        const FastDiv f(d);
        auto check = [&](unsigned n) {
            unsigned q;
            const unsigned r = f.divmod(n, q);
            if (q != n / d || r != n % d || f.mod(n) != n % d) ++bad;
        };
        for (unsigned k = 0; k < 40; ++k) {
            const unsigned long long base = (unsigned long long)d * k;
            for (int o = -2; o <= 2; ++o) {
                const long long n = (long long)base + o;
                if (n >= 0 && n <= 0xffffffffll) check((unsigned)n);
            }
        }
        for (int b = 0; b < 32; ++b)
            for (int o = -2; o <= 2; ++o) {
                const long long n = (1ll << b) + o;
                if (n >= 0 && n <= 0xffffffffll) check((unsigned)n);
            }
        for (unsigned n = 0xfffffff0u; n != 0u; ++n) check(n);          // up to 2^32 - 1
        for (int i = 0; i < random_per_divisor; ++i) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            check((unsigned)(lcg >> 32));
        }
    }
    return bad;
}
