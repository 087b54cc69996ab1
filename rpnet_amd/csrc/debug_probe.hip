// Diagnostic entry points (not on the hot path): probes for the pooled-pass fault (lds_dma.h, profiles/r06_pool_fault.txt).
#include "common.h"

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

}  // namespace rpnet

extern "C" int rpnet_debug_lds_canary(int blocks, int lds_bytes, long long spin_ticks, unsigned* mismatches, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(mismatches && blocks > 0 && lds_bytes >= 1024 && lds_bytes <= 64 * 1024 && lds_bytes % 4 == 0, RPNET_ERR_ARG,
                  "debug_lds_canary: blocks %d lds_bytes %d (1 KB .. 64 KB)", blocks, lds_bytes);
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, lds_bytes / 4, spin_ticks, mismatches);
    return check_launch("debug_lds_canary");
}
