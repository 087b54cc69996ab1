// Can ds_read_b128 fragment loads run under v_mfma_f32_32x32x16_f16 on gfx950, and what does it take?
// One block of 512 threads per CU (2 waves per SIMD), the per-slice instruction mix of the fp16 patch kernel:
// 8 ds_read_b128 (4 A + 4 B fragments) feeding 12 MFMAs on 4 accumulators.
//   mode 0: MFMAs only        mode 1: reads only
//   mode 2: reads, then MFMAs (what the compiler makes of the straightforward loop)
//   mode 3: register double-buffered: the reads of slice i+1 are issued between the MFMAs of slice i (sched_group_barrier)
// hipcc --offload-arch=gfx950 -O3 -o lds_mfma_probe lds_mfma_probe.cpp && ./lds_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE, int NT = 512, bool ZERO = false>
__global__ __launch_bounds__(NT, 1) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // ZERO: all-zero operands (what the chip clocks a pure MFMA stream at depends on the data: DVFS)
    for (int e = t; e < 64 * 1024 / 4; e += NT) reinterpret_cast<float*>(smem)[e] = ZERO ? 0.f : 0.001f * (e & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned char* base = smem + (wv & 3) * 8192 + lane * 16;
    auto rd = [&](int k, int it) { return *reinterpret_cast<const f16x8*>(base + ((k * 1024 + it * 64) & 8191) + ((wv >> 2) & 1) * 32768); };
    f16x8 a[2][4], b[2][4];
    if (MODE == 3 || MODE == 7) { for (int k = 0; k < 4; ++k) { a[0][k] = rd(k, 0); b[0][k] = rd(k + 4, 0); } }
    const f16x8 z = rd(lane & 3, wv);      // real (non-zero) operands for the modes whose MFMAs do not depend on the loop's reads
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            for (int q = 0; q < 3; ++q) for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z, z, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
            for (int k = 0; k < 4; ++k) { a[0][k] = rd(k, it); b[0][k] = rd(k + 4, it); }
            for (int k = 0; k < 4; ++k) acc[k][0] += (float)a[0][k][0] + (float)b[0][k][1];
        } else if (MODE == 4 || MODE == 5) {      // the same bytes as 16 x ds_read_b64
            using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
            for (int k = 0; k < 4; ++k) {
                const unsigned char* pa = base + ((k * 1024 + it * 64) & 8191) + ((wv >> 2) & 1) * 32768;
                const unsigned char* pb = base + (((k + 4) * 1024 + it * 64) & 8191) + ((wv >> 2) & 1) * 32768;
                const f16x4 a0 = *reinterpret_cast<const f16x4*>(pa - lane * 8), a1 = *reinterpret_cast<const f16x4*>(pa - lane * 8 + 512);
                const f16x4 b0 = *reinterpret_cast<const f16x4*>(pb - lane * 8), b1 = *reinterpret_cast<const f16x4*>(pb - lane * 8 + 512);
                a[0][k] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                b[0][k] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (MODE == 4) { for (int k = 0; k < 4; ++k) acc[k][0] += (float)a[0][k][0] + (float)b[0][k][5]; }
            else
                for (int q = 0; q < 3; ++q) for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][(i >> 1) + 2 * (q & 1)], b[0][(i & 1) + 2 * (q >> 1)], acc[i], 0, 0, 0);
        } else if (MODE == 7) {      // coarse double buffering: all reads of slice i+1 first, then the MFMAs of slice i
#define SLICE7(CUR, NXT, IT)                                                                                             \
    for (int k = 0; k < 4; ++k) { a[NXT][k] = rd(k, (IT) + 1); b[NXT][k] = rd(k + 4, (IT) + 1); }                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    for (int q = 0; q < 3; ++q) for (int i = 0; i < 4; ++i)                                                             \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[CUR][(i >> 1) + 2 * (q & 1)], b[CUR][(i & 1) + 2 * (q >> 1)], acc[i], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
            SLICE7(0, 1, it)
            ++it;
            SLICE7(1, 0, it)
#undef SLICE7
        } else if (MODE == 6) {      // the reads are issued and kept alive, the MFMAs do not depend on them
            for (int k = 0; k < 4; ++k) { a[0][k] = rd(k, it); b[0][k] = rd(k + 4, it); }
            for (int q = 0; q < 3; ++q) for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z, z, acc[i], 0, 0, 0);
            for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(a[0][k]), "v"(b[0][k]));
        } else if (MODE == 2) {
            for (int k = 0; k < 4; ++k) { a[0][k] = rd(k, it); b[0][k] = rd(k + 4, it); }
            for (int q = 0; q < 3; ++q) for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][(i >> 1) + 2 * (q & 1)], b[0][(i & 1) + 2 * (q >> 1)], acc[i], 0, 0, 0);
        } else {
            // two slices per trip so that the register buffers are indexed statically
#define SLICE(CUR, NXT, IT)                                                                                              \
    for (int k = 0; k < 4; ++k) { a[NXT][k] = rd(k, (IT) + 1); b[NXT][k] = rd(k + 4, (IT) + 1); }                       \
    for (int q = 0; q < 3; ++q) for (int i = 0; i < 4; ++i)                                                             \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[CUR][(i >> 1) + 2 * (q & 1)], b[CUR][(i & 1) + 2 * (q >> 1)], acc[i], 0, 0, 0); \
    for (int g = 0; g < 8; ++g) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            SLICE(0, 1, it)
            ++it;
            SLICE(1, 0, it)
#undef SLICE
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int NT = 512, bool ZERO = false>
void run(const char* name, int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, NT, ZERO>), dim3(256), dim3(NT), 0, 0, out, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, NT, ZERO>), dim3(256), dim3(NT), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_it_ns = ms * 1e6 / iters;
    const double ideal = 12 * (NT / 256) * 32 / 2.4;
    printf("%-52s %8.1f ns per slice-iteration  (MFMA-only ideal at 2.4 GHz: %.1f ns; MFMA-equivalent clock %.2f GHz)\n", name, per_it_ns,
           ideal, 2.4 * ideal / per_it_ns);
}

int main() {
    const int iters = 20000;
    run<0, 512, true>("MFMAs only, ALL-ZERO operands (2 waves/SIMD)", iters);
    run<0>("MFMAs only (12 per wave, 2 waves/SIMD)", iters);
    run<0, 256, true>("1 wave/SIMD: MFMAs only, ALL-ZERO operands", iters);
    run<1>("ds_read_b128 only (8 per wave)", iters);
    run<2>("reads then MFMAs (compiler order)", iters);
    run<3>("double-buffered, reads between MFMAs", iters);
    run<6>("8 reads + 12 independent MFMAs", iters);
    run<7>("double-buffered, all reads first, then MFMAs", iters);
    run<7, 256>("1 wave/SIMD: same", iters);
    run<4>("16 x ds_read_b64 only", iters);
    run<5>("16 x ds_read_b64, then MFMAs", iters);
    run<0, 256>("1 wave/SIMD: MFMAs only", iters);
    run<1, 256>("1 wave/SIMD: reads only", iters);
    run<2, 256>("1 wave/SIMD: reads then MFMAs", iters);
    run<3, 256>("1 wave/SIMD: double-buffered interleaved", iters);
    run<0, 1024>("4 waves/SIMD: MFMAs only", iters);
    run<1, 1024>("4 waves/SIMD: reads only", iters);
    run<2, 1024>("4 waves/SIMD: reads then MFMAs", iters);
    return 0;
}
