// fp32 MFMA peak probe: what does v_mfma_f32_32x32x2_f32 sustain on this box (clock included)?
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 768, 1024}) for (int iters : {2000, 20000}) {
        probe<4><<<blocks, 256>>>(out, 100, 1.f, 1.f);
        hipEventRecord(e0);
        probe<4><<<blocks, 256>>>(out, iters, 1.f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
        printf("blocks %4d iters %6d: %.3f ms  %.1f TF\n", blocks, iters, ms, fl / ms / 1e9);
    }
    return 0;
}
