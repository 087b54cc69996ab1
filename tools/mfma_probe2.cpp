// Which ingredient of the igemm K-step costs MFMA throughput?  Variants add, one at a time:
//  1 = MFMAs fed from LDS fragment reads (ds_read_b128), 2 = + two barriers per step,
//  3 = + LDS tile stores (ds_write_b128), 4 = + global tile loads (L2-resident source)
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef RANDOMIZE
#define RANDOMIZE 0
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int V>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* out, int steps) {
    constexpr int BM = 128, BN = 128, ASTR = 36;
    __shared__ __attribute__((aligned(16))) float smem[BM * ASTR + 32 * BN];
    float* As = smem; float* Bs = smem + BM * ASTR;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5, wm = wv >> 1, wn = wv & 1;
    for (int i = t; i < BM * ASTR + 32 * BN; i += 256) { unsigned hsh = (i * 2654435761u) ^ (blockIdx.x * 40503u); smem[i] = RANDOMIZE ? ((hsh >> 8) & 0xFFFF) * (1.0f / 32768.f) - 1.0f : 0.001f * (i & 63); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4], rb[4];
    for (int j = 0; j < 4; ++j) { ra[j] = f32x4{1, 2, 3, 4}; rb[j] = f32x4{1, 2, 3, 4}; }
    const float* gp = src + ((size_t)blockIdx.x * 256 + t) * 4;
    for (int s = 0; s < steps; ++s) {
        if (V >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[j] = *(const f32x4*)(gp + (size_t)((s * 8 + j) & 63) * 262144); rb[j] = *(const f32x4*)(gp + (size_t)((s * 8 + 4 + j) & 63) * 262144); }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const f32x4*)&As[(wm * 64 + i * 32 + li) * ASTR + g * 8 + h * 4];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const f32x4*)&Bs[((g * 2 + h) * BN + wn * 64 + j * 32 + li) * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
        }
        if (V >= 2) __syncthreads();
        if (V >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)&As[((t >> 3) + 32 * j) * ASTR + (t & 7) * 4] = ra[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)&Bs[(t + 256 * j) * 4] = rb[j];
        }
        if (V >= 2) __syncthreads();
    }
    float sum = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + t] = sum;
}
template <int V> void run(const float* src, float* out, int blocks, int steps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256>>>(src, out, 8);
    hipEventRecord(e0);
    probe<V><<<blocks, 256>>>(src, out, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * steps * 64 * 4096.0;
    printf("variant %d blocks %4d: %.3f ms  %.1f TF\n", V, blocks, ms, fl / ms / 1e9);
}
int main() {
    float *src, *out; hipMalloc(&src, (size_t)64 * 262144 * 4 + (1 << 25)); hipMemset(src, RANDOMIZE ? 0x3b : 0, (size_t)64 * 262144 * 4 + (1 << 25)); hipMalloc(&out, 4096 * 256 * 4);
    for (int blocks : {512, 768, 1536}) { run<1>(src, out, blocks, 288); run<2>(src, out, blocks, 288); run<3>(src, out, blocks, 288); run<4>(src, out, blocks, 288); }
    printf("-- real grid shapes\n");
    run<3>(src, out, 1024, 72); run<4>(src, out, 1024, 72); run<3>(src, out, 512, 144); run<4>(src, out, 512, 144); run<3>(src, out, 2048, 72); run<4>(src, out, 4096, 72);
    return 0;
}
