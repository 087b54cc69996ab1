// Mechanism probe for round 4's pooled-pass fault: does a VALU read of element 1 of a 16-byte global load,
// issued right behind the counted `s_waitcnt vmcnt(N)` that covers the load, ever see the OLD register value when the wave shares
// its CU with a kernel that streams through the LDS-DMA path (buffer_load_dwordx4 ... lds)?
//   kernel B (the victim): per iteration eight global_load_dwordx4 from eight distant addresses back to back (inline asm, so that
//     hipcc adds no wait of its own), `s_waitcnt vmcnt(6)` — the two oldest have returned if loads return in order — then at once
//     element .y of the oldest, compared with the value the address must hold; afterwards a full wait and a check of everything
//     (which must never fail).  The registers still hold the previous iteration's (different) values, so a stale read shows.
//   kernel A (the neighbour): one 1024-byte-per-wave LDS-DMA piece after the other from a large buffer into 144 KB of LDS, one
//     block per CU, for the whole time B runs (its own stream).
// Prints the mismatch counts of B alone and of B beside A.  hipcc --offload-arch=gfx950 -O3 -o vmcnt_probe vmcnt_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ __launch_bounds__(256) void victim(const u32x4* __restrict__ data, const size_t n4, const int iters, unsigned* __restrict__ early_bad,
                                              unsigned* __restrict__ late_bad) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    const size_t stride = n4 / 8;
    u32x4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = r0, r7 = r0;
    unsigned eb = 0, lb = 0, second;
    for (int it = 0; it < iters; ++it) {
        const size_t i0 = (tid + (size_t)it * nthreads) % stride;
        const u32x4 *p0 = data + i0, *p1 = p0 + stride, *p2 = p1 + stride, *p3 = p2 + stride, *p4 = p3 + stride, *p5 = p4 + stride,
                    *p6 = p5 + stride, *p7 = p6 + stride;
        unsigned first;
        // r0 / r1 live in FIXED registers so that the statement itself can read element 1 of the oldest load in the very
        // instruction behind the counted wait (as the compiled pass did: `s_waitcnt vmcnt(7); v_mov_b32 v46, v29`)
        asm volatile(
            "global_load_dwordx4 v[100:103], %8, off\n\t"
            "global_load_dwordx4 v[104:107], %9, off\n\t"
            "global_load_dwordx4 %0, %10, off\n\t"
            "global_load_dwordx4 %1, %11, off\n\t"
            "global_load_dwordx4 %2, %12, off\n\t"
            "global_load_dwordx4 %3, %13, off\n\t"
            "global_load_dwordx4 %4, %14, off\n\t"
            "global_load_dwordx4 %5, %15, off\n\t"
            "s_waitcnt vmcnt(6)\n\t"
            "v_mov_b32 %6, v101\n\t"
            "v_mov_b32 %7, v107\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            : "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(first), "=&v"(second)
            : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
        asm volatile("v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
                     "v_mov_b32 %4, v104\n\tv_mov_b32 %5, v105\n\tv_mov_b32 %6, v106\n\tv_mov_b32 %7, v107"
                     : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r1.w)
                     :
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
        // `first` / `second`: element .y of the oldest and .w of the second-oldest load, read in the two instructions behind vmcnt(6)
        if (first != (unsigned)(4 * i0 + 1)) ++eb;
        if (second != (unsigned)(4 * (i0 + stride) + 3)) ++eb;
        // poison the fixed registers so that the next iteration's early read cannot pass on a stale but equal value
        asm volatile("v_mov_b32 v101, -1\n\tv_mov_b32 v107, -1" : : : "v101", "v107");
        const u32x4 rr[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned b = (unsigned)(4 * (i0 + k * stride));
            if (rr[k].x != b || rr[k].y != b + 1 || rr[k].z != b + 2 || rr[k].w != b + 3) ++lb;
        }
    }
    if (eb) atomicAdd(early_bad, eb);
    if (lb) atomicAdd(late_bad, lb);
}

// the LDS-DMA neighbour: every wave moves 1 KB pieces from `src` into its quarter of a 144 KB LDS array, `pieces` times
__global__ __launch_bounds__(256, 1) void neighbour(const unsigned* __restrict__ src, const unsigned bytes, const int pieces, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[147456];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), (short)0, (int)bytes, 0x00020000);
    unsigned off = ((blockIdx.x * 4 + wv) * 4096u) % (bytes - 65536u);
    for (int i = 0; i < pieces; ++i) {
        auto* dst = (__attribute__((address_space(3))) void*)(smem + wv * 36864 + (i % 36) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, lane * 16, (int)off, 0, 0);
        off += 1024u * 1024u;
        if (off >= bytes - 65536u) off -= bytes - 65536u;
        if ((i & 15) == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[5] == 0xff && smem[7777] == 0xfe) *sink = smem[9];
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const size_t n4 = (size_t)64 << 20;                     // 1 GiB of 16-byte records: the eight loads of a thread are 128 MiB apart
    u32x4* data; unsigned *cnt, *nsrc;
    CK(hipMalloc(&data, n4 * 16)); CK(hipMalloc(&cnt, 16)); CK(hipMalloc(&nsrc, (size_t)256 << 20));
    {
        unsigned* h = (unsigned*)malloc(n4 * 16);
        for (size_t i = 0; i < n4 * 4; ++i) h[i] = (unsigned)i;
        CK(hipMemcpy(data, h, n4 * 16, hipMemcpyHostToDevice));
        free(h);
    }
    CK(hipMemset(nsrc, 1, (size_t)256 << 20));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    for (int with_a = 0; with_a < 2; ++with_a) {
        unsigned tot[2] = {0, 0};
        for (int r = 0; r < rounds; ++r) {
            CK(hipMemsetAsync(cnt, 0, 16, sb));
            CK(hipStreamSynchronize(sb));
            if (with_a) hipLaunchKernelGGL(neighbour, dim3(256), dim3(256), 0, sa, (const unsigned*)nsrc, (unsigned)(256u << 20), 60000, cnt + 2);
            hipLaunchKernelGGL(victim, dim3(2048), dim3(256), 0, sb, (const u32x4*)data, n4, 64, cnt, cnt + 1);
            CK(hipDeviceSynchronize());
            unsigned h[2];
            CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
            tot[0] += h[0]; tot[1] += h[1];
        }
        printf("%s: %d rounds x 2048 blocks x 256 threads x 64 iterations: reads right behind the counted wait that saw an old value: %u; "
               "values wrong after the full wait: %u\n", with_a ? "victim beside the LDS-DMA neighbour" : "victim alone", rounds, tot[0], tot[1]);
    }
    return 0;
}
