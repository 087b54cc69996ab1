// LDS-DMA (buffer_load_dwordx4 ... lds) as inline assembly, hidden from hipcc.
//
// Why not __builtin_amdgcn_raw_ptr_buffer_load_lds everywhere: hipcc (ROCm 7.2) tracks a builtin DMA as a pending LDS write and
// puts `s_waitcnt vmcnt(0)` in front of every later LDS access it cannot prove disjoint.  Plain loads of the one __shared__
// array pass (conv_split_dma.hip uses the builtin), but ds_read_b64_tr_b16 — an intrinsic without alias information — gets
// the full drain before EVERY read, which serialises a pipeline whose point is DMAs in flight across barriers
// (cdna_hip_programming.md §5 "Three .s-level traps", §5.7).  An asm statement is invisible to that bookkeeping: no VGPR
// destination, so nothing for the compiler to protect; the data is ordered by the kernel's own counted
// `s_waitcnt vmcnt(N)` + s_barrier (which it needs anyway).
// M0 (the LDS destination base) is written inside the statement that uses it.
#pragma once
#include "common.h"

namespace rpnet {

// The pooled BatchNorm passes (bn.hip: bn_relu_pool_split_kernel, bn_bwd_apply_pool_split) made wrong 2 x 2 window decisions when their
// blocks shared a CU with a block of an LDS-DMA kernel (profiles/r04_pool_apply_fault.txt, r05_pool_fault_repro.txt, r06_pool_fault.txt:
// mechanism unknown; not a host-side lifetime race).  The guard is STRUCTURAL: those passes reserve kGuardedPassLds bytes of LDS they do
// not use, and every kernel that issues `buffer_load ... lds` asserts at compile time (RPNET_ASSERT_NO_CORESIDENCE) that its own
// allocation plus that reservation exceeds a CU's LDS — a new tile variant that would fit beside a guarded block does not build.
constexpr int kLdsPerCu = 160 * 1024;
constexpr int kGuardedPassLds = 52 * 1024;
#define RPNET_ASSERT_NO_CORESIDENCE(lds_bytes)                                                                                    \
    static_assert((lds_bytes) + rpnet::kGuardedPassLds > rpnet::kLdsPerCu,                                                         \
                  "an LDS-DMA kernel form that fits on a CU beside a guarded pooled BatchNorm block (bn.hip pool_alone_bytes)")

using srd_t = __attribute__((ext_vector_type(4))) unsigned;

// raw buffer descriptor over `bytes` bytes at p (stride 0, bounds-checked: offsets >= bytes read as zero)
__device__ __forceinline__ srd_t make_srd(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    return srd_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

// LDS byte address of a pointer into a __shared__ array
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

// 64 lanes x 16 bytes: lane l reads srd[voff_l + soff .. +16) and the DMA engine writes it to LDS at lds_base + 16 l
// (lds_base, soff and srd wave-uniform).  Counts on vmcnt; no register result.
__device__ __forceinline__ void lds_dma16(const srd_t srd, const unsigned lds_base, const int voff, const int soff) {
    // M0 is not restored: hipcc keeps nothing live in M0 across statements in these kernels (no LDS-DMA builtins, no
    // ds_*_addtid, no movrel) — checked in the .s: every other M0 write is followed by its own use
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(lds_base), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
}

// the same with the LDS destination as scalar base + compile-time offset (one s_add into M0 instead of an add and a move)
template <int OFF>
__device__ __forceinline__ void lds_dma16_at(const srd_t srd, const unsigned lds_base, const int voff, const int soff) {
    asm volatile(
        "s_add_u32 m0, %0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(lds_base), "v"(voff), "s"(srd), "s"(soff), "n"(OFF)
        : "memory", "scc");
}

}  // namespace rpnet
