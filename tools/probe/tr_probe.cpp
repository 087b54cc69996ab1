// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.cpp -o tools/probe/tr_probe; run: tr_probe <row bytes> 0
// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 value = its own element index; every lane passes the
// byte address  (lane >> 2) * ROWB + (lane & 3) * 8  [+ 16-lane group offsets]  and we print what each lane gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(unsigned* out, int rowb, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr;
    if (mode == 0) {        // group g = l>>4: rows 4g..4g+3 ; lane in group L: row L>>2, col quad L&3
        const int g = l >> 4, L = l & 15;
        addr = (4 * g + (L >> 2)) * rowb + (L & 3) * 8;
    } else {                // all lanes same address
        addr = 0;
    }
    unsigned lo, hi;
    unsigned base = (unsigned)(size_t)lds + addr;
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base));
    lo = v[0]; hi = v[1];
    out[l * 2] = lo; out[l * 2 + 1] = hi;
}
int main(int argc, char** argv) {
    int rowb = argc > 1 ? atoi(argv[1]) : 128, mode = argc > 2 ? atoi(argv[2]) : 0;
    unsigned* d; hipMalloc(&d, 64 * 8);
    k<<<1, 64>>>(d, rowb, mode);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        unsigned e0 = h[2*l] & 0xffff, e1 = h[2*l] >> 16, e2 = h[2*l+1] & 0xffff, e3 = h[2*l+1] >> 16;
        int rb2 = rowb / 2;
        printf("lane %2d: elems %5u %5u %5u %5u  -> (row,col) (%u,%u) (%u,%u) (%u,%u) (%u,%u)\n", l, e0, e1, e2, e3,
               e0 / rb2, e0 % rb2, e1 / rb2, e1 % rb2, e2 / rb2, e2 % rb2, e3 / rb2, e3 % rb2);
    }
    return 0;
}
