// probe: buffer_load_dwordx4 ... offen lds on gfx950 -- destination addressing (M0 + lane*16), out-of-range lanes
// (do they write zeros?), soffset and counted vmcnt.  hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o /tmp/probe_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, i32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc),
                 "s"(soff)
                 : "memory");
}

__global__ void probe(const unsigned* src, unsigned bytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    // poison
    for (int i = lane; i < 1024; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long pa = (unsigned long long)src;
    i32x4 rs;
    rs[0] = (int)(unsigned)pa;
    rs[1] = (int)((unsigned)(pa >> 32) & 0xffffu);
    rs[2] = (int)bytes;
    rs[3] = 0x00020000;
    const unsigned lbase = (unsigned)(size_t)smem;   // LDS byte address of the array
    // piece 0: lane l reads 16 B at (63-l)*16 (reversed), odd lanes out of range
    unsigned voff = (63 - lane) * 16;
    if (lane & 1) voff = 0x80000000u;
    dma16(lbase, voff, rs, 0);
    // piece 1: at LDS +1024 (via m0), soffset 1024
    dma16(lbase + 1024, (unsigned)lane * 16, rs, 1024);
    // piece 2: at +2048 with every lane out of range
    dma16(lbase + 2048, 0x80000000u + lane * 16, rs, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = lane; i < 1024; i += 64) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}

int main() {
    const int n = 4096;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x10000000u + i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, 2048u, o);
    std::vector<unsigned> r(1024);
    hipError_t e = hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    printf("status %d\n", (int)e);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            unsigned want = (l & 1) ? 0u : 0x10000000u + (63 - l) * 4 + k;
            if (r[l * 4 + k] != want) { if (bad < 8) printf("piece0 lane %d dw %d: got %08x want %08x\n", l, k, r[l * 4 + k], want); ++bad; }
        }
    printf("piece0 (reversed, odd lanes OOB->zero): %s\n", bad ? "MISMATCH" : "ok");
    bad = 0;
    for (int i = 0; i < 256; ++i) {
        // bytes limit is 2048 -> soffset 1024 + lane*16 < 2048 valid
        unsigned want = 0x10000000u + 256 + i;
        if (r[256 + i] != want) { if (bad < 8) printf("piece1 dw %d: got %08x want %08x\n", i, r[256 + i], want); ++bad; }
    }
    printf("piece1 (m0 +1024, soffset 1024): %s\n", bad ? "MISMATCH" : "ok");
    bad = 0;
    for (int i = 0; i < 256; ++i)
        if (r[512 + i] != 0u) { if (bad < 8) printf("piece2 dw %d: got %08x\n", i, r[512 + i]); ++bad; }
    printf("piece2 (all OOB -> zeros): %s\n", bad ? "MISMATCH" : "ok");
    printf("untouched tail: %08x\n", r[800]);
    return 0;
}
