// Shared pieces of the direct-to-LDS convolution pipelines (fprop_dma.hip, fprop_rw.hip): LDS-DMA issue, raw barriers
// with counted waits, order-pinned fragment reads / MFMAs.  See fprop_dma.hip for the scheme.
#pragma once
#include "common.h"

#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct FdArgs {
    const bf16_t* x;
    const bf16_t* w;
    unsigned x_bytes, w_bytes, out_bytes;
    const float* bias;
    int bias_n;
    bf16_t* out;
    double* stats;
    int N, H, W;          // output grid
    int Hi, Wi;           // input tensor
    int Ci, Co, ld_x, ld_out, Ktot;
    int dhmin, dwmin;
    int dh[9], dw[9];     // tap offsets minus (dhmin, dwmin): 0..2
    int HB, WB, IT, NTL, GM, NCH;
    int NCHP;             // plane gather (WsCfg::UPD): 64-channel chunks per source plane (NCH = 4 * NCHP)
    // virtual concat (segnb_conv_fprop_upcat): the first NCHU channel chunks of the input are the nearest-x2 upsample of
    // u [N][Hi/2][Wi/2][ld_u] -- halo pixel (hi, wi) is fetched from u pixel (hi >> 1, wi >> 1) -- the others come from x
    // (the skip tensor); the upsampled copy is never materialised (lib/models/zf_unet.py:42,78-90).  u == NULL: off
    const bf16_t* u;
    unsigned u_bytes;
    int ld_u, NCHU, Hu, Wu;
    // fused Upsample(x2) backward in the data gradient's store pass (segnb_conv_fprop_upsum, fprop_rw.hip): the first up_C
    // output channels are NOT stored at this resolution -- each 2 x 2 window of the staged tile is summed (fp32 sum of the
    // four bf16 values, rounded once: what the consumer's 2 x 2 sum of the stored slice computed) and written to up_out
    // [N][H/2][W/2][up_ld]; the other channels are stored as usual.  up_out == NULL: off
    bf16_t* up_out;
    int up_C, up_ld;
    int no_prev;    // UPF: 1 = nothing to accumulate into (a transposed convolution's own forward: segnb_upconv_fprop)
    int P32;              // plane gather with 32-channel planes: a K chunk = two planes (halves of every LDS row)
    int RPS;              // 2 x 2-window forms: store rows of the previous tile carried per step (1 or 2)
    int NTLR, CoW;        // phase forward (WsCfg::UPF): channel tiles per phase (NTL = 4 * NTLR), weight rows per phase
    // fused BatchNorm-backward reduction in the epilogue of a DATA-GRADIENT launch (segnb_conv_fprop_bnreduce): the output
    // tile is the gradient g of the producing layer's activation; with that layer's pre-BatchNorm output y the store
    // threads accumulate sum dz and sum dz * yhat (dz = g * act'(z)) -- segnb_bn_act_bwd_reduce without its own pass
    const bf16_t* bn_y;   // NULL: off
    unsigned bn_y_bytes;
    int bn_ld;
    const float* bn_coef; // [4][Co]: scale, shift, mean, invstd
    double* bn_sums;      // [REPL][2][Co]
    int bn_act;
    float bn_slope;
    // affine + activation epilogue (segnb_conv_fprop_act): out = act((acc + bias) affine-mapped); ep_act < 0: off.
    // ep_coef NULL: v = acc + bias;  else [4][Co] of segnb_bn_finalize: v = (acc + bias - mean) * scale + shift
    const float* ep_coef;
    int ep_act;
    float ep_slope;
    // split K (conv_fprop_ws_kernel<..., SPLITK>): KS blocks share one (pixel tile, channel tile), each walks NCH (= Ci / 64 / KS)
    // input-channel chunks starting at chunk ks * NCH, publishes its fp32 accumulator tile as a 64 KB slab in ks_slab
    // [tile][KS][4 waves][16 pieces][64 lanes][4] (write-through stores) and draws a ticket from ks_cnt[tile]; the block that draws
    // KS - 1 adds the slabs in slice order and runs the ordinary epilogue (bias, rounding, statistics, stores).  KS = 1: off
    int KS;
    int ntmajor;          // block order: 1 = (channel tile, slice, pixel tile) -- the blocks of an XCD share WEIGHTS (7 x 7 level: the
                          // weights are 6 x the activations); 0 = (pixel tile, channel tile) -- they share the input tile
    float* ks_slab;
    unsigned ks_slab_bytes;
    int* ks_cnt;
    int dbg;              // timing builds (segnb_tune "fprop_dma_dbg"): 1 no weight fetches, 2 no halo fetches, 4 no MFMA +
                          // fragment reads, 8 no stores, 16 no fragment reads, 32 in-kernel stamps, 64 one stamp per tap
};

constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ int xcd_remap_fd(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7, j = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

__device__ __forceinline__ i32x4_t make_rsrc4(const void* base, unsigned bytes) {
    const unsigned long long pa = (unsigned long long)base;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// one LDS-DMA piece: lane l's 16 bytes at (descriptor base + voff + soff) land at LDS byte lds_dst + 16*l.
// hipcc does not count these (no s_waitcnt of its own for them): completion is the counted vmcnt of step_sync.
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, const i32x4_t& rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// LDS-only barrier: __syncthreads() would also wait for the global stores of the epilogue and for every DMA in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define FD_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
// (accumulators in arch VGPRs: "a"-constrained AGPR accumulators measured 1-2 % slower here, same-box A/B)
#define FD_MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// first MFMA of an accumulator: C = 0 (no 16 v_mov per accumulator tile)
#define FD_MFMA0(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
// 16 x 16 x 32 form (4 accumulator registers per 16-channel x 16-pixel tile): at equal cycles per FLOP the chip holds a
// higher clock under this shape than under 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x FLOP/s)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define FD_MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FD_MFMA16_0(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
// counted LDS wait that names the fragment registers it completes
template <int N>
__device__ __forceinline__ void ws_wait1(bf16x8_t& f0) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f0) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ws_wait5(bf16x8_t& f0, bf16x8_t& f1, bf16x8_t& f2, bf16x8_t& f3, bf16x8_t& f4) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "n"(N));
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void step_sync() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// counted LDS wait naming a fragment set of NF registers
template <int N, int NF>
__device__ __forceinline__ void ws_wait(bf16x8_t (&f)[NF]) {
    if constexpr (NF == 6)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N));
    else if constexpr (NF == 5)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]) : "n"(N));
    else if constexpr (NF == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
    else if constexpr (NF == 3)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : "n"(N));
    else {
        static_assert(NF == 2, "fragment set");
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N));
    }
}


}  // namespace
