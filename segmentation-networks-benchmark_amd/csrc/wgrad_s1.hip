// Weight gradient of stride-1 3x3 convolutions (bf16), gfx950 -- the fast path behind segnb_conv_wgrad.
//
//   dW[co][t][ci] += sum_{n,h,w} dy[n,h,w,co] * x[n, h+dh[t], w+dw[t], ci]        (aten::convolution_backward,
//   weight part, for nn.Conv2d(.., 3, padding=1) of lib/models/zf_unet.py:8 and its siblings)
//
// The reduction index is the PIXEL, but both operands live pixel-major (NHWC) in HBM.  Instead of
// transposing on the way into LDS, tiles stay pixel-major in LDS -- [pixel][channel], exactly as loaded,
// 16-byte stores -- and the MFMA fragments are fetched with ds_read_b64_tr_b16, the hardware transposing
// read (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane).  A tap shift (dh, dw) is then
// just a different pixel ROW of the same x tile, so ONE x tile (with halo) and ONE dy tile feed all nine
// taps: x and dy are read once per (co-tile, ci-tile) instead of once per tap.
//
// Block = 256 threads.  Work item ("iteration") = R rows x WT columns of output pixels of one image.
//   x tile  : (R+2) x (WT+2) pixels x BCI channels,   dy tile : R x WT pixels x BCO channels
//   K slab  = 16 consecutive pixels of one row = one v_mfma_f32_32x32x16_bf16 per (32co x 32ci x tap)
//   tile 32x32 : the four waves split the slabs (K) of an iteration, 9 x 16 accumulators each
//   tile 64x64 : wave (i,j) owns sub-tile (32i.., 32j..) for all slabs
// Row strides are = 64 or 192 (mod 256) bytes so the four pixel rows a half-wave reads hit disjoint banks.
// Single LDS buffer + register prefetch of the next iteration; iterations of a block are a contiguous
// range (split over blocks), results merged with fp32 atomics into the packed [Co][9*Ci] workspace.
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_bf16x4_ptr;

// 8 consecutive reduction-index (pixel) values of one channel per lane = one MFMA operand fragment:
// two transposing reads of 4 pixel rows each.  (Keep the bf16-typed builtin + shufflevector: assembling the
// fragment element by element from the v4i16 form is mis-lowered by hipcc 7.2 -- half the elements dropped.)
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p, int row4_bytes) {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + row4_bytes));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

typedef __attribute__((address_space(3))) unsigned char* lds_u8_ptr;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lds_u8_ptr)p; }

// The slab loop is written as volatile asm so that the issue order is exactly the program order: the fragments
// of the NEXT 16-pixel slab are requested (ds_read_b64_tr_b16) between the MFMAs of the current one and waited
// for with explicit s_waitcnt.  (Left to the compiler, every transposing read is sunk next to its use --
// read latency and matrix pipe fully serialised, measured 12 % MFMA utilisation.)
#define SEGNB_TR_READ2(lo, hi, addr, off0, off1)                                           \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" \
                 : "=&v"(lo), "=&v"(hi)                                                    \
                 : "v"(addr), "n"(off0), "n"(off1))
#define SEGNB_MFMA(acc, af, bf) \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(af), "v"(bf))
// specialised blocks run two waves per SIMD = 256 registers per wave: accumulators in ordinary VGPRs there (with "a" the
// allocator's VGPR/AGPR split of the halved budget shuttled them through v_accvgpr moves and scratch)
#define SEGNB_MFMA_V(acc, af, bf) \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(af), "v"(bf))
#define SEGNB_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n))

#ifndef SEGNB_WG_TS
#define SEGNB_WG_TS 1
#endif
#ifndef SEGNB_WG_WS_DB
#define SEGNB_WG_WS_DB 0     // specialised blocks: 1 = two tile buffers (one barrier per iteration; same speed alone, but the
                            // 128 KB block keeps other kernels off the CU: +0.8 % step time beside the main stream)
#endif
#ifndef SEGNB_WG_PACE
#define SEGNB_WG_PACE 0      // s_sleep argument (x 64 clocks) after each paced tile store (0: none)
#endif
#ifndef SEGNB_WG_TALL
#define SEGNB_WG_TALL 1
#endif
#ifndef SEGNB_WG_EXP
#define SEGNB_WG_EXP 0       // timing experiments (wrong results): 1 = no global loads after the first tile, 2 = no MFMA loop,
                              // 4 = no slab stores
#endif

struct WgS1Args {
    const bf16_t* x;
    const bf16_t* dy;
    float* dwp;
    int N, H, W;            // output == input spatial size is NOT assumed: Hi/Wi below
    int Hi, Wi;
    int Ci, Co, ld_x, ld_dy;
    int dhmin, dwmin;
    int dh[9], dw[9];       // tap offsets minus (dhmin, dwmin): 0..2
    int HB, WB, IT;         // row blocks, column segments, total iterations
    int TCI_TILES;          // number of ci tiles
    int its_per_split;
    long long slab_stride;  // floats between the partial slabs of consecutive pixel splits
    int Ktot;               // 9 * Ci
    segnb_wgrad_bnapply bna;    // BNA instantiations: dy recomputed from (g, y) while staging (bna.g != NULL)
    // virtual concat (segnb_conv_wgrad_upcat): input channels [0, Cu) are the nearest-x2 upsample of u [N][Hi/2][Wi/2][ld_u]
    // -- x-tile pixel (hi, wi) of those channels is u pixel (hi >> 1, wi >> 1) -- channels Cu.. come from x (the skip tensor)
    const bf16_t* u;            // NULL: off
    int Cu, ld_u;
    // segnb_wgrad_target of a SINGLE-slab launch (gw != NULL): the block's tile goes straight into the parameter's gradient
    //     gw[co * gw_s_out + (gw_ci_off + ci) * gw_s_in + gw_kpos[t]]  (+)=  acc        co < gw_Co, ci < gw_Ci
    float* gw;
    long long gw_s_out;
    int gw_s_in, gw_ci_off, gw_Ci, gw_Co, gw_acc;
    int gw_kpos[9];
};

__device__ __forceinline__ float wg_round_bf16(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }

// one tap of one slab: wait for B_t of the current fragment set, MFMA, request B_t of the next set
template <int T, int WAITN, bool MORE, int OB, int SXB>
__device__ __forceinline__ void wg_tap(f32x16_t (&acc)[9], const bf16x4_t (&fa)[2], bf16x4_t (&fbc)[9][2],
                                       bf16x4_t (&fbn)[9][2], const unsigned (&vb)[9]) {
    SEGNB_WAIT_LGKM(WAITN);
    const bf16x8_t af = __builtin_shufflevector(fa[0], fa[1], 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8_t bf = __builtin_shufflevector(fbc[T][0], fbc[T][1], 0, 1, 2, 3, 4, 5, 6, 7);
    SEGNB_MFMA(acc[T], af, bf);
    if constexpr (MORE) SEGNB_TR_READ2(fbn[T][0], fbn[T][1], vb[T], OB, OB + 4 * SXB);
}

// slab S0 of SPW: request A of slab S0+1, then the nine taps.  Reads still allowed in flight when B_t of the
// current slab is needed = everything issued after it = 2*(8-t) [rest of this slab] + 2 [next A] + 2*t [next
// B_0..t-1] = 18 -> the 4-bit counter clamps it to 15; on the last slab 2*(8-t).
template <int S0, int SPW, int KSPLIT, int SEGS, int WT, int XC, int SX, int SY, int MID, typename F>
__device__ __forceinline__ void wg_slabs(f32x16_t (&acc)[9], bf16x4_t (&fa)[2][2], bf16x4_t (&fb)[2][9][2],
                                         unsigned va, const unsigned (&vb)[9], F& mid) {
    if constexpr (S0 < SPW) {
        if constexpr (S0 == MID) mid();          // double-buffered tiles: the next iteration's LDS stores go here
        constexpr int cur = S0 & 1, nxt = cur ^ 1;
        constexpr bool more = S0 + 1 < SPW;
        constexpr int sn = (S0 + 1) * KSPLIT;
        constexpr int oa = ((sn / SEGS) * WT + (sn % SEGS) * 16) * SY;
        constexpr int ob = ((sn / SEGS) * XC + (sn % SEGS) * 16) * SX;
        if constexpr (more) SEGNB_TR_READ2(fa[nxt][0], fa[nxt][1], va, oa, oa + 4 * SY);
        wg_tap<0, 15, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<1, more ? 15 : 14, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<2, more ? 15 : 12, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<3, more ? 15 : 10, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<4, more ? 15 : 8, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<5, more ? 15 : 6, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<6, more ? 15 : 4, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<7, more ? 15 : 2, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_tap<8, more ? 15 : 0, more, ob, SX>(acc, fa[cur], fb[cur], fb[nxt], vb);
        wg_slabs<S0 + 1, SPW, KSPLIT, SEGS, WT, XC, SX, SY, MID>(acc, fa, fb, va, vb, mid);
    }
}

// ---- tap-split wave tiles (64x64 block tiles) ------------------------------------------------------------------
// With wave (i, j) owning sub-tile (32 co, 32 ci) for all nine taps, a 16-pixel slab costs 1 A + 9 B fragments per
// 9 MFMAs: 4 waves x 20 ds_read_b64_tr_b16 x 512 B = 40 KB of LDS reads per 288 matrix-pipe cycles -- the loop ran
// at the LDS's ~100 B/clk, not at the MFMA rate (measured 30 ns per MFMA against 21 ns in the forward kernel).
// Here wave (G, j) owns ALL 64 co x 32 ci for half of the taps instead: nine (co half c, tap) units
//     G = 0: (0,t0) (1,t0) (0,t1) (1,t1) (0,t2) (1,t2) (0,t3) (1,t3) (0,t4)
//     G = 1: (1,t4) (0,t5) (1,t5) (0,t6) (1,t6) (0,t7) (1,t7) (0,t8) (1,t8)
// = 2 A + 5 B fragments per 9 MFMAs (28 KB per slab round), still 9 x 16 accumulators and 9 MFMAs per wave.
// unit u -> co half (u+G)&1, local tap b = (u+G)>>1 (tap 4G + b).  Fragments are requested in first-use order
// F0..F6 = A_G, B0, A_(1-G), B1, B2, B3, B4: fragment Fj of the NEXT slab right after MFMA j of the current one.
constexpr int ts_c(int G, int u) { return (u + G) & 1; }
constexpr int ts_b(int G, int u) { return (u + G) >> 1; }
constexpr int ts_fa(int G, int c) { return c == G ? 0 : 2; }
constexpr int ts_fb(int b) { return b == 0 ? 1 : b + 2; }
constexpr int ts_k(int G, int u) {
    const int ka = ts_fa(G, ts_c(G, u)), kb = ts_fb(ts_b(G, u));
    return ka > kb ? ka : kb;
}
// reads still allowed in flight when unit u issues: the fragments requested after the last one it needs (rest of the
// previous slab's requests) plus the next slab's requests made so far in this slab (LDS returns in order)
constexpr int ts_wait(int G, int u, bool more) {
    const int w = (6 - ts_k(G, u)) * 2 + (more ? 2 * (u < 7 ? u : 7) : 0);
    return w > 15 ? 15 : w;
}

template <int G, int J, int OA, int OB, int SX, int SY>
__device__ __forceinline__ void ts_request(bf16x4_t (&fan)[2][2], bf16x4_t (&fbn)[5][2], unsigned va,
                                           const unsigned (&vb)[5]) {
    if constexpr (J == 0) {
        SEGNB_TR_READ2(fan[G][0], fan[G][1], va, OA + 64 * G, OA + 64 * G + 4 * SY);
    } else if constexpr (J == 2) {
        SEGNB_TR_READ2(fan[1 - G][0], fan[1 - G][1], va, OA + 64 * (1 - G), OA + 64 * (1 - G) + 4 * SY);
    } else {
        constexpr int B = J == 1 ? 0 : J - 2;
        SEGNB_TR_READ2(fbn[B][0], fbn[B][1], vb[B], OB, OB + 4 * SX);
    }
}

template <int G, int U, bool MORE, int OA, int OB, int SX, int SY>
__device__ __forceinline__ void ts_units(f32x16_t (&acc)[9], bf16x4_t (&fac)[2][2], bf16x4_t (&fbc)[5][2],
                                         bf16x4_t (&fan)[2][2], bf16x4_t (&fbn)[5][2], unsigned va,
                                         const unsigned (&vb)[5]) {
    if constexpr (U < 9) {
        constexpr int C = ts_c(G, U), B = ts_b(G, U);
        SEGNB_WAIT_LGKM(ts_wait(G, U, MORE));
        const bf16x8_t af = __builtin_shufflevector(fac[C][0], fac[C][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8_t bf = __builtin_shufflevector(fbc[B][0], fbc[B][1], 0, 1, 2, 3, 4, 5, 6, 7);
        SEGNB_MFMA_V(acc[U], af, bf);
        if constexpr (MORE && U < 7) ts_request<G, U, OA, OB, SX, SY>(fan, fbn, va, vb);
        ts_units<G, U + 1, MORE, OA, OB, SX, SY>(acc, fac, fbc, fan, fbn, va, vb);
    }
}

template <int G, int J, int SX, int SY>
__device__ __forceinline__ void ts_first(bf16x4_t (&fa0)[2][2], bf16x4_t (&fb0)[5][2], unsigned va,
                                         const unsigned (&vb)[5]) {
    if constexpr (J < 7) {
        ts_request<G, J, 0, 0, SX, SY>(fa0, fb0, va, vb);
        ts_first<G, J + 1, SX, SY>(fa0, fb0, va, vb);
    }
}

template <int G, int S0, int SPW, int SEGS, int WT, int XC, int SX, int SY, int MID, typename F>
__device__ __forceinline__ void ts_slabs(f32x16_t (&acc)[9], bf16x4_t (&fa)[2][2][2], bf16x4_t (&fb)[2][5][2],
                                         unsigned va, const unsigned (&vb)[5], F& mid) {
    if constexpr (S0 < SPW) {
        if constexpr (S0 == MID) mid();
        constexpr int cur = S0 & 1, nxt = cur ^ 1;
        constexpr bool more = S0 + 1 < SPW;
        constexpr int sn = S0 + 1;
        constexpr int oa = ((sn / SEGS) * WT + (sn % SEGS) * 16) * SY;
        constexpr int ob = ((sn / SEGS) * XC + (sn % SEGS) * 16) * SX;
        ts_units<G, 0, more, oa, ob, SX, SY>(acc, fa[cur], fb[cur], fa[nxt], fb[nxt], va, vb);
        ts_slabs<G, S0 + 1, SPW, SEGS, WT, XC, SX, SY, MID>(acc, fa, fb, va, vb, mid);
    }
}

// one iteration's slab loop of a tap-split wave (prologue requests + SPW slabs)
template <int G, int SPW, int SEGS, int WT, int XC, int SX, int SY, int MID, typename F>
__device__ __forceinline__ void ts_iteration(f32x16_t (&acc)[9], unsigned va, const unsigned (&vb)[5], F& mid) {
    bf16x4_t fa[2][2][2], fb[2][5][2];
    ts_first<G, 0, SX, SY>(fa[0], fb[0], va, vb);
    ts_slabs<G, 0, SPW, SEGS, WT, XC, SX, SY, MID>(acc, fa, fb, va, vb, mid);
}

template <int T, int SXB>
__device__ __forceinline__ void wg_first(bf16x4_t (&fb0)[9][2], const unsigned (&vb)[9]) {
    if constexpr (T < 9) {
        SEGNB_TR_READ2(fb0[T][0], fb0[T][1], vb[T], 0, 4 * SXB);
        wg_first<T + 1, SXB>(fb0, vb);
    }
}

constexpr bool wg_double_buffered(int tile_bytes) { return 2 * tile_bytes <= 160 * 1024; }

constexpr int lds_stride(int channels) {
    // bytes; multiple of 16, >= 2*channels, == 64 or 192 (mod 256)
    int s = channels * 2;
    while (!((s % 256) == 64 || (s % 256) == 192)) s += 16;
    return s;
}

// Tile shapes.  FLAT = false: R rows x WT columns of one image (+ halo).  FLAT = true (small square images,
// WT = H = W of 7 or 14): R whole images per iteration in the PADDED-ROW FLATTENING -- output pixel (h, w) sits at
// position p = h*(W+2) + w of a dy tile with two zero columns per row, input pixel (h + dh, w + dw) at position
// p + dh*(W+2) + dw of the zero-padded x image, so a tap is still one constant LDS offset, a K slab is any 16
// consecutive positions, and a 7-pixel row no longer wastes 9 of the 16 pixels of a slab.
template <int R, int WT, bool FLAT>
struct WgTile {
    static constexpr int W2 = WT + 2;
    static constexpr int PD = FLAT ? (WT * W2 + 15) / 16 * 16 : R * WT;        // dy positions per image / tile
    static constexpr int XT = FLAT ? PD + 2 * W2 + 2 : (R + 2) * (WT + 2);      // x positions per image / tile
    static constexpr int XROWS = FLAT ? R * XT : XT, YROWS = FLAT ? R * PD : PD;
    static constexpr int NSLAB = YROWS / 16;
    static constexpr int SEGS = FLAT ? PD / 16 : WT / 16;                       // slabs per dy "row"
    static constexpr int YSTEP = FLAT ? PD : WT, XSTEP = FLAT ? XT : WT + 2;    // positions between those rows
};

// 64x64 tiles whose two buffers fit run WAVE-SPECIALISED: 4 matrix waves (tap-split units) + 4 fetch waves that stage the
// next tile (global -> registers -> other LDS buffer).  With one wave per SIMD doing both, every staging instruction
// (~300 VALU/VMEM/DS per iteration, mostly address arithmetic) sat between two MFMAs of an in-order wave: measured
// 2.1 us per iteration = 1.5 us of MFMAs + 0.6 us of staging, whatever the prefetch distance or LDS read count.
template <int BCO, int BCI, int R, int WT, bool FLAT>
constexpr bool wg_specialised() {
    using TL = WgTile<R, WT, FLAT>;
    return SEGNB_WG_TS && BCO == 64 && BCI == 64;
}
// ... and with TWO tile buffers where they fit (SEGNB_WG_WS_DB): the fetch waves keep two tiles in flight in registers and store
// the next one into the other buffer under the matrix waves' slab loop
template <int BCO, int BCI, int R, int WT, bool FLAT>
constexpr bool wg_ws_db() {
    using TL = WgTile<R, WT, FLAT>;
    return SEGNB_WG_WS_DB && wg_specialised<BCO, BCI, R, WT, FLAT>() &&
           wg_double_buffered(TL::XROWS * lds_stride(BCI) + TL::YROWS * lds_stride(BCO));
}

template <int BCO, int BCI, int R, int WT, bool FLAT = false, bool BNA = false>
__global__ __launch_bounds__((wg_specialised<BCO, BCI, R, WT, FLAT>() ? 512 : 256), 1) void conv_wgrad_s1x9_kernel(
    const WgS1Args a) {
    static_assert(!BNA || (!FLAT && BCO == 32), "the recomputed dy operand is built for the thin 32 x 32 tiles");
    using TL = WgTile<R, WT, FLAT>;
    constexpr int TCO = BCO / 32, TCI = BCI / 32;
    constexpr int NSUB = TCO * TCI;
    static_assert(NSUB == 1 || NSUB == 4, "tile is 32x32 or 64x64");
    constexpr int KSPLIT = 4 / NSUB;
    constexpr int XC = FLAT ? TL::W2 : WT + 2;
    constexpr int SX = lds_stride(BCI), SY = lds_stride(BCO);
    constexpr int NSLAB = TL::NSLAB;
    constexpr int SEGS = TL::SEGS;
    constexpr int XCH = TL::XROWS * (BCI / 8);      // 16-byte chunks of the x tile
    constexpr int YCH = TL::YROWS * (BCO / 8);
    constexpr int XPT = (XCH + 255) / 256, YPT = (YCH + 255) / 256;

    // two tile buffers when they fit (every non-FLAT shape): the LDS stores of iteration it+1 are issued in the middle
    // of iteration it's slab loop, under its MFMAs, and one barrier per iteration separates the buffers' roles
    // (single buffer: MFMA drain -> barrier -> stores -> barrier -> first fragment reads, all exposed, every iteration)
    constexpr int BUF = TL::XROWS * SX + TL::YROWS * SY;
    constexpr bool DB = wg_ws_db<BCO, BCI, R, WT, FLAT>();

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sY = smem + TL::XROWS * SX;

    constexpr bool TS = wg_specialised<BCO, BCI, R, WT, FLAT>();
    constexpr int NT = TS ? 512 : 256;
    const bool fetcher = TS && threadIdx.x >= 256;                 // waves 4..7 of a specialised block
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    // blocks b, b+8, ... share an XCD (round-robin dispatch) and so an L2: give each XCD a contiguous range of
    // logical blocks, so that all (co, ci) tiles of one pixel range fetch their x / dy tiles through ONE L2
    // (without it every tile pair re-read both tensors from HBM: 249 MB per launch measured)
    const int G = gridDim.x, xq = G >> 3, xr = G & 7, xx = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int L = (xx < xr ? xx * (xq + 1) : xr * (xq + 1) + (xx - xr) * xq) + xj;
    const int ntile = L % a.TCI_TILES;              // consecutive logical blocks share dy, differ in ci tile
    const int rest = L / a.TCI_TILES;
    const int ncot = (a.Co + BCO - 1) / BCO;
    const int mtile = rest % ncot;
    const int split = rest / ncot;
    const int co0 = mtile * BCO, ci0 = ntile * BCI;

    const int it_begin = split * a.its_per_split;
    int it_end = it_begin + a.its_per_split;
    if (it_end > a.IT) it_end = a.IT;
    // every (tile, split) block OWNS its [BCO][9][BCI] piece of partial slab `split` and writes it with plain
    // stores: no atomics (global float atomics run at ~1.3 TB/s chip-wide: 512 blocks x 147 KB = 58 us per layer,
    // measured as the floor of the atomic version), no zeroing; segnb_unpack_wgrad sums the slabs.
    float* __restrict__ slab = a.dwp + (long long)split * a.slab_stride;
    if (it_begin >= it_end) {                       // more slabs than pixel ranges (tiny inputs): zero piece
        for (int i = threadIdx.x; i < BCO * 9 * BCI; i += NT) {
            const int ci = ci0 + i % BCI, t = (i / BCI) % 9, co = co0 + i / (9 * BCI);
            if (co < a.Co && ci < a.Ci) slab[(long long)co * a.Ktot + t * a.Ci + ci] = 0.f;
        }
        return;
    }

    // BNA: constants of this thread's dy channel chunk (chunk index = tid % (BCO / 8) for every staged vector: 256 % (BCO / 8) == 0)
    float k_mu[8], k_sc[8], k_sh[8], k_is[8], k_a[8], k_c1[8], k_c2[8], k_neg = 0.f;
    if constexpr (BNA) {
        static_assert(256 % (BCO / 8) == 0, "fixed chunk per thread");
        const int chk = co0 + (tid % (BCO / 8)) * 8;
        k_neg = a.bna.act == SEGNB_ACT_RELU ? 0.f : (a.bna.act == SEGNB_ACT_LEAKY ? a.bna.slope : 1.f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cch = chk + e < a.bna.Cp ? chk + e : 0;
            k_sc[e] = a.bna.coef[cch];
            k_sh[e] = a.bna.coef[a.bna.Cp + cch];
            k_mu[e] = a.bna.coef[2 * a.bna.Cp + cch];
            k_is[e] = a.bna.coef[3 * a.bna.Cp + cch];
            k_a[e] = a.bna.bcoef[cch];
            k_c1[e] = a.bna.bcoef[a.bna.Cp + cch];
            k_c2[e] = a.bna.bcoef[2 * a.bna.Cp + cch];
        }
    }
    uint4 rx[XPT], ry[YPT];
    uint4 rx2[DB ? XPT : 1], ry2[DB ? YPT : 1];        // second tile in flight (double-buffered fetch waves)
    auto gload_to = [&](int it, auto& rx, auto& ry) {
        if constexpr (FLAT) {
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                const int c = tid + u * 256;
                const int row = c / (BCI / 8), cc = c - row * (BCI / 8);
                const int g = row / TL::XT, q = row - g * TL::XT;
                const int n = it * R + g;
                const int hi = q / TL::W2 + a.dhmin, wi = q % TL::W2 + a.dwmin;
                const int ch = ci0 + cc * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c < XCH && n < a.N && ch < a.Ci && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi) {
                    if (a.u != nullptr && ch < a.Cu)
                        v = *reinterpret_cast<const uint4*>(a.u + ((long long)(n * (a.Hi >> 1) + (hi >> 1)) * (a.Wi >> 1) + (wi >> 1)) * a.ld_u + ch);
                    else
                        v = *reinterpret_cast<const uint4*>(a.x + ((long long)(n * a.Hi + hi) * a.Wi + wi) * a.ld_x + (ch - (a.u != nullptr ? a.Cu : 0)));
                }
                rx[u] = v;
            }
#pragma unroll
            for (int u = 0; u < YPT; ++u) {
                const int c = tid + u * 256;
                const int row = c / (BCO / 8), cc = c - row * (BCO / 8);
                const int g = row / TL::PD, pp = row - g * TL::PD;
                const int n = it * R + g;
                const int ho = pp / TL::W2, wo = pp % TL::W2;
                const int ch = co0 + cc * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c < YCH && n < a.N && ch < a.Co && ho < a.H && wo < a.W)
                    v = *reinterpret_cast<const uint4*>(a.dy + ((long long)(n * a.H + ho) * a.W + wo) * a.ld_dy + ch);
                ry[u] = v;
            }
            return;
        }
        const int n = it / (a.HB * a.WB);
        const int rem = it - n * (a.HB * a.WB);
        const int hb = rem / a.WB, wb = rem - hb * a.WB;
        const int h0 = hb * R, w0 = wb * WT;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int c = tid + u * 256;                       // chunk -> (pixel of the x tile, 8-channel chunk)
            const int pix = c / (BCI / 8), cc = c - pix * (BCI / 8);
            const int xr = pix / XC, xc = pix - xr * XC;
            const int hi = h0 + a.dhmin + xr, wi = w0 + a.dwmin + xc;
            const int ch = ci0 + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < XCH && ch < a.Ci && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi) {
                if (a.u != nullptr && ch < a.Cu)
                    v = *reinterpret_cast<const uint4*>(a.u + ((long long)(n * (a.Hi >> 1) + (hi >> 1)) * (a.Wi >> 1) + (wi >> 1)) * a.ld_u + ch);
                else
                    v = *reinterpret_cast<const uint4*>(a.x + ((long long)(n * a.Hi + hi) * a.Wi + wi) * a.ld_x + (ch - (a.u != nullptr ? a.Cu : 0)));
            }
            rx[u] = v;
        }
#pragma unroll
        for (int u = 0; u < YPT; ++u) {
            const int c = tid + u * 256;
            const int pix = c / (BCO / 8), cc = c - pix * (BCO / 8);
            const int yr = pix / WT, yc = pix - yr * WT;
            const int ho = h0 + yr, wo = w0 + yc;
            const int ch = co0 + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < YCH && ch < a.Co && ho < a.H && wo < a.W) {
                if constexpr (BNA) {
                    // dy = BatchNorm-backward apply of (g, y), bit for bit what bn_bwd_apply_kernel (direct form) would have
                    // stored: round(g act'(z)), then round(a (. - c1 - yhat c2)); the per-channel constants of the thread's
                    // (fixed) channel chunk were loaded once, before the pixel loop
                    const long long pix = (long long)(n * a.H + ho) * a.W + wo;
                    float gv[8], yv[8], o[8];
                    load8(reinterpret_cast<const bf16_t*>(a.bna.g) + pix * a.bna.ld_g + ch, gv);
                    load8(reinterpret_cast<const bf16_t*>(a.bna.y) + pix * a.bna.ld_y + ch, yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = (yv[e] - k_mu[e]) * k_sc[e] + k_sh[e] + 0.f;
                        const float d = wg_round_bf16(gv[e] * 1.f * (z > 0.f ? 1.f : k_neg));
                        const float yh = (yv[e] - k_mu[e]) * k_is[e];
                        o[e] = k_a[e] * (d - k_c1[e] - yh * k_c2[e]);
                    }
                    bf16_t packed[8];
                    store8(packed, o);
                    v = *reinterpret_cast<const uint4*>(packed);
                } else {
                    v = *reinterpret_cast<const uint4*>(a.dy + ((long long)(n * a.H + ho) * a.W + wo) * a.ld_dy + ch);
                }
            }
            ry[u] = v;
        }
    };
    auto gload = [&](int it) { gload_to(it, rx, ry); };
    // paced: (double-buffered fetch waves) a pause after every store, so that the tile trickles into the other buffer under the
    // matrix waves' slab loop instead of arriving as one burst their fragment reads queue behind
    auto lstore_from = [&](int bo, auto paced, const auto& rx, const auto& ry) {
        constexpr bool PACED = decltype(paced)::value;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int c = tid + u * 256;
            if (c < XCH) {
                const int pix = c / (BCI / 8), cc = c - pix * (BCI / 8);
                *reinterpret_cast<uint4*>(sX + bo + pix * SX + cc * 16) = rx[u];
            }
            if constexpr (PACED && SEGNB_WG_PACE > 0) __builtin_amdgcn_s_sleep(SEGNB_WG_PACE);
        }
#pragma unroll
        for (int u = 0; u < YPT; ++u) {
            const int c = tid + u * 256;
            if (c < YCH) {
                const int pix = c / (BCO / 8), cc = c - pix * (BCO / 8);
                *reinterpret_cast<uint4*>(sY + bo + pix * SY + cc * 16) = ry[u];
            }
            if constexpr (PACED && SEGNB_WG_PACE > 0) __builtin_amdgcn_s_sleep(SEGNB_WG_PACE);
        }
    };
    auto lstore = [&](int bo, auto paced) { lstore_from(bo, paced, rx, ry); };
    constexpr std::false_type burst{};
    constexpr std::true_type paced{};

    // fragment addressing: 16-lane group g reads 4 pixel rows x 16 channels; lane 4q+p supplies the address of
    // pixel row q, channels 4p..4p+3 and receives channel (l&15) of the 4 rows (probe: tools/probe_tr.hip)
    // TS (64x64 tiles): wave -> (tap group tg, ci half); otherwise wave -> (32x32 sub-tile, K part)
    const int tg = TS ? __builtin_amdgcn_readfirstlane(wave >> 1) : 0;
    const int sub = TS ? 0 : wave / KSPLIT, kpart = TS ? 0 : wave - sub * KSPLIT;
    const int sco = TS ? 0 : sub / TCI, sci = TS ? (wave & 1) : sub - sco * TCI;
    const int q = (lane & 15) >> 2, p = lane & 3, h = lane >> 5, cbase = 16 * ((lane >> 4) & 1);
    const int a_off = (8 * h + q) * SY + (sco * 32 + cbase + 4 * p) * 2;
    const int b_off = (8 * h + q) * SX + (sci * 32 + cbase + 4 * p) * 2;
    int tap_off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap_off[t] = (a.dh[t] * XC + a.dw[t]) * SX;

    if constexpr (TS) {
        if (fetcher) {
            // fetch waves: tile it+1 -> registers -> the buffer the matrix waves are not reading; same barrier sequence
            gload(it_begin);
            lstore(0, burst);
            if constexpr (DB) {
                // tiles it+1 (set A) and it+2 (set B) in flight; each iteration stores the older set into the buffer the matrix
                // waves are not reading and re-issues that set two tiles ahead: a tile has two slab loops to land
                if (it_begin + 1 < it_end) gload_to(it_begin + 1, rx, ry);
                if (it_begin + 2 < it_end) gload_to(it_begin + 2, rx2, ry2);
                __syncthreads();
                int cur = 0;
                for (int it = it_begin; it < it_end; it += 2) {
                    if (it + 1 < it_end) lstore_from((cur ^ 1) * BUF, paced, rx, ry);
                    if (it + 3 < it_end) gload_to(it + 3, rx, ry);
                    __syncthreads();
                    if (it + 1 >= it_end) break;
                    if (it + 2 < it_end) lstore_from(cur * BUF, paced, rx2, ry2);
                    if (it + 4 < it_end) gload_to(it + 4, rx2, ry2);
                    __syncthreads();
                }
                return;
            }
            __syncthreads();
            for (int it = it_begin; it < it_end; ++it) {
#if !(SEGNB_WG_EXP & 1)
                if (it + 1 < it_end) gload(it + 1);
#endif
                __syncthreads();                    // the matrix waves are done with this tile
                if (it + 1 < it_end) {
                    lstore(0, burst);
                    __syncthreads();
                }
            }
            return;
        }
    }

    // accumulators are defined and updated only by "a"-constrained asm, so they live in AGPRs across the whole
    // pixel loop (a VALU zero-init makes the allocator keep them in VGPRs and copy 144 registers into and out of
    // AGPRs around every iteration)
    f32x16_t acc[9];
    if constexpr (TS) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    } else {
        const bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t)      // s_nop: the hazard recognizer does not see into asm (VALU write of z -> MFMA read)
            asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[t]) : "v"(z));
    }

    if constexpr (!TS) {
        gload(it_begin);
        lstore(0, burst);
    }
    __syncthreads();
    constexpr int SPW = NSLAB / KSPLIT;        // slabs per wave and iteration
    // slab s0 of this wave: s = s0*KSPLIT + kpart -> row s / SEGS, column segment (s % SEGS)*16.  KSPLIT is 1 or
    // a multiple of SEGS, so the kpart part of the offset is the same for every s0: folded into va / vb.
    static_assert(KSPLIT == 1 || KSPLIT % SEGS == 0, "slab -> (row, segment) split");
    const int krow = kpart / SEGS, kseg = kpart - krow * SEGS;
    const unsigned va = lds_addr(sY + a_off) + (unsigned)((krow * TL::YSTEP + kseg * 16) * SY);
    unsigned vb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) vb[t] = lds_addr(sX + b_off + tap_off[t]) + (unsigned)((krow * TL::XSTEP + kseg * 16) * SX);
    // the whole pixel loop is instantiated per tap group (the MFMA / request order differs) and branched ONCE, so the
    // accumulators stay in their AGPRs across it
    auto pixel_loop = [&](auto Gc) {
        constexpr int G = decltype(Gc)::value;
        if constexpr (DB) {
            int cur = 0;
            for (int it = it_begin; it < it_end; ++it) {
                const bool more = it + 1 < it_end;
                if constexpr (!TS) {
                    if (more) gload(it + 1);
                }
                const unsigned bo = (unsigned)(cur * BUF);
                const unsigned vac = va + bo;
                auto mid = [&]() {
                    if constexpr (!TS) {
                        if (more) lstore((cur ^ 1) * BUF, burst);
                    }
                };
#if SEGNB_WG_EXP & 2
                mid();
                if (false) {
#else
                if constexpr (TS) {
#endif
                    unsigned vtc[5];
#pragma unroll
                    for (int b = 0; b < 5; ++b) vtc[b] = vb[4 * G + b] + bo;
                    ts_iteration<G, SPW, SEGS, TL::YSTEP, TL::XSTEP, SX, SY, (SPW + 1) / 2>(acc, vac, vtc, mid);
                } else if constexpr (!(SEGNB_WG_EXP & 2)) {
                    unsigned vbc[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) vbc[t] = vb[t] + bo;
                    bf16x4_t fa[2][2], fb[2][9][2];
                    SEGNB_TR_READ2(fa[0][0], fa[0][1], vac, 0, 4 * SY);          // prologue: slab 0 -> fragment set 0
                    wg_first<0, SX>(fb[0], vbc);
                    wg_slabs<0, SPW, KSPLIT, SEGS, TL::YSTEP, TL::XSTEP, SX, SY, (SPW + 1) / 2>(acc, fa, fb, vac, vbc,
                                                                                                mid);
                }
                __syncthreads();                // this buffer read by everyone, the other one written by everyone
                cur ^= 1;
            }
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::);      // last MFMA results land before anyone reads the accumulators
        } else {
            auto mid = [] {};
            for (int it = it_begin; it < it_end; ++it) {
                if constexpr (!TS) {
                    if (it + 1 < it_end) gload(it + 1);
                }
                if constexpr (TS) {
#if !(SEGNB_WG_EXP & 2)
                    unsigned vtc[5];
#pragma unroll
                    for (int b = 0; b < 5; ++b) vtc[b] = vb[4 * G + b];
                    ts_iteration<G, SPW, SEGS, TL::YSTEP, TL::XSTEP, SX, SY, -1>(acc, va, vtc, mid);
#endif
                } else {
                    bf16x4_t fa[2][2], fb[2][9][2];
                    SEGNB_TR_READ2(fa[0][0], fa[0][1], va, 0, 4 * SY);          // prologue: slab 0 -> fragment set 0
                    wg_first<0, SX>(fb[0], vb);
                    wg_slabs<0, SPW, KSPLIT, SEGS, TL::YSTEP, TL::XSTEP, SX, SY, -1>(acc, fa, fb, va, vb, mid);
                }
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::);      // last MFMA results land before anyone reads the accumulators
                __syncthreads();                    // everyone done reading this iteration's tiles
                if (it + 1 < it_end) {
                    if constexpr (!TS) lstore(0, burst);
                    __syncthreads();
                }
            }
        }
    };
    if (TS && tg != 0)
        pixel_loop(std::integral_constant<int, 1>{});
    else
        pixel_loop(std::integral_constant<int, 0>{});

    // D[i = co][j = ci]: lane holds column ci = lane&31, rows co = (e&3) + 8*(e>>2) + 4*h
    if (KSPLIT > 1) {
        // the waves hold partial sums of the SAME 32x32x9 tile: merge them in LDS (tiles are dead by now: the
        // loop ended on a barrier)
        float* sAcc = reinterpret_cast<float*>(smem);          // [9][16][64] floats = 36 KB
        for (int w = 1; w < KSPLIT; ++w) {
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* dst = sAcc + (t * 16 + e) * 64 + lane;
                        *dst = (w == 1) ? acc[t][e] : (*dst + acc[t][e]);
                    }
            }
            __syncthreads();
        }
        if (wave != 0) return;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] += sAcc[(t * 16 + e) * 64 + lane];
    }
    const int ci = ci0 + sci * 32 + (lane & 31);
#if SEGNB_WG_EXP & 4
    {
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[u][e];
        if (sum == 12345.f) slab[0] = sum;
        return;
    }
#endif
    if (a.gw != nullptr) {
        // one slab (no pixel split): this block holds the FINAL sums of its tile -- they leave in the parameter's own layout
        // [Co][Ci][3][3], no workspace, no unpack pass.
        if constexpr (TS) {
            // 64 x 64 tiles: four rounds of 16 output channels through LDS (the tiles are dead: the loop ended on a barrier; the
            // fetch waves have left).  A lane's accumulators of a round -- input channel `ci`, 8 output channels, the taps its wave
            // owns -- go to stg[co][ci][tap] (lane stride 9 floats: conflict-free), then the 256 threads move whole rows: 64 input
            // channels x 9 taps of one output channel are 2304 contiguous bytes of the parameter.
            float* const stg = reinterpret_cast<float*>(smem);
            int civ = a.gw_Ci - ci0;
            civ = civ > BCI ? BCI : civ;
            const long long col0 = (long long)(a.gw_ci_off + ci0) * 9;
            const bool vec = a.gw_s_in == 9 && (a.gw_s_out & 3) == 0 && (col0 & 3) == 0 && civ > 0 && ((civ * 9) & 3) == 0 &&
                             (reinterpret_cast<uintptr_t>(a.gw) & 15) == 0;
            const int nrow4 = vec ? civ * 9 / 4 : 0;
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                constexpr int dummy = 0;
                (void)dummy;
                const int c = rd >> 1, r = rd & 1;
                if (rd > 0) __syncthreads();                     // the previous round has been copied out
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    if (((u + tg) & 1) == c) {                   // (wave-uniform)
                        const int kp = a.gw_kpos[4 * tg + ((u + tg) >> 1)];
#pragma unroll
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int e = 8 * r + e8;
                            const int row = (e & 3) + 8 * ((e >> 2) & 1) + 4 * h;
                            stg[(row * 64 + sci * 32 + (lane & 31)) * 9 + kp] = acc[u][e];
                        }
                    }
                }
                __syncthreads();
                const int cobase = co0 + 16 * rd;
                if (vec) {
#pragma unroll 1
                    for (int kb = 0; kb < 9; kb += 3) {          // three rows of float4 in flight (registers: the accumulators are live)
                        float4 o[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int i = tid + 256 * (kb + k), row = i / 144, q = i - row * 144;
                            o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (q < nrow4 && cobase + row < a.gw_Co && a.gw_acc)
                                o[k] = *(reinterpret_cast<const float4*>(a.gw + (long long)(cobase + row) * a.gw_s_out + col0) + q);
                        }
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int i = tid + 256 * (kb + k), row = i / 144, q = i - row * 144;
                            if (q < nrow4 && cobase + row < a.gw_Co) {
                                float4 w = *reinterpret_cast<const float4*>(stg + row * 576 + 4 * q);
                                w.x += o[k].x; w.y += o[k].y; w.z += o[k].z; w.w += o[k].w;
                                *(reinterpret_cast<float4*>(a.gw + (long long)(cobase + row) * a.gw_s_out + col0) + q) = w;
                            }
                        }
                    }
                } else {
                    for (int i = tid; i < 16 * 576; i += 256) {
                        const int row = i / 576, q = i - row * 576;
                        const int cil = q / 9, kp = q - cil * 9;
                        if (cil < civ && cobase + row < a.gw_Co) {
                            float* const d = a.gw + (long long)(cobase + row) * a.gw_s_out + (long long)(a.gw_ci_off + ci0 + cil) * a.gw_s_in + kp;
                            *d = a.gw_acc ? *d + stg[i] : stg[i];
                        }
                    }
                }
            }
            return;
        }
        // (thin 32 x 32 tiles reach here only on inputs too small for a pixel split: element stores)
        float* const gcol = a.gw + (long long)(a.gw_ci_off + ci) * a.gw_s_in;
        const bool civ = ci < a.gw_Ci;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int kp = a.gw_kpos[u];
            const int cob = co0 + sco * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = cob + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < a.gw_Co && civ) {
                    float* const dst = gcol + (long long)co * a.gw_s_out + kp;
                    *dst = a.gw_acc ? *dst + acc[u][e] : acc[u][e];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        // accumulator u: tap u of sub-tile sco, or (tap-split) unit u = (co half (u+tg)&1, tap 4*tg + ((u+tg)>>1))
        const int t = TS ? 4 * tg + ((u + tg) >> 1) : u;
        const int cob = co0 + (TS ? ((u + tg) & 1) : sco) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = cob + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (co < a.Co && ci < a.Ci) slab[(long long)co * a.Ktot + t * a.Ci + ci] = acc[u][e];
        }
    }
}

// ---- slabs -> the parameter's gradient (segnb_wgrad_target): sum of the nslab partial slabs [Cop][nt][Cip] in a fixed
// order, transposed to the parameter's [Co][Ci][KH][KW] and added to (or stored into) the flat gradient buffer, ONE pass.
// A thread owns one (co, ci) pair: for every tap its lanes read 64 consecutive input channels of a slab row (256 B) and
// write the pair's taps as neighbours.  GROUPS: slab groups per block that sum concurrently (many slabs, few pairs: the
// thin layers' 256 slabs x 1024 pairs), each with four independent partial sums; combined in group order.
struct ParamReduceArgs {
    float* dwp;
    long long total;        // Cop * nt * Cip: floats per slab
    int nslab, nt, Cip, Cop;
    float* gw;
    long long s_out;
    int s_in, ci_off, Ci, Co, acc, rezero;
    int kpos[SEGNB_MAX_TAPS];
};

// NT9: 3 x 3 windows (nt == 9 && s_in == 9) -- the loads of ALL nine taps of a slab batch are issued together (36 in flight per
// thread: a thread's time is nslab / (4 GROUPS) round trips, not 9 x that), the block's output is staged as [pair][position] and
// leaves as consecutive floats.  !NT9 (any window): tap by tap, element stores.
template <int GROUPS, bool NT9>
__global__ __launch_bounds__(GROUPS == 1 ? 256 : 64 * GROUPS) void slab_reduce_param_kernel(const ParamReduceArgs a) {
    constexpr int PPB = GROUPS == 1 ? 256 : 64;                 // (co, ci) pairs per block
    constexpr int NTH = GROUPS == 1 ? 256 : 64 * GROUPS;
    __shared__ float part[GROUPS == 1 ? 1 : GROUPS][NT9 ? 9 : 1][64];
    __shared__ float stg[NT9 ? PPB * 9 : 1];                     // [pair][kernel position]
    const int lane = GROUPS == 1 ? threadIdx.x : (threadIdx.x & 63), grp = GROUPS == 1 ? 0 : (threadIdx.x >> 6);
    const long long pair0 = (long long)blockIdx.x * PPB, pair = pair0 + lane;
    const long long npair = (long long)a.Cop * a.Cip;
    const bool in = pair < npair;
    const int co = in ? (int)(pair / a.Cip) : 0, ci = in ? (int)(pair % a.Cip) : 0;
    const float* const src = a.dwp + (long long)co * a.nt * a.Cip + ci;      // tap t of slab s: src[s * total + t * Cip]
    if constexpr (NT9) {
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = 0.f;
        if (in) {
            if constexpr (GROUPS == 1) {
                // few slabs: in slab order, four slabs x nine taps in flight
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = src[t * a.Cip];
                int sl = 1;
                for (; sl + 3 < a.nslab; sl += 4) {
                    float b[4][9];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int t = 0; t < 9; ++t) b[k][t] = src[(long long)(sl + k) * a.total + t * a.Cip];
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] = (((v[t] + b[0][t]) + b[1][t]) + b[2][t]) + b[3][t];
                }
                for (; sl < a.nslab; ++sl) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] += src[(long long)sl * a.total + t * a.Cip];
                }
            } else {
                constexpr int KF = GROUPS >= 16 ? 2 : 4;           // slab loads in flight per tap (1024-thread blocks: 128 registers)
                float w[KF][9];
#pragma unroll
                for (int k = 0; k < KF; ++k)
#pragma unroll
                    for (int t = 0; t < 9; ++t) w[k][t] = 0.f;
                int sl = grp;
                for (; sl + (KF - 1) * GROUPS < a.nslab; sl += KF * GROUPS) {
#pragma unroll
                    for (int k = 0; k < KF; ++k)
#pragma unroll
                        for (int t = 0; t < 9; ++t) w[k][t] += src[(long long)(sl + k * GROUPS) * a.total + t * a.Cip];
                }
                for (; sl < a.nslab; sl += GROUPS) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) w[0][t] += src[(long long)sl * a.total + t * a.Cip];
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = KF == 4 ? (w[0][t] + w[1][t]) + (w[2 % KF][t] + w[3 % KF][t]) : w[0][t] + w[1][t];
            }
        }
        if constexpr (GROUPS > 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t) part[grp][t][lane] = v[t];
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float sum = 0.f;
#pragma unroll
                    for (int g2 = 0; g2 < GROUPS; ++g2) sum += part[g2][t][lane];
                    v[t] = sum;
                }
            }
        }
        if (grp == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (in && a.rezero) a.dwp[(long long)co * 9 * a.Cip + t * a.Cip + ci] = 0.f;
                stg[lane * 9 + a.kpos[t]] = v[t];
            }
        }
        __syncthreads();
        // the block's pairs are consecutive (co, ci): their 3 x 3 windows are consecutive in the parameter wherever the channels are --
        // consecutive threads write consecutive floats
        for (int f = threadIdx.x; f < PPB * 9; f += NTH) {
            const int pl = f / 9, k = f - pl * 9;
            const long long pr = pair0 + pl;
            if (pr >= npair) break;
            const int co2 = (int)(pr / a.Cip), ci2 = (int)(pr % a.Cip);
            if (co2 < a.Co && ci2 < a.Ci) {
                float* const d = a.gw + (long long)co2 * a.s_out + (long long)(a.ci_off + ci2) * 9 + k;
                *d = a.acc ? *d + stg[f] : stg[f];
            }
        }
    } else {
        // tap by tap; gridDim.y > 1: the taps are spread over the blocks of a column too (many slabs, few pairs -- the thin layers'
        // 128 - 256 slabs of 1024 - 4096 pairs: 16 - 64 blocks walking nine taps each took 40 - 90 us beside the dependent chain)
        const bool keep = in && co < a.Co && ci < a.Ci;
        float* const dst = a.gw + (long long)co * a.s_out + (long long)(a.ci_off + ci) * a.s_in;
        const int tpb = (a.nt + (int)gridDim.y - 1) / (int)gridDim.y;
        const int t_lo = (int)blockIdx.y * tpb, t_hi = t_lo + tpb < a.nt ? t_lo + tpb : a.nt;
        for (int t = t_lo; t < t_hi; ++t) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if (in) {
                int sl = grp;
                for (; sl + 3 * GROUPS < a.nslab; sl += 4 * GROUPS) {
                    a0 += src[(long long)sl * a.total + t * a.Cip];
                    a1 += src[(long long)(sl + GROUPS) * a.total + t * a.Cip];
                    a2 += src[(long long)(sl + 2 * GROUPS) * a.total + t * a.Cip];
                    a3 += src[(long long)(sl + 3 * GROUPS) * a.total + t * a.Cip];
                }
                for (; sl < a.nslab; sl += GROUPS) a0 += src[(long long)sl * a.total + t * a.Cip];
            }
            float v = (a0 + a1) + (a2 + a3);
            if constexpr (GROUPS > 1) {
                if (t > t_lo) __syncthreads();                     // the previous tap's sums were read
                part[grp][0][lane] = v;
                __syncthreads();
                v = 0.f;
                if (grp == 0) {
#pragma unroll
                    for (int g2 = 0; g2 < GROUPS; ++g2) v += part[g2][0][lane];
                }
            }
            if (grp == 0) {
                if (in && a.rezero) a.dwp[(long long)co * a.nt * a.Cip + t * a.Cip + ci] = 0.f;
                if (keep) {
                    float* const d = dst + a.kpos[t];
                    *d = a.acc ? *d + v : v;
                }
            }
        }
    }
}

// slab 0 += slabs 1..nslab-1 (fixed order: bitwise reproducible).  256 threads = 64 consecutive elements x 4
// slab groups, four independent partial sums per thread so that 16 loads are in flight per lane.
__global__ __launch_bounds__(256) void slab_reduce_kernel(float* __restrict__ dwp, long long total, int nslab) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < total) {
        int s = grp;
        for (; s + 12 < nslab; s += 16) {
            a0 += dwp[(long long)s * total + i];
            a1 += dwp[(long long)(s + 4) * total + i];
            a2 += dwp[(long long)(s + 8) * total + i];
            a3 += dwp[(long long)(s + 12) * total + i];
        }
        for (; s < nslab; s += 4) a0 += dwp[(long long)s * total + i];
    }
    part[grp][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && i < total) dwp[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// few elements x many slabs (the first layer: 2304 x 256): the 36 blocks above would each walk 64 slabs per thread, 16 dependent
// round trips; 16 slab groups per block (1024 threads) and eight loads in flight per lane instead
__global__ __launch_bounds__(1024) void slab_reduce_deep_kernel(float* __restrict__ dwp, long long total, int nslab) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < total) {
        int s = grp;
        for (; s + 7 * 16 < nslab; s += 8 * 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += dwp[(long long)(s + 16 * k) * total + i];
        }
        for (; s < nslab; s += 16) a[0] += dwp[(long long)s * total + i];
    }
    part[grp][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (grp == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += part[g][lane];
        dwp[i] = t;
    }
}

// few slabs (<= 16: the layers with many tiles): one lane per FOUR consecutive elements, slabs summed in order.  The
// 64-element blocks above are launch-rate bound there (4.7 M elements x 2 slabs = 74 k blocks: 23 us for 57 MB).
__global__ __launch_bounds__(256) void slab_reduce_few_kernel(float4* __restrict__ dwp, long long total4, int nslab) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 a = dwp[i];
    int s = 1;
    for (; s + 3 < nslab; s += 4) {
        const float4 b0 = dwp[(long long)s * total4 + i], b1 = dwp[(long long)(s + 1) * total4 + i];
        const float4 b2 = dwp[(long long)(s + 2) * total4 + i], b3 = dwp[(long long)(s + 3) * total4 + i];
        a.x = (((a.x + b0.x) + b1.x) + b2.x) + b3.x;
        a.y = (((a.y + b0.y) + b1.y) + b2.y) + b3.y;
        a.z = (((a.z + b0.z) + b1.z) + b2.z) + b3.z;
        a.w = (((a.w + b0.w) + b1.w) + b2.w) + b3.w;
    }
    for (; s < nslab; ++s) {
        const float4 b = dwp[(long long)s * total4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    dwp[i] = a;
}

// number of pixel splits (= partial slabs) for a tile count, independent of the batch / image size so that the host
// can size the workspace per convolution.  One resident block per CU would be tiles * S = all CUs -- but the launch
// runs on the side stream beside the main stream's BatchNorm passes and data gradients, whose blocks cannot share
// a CU with a weight-gradient block (registers): the 64x64-tile launches take HALF of the CUs and leave the other
// half to the dependent chain (measured in situ: fraction 1 -> 5.69 ms/step, 0.75 -> 5.60, 0.5 -> 5.59, 0.4 -> 5.75,
// 0.25 -> 6.3; two blocks per CU 5.91).  Half the blocks is also half the slab traffic.  The thin 32x32-tile launches
// (224x224 / 112x112, HBM-bound) keep all CUs (1 -> 5.56, 0.5 -> 5.60, 0.25 -> 5.72 with the wide ones at 0.5).
// SEGNB_WG_CU_FRACTION overrides the wide share (read once); the thin (HBM-bound) launches keep all CUs, the flattened 7x7 /
// 14x14 tiles take the wide share (0.25 .. 1 measured within +-0.5 %).
int s1_slabs(int tiles, bool thin, bool flat = false) {
    static const double frac_wide = [] {
        const char* e = getenv("SEGNB_WG_CU_FRACTION");
        const double r = e ? atof(e) : 0.5;
        return r <= 0.0 ? 1.0 : r;
    }();
    static const double frac_thin = [] {
        const char* e = getenv("SEGNB_WG_CU_FRACTION_THIN");
        const double r = e ? atof(e) : 1.0;
        return r <= 0.0 ? 1.0 : r;
    }();
    constexpr double frac_flat = -1.0;
    const int pct = segnb_knob_wg_cu_pct();
    const double wide = pct > 0 ? pct / 100.0 : frac_wide;
    const double frac = thin ? frac_thin : ((flat && frac_flat > 0.0) ? frac_flat : wide);
    int S = (int)(segnb_num_cus() * frac) / tiles;
    return S < 1 ? 1 : S;
}

template <int BCO, int BCI, int R, int WT, bool FLAT = false, bool BNA = false>
int launch_s1(WgS1Args& a, int nslab, hipStream_t stream, bool partial, const segnb_wgrad_target* tgt = nullptr) {
    using TL = WgTile<R, WT, FLAT>;
    constexpr int tile_bytes = TL::XROWS * lds_stride(BCI) + TL::YROWS * lds_stride(BCO);
    constexpr int smem = wg_ws_db<BCO, BCI, R, WT, FLAT>() ? 2 * tile_bytes : tile_bytes;
    static_assert(smem <= 160 * 1024, "tiles fit the LDS");
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_s1x9_kernel<BCO, BCI, R, WT, FLAT, BNA>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) segnb_set_error("wgrad_s1 hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + R - 1) / R;
    a.WB = (a.W + WT - 1) / WT;
    a.IT = FLAT ? (a.N + R - 1) / R : a.N * a.HB * a.WB;         // FLAT: R whole images per iteration
    const int ncot = (a.Co + BCO - 1) / BCO;
    a.TCI_TILES = (a.Ci + BCI - 1) / BCI;
    const int tiles = ncot * a.TCI_TILES;
    const int S = s1_slabs(tiles, BCO == 32, FLAT);
    if (S != nslab) {
        segnb_set_error("segnb_conv_wgrad: workspace has %d slabs, this geometry needs %d (segnb_conv_wgrad_slabs)",
                        nslab, S);
        return SEGNB_E_BADARG;
    }
    a.its_per_split = (a.IT + S - 1) / S;
    a.slab_stride = (long long)a.Co * a.Ktot;
    a.gw = nullptr;
    if (tgt != nullptr && S == 1) {
        a.gw = tgt->gw;
        a.gw_s_out = tgt->s_out;
        a.gw_s_in = tgt->s_in; a.gw_ci_off = tgt->ci_off; a.gw_Ci = tgt->Ci; a.gw_Co = tgt->Co; a.gw_acc = tgt->accumulate;
        for (int t = 0; t < 9; ++t) a.gw_kpos[t] = tgt->kpos[t];
    }
    hipLaunchKernelGGL((conv_wgrad_s1x9_kernel<BCO, BCI, R, WT, FLAT, BNA>), dim3(tiles * S),
                       dim3(wg_specialised<BCO, BCI, R, WT, FLAT>() ? 512 : 256), smem, stream, a);
    if (S > 1 && !partial) {
        const long long total = a.slab_stride;
        if (S <= 16 && total % 4 == 0)
            hipLaunchKernelGGL(slab_reduce_few_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream,
                               reinterpret_cast<float4*>(a.dwp), total / 4, S);
        else
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, a.dwp,
                               total, S);
    }
    return a.gw != nullptr ? -1000 : 0;       // (-1000: written to the target by the kernel itself)
}

// tile configuration of the fast path for a geometry: 0 = not handled here (general kernel, one slab)
struct S1Choice {
    int cfg;        // 1: 32x32 R8 WT32, 2: 64x64 R4 WT32, 3: 64x64 R8 WT16, 4: flat 7x7 x4 images, 5: flat 14x14,
                    // 6: 32x32 R16 WT32 (halo rows 18/16 instead of 10/8 of the HBM-bound thin layers)
    int bco, bci;
};
S1Choice s1_choose(const segnb_conv_geom* g) {
    S1Choice c = {0, 0, 0};
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return c;
    if (g->QH != g->Ho || g->QW != g->Wo) return c;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return c;
    const bool thin = g->Co <= 32 || g->Ci <= 32;
    if (thin) {
        if (g->Wo < 24) return c;
        c = {(g->Ho % 16 == 0 && g->Ho >= 64 && SEGNB_WG_TALL) ? 6 : 1, 32, 32};
    } else if (g->Wo > 16) {
        // rows per iteration: 8, or 7, where they divide the image height (x tile 10 rows for 8, 9 for 7; half as many tile
        // hand-overs per pixel), else 4 (6 for 4).  Measured alone over the 14 such layers of the timed configuration: 1788 ->
        // 1667 us; in situ -0.6 % step time.  SEGNB_WG_ROWS=4 restores the short tiles.
        constexpr int rows = 8;
        // 16-column tiles where they cover the width exactly and 32-column tiles do not (112 = 7 x 16 against 4 x 32: an
        // eighth of the slabs were padding); SEGNB_WG_W16=0: off
        constexpr bool w16 = true;
        if (w16 && g->Wo % 16 == 0 && g->Wo % 32 != 0)
            c = {3, 64, 64};
        else
            c = {(rows == 8 && g->Ho % 8 == 0) ? 7 : ((rows >= 7 && g->Ho % 7 == 0) ? 8 : 2), 64, 64};
    } else if (g->Wo == 14 && g->Ho == 14) {
        c = {5, 64, 64};
    } else if (g->Wo >= 12) {
        c = {3, 64, 64};
    } else if (g->Wo == 7 && g->Ho == 7) {
        c = {4, 64, 64};
    }
    return c;
}


// ================================================================================================================
// Strided / wide-window weight gradients on the same tile scheme (LinkNet34's stem, linknet.py:16 -- ResNet34 conv1 7x7
// stride 2, 3 -> 64 --, its transposed 3x3 stride-2 "finaldeconv1", linknet.py:41, and the 3x3 stride-2 first convolutions
// of ResNet34's layer2..4): the general gather kernel fetched every x pixel once per tap as a separate 16-byte load.
//   dW[co][t][ci] = sum_{n,h,w} dout[n,h,w,co] * in[n, S h + dh[t], S w + dw[t], ci]
// ONE x tile with its (S (R-1) + KH) x (S (WT-1) + KW) halo and one dout tile per iteration feed ALL taps.  The GEMM's N
// dimension is the flattened (tap, channel) index k' = t * BCI + ci: ds_read_b64_tr_b16 takes a per-lane address, so the 16
// columns a lane group fetches may belong to different taps (BCI = 8: two taps) and consecutive K rows (output pixels) are
// S pixels apart in the tile.  Wave w owns the 32-column tiles w, w + 4, ... of k' for all 64 dout channels.
// ================================================================================================================
struct WgSxArgs {
    const bf16_t* x;
    const bf16_t* dy;
    float* dwp;
    int N, H, W, Hi, Wi;
    int Ci, Co, ld_x, ld_dy;
    int dhmin, dwmin;
    int ntaps;
    int HB, WB, IT, TCI_TILES, its_per_split;
    long long slab_stride;
    int Ktot;
    signed char dh[SEGNB_MAX_TAPS], dw[SEGNB_MAX_TAPS];     // minus (dhmin, dwmin)
};

constexpr int lds_stride_step(int channels, int step) {
    // bytes; multiple of 16, >= 2*channels, step * stride == 64 or 192 (mod 256): the four K rows of a fragment read, `step`
    // pixels apart, hit disjoint banks
    int s = channels * 2;
    while (!(((s * step) % 256) == 64 || ((s * step) % 256) == 192)) s += 16;
    return s;
}

template <int S, int BCI, int R, int WT, int KH, int KW, int BCO = 64>
__global__ __launch_bounds__(256, 1) void conv_wgrad_sx_kernel(const WgSxArgs a) {
    constexpr int TCO = BCO / 32;
    static_assert(BCO == 32 || BCO == 64, "dout channel tile");
    constexpr int XR = (R - 1) * S + KH, XC = (WT - 1) * S + KW;
    constexpr int SX = lds_stride_step(BCI, S), SY = lds_stride(BCO);
    constexpr int NK = KH * KW * BCI;                  // columns k' of this block
    constexpr int NT32 = (NK + 31) / 32, NTW = (NT32 + 3) / 4;
    constexpr int NSLAB = R * WT / 16, SEGS = WT / 16;
    constexpr int XCH = XR * XC * (BCI / 8), YCH = R * WT * (BCO / 8);
    constexpr int XPT = (XCH + 255) / 256, YPT = (YCH + 255) / 256;
    static_assert(WT % 16 == 0 && NTW <= 4, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sY = smem + XR * XC * SX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = blockIdx.x % a.TCI_TILES;
    const int rest = blockIdx.x / a.TCI_TILES;
    const int ncot = (a.Co + BCO - 1) / BCO;
    const int mtile = rest % ncot, split = rest / ncot;
    const int co0 = mtile * BCO, ci0 = ntile * BCI;
    const int it_begin = split * a.its_per_split;
    int it_end = it_begin + a.its_per_split;
    if (it_end > a.IT) it_end = a.IT;
    float* __restrict__ slab = a.dwp + (long long)split * a.slab_stride;

    uint4 rx[XPT], ry[YPT];
    auto gload = [&](int it) {
        const int n = it / (a.HB * a.WB);
        const int rem = it - n * (a.HB * a.WB);
        const int hb = rem / a.WB, wb = rem - hb * a.WB;
        const int h0 = hb * R, w0 = wb * WT;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int c = tid + u * 256;
            const int pix = c / (BCI / 8), cc = c - pix * (BCI / 8);
            const int xr = pix / XC, xc = pix - xr * XC;
            const int hi = h0 * S + a.dhmin + xr, wi = w0 * S + a.dwmin + xc;
            const int ch = ci0 + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < XCH && ch < a.Ci && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
                v = *reinterpret_cast<const uint4*>(a.x + ((long long)(n * a.Hi + hi) * a.Wi + wi) * a.ld_x + ch);
            rx[u] = v;
        }
#pragma unroll
        for (int u = 0; u < YPT; ++u) {
            const int c = tid + u * 256;
            const int pix = c / (BCO / 8), cc = c - pix * (BCO / 8);
            const int yr = pix / WT, yc = pix - yr * WT;
            const int ho = h0 + yr, wo = w0 + yc;
            const int ch = co0 + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < YCH && ch < a.Co && ho < a.H && wo < a.W)
                v = *reinterpret_cast<const uint4*>(a.dy + ((long long)(n * a.H + ho) * a.W + wo) * a.ld_dy + ch);
            ry[u] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int c = tid + u * 256;
            if (c < XCH) {
                const int pix = c / (BCI / 8), cc = c - pix * (BCI / 8);
                *reinterpret_cast<uint4*>(sX + pix * SX + cc * 16) = rx[u];
            }
        }
#pragma unroll
        for (int u = 0; u < YPT; ++u) {
            const int c = tid + u * 256;
            if (c < YCH) {
                const int pix = c / (BCO / 8), cc = c - pix * (BCO / 8);
                *reinterpret_cast<uint4*>(sY + pix * SY + cc * 16) = ry[u];
            }
        }
    };

    // fragment addressing (see conv_wgrad_s1x9_kernel): lane 4q+p of a 16-lane group supplies the address of K row q, columns
    // 4p..4p+3 of the group's 16; here a column is k' = (tap, channel)
    const int q = (lane & 15) >> 2, p = lane & 3, h = lane >> 5, cbase = 16 * ((lane >> 4) & 1);
    const int a_off = (8 * h + q) * SY + (cbase + 4 * p) * 2;
    int b_off[NTW];
#pragma unroll
    for (int jj = 0; jj < NTW; ++jj) {
        int kp = 32 * (wave + 4 * jj) + cbase + 4 * p;
        if (kp >= a.ntaps * BCI) kp = 0;                  // (columns past the last tap: computed on tap 0's data, never stored)
        const int t = kp / BCI, ci = kp - t * BCI;
        b_off[jj] = ((8 * h + q) * S + (int)a.dh[t] * XC + (int)a.dw[t]) * SX + ci * 2;
    }
    const int ntw = wave < NT32 - 4 * (NTW - 1) ? NTW : NTW - 1;      // 32-column tiles of this wave (wave-uniform)

    f32x16_t acc[TCO][NTW];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int jj = 0; jj < NTW; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    if (it_begin < it_end) {
        gload(it_begin);
        lstore();
    }
    __syncthreads();
    for (int it = it_begin; it < it_end; ++it) {
        if (it + 1 < it_end) gload(it + 1);
#pragma unroll 2
        for (int s = 0; s < NSLAB; ++s) {
            const int row = s / SEGS, seg = s - row * SEGS;
            const unsigned char* pa = sY + a_off + (row * WT + seg * 16) * SY;
            const unsigned char* pb = sX + (row * S * XC + seg * 16 * S) * SX;
            bf16x8_t fa[TCO];
#pragma unroll
            for (int i = 0; i < TCO; ++i) fa[i] = tr_frag(pa + 64 * i, 4 * SY);
#pragma unroll
            for (int jj = 0; jj < NTW; ++jj) {
                if (jj < ntw) {
                    const bf16x8_t fb = tr_frag(pb + b_off[jj], 4 * S * SX);
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[i][jj], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (it + 1 < it_end) {
            lstore();
            __syncthreads();
        }
    }
    // D[i = co][j = k']: lane holds column lane & 31, rows (e & 3) + 8 (e >> 2) + 4 h.  Every (tile, split) block owns its piece
    // of partial slab `split` (plain stores; an empty pixel range stores zeros)
#pragma unroll
    for (int jj = 0; jj < NTW; ++jj) {
        if (jj >= ntw) continue;
        const int kp = 32 * (wave + 4 * jj) + (lane & 31);
        const int t = kp / BCI, ci = ci0 + kp - t * BCI;
        if (kp >= a.ntaps * BCI || ci >= a.Ci) continue;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < a.Co) slab[(long long)co * a.Ktot + t * a.Ci + ci] = acc[i][jj][e];
            }
    }
}

struct SxChoice {
    int cfg;        // 0: not served.  1: 7x7 window stride 2, 8 channels (stem); 2: 3x3 stride 2, 32-channel tiles, 32 columns;
                    // 3: the same on 16-column tiles (outputs narrower than 32); 9 / 10: 4x4 stride 2 likewise; 4..7: 1x1 stride 1 (a plain dy^T x GEMM: the
                    // pixels of the whole batch as ONE row of 256-pixel tiles) with 16 / 32 / 64 / 128 input channels per block
    int bci;        // 8: 2x2 window stride 1, at most 32 output channels (LinkNet34's finalconv3, linknet.py:45)
    int bco = 64;
};
SxChoice sx_choose(const segnb_conv_geom* g) {
    SxChoice c = {0, 0, 64};
    static const bool off = getenv("SEGNB_WGRAD_SX") != nullptr && getenv("SEGNB_WGRAD_SX")[0] == '0';
    if (off || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0 || g->QH != g->Ho || g->QW != g->Wo) return c;
    if (g->in_step == 1 && g->ntaps == 1 && g->dh[0] == 0 && g->dw[0] == 0 && g->Hi == g->Ho && g->Wi == g->Wo &&
        g->Ho * g->Wo >= 256) {                      // (per image: the choice may not depend on the batch size)
        if (g->Ci <= 16) c = {4, 16};
        else if (g->Ci <= 32) c = {5, 32};
        else if (g->Ci < 96) c = {6, 64};
        else c = {7, 128};
        return c;
    }
    if (g->in_step == 1) {
        int hmin = g->dh[0], hmax = g->dh[0], wmin = g->dw[0], wmax = g->dw[0];
        for (int t = 1; t < g->ntaps; ++t) {
            hmin = g->dh[t] < hmin ? g->dh[t] : hmin;
            hmax = g->dh[t] > hmax ? g->dh[t] : hmax;
            wmin = g->dw[t] < wmin ? g->dw[t] : wmin;
            wmax = g->dw[t] > wmax ? g->dw[t] : wmax;
        }
        if (g->ntaps == 4 && hmax - hmin == 1 && wmax - wmin == 1 && g->Ci % 32 == 0 && g->Co <= 32 && g->Wo >= 24) c = {8, 32, 32};
        return c;
    }
    if (g->in_step != 2) return c;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < g->ntaps; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    const int kh = dhmax - dhmin + 1, kw = dwmax - dwmin + 1;
    if (g->ntaps != kh * kw) return c;
    if (kh == 7 && kw == 7 && g->Ci == 8 && g->Wo >= 32) c = {1, 8};
    else if (kh == 3 && kw == 3 && g->Ci % 32 == 0 && g->Wo >= 24) c = {2, 32};
    else if (kh == 3 && kw == 3 && g->Ci % 32 == 0 && g->Wo >= 12) c = {3, 32};
    else if (kh == 4 && kw == 4 && g->Ci % 32 == 0 && g->Wo >= 24) c = {9, 32};        // (UNet16's ConvTranspose2d(4, 2, 1), unet16.py:30)
    else if (kh == 4 && kw == 4 && g->Ci % 32 == 0 && g->Wo >= 12) c = {10, 32};
    return c;
}

template <int S, int BCI, int R, int WT, int KH, int KW, int BCO = 64>
int launch_sx(WgSxArgs& a, int nslab, hipStream_t stream, bool partial) {
    constexpr int XR = (R - 1) * S + KH, XC = (WT - 1) * S + KW;
    constexpr int smem = XR * XC * lds_stride_step(BCI, S) + R * WT * lds_stride(BCO);
    static_assert(smem <= 160 * 1024, "tiles fit the LDS");
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_sx_kernel<S, BCI, R, WT, KH, KW, BCO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) segnb_set_error("wgrad_sx hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + R - 1) / R;
    a.WB = (a.W + WT - 1) / WT;
    a.IT = a.N * a.HB * a.WB;
    a.TCI_TILES = (a.Ci + BCI - 1) / BCI;
    const int tiles = ((a.Co + BCO - 1) / BCO) * a.TCI_TILES;
    const int S_ = s1_slabs(tiles, true);
    if (S_ != nslab) {
        segnb_set_error("segnb_conv_wgrad: workspace has %d slabs, this geometry needs %d (segnb_conv_wgrad_slabs)", nslab, S_);
        return SEGNB_E_BADARG;
    }
    a.its_per_split = (a.IT + S_ - 1) / S_;
    a.slab_stride = (long long)a.Co * a.Ktot;
    hipLaunchKernelGGL((conv_wgrad_sx_kernel<S, BCI, R, WT, KH, KW, BCO>), dim3(tiles * S_), dim3(256), smem, stream, a);
    if (S_ > 1 && !partial) {
        const long long total = a.slab_stride;
        if (S_ <= 16 && total % 4 == 0)
            hipLaunchKernelGGL(slab_reduce_few_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream,
                               reinterpret_cast<float4*>(a.dwp), total / 4, S_);
        else
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, a.dwp, total, S_);
    }
    return 0;
}

}  // namespace

// slab 0 += slabs 1 .. nslab - 1 of a [nslab][total] fp32 workspace, fixed order (shared with wgrad_roll.hip)
void segnb_slab_reduce(float* dwp, long long total, int nslab, hipStream_t stream) {
    if (nslab <= 16 && total % 4 == 0)
        hipLaunchKernelGGL(slab_reduce_few_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<float4*>(dwp), total / 4, nslab);
    else if (total <= 4096 && nslab >= 128)
        hipLaunchKernelGGL(slab_reduce_deep_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, stream, dwp, total, nslab);
    else
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, dwp, total, nslab);
}

void segnb_wgrad_to_param(float* dwp, int Cop, int ntaps, int Cip, int nslab, const segnb_wgrad_target* tgt, bool rezero,
                          hipStream_t stream) {
    ParamReduceArgs p;
    p.dwp = dwp;
    p.total = (long long)Cop * ntaps * Cip;
    p.nslab = nslab; p.nt = ntaps; p.Cip = Cip; p.Cop = Cop;
    p.gw = tgt->gw;
    p.s_out = tgt->s_out;
    p.s_in = tgt->s_in; p.ci_off = tgt->ci_off; p.Ci = tgt->Ci; p.Co = tgt->Co; p.acc = tgt->accumulate;
    p.rezero = rezero ? 1 : 0;
    for (int t = 0; t < ntaps; ++t) p.kpos[t] = tgt->kpos[t];
    const long long pairs = (long long)Cop * Cip;
    // few slabs (the layers with many channel tiles = most of the parameters): one thread per pair walks the slabs in order, the
    // block's 3 x 3 windows leave as consecutive floats.  Many slabs (>= 32: few pairs): the slabs are summed by 4 / 16 groups per
    // block AND the taps are spread over the grid's second dimension; the few outputs leave as element stores
    const bool nt9 = ntaps == 9 && tgt->s_in == 9;
    const bool tappar = nslab >= 32;
    const int groups = nslab <= 16 ? 1 : (nslab < 128 ? 4 : 16);      // (16 / 8 / 4 groups at >= 128 slabs: the same step time, r06_ab.txt)
    const unsigned gy = tappar ? (unsigned)ntaps : 1u;
#define SEGNB_PR_LAUNCH(G, N9)                                                                                            \
    hipLaunchKernelGGL((slab_reduce_param_kernel<G, N9>), dim3((unsigned)((pairs + (G == 1 ? 255 : 63)) / (G == 1 ? 256 : 64)), gy), \
                       dim3(G == 1 ? 256 : 64 * G), 0, stream, p)
    if (groups == 1) { if (nt9) SEGNB_PR_LAUNCH(1, true); else SEGNB_PR_LAUNCH(1, false); }
    else if (groups == 4) { if (nt9 && !tappar) SEGNB_PR_LAUNCH(4, true); else SEGNB_PR_LAUNCH(4, false); }
    else SEGNB_PR_LAUNCH(16, false);
#undef SEGNB_PR_LAUNCH
}

// partial slabs segnb_conv_wgrad writes for this geometry on the fast path (0: not a fast-path geometry)
int segnb_wgrad_s1_slabs(const segnb_conv_geom* g) {
    const S1Choice c = s1_choose(g);
    if (!c.cfg) return 0;
    return s1_slabs(((g->Co + c.bco - 1) / c.bco) * ((g->Ci + c.bci - 1) / c.bci), c.bco == 32, c.cfg == 4 || c.cfg == 5);
}

// returns 1 when the launch was handled here, 0 when the geometry is not a stride-1 3x3 bf16 case
// (caller falls through to the general kernel), <0 / hipError on failure
int segnb_wgrad_s1_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab,
                       hipStream_t stream, bool partial, const segnb_wgrad_bnapply* bna, const segnb_upcat_src* uc,
                       const segnb_wgrad_target* tgt) {
    const S1Choice c = s1_choose(g);
    if (!c.cfg) return 0;
    if (tgt != nullptr) partial = true;           // the slabs are summed by segnb_wgrad_to_param
    if (uc != nullptr && (uc->Cu % 8 != 0 || uc->Cu <= 0 || uc->Cu >= g->Ci || (g->Hi & 1) || (g->Wi & 1))) return 0;
    if (bna != nullptr && c.cfg != 1 && c.cfg != 6) return 0;      // (thin 32 x 32 tiles only)
    int dhmin = g->dh[0], dwmin = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
    }
    WgS1Args a;
    a.x = (const bf16_t*)in;
    a.dy = (const bf16_t*)dout;
    a.dwp = dwp;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Hi = g->Hi; a.Wi = g->Wi;
    a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_dy = g->ld_out;
    a.dhmin = dhmin; a.dwmin = dwmin;
    for (int t = 0; t < 9; ++t) {
        a.dh[t] = g->dh[t] - dhmin;
        a.dw[t] = g->dw[t] - dwmin;
    }
    a.Ktot = 9 * g->Ci;
    a.u = uc != nullptr ? (const bf16_t*)uc->u : nullptr;
    a.Cu = uc != nullptr ? uc->Cu : 0;
    a.ld_u = uc != nullptr ? uc->ld_u : 0;
    a.bna = segnb_wgrad_bnapply{};
    int rc;
    if (bna != nullptr) {
        a.bna = *bna;
        rc = c.cfg == 1 ? launch_s1<32, 32, 8, 32, false, true>(a, nslab, stream, partial, tgt)
                        : launch_s1<32, 32, 16, 32, false, true>(a, nslab, stream, partial, tgt);
        return rc == -1000 ? 2 : (rc ? rc : 1);
    }
    if (c.cfg == 1) rc = launch_s1<32, 32, 8, 32>(a, nslab, stream, partial, tgt);
    else if (c.cfg == 6) rc = launch_s1<32, 32, 16, 32>(a, nslab, stream, partial, tgt);
    else if (c.cfg == 2) rc = launch_s1<64, 64, 4, 32>(a, nslab, stream, partial, tgt);
    else if (c.cfg == 7) rc = launch_s1<64, 64, 8, 32>(a, nslab, stream, partial, tgt);
    else if (c.cfg == 8) rc = launch_s1<64, 64, 7, 32>(a, nslab, stream, partial, tgt);
    else if (c.cfg == 3) rc = launch_s1<64, 64, 8, 16>(a, nslab, stream, partial, tgt);
    else if (c.cfg == 4) rc = launch_s1<64, 64, 4, 7, true>(a, nslab, stream, partial, tgt);
    else rc = launch_s1<64, 64, 1, 14, true>(a, nslab, stream, partial, tgt);
    return rc == -1000 ? 2 : (rc ? rc : 1);           // 2: the single-slab launch wrote the target itself
}


// the strided / wide-window tile kernel (conv_wgrad_sx_kernel): slabs it writes (0: geometry not served) and the launch
int segnb_wgrad_sx_slabs(const segnb_conv_geom* g) {
    const SxChoice c = sx_choose(g);
    if (!c.cfg) return 0;
    return s1_slabs(((g->Co + c.bco - 1) / c.bco) * ((g->Ci + c.bci - 1) / c.bci), true);
}

int segnb_wgrad_sx_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab, hipStream_t stream,
                       bool partial) {
    const SxChoice c = sx_choose(g);
    if (!c.cfg) return 0;
    int dhmin = g->dh[0], dwmin = g->dw[0];
    for (int t = 1; t < g->ntaps; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
    }
    WgSxArgs a;
    a.x = (const bf16_t*)in;
    a.dy = (const bf16_t*)dout;
    a.dwp = dwp;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Hi = g->Hi; a.Wi = g->Wi;
    a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_dy = g->ld_out;
    a.dhmin = dhmin; a.dwmin = dwmin;
    a.ntaps = g->ntaps;
    for (int t = 0; t < g->ntaps; ++t) {
        a.dh[t] = (signed char)(g->dh[t] - dhmin);
        a.dw[t] = (signed char)(g->dw[t] - dwmin);
    }
    a.Ktot = g->ntaps * g->Ci;
    int rc;
    if (c.cfg >= 4 && c.cfg <= 7) {
        a.W = a.Wi = g->N * g->Ho * g->Wo;          // 1x1: no halo, the batch is one row of pixels
        a.N = a.H = a.Hi = 1;
        rc = c.cfg == 4 ? launch_sx<1, 16, 1, 256, 1, 1>(a, nslab, stream, partial)
           : c.cfg == 5 ? launch_sx<1, 32, 1, 256, 1, 1>(a, nslab, stream, partial)
           : c.cfg == 6 ? launch_sx<1, 64, 1, 256, 1, 1>(a, nslab, stream, partial)
                        : launch_sx<1, 128, 1, 256, 1, 1>(a, nslab, stream, partial);
        return rc ? rc : 1;
    }
    if (c.cfg == 8) rc = launch_sx<1, 32, 8, 32, 2, 2, 32>(a, nslab, stream, partial);
    else if (c.cfg == 1) rc = launch_sx<2, 8, 8, 32, 7, 7>(a, nslab, stream, partial);
    else if (c.cfg == 2) rc = launch_sx<2, 32, 4, 32, 3, 3>(a, nslab, stream, partial);
    else if (c.cfg == 9) rc = launch_sx<2, 32, 4, 32, 4, 4>(a, nslab, stream, partial);
    else if (c.cfg == 10) rc = launch_sx<2, 32, 8, 16, 4, 4>(a, nslab, stream, partial);
    else rc = launch_sx<2, 32, 8, 16, 3, 3>(a, nslab, stream, partial);
    return rc ? rc : 1;
}
