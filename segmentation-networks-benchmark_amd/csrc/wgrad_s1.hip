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
#define SEGNB_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n))

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
};

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
template <int S0, int SPW, int KSPLIT, int SEGS, int WT, int XC, int SX, int SY>
__device__ __forceinline__ void wg_slabs(f32x16_t (&acc)[9], bf16x4_t (&fa)[2][2], bf16x4_t (&fb)[2][9][2],
                                         unsigned va, const unsigned (&vb)[9]) {
    if constexpr (S0 < SPW) {
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
        wg_slabs<S0 + 1, SPW, KSPLIT, SEGS, WT, XC, SX, SY>(acc, fa, fb, va, vb);
    }
}

template <int T, int SXB>
__device__ __forceinline__ void wg_first(bf16x4_t (&fb0)[9][2], const unsigned (&vb)[9]) {
    if constexpr (T < 9) {
        SEGNB_TR_READ2(fb0[T][0], fb0[T][1], vb[T], 0, 4 * SXB);
        wg_first<T + 1, SXB>(fb0, vb);
    }
}

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

template <int BCO, int BCI, int R, int WT, bool FLAT = false>
__global__ __launch_bounds__(256, 1) void conv_wgrad_s1x9_kernel(const WgS1Args a) {
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

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sY = smem + TL::XROWS * SX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        for (int i = tid; i < BCO * 9 * BCI; i += 256) {
            const int ci = ci0 + i % BCI, t = (i / BCI) % 9, co = co0 + i / (9 * BCI);
            if (co < a.Co && ci < a.Ci) slab[(long long)co * a.Ktot + t * a.Ci + ci] = 0.f;
        }
        return;
    }

    uint4 rx[XPT], ry[YPT];
    auto gload = [&](int it) {
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
                if (c < XCH && n < a.N && ch < a.Ci && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
                    v = *reinterpret_cast<const uint4*>(a.x + ((long long)(n * a.Hi + hi) * a.Wi + wi) * a.ld_x + ch);
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

    // fragment addressing: 16-lane group g reads 4 pixel rows x 16 channels; lane 4q+p supplies the address of
    // pixel row q, channels 4p..4p+3 and receives channel (l&15) of the 4 rows (probe: tools/probe_tr.hip)
    const int sub = wave / KSPLIT, kpart = wave - sub * KSPLIT;
    const int sco = sub / TCI, sci = sub - sco * TCI;
    const int q = (lane & 15) >> 2, p = lane & 3, h = lane >> 5, cbase = 16 * ((lane >> 4) & 1);
    const int a_off = (8 * h + q) * SY + (sco * 32 + cbase + 4 * p) * 2;
    const int b_off = (8 * h + q) * SX + (sci * 32 + cbase + 4 * p) * 2;
    int tap_off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap_off[t] = (a.dh[t] * XC + a.dw[t]) * SX;

    // accumulators are defined and updated only by "a"-constrained asm, so they live in AGPRs across the whole
    // pixel loop (a VALU zero-init makes the allocator keep them in VGPRs and copy 144 registers into and out of
    // AGPRs around every iteration)
    f32x16_t acc[9];
    {
        const bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t)      // s_nop: the hazard recognizer does not see into asm (VALU write of z -> MFMA read)
            asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[t]) : "v"(z));
    }

    gload(it_begin);
    lstore();
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
    for (int it = it_begin; it < it_end; ++it) {
        if (it + 1 < it_end) gload(it + 1);
        bf16x4_t fa[2][2], fb[2][9][2];
        SEGNB_TR_READ2(fa[0][0], fa[0][1], va, 0, 4 * SY);          // prologue: slab 0 -> fragment set 0
        wg_first<0, SX>(fb[0], vb);
        wg_slabs<0, SPW, KSPLIT, SEGS, TL::YSTEP, TL::XSTEP, SX, SY>(acc, fa, fb, va, vb);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::);      // last MFMA results land before anyone reads the accumulators
        __syncthreads();                    // everyone done reading this iteration's tiles
        if (it + 1 < it_end) {
            lstore();
            __syncthreads();
        }
    }

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
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + sco * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (co < a.Co && ci < a.Ci) slab[(long long)co * a.Ktot + t * a.Ci + ci] = acc[t][e];
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

// number of pixel splits (= partial slabs) for a tile count: one resident block per CU (the kernel holds 144
// accumulator AGPRs + 160 VGPRs per lane), independent of the batch / image size so that the host can size the
// workspace per convolution
int s1_slabs(int tiles) {
    int S = segnb_num_cus() / tiles;
    return S < 1 ? 1 : S;
}

template <int BCO, int BCI, int R, int WT, bool FLAT = false>
int launch_s1(WgS1Args& a, int nslab, hipStream_t stream) {
    using TL = WgTile<R, WT, FLAT>;
    constexpr int smem = TL::XROWS * lds_stride(BCI) + TL::YROWS * lds_stride(BCO);
    static_assert(smem <= 160 * 1024, "tiles fit the LDS");
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_s1x9_kernel<BCO, BCI, R, WT, FLAT>),
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
    const int S = s1_slabs(tiles);
    if (S != nslab) {
        segnb_set_error("segnb_conv_wgrad: workspace has %d slabs, this geometry needs %d (segnb_conv_wgrad_slabs)",
                        nslab, S);
        return SEGNB_E_BADARG;
    }
    a.its_per_split = (a.IT + S - 1) / S;
    a.slab_stride = (long long)a.Co * a.Ktot;
    hipLaunchKernelGGL((conv_wgrad_s1x9_kernel<BCO, BCI, R, WT, FLAT>), dim3(tiles * S), dim3(256), smem, stream, a);
    if (S > 1) {
        const long long total = a.slab_stride;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, a.dwp, total,
                           S);
    }
    return 0;
}

// tile configuration of the fast path for a geometry: 0 = not handled here (general kernel, one slab)
struct S1Choice {
    int cfg;        // 1: 32x32 R8 WT32, 2: 64x64 R4 WT32, 3: 64x64 R8 WT16, 4: flat 7x7 x4 images, 5: flat 14x14
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
        c = {1, 32, 32};
    } else if (g->Wo > 16) {
        c = {2, 64, 64};
    } else if (g->Wo == 14 && g->Ho == 14) {
        c = {5, 64, 64};
    } else if (g->Wo >= 12) {
        c = {3, 64, 64};
    } else if (g->Wo == 7 && g->Ho == 7) {
        c = {4, 64, 64};
    }
    return c;
}

}  // namespace

// partial slabs segnb_conv_wgrad writes for this geometry on the fast path (0: not a fast-path geometry)
int segnb_wgrad_s1_slabs(const segnb_conv_geom* g) {
    const S1Choice c = s1_choose(g);
    if (!c.cfg) return 0;
    return s1_slabs(((g->Co + c.bco - 1) / c.bco) * ((g->Ci + c.bci - 1) / c.bci));
}

// returns 1 when the launch was handled here, 0 when the geometry is not a stride-1 3x3 bf16 case
// (caller falls through to the general kernel), <0 / hipError on failure
int segnb_wgrad_s1_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab,
                       hipStream_t stream) {
    const S1Choice c = s1_choose(g);
    if (!c.cfg) return 0;
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
    int rc;
    if (c.cfg == 1) rc = launch_s1<32, 32, 8, 32>(a, nslab, stream);
    else if (c.cfg == 2) rc = launch_s1<64, 64, 4, 32>(a, nslab, stream);
    else if (c.cfg == 3) rc = launch_s1<64, 64, 8, 16>(a, nslab, stream);
    else if (c.cfg == 4) rc = launch_s1<64, 64, 4, 7, true>(a, nslab, stream);
    else rc = launch_s1<64, 64, 1, 14, true>(a, nslab, stream);
    return rc ? rc : 1;
}
