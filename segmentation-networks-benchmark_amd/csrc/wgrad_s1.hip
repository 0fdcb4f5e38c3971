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
    int Ktot;               // 9 * Ci
};

constexpr int lds_stride(int channels) {
    // bytes; multiple of 16, >= 2*channels, == 64 or 192 (mod 256)
    int s = channels * 2;
    while (!((s % 256) == 64 || (s % 256) == 192)) s += 16;
    return s;
}

template <int BCO, int BCI, int R, int WT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_s1x9_kernel(const WgS1Args a) {
    constexpr int TCO = BCO / 32, TCI = BCI / 32;
    constexpr int NSUB = TCO * TCI;
    static_assert(NSUB == 1 || NSUB == 4, "tile is 32x32 or 64x64");
    constexpr int KSPLIT = 4 / NSUB;
    constexpr int XR = R + 2, XC = WT + 2;
    constexpr int SX = lds_stride(BCI), SY = lds_stride(BCO);
    constexpr int NSLAB = R * WT / 16;
    constexpr int SEGS = WT / 16;
    constexpr int XCH = XR * XC * (BCI / 8);        // 16-byte chunks of the x tile
    constexpr int YCH = R * WT * (BCO / 8);
    constexpr int XPT = (XCH + 255) / 256, YPT = (YCH + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sY = smem + XR * XC * SX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = blockIdx.x % a.TCI_TILES;     // consecutive blocks share dy, differ in ci tile
    const int rest = blockIdx.x / a.TCI_TILES;
    const int ncot = (a.Co + BCO - 1) / BCO;
    const int mtile = rest % ncot;
    const int split = rest / ncot;
    const int co0 = mtile * BCO, ci0 = ntile * BCI;

    const int it_begin = split * a.its_per_split;
    int it_end = it_begin + a.its_per_split;
    if (it_end > a.IT) it_end = a.IT;
    if (it_begin >= it_end) return;

    uint4 rx[XPT], ry[YPT];
    auto gload = [&](int it) {
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

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    gload(it_begin);
    lstore();
    __syncthreads();
    for (int it = it_begin; it < it_end; ++it) {
        if (it + 1 < it_end) gload(it + 1);
#pragma unroll
        for (int s0 = 0; s0 < NSLAB / KSPLIT; ++s0) {
            const int s = s0 * KSPLIT + kpart;
            const int rr = s / SEGS, cs = (s - rr * SEGS) * 16;
            const unsigned char* pa = sY + a_off + (rr * WT + cs) * SY;
            const bf16x8_t af = tr_frag(pa, 4 * SY);
            const unsigned char* pb = sX + b_off + (rr * XC + cs) * SX;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16x8_t bfr = tr_frag(pb + tap_off[t], 4 * SX);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();                    // everyone done reading this iteration's tiles
        if (it + 1 < it_end) {
            lstore();
            __syncthreads();
        }
    }

    // D[i = co][j = ci]: lane holds column ci = lane&31, rows co = (e&3) + 8*(e>>2) + 4*h
    if (KSPLIT > 1) {
        // the waves hold partial sums of the SAME 32x32x9 tile: merge them in LDS (tiles are dead by now: the
        // loop ended on a barrier), one atomic per element per block instead of four
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
            if (co < a.Co && ci < a.Ci)
                atomicAdd(&a.dwp[(long long)co * a.Ktot + t * a.Ci + ci], acc[t][e]);
        }
}

template <int BCO, int BCI, int R, int WT>
int launch_s1(WgS1Args& a, hipStream_t stream) {
    constexpr int smem = (R + 2) * (WT + 2) * lds_stride(BCI) + R * WT * lds_stride(BCO);
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_s1x9_kernel<BCO, BCI, R, WT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) segnb_set_error("wgrad_s1 hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + R - 1) / R;
    a.WB = (a.W + WT - 1) / WT;
    a.IT = a.N * a.HB * a.WB;
    const int ncot = (a.Co + BCO - 1) / BCO;
    a.TCI_TILES = (a.Ci + BCI - 1) / BCI;
    const int tiles = ncot * a.TCI_TILES;
    // two resident blocks per CU; every block ends with an atomic merge of its whole [BCO][9*BCI] partial, so
    // keep the pixel split coarse: at least 6 iterations per block
    int S = (segnb_num_cus() * 2 + tiles - 1) / tiles;
    if (S > a.IT / 6) S = a.IT / 6;
    if (S < 1) S = 1;
    a.its_per_split = (a.IT + S - 1) / S;
    S = (a.IT + a.its_per_split - 1) / a.its_per_split;
    hipLaunchKernelGGL((conv_wgrad_s1x9_kernel<BCO, BCI, R, WT>), dim3(tiles * S), dim3(256), smem, stream, a);
    return 0;
}

}  // namespace

// returns 1 when the launch was handled here, 0 when the geometry is not a stride-1 3x3 bf16 case
// (caller falls through to the general kernel), <0 / hipError on failure
int segnb_wgrad_s1_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, hipStream_t stream) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return 0;
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
    const bool thin = g->Co <= 32 || g->Ci <= 32;
    if (thin) {
        if (g->Wo < 24) return 0;
        rc = launch_s1<32, 32, 8, 32>(a, stream);
    } else if (g->Wo > 16) {
        rc = launch_s1<64, 64, 4, 32>(a, stream);
    } else if (g->Wo >= 12) {
        rc = launch_s1<64, 64, 8, 16>(a, stream);
    } else {
        return 0;
    }
    return rc ? rc : 1;
}
