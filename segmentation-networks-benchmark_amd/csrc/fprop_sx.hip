// Forward of a strided wide-window convolution with few input channels -- LinkNet34's stem, linknet.py:16: ResNet34 conv1, 7x7
// stride 2, 3 (padded 8) -> 64 -- on halo tiles.  (Written for every geometry the stride-1 3x3 pipelines do not serve; only
// the stem beat the general gather kernel: see segnb_fprop_sx_try.)
//
//     out[n, qh*os + oh0, qw*os + ow0, co] = bias[co] + sum_{t, ci} in[n, qh*S + dh[t], qw*S + dw[t], ci] * W[co][t][ci]
//
// The general gather kernel (conv_igemm.hip) fetches every input pixel once per tap as a separate 16-byte load.  Here ONE
// input tile with its (S (R-1) + KH) x (S (WT-1) + KW) halo per (pixel tile, channel chunk) feeds all taps: the A operand of
// v_mfma_f32_32x32x16_bf16 (rows = output pixels, K = 8 consecutive channels of one tap per lane) is a 16-byte LDS read at a
// per-lane address -- pixel base + tap offset -- so stride, window and tap subset are address arithmetic; for 8-channel
// inputs the two K halves of an instruction are two taps.  B = the packed weights [co][tap][ci] of the chunk, K-contiguous
// as they are.  Accumulators leave through an LDS staging tile as 16-byte channel rows; BatchNorm statistics (sum, sum of
// squares of the stored values) ride along in four registers per lane.  Block = 256 threads, tile = R x WT output-grid
// pixels x BCO channels; register prefetch of the next (tile, chunk) under the MFMAs of the current one.
#include "common.h"

#include <cstdlib>

namespace {

struct FxArgs {
    const bf16_t* x;
    const bf16_t* w;
    const float* bias;
    bf16_t* out;
    double* stats;
    int bias_n;
    int N, QH, QW, Hi, Wi, Ho, Wo;
    int out_step, oh0, ow0;
    int Ci, Co, ld_x, ld_out;
    int dhmin, dwmin, ntaps;
    int HB, WB, IT, NCHUNK, NCOT;
    signed char dh[SEGNB_MAX_TAPS], dw[SEGNB_MAX_TAPS];     // minus (dhmin, dwmin)
};

template <int S, int BCI, int R, int WT, int KH, int KW, int BCO>
struct FxCfg {
    static constexpr int TP = R * WT;                          // output-grid pixels per tile
    static constexpr int MPW = TP / 128;                       // 32-pixel MFMA row tiles per wave
    static constexpr int TCO = BCO / 32;
    static constexpr int XR = (R - 1) * S + KH, XC = (WT - 1) * S + KW;
    static constexpr int SXB = BCI * 2 + 16;                   // bytes per staged input pixel (pad: rows fall on different banks)
    static constexpr int KMAX = KH * KW * BCI;                 // K of a chunk when every tap of the window is present
    static constexpr int KPAD = (KMAX + 15) / 16 * 16;
    static constexpr int WROW = KPAD * 2 + 16;                 // bytes per staged weight row
    static constexpr int OROW = BCO * 2 + 16;                  // bytes per staged output pixel
    static constexpr int OFF_W = XR * XC * SXB;
    static constexpr int OFF_O = OFF_W + BCO * WROW;
    static constexpr int OFF_TAP = OFF_O + TP * OROW;
    static constexpr int OFF_STAT = OFF_TAP + 64 * 4;
    static constexpr int SMEM = OFF_STAT + 4 * 2 * BCO * 4;     // one statistics row per wave (summed in a fixed order)
    static_assert(TP == 128 || TP == 256, "tile pixels");
    static_assert(WT % 32 == 0 || (WT == 16 && R % 2 == 0), "an MFMA row tile is 32 consecutive grid pixels of a tile row (or two 16-pixel rows)");
    static_assert(BCI == 8 || BCI % 16 == 0, "channel chunk");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int S, int BCI, int R, int WT, int KH, int KW, int BCO>
__global__ __launch_bounds__(256, 1) void conv_fprop_sx_kernel(const FxArgs a) {
    using C = FxCfg<S, BCI, R, WT, KH, KW, BCO>;
    constexpr int TP = C::TP, MPW = C::MPW, TCO = C::TCO, XR = C::XR, XC = C::XC, SXB = C::SXB, WROW = C::WROW, OROW = C::OROW;
    constexpr int XCH = XR * XC * (BCI / 8);                   // 16-byte pieces of an input tile
    constexpr int XPT = (XCH + 255) / 256;
    constexpr int WMAX = BCO * KH * KW * (BCI / 8);            // ... of a weight chunk (all taps of the window)
    constexpr int WPT = (WMAX + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sW = smem + C::OFF_W;
    unsigned char* sO = smem + C::OFF_O;
    int* sTap = reinterpret_cast<int*>(smem + C::OFF_TAP);
    float* sStat = reinterpret_cast<float*>(smem + C::OFF_STAT);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cot = blockIdx.x % a.NCOT;                       // consecutive blocks share the pixel range, differ in co tile
    const int pb = blockIdx.x / a.NCOT, npb = gridDim.x / a.NCOT;
    const int co0 = cot * BCO;
    const int ksteps = (a.ntaps * BCI + 15) / 16;
    const int wpieces = BCO * a.ntaps * (BCI / 8);

    if (tid < 64) sTap[tid] = tid < a.ntaps ? ((int)a.dh[tid] * XC + (int)a.dw[tid]) * SXB : 0;
    for (int i = tid; i < 4 * 2 * BCO; i += 256) sStat[i] = 0.f;
    // weight rows end in zeros up to KPAD (+ pad): the K steps past the last tap multiply them
    for (int i = tid; i < BCO * (WROW / 16); i += 256) *reinterpret_cast<uint4*>(sW + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    uint4 rx[XPT], rw[WPT];
    auto gload_x = [&](int it, int chunk) {
        const int n = it / (a.HB * a.WB);
        const int rem = it - n * (a.HB * a.WB);
        const int hb = rem / a.WB, wb = rem - hb * a.WB;
        const int hi0 = hb * R * S + a.dhmin, wi0 = wb * WT * S + a.dwmin;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int c = tid + u * 256;
            const int pix = c / (BCI / 8), cc = c - pix * (BCI / 8);
            const int xr = pix / XC, xc = pix - xr * XC;
            const int hi = hi0 + xr, wi = wi0 + xc;
            const int ch = chunk * BCI + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < XCH && ch < a.Ci && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
                v = *reinterpret_cast<const uint4*>(a.x + ((long long)(n * a.Hi + hi) * a.Wi + wi) * a.ld_x + ch);
            rx[u] = v;
        }
    };
    auto gload_w = [&](int chunk) {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int c = tid + u * 256;
            const int row = c / (a.ntaps * (BCI / 8)), r2 = c - row * (a.ntaps * (BCI / 8));
            const int t = r2 / (BCI / 8), cc = r2 - t * (BCI / 8);
            const int co = co0 + row, ch = chunk * BCI + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < wpieces && co < a.Co && ch < a.Ci)
                v = *reinterpret_cast<const uint4*>(a.w + ((long long)co * a.ntaps + t) * a.Ci + ch);
            rw[u] = v;
        }
    };
    auto lstore_x = [&]() {
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int c = tid + u * 256;
            if (c < XCH) {
                const int pix = c / (BCI / 8), cc = c - pix * (BCI / 8);
                *reinterpret_cast<uint4*>(sX + pix * SXB + cc * 16) = rx[u];
            }
        }
    };
    auto lstore_w = [&]() {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int c = tid + u * 256;
            if (c < wpieces) {
                const int row = c / (a.ntaps * (BCI / 8)), r2 = c - row * (a.ntaps * (BCI / 8));
                *reinterpret_cast<uint4*>(sW + row * WROW + r2 * 16) = rw[u];
            }
        }
    };

    // MFMA operands: lane = (row / column l & 31, K half h).  A row tile mi of this wave = grid pixels 32 (MPW wave + mi) ..+31 of
    // the tile in row-major order; its input pixel is (r S, c S) of the halo tile
    const int l31 = lane & 31, h = lane >> 5;
    int xpix[MPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int p = 32 * (MPW * wave + mi) + l31;
        const int r = p / WT, c = p - r * WT;
        xpix[mi] = (r * S * XC + c * S) * SXB;
    }
    float bia[TCO], s1[TCO], s2[TCO];
#pragma unroll
    for (int j = 0; j < TCO; ++j) {
        const int co = co0 + 32 * j + l31;
        bia[j] = (a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;
        s1[j] = s2[j] = 0.f;
    }

    // work items of this block: (tile it, chunk) in order, chunk innermost
    const int nwork = a.NCHUNK;
    int it = pb;
    bool have = it < a.IT;
    if (have) {
        gload_x(it, 0);
        gload_w(0);
    }
    f32x16_t acc[MPW][TCO];
    while (have) {
        for (int chunk = 0; chunk < nwork; ++chunk) {
            // the LDS tiles are free (barrier at the end of the previous item): stage this item, then request the next one
            lstore_x();
            if (nwork > 1 || it == pb) lstore_w();                      // a single chunk's weights stay resident
            __syncthreads();
            const bool last_chunk = chunk + 1 == nwork;
            const int nit = last_chunk ? it + npb : it, nchunk = last_chunk ? 0 : chunk + 1;
            const bool more = nit < a.IT;
            if (more) {
                gload_x(nit, nchunk);
                if (nwork > 1) gload_w(nchunk);
            }
            if (chunk == 0) {
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                    for (int j = 0; j < TCO; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[mi][j][e] = 0.f;
            }
            for (int s = 0; s < ksteps; ++s) {
                int toff, coff;
                if constexpr (BCI == 8) {
                    toff = sTap[2 * s + h];                              // two taps per instruction
                    coff = 0;
                } else {
                    const int k0 = 16 * s;
                    const int t = k0 / BCI;
                    toff = sTap[t];
                    coff = (k0 - t * BCI + 8 * h) * 2;
                }
                bf16x8_t fa[MPW], fb[TCO];
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(sX + xpix[mi] + toff + coff);
#pragma unroll
                for (int j = 0; j < TCO; ++j)
                    fb[j] = *reinterpret_cast<const bf16x8_t*>(sW + (32 * j + l31) * WROW + 32 * s + 16 * h);
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                    for (int j = 0; j < TCO; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[j], acc[mi][j], 0, 0, 0);
            }
            if (last_chunk) {
                // D[i = pixel][j = co]: lane holds channel l31 of pixels (e & 3) + 8 (e >> 2) + 4 h of its row tiles
                const int n = it / (a.HB * a.WB);
                const int rem = it - n * (a.HB * a.WB);
                const int hb = rem / a.WB, wb = rem - hb * a.WB;
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int p = 32 * (MPW * wave + mi) + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const int r = p / WT, c = p - r * WT;
                        const bool pok = hb * R + r < a.QH && wb * WT + c < a.QW;
#pragma unroll
                        for (int j = 0; j < TCO; ++j) {
                            const unsigned short bv = f32_to_bf16_bits(acc[mi][j][e] + bia[j]);
                            *reinterpret_cast<unsigned short*>(sO + p * OROW + (32 * j + l31) * 2) = bv;
                            const float fv = pok ? bf16_bits_to_f32(bv) : 0.f;
                            s1[j] += fv;
                            s2[j] += fv * fv;
                        }
                    }
                __syncthreads();
                for (int i = tid; i < TP * (BCO / 8); i += 256) {
                    const int p = i / (BCO / 8), cc = i - p * (BCO / 8);
                    const int r = p / WT, c = p - r * WT;
                    const int qh = hb * R + r, qw = wb * WT + c, ch = co0 + cc * 8;
                    if (qh < a.QH && qw < a.QW && ch < a.Co) {
                        const long long opix = ((long long)n * a.Ho + qh * a.out_step + a.oh0) * a.Wo + qw * a.out_step + a.ow0;
                        *reinterpret_cast<uint4*>(a.out + opix * a.ld_out + ch) = *reinterpret_cast<const uint4*>(sO + p * OROW + cc * 16);
                    }
                }
            }
            __syncthreads();                                            // everyone is done with the staged tiles
            if (last_chunk) it += npb;
            have = more;
        }
    }
    if (a.stats != nullptr) {
#pragma unroll
        for (int j = 0; j < TCO; ++j) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32), t2 = s2[j] + __shfl_xor(s2[j], 32);
            if (h == 0) {
                // one slot per (wave, column): plain stores, summed below in wave order -- LDS float atomics from the four waves
                // arrive in any order, and the rounding of that sum reached the layer's scale / shift (LinkNet34's stem: one
                // float ulp in a tenth of the channels, a handful of flipped bf16 roundings, two different gradients from the
                // same step: profiles/r06_ab.txt section 19)
                sStat[wave * 2 * BCO + 32 * j + l31] = t1;
                sStat[wave * 2 * BCO + BCO + 32 * j + l31] = t2;
            }
        }
        __syncthreads();
        if (tid < 2 * BCO) {
            const int which = tid / BCO, co = co0 + tid - which * BCO;
            const float sum = ((sStat[tid] + sStat[2 * BCO + tid]) + sStat[4 * BCO + tid]) + sStat[6 * BCO + tid];
            if (co < a.Co)
                atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + co], (double)sum);
        }
    }
}

template <int S, int BCI, int R, int WT, int KH, int KW, int BCO>
int launch_fx(FxArgs& a, hipStream_t stream) {
    using C = FxCfg<S, BCI, R, WT, KH, KW, BCO>;
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_sx_kernel<S, BCI, R, WT, KH, KW, BCO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) segnb_set_error("fprop_sx hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.QH + R - 1) / R;
    a.WB = (a.QW + WT - 1) / WT;
    a.IT = a.N * a.HB * a.WB;
    a.NCHUNK = (a.Ci + BCI - 1) / BCI;
    a.NCOT = (a.Co + BCO - 1) / BCO;
    // persistent blocks per co tile: two blocks per CU where two tiles fit the LDS (one's loads under the other's MFMAs)
    constexpr int per_cu = C::SMEM <= 80 * 1024 ? 2 : 1;
    int pbn = segnb_knob_conv_cus() * per_cu / a.NCOT;
    if (pbn < 1) pbn = 1;
    if (pbn > a.IT) pbn = a.IT;
    hipLaunchKernelGGL((conv_fprop_sx_kernel<S, BCI, R, WT, KH, KW, BCO>), dim3(pbn * a.NCOT), dim3(256), C::SMEM, stream, a);
    return 0;
}

}  // namespace

// 1 = launched, 0 = geometry not served (the caller falls through to the general gather kernel), else an error
int segnb_fprop_sx_try(const segnb_conv_geom* g, const void* in, const void* wpacked, const float* bias, int bias_n, void* out,
                       double* stats, hipStream_t stream) {
    if ((g->in_step != 1 && g->in_step != 2) || g->QW < 24 || g->ntaps > 49) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < g->ntaps; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    const int kh = dhmax - dhmin + 1, kw = dwmax - dwmin + 1;
    if (g->ntaps > kh * kw) return 0;                               // (a repeated tap: not a window)
    // Served: the 8-channel stride-2 stem (linknet.py:16: 183 -> 148 us at 512x512 bs=16).  The template also instantiates for
    // 3x3 / 1x1 stride 2, the 2x2-window phases of transposed convolutions, 2x2 and 1x1 stride 1 and 16-channel-multiple 3x3
    // (all parity-tested once), but there one (tile, chunk) item is 8-40 MFMAs behind three barriers and a staged store, and
    // the general gather kernel with five resident blocks per CU was 10-120 us faster per launch (profiles/r03_ab.txt):
    // LinkNet34 1955 -> 1869 images/s with all of them on.  They are not dispatched.
    if (g->in_step != 2 || kh > 7 || kw > 7 || g->Ci != 8) return 0;
    FxArgs a;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.bias = bias_n > 0 ? bias : nullptr;
    a.bias_n = bias_n;
    a.out = (bf16_t*)out;
    a.stats = stats;
    a.N = g->N; a.QH = g->QH; a.QW = g->QW; a.Hi = g->Hi; a.Wi = g->Wi; a.Ho = g->Ho; a.Wo = g->Wo;
    a.out_step = g->out_step; a.oh0 = g->oh0; a.ow0 = g->ow0;
    a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
    a.dhmin = dhmin; a.dwmin = dwmin; a.ntaps = g->ntaps;
    for (int t = 0; t < g->ntaps; ++t) {
        a.dh[t] = (signed char)(g->dh[t] - dhmin);
        a.dw[t] = (signed char)(g->dw[t] - dwmin);
    }
    const int rc = launch_fx<2, 8, 8, 32, 7, 7, 64>(a, stream);
    return rc ? rc : 1;
}
