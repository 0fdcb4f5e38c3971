// Stride-1 3x3 convolution forward (bf16) for an 8-channel input -- the FIRST layer of every model here: 3 image
// channels padded to 8 (lib/models/zf_unet.py:37 conv_224.l1, 3 -> 32).  The general gather kernel ran it at 35 TFLOP/s
// (80 us at bs=32, 224x224) because K = 72 is less than two of its 64-deep K steps; the layer is HBM-bound (26 MB in,
// 103 MB out), so this kernel is about bytes:
//   * a pixel is ONE 16-byte vector (8 bf16 channels): the 18 x 18 halo tile of a 16 x 16 pixel tile is 324 coalesced
//     16-byte loads, staged in LDS pixel-major (consecutive pixels = consecutive 16-byte slots: conflict-free reads);
//   * K is ordered (tap, channel): one MFMA K step = two taps, and the A fragment of lane (r, h) for step s is exactly the
//     16 bytes of pixel (row r shifted by tap 2s + h) -- one ds_read_b128, no gather; five steps cover the nine taps, the
//     tenth half-step meets zero weights;
//   * the 32 x 80 weight matrix lives in registers (five B fragments per lane) for the whole kernel;
//   * persistent blocks, one tile per iteration, next tile's halo loads issued before the current tile's MFMAs;
//     transposed accumulators, LDS-staged 16-byte stores, BatchNorm statistics by the store threads (as fprop_dma.hip).
#include "fprop_dma.h"

namespace {

// U8 input: the network input as the dataset holds it -- uint8 HWC (lib/common.py:44 cv2.imread) -- normalised in
// registers on the way in: v = (u8 * scale - mean[c]) / std[c]  (NormalizeImage, lib/augmentations.py:452-460), rounded to
// bf16 exactly as segnb_pack_input_u8 does.  3 bytes per pixel from HBM instead of 12 (fp32 CHW) + 16 + 16 (pack, re-read).
struct C8Norm {
    const unsigned char* img;     // [N][Hi][Wi][C] uint8
    bf16_t* packed_out;           // optional: the normalised 8-channel bf16 pixels, [N][Hi][Wi][ld_p] (the weight gradient's x)
    int C, ld_p;
    float scale, mean[8], inv_std[8];
};

struct C8Args {
    C8Norm u8;
    const float* ep_coef;      // affine + activation epilogue (segnb_conv_fprop_act), as FdArgs
    int ep_act;
    float ep_slope;
    const bf16_t* x;
    const bf16_t* w;       // [Co][9][8]
    const float* bias;
    int bias_n;
    bf16_t* out;
    double* stats;
    int N, H, W, Hi, Wi, Co, ld_x, ld_out;
    int dhmin, dwmin;
    int dh[9], dw[9];
    int HB, WB, IT;
};

constexpr int C8_R = 16, C8_WT = 16, C8_XC = 18, C8_NPIX = 18 * 18, C8_BM = 256, C8_BN = 32;
constexpr int C8_OUT_ROW = C8_BN * 2 + 16;
constexpr int C8_OFF_STG = ((C8_NPIX * 16 + 1023) / 1024) * 1024;          // halo tile: 5184 B
constexpr int C8_OFF_PIX = C8_OFF_STG + C8_BM * C8_OUT_ROW;
constexpr int C8_SMEM_TILE = C8_OFF_PIX + C8_BM * 4;                        // 27 KiB
constexpr int C8_SMEM = C8_SMEM_TILE > 256 * 16 * 8 ? C8_SMEM_TILE : 256 * 16 * 8;   // >= the statistics reduction scratch

__device__ __forceinline__ uint4 c8_norm_pixel(const C8Norm& u, long long pix) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const unsigned char* p = u.img + pix * u.C;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e < u.C) v[e] = ((float)p[e] * u.scale - u.mean[e]) * u.inv_std[e];
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    return o;
}

template <bool U8, bool EP = false>      // EP: affine + activation epilogue, a separate instantiation (see conv_fprop_ws_kernel)
__global__ __launch_bounds__(256) void conv_fprop_c8_kernel(const C8Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[C8_SMEM];
    int* sPix = reinterpret_cast<int*>(smem + C8_OFF_PIX);
    unsigned char* sOut = smem + C8_OFF_STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // weights: B fragment of K step s for lane (r, h) = channel r, tap 2s + h, 8 input channels (zeros past tap 8 / Co)
    bf16x8_t wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + h;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tap < 9 && r < a.Co) v = *reinterpret_cast<const uint4*>(a.w + ((long long)r * 9 + tap) * 8);
        wf[s] = __builtin_bit_cast(bf16x8_t, v);
    }
    float4 bias4[4], scale4[4];         // transposed accumulators: lane holds channels 8g + 4h .. +3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float b[4], sc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 8 * g + 4 * h + e;
            b[e] = (a.bias != nullptr && c < a.bias_n) ? a.bias[c] : 0.f;
            sc[e] = 1.f;
            if constexpr (EP) {
                if (a.ep_coef != nullptr && c < a.Co) {
                    sc[e] = a.ep_coef[c];
                    b[e] = (b[e] - a.ep_coef[2 * a.Co + c]) * sc[e] + a.ep_coef[a.Co + c];
                }
            }
        }
        bias4[g] = make_float4(b[0], b[1], b[2], b[3]);
        scale4[g] = make_float4(sc[0], sc[1], sc[2], sc[3]);
    }
    const float ep_neg = a.ep_act == SEGNB_ACT_RELU ? 0.f : (a.ep_act == SEGNB_ACT_LEAKY ? a.ep_slope : 1.f);
    auto ep = [&](float acc, float sc, float sh) {
        if constexpr (EP) {
            const float v = acc * sc + sh;
            return v < 0.f ? v * ep_neg + 0.f : v;
        } else {
            return acc + sh;
        }
    };
    // A fragment LDS offsets: wave w owns tile rows 64w .. 64w+63 (two MFMA row tiles)
    int a_off[5][2];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + h < 9 ? 2 * s + h : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = wave * 64 + 32 * i + r;
            a_off[s][i] = ((m / C8_WT + a.dh[tap]) * C8_XC + (m % C8_WT) + a.dw[tap]) * 16;
        }
    }
    // store pass: thread = (row group, 8-channel chunk); 4 chunks per pixel
    const int cc = tid & 3, row0 = tid >> 2;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;

    auto halo_load = [&](int it, uint4 (&hv)[2]) {
        const int n = it / (a.HB * a.WB);
        const int rem = it - n * (a.HB * a.WB);
        const int hb = rem / a.WB, wb = rem - hb * a.WB;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = tid + u * 256;
            const int xr = pix / C8_XC, xc = pix - xr * C8_XC;
            const int hi = hb * C8_R + a.dhmin + xr, wi = wb * C8_WT + a.dwmin + xc;
            hv[u] = make_uint4(0, 0, 0, 0);
            if (pix < C8_NPIX && it < a.IT && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi) {
                const long long ipix = (long long)(n * a.Hi + hi) * a.Wi + wi;
                if constexpr (U8) {
                    hv[u] = c8_norm_pixel(a.u8, ipix);
                    // the tile that OWNS the pixel (not a halo copy of it) also publishes the packed form
                    if (a.u8.packed_out != nullptr && (unsigned)(xr + a.dhmin) < (unsigned)C8_R &&
                        (unsigned)(xc + a.dwmin) < (unsigned)C8_WT)
                        *reinterpret_cast<uint4*>(a.u8.packed_out + ipix * a.u8.ld_p) = hv[u];
                } else {
                    hv[u] = *reinterpret_cast<const uint4*>(a.x + ipix * a.ld_x);
                }
            }
        }
    };

    uint4 hv[2];
    halo_load(blockIdx.x, hv);
    for (int it = blockIdx.x; it < a.IT; it += gridDim.x) {
        __syncthreads();                                  // previous tile's halo / staging consumed
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + u * 256 < C8_NPIX) *reinterpret_cast<uint4*>(smem + (tid + u * 256) * 16) = hv[u];
        {
            const int n = it / (a.HB * a.WB);
            const int rem = it - n * (a.HB * a.WB);
            const int hb = rem / a.WB, wb = rem - hb * a.WB;
            const int ho = hb * C8_R + tid / C8_WT, wo = wb * C8_WT + tid % C8_WT;
            sPix[tid] = (ho < a.H && wo < a.W) ? (n * a.H + ho) * a.W + wo : -1;
        }
        halo_load(it + gridDim.x, hv);                    // next tile: in flight under this tile's work
        __syncthreads();
        f32x16_t acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(smem + a_off[s][i]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], af, acc[i], 0, 0, 0);
            }
        // stage: lane holds pixel (64w + 32i + r), channels 8g + 4h .. +3
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 64 + 32 * i + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack2bf(ep(acc[i][4 * g + 0], scale4[g].x, bias4[g].x), ep(acc[i][4 * g + 1], scale4[g].y, bias4[g].y));
                pk.y = pack2bf(ep(acc[i][4 * g + 2], scale4[g].z, bias4[g].z), ep(acc[i][4 * g + 3], scale4[g].w, bias4[g].w));
                *reinterpret_cast<uint2*>(sOut + row * C8_OUT_ROW + (8 * g + 4 * h) * 2) = pk;
            }
        }
        __syncthreads();
        // coalesced stores: 4 threads per pixel, 64 pixels per pass
        int opix[4];
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) opix[k] = sPix[row0 + 64 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(sOut + (row0 + 64 * k) * C8_OUT_ROW + cc * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = opix[k] >= 0 && cc * 8 < a.Co;
            if (ok) *reinterpret_cast<uint4*>(a.out + (long long)opix[k] * a.ld_out + cc * 8) = v[k];
            if (a.stats != nullptr) {
                const float m = ok ? 1.f : 0.f;
                float f[8];
                f[0] = __uint_as_float(v[k].x << 16); f[1] = __uint_as_float(v[k].x & 0xffff0000u);
                f[2] = __uint_as_float(v[k].y << 16); f[3] = __uint_as_float(v[k].y & 0xffff0000u);
                f[4] = __uint_as_float(v[k].z << 16); f[5] = __uint_as_float(v[k].z & 0xffff0000u);
                f[6] = __uint_as_float(v[k].w << 16); f[7] = __uint_as_float(v[k].w & 0xffff0000u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float fm = f[e] * m;
                    s1[e] += fm;
                    s2[e] += fm * fm;
                }
            }
        }
    }
    // statistics: fixed-order block reduction, one fp64 atomic per channel and block
    if (a.stats != nullptr) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);        // [256][16] = 32 KiB
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid * 16 + e] = (double)s1[e];
            red[tid * 16 + 8 + e] = (double)s2[e];
        }
        __syncthreads();
        if (tid < 2 * C8_BN) {
            const int which = tid / C8_BN, col = tid - which * C8_BN;
            const int c8 = col >> 3, e = col & 7;
            double sum = 0.0;
            for (int k = 0; k < 64; ++k) sum += red[(k * 4 + c8) * 16 + which * 8 + e];
            if (col < a.Co)
                atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + col], sum);
        }
    }
}
static_assert(256 * 16 * 8 <= C8_SMEM, "statistics reduction scratch");

}  // namespace

static int c8_launch(const segnb_conv_geom* g, const void* in, const C8Norm* u8, const void* wpacked, const float* bias,
                     int bias_n, void* out, double* stats, hipStream_t stream, const segnb_act_epilogue* ep = nullptr);

// 1 = handled, 0 = not applicable, else error
int segnb_fprop_c8_try(const segnb_conv_geom* g, const void* in, const void* wpacked, const float* bias, int bias_n,
                       void* out, double* stats, hipStream_t stream, const segnb_act_epilogue* ep) {
    if (!segnb_knob_fprop_dma()) return 0;
    return c8_launch(g, in, nullptr, wpacked, bias, bias_n, out, stats, stream, ep);
}

static int c8_launch(const segnb_conv_geom* g, const void* in, const C8Norm* u8, const void* wpacked, const float* bias,
                     int bias_n, void* out, double* stats, hipStream_t stream, const segnb_act_epilogue* ep) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->Co > 32 && g->Co <= 64 && g->Co % 8 == 0 && g->Ci == 8 && stats == nullptr && u8 == nullptr &&
        (ep == nullptr || ep->coef == nullptr)) {
        // 33..64 output channels (unet16.py:73: VGG's 3 -> 64 at 1024 x 1024) as two launches over channel halves: the input is
        // 16 bytes per pixel, the output 128 -- each launch re-reads the former and writes its half of the latter (285 us on the
        // general kernel against 2 x ~75).  Without per-channel side tables only (statistics and a folded BatchNorm index by Co)
        segnb_conv_geom h = *g;
        h.Co = 32;
        const int rc = c8_launch(&h, in, nullptr, wpacked, bias, bias_n < 32 ? bias_n : 32, out, nullptr, stream, ep);
        if (rc != 1) return rc;
        h.Co = g->Co - 32;
        return c8_launch(&h, in, nullptr, (const bf16_t*)wpacked + 32 * 9 * 8, bias != nullptr && bias_n > 32 ? bias + 32 : nullptr,
                         bias_n > 32 ? bias_n - 32 : 0, (bf16_t*)out + 32, nullptr, stream, ep);
    }
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci != 8 || g->Co > 32 || g->Wo < 12) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return 0;
    C8Args a;
    a.ep_act = ep != nullptr ? ep->act : -1;
    a.ep_coef = ep != nullptr ? ep->coef : nullptr;
    a.ep_slope = ep != nullptr ? ep->slope : 0.f;
    if (u8 != nullptr) a.u8 = *u8;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.bias = bias;
    a.bias_n = bias_n;
    a.out = (bf16_t*)out;
    a.stats = stats;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Hi = g->Hi; a.Wi = g->Wi;
    a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
    a.dhmin = dhmin; a.dwmin = dwmin;
    for (int t = 0; t < 9; ++t) {
        a.dh[t] = g->dh[t] - dhmin;
        a.dw[t] = g->dw[t] - dwmin;
    }
    a.HB = (a.H + C8_R - 1) / C8_R;
    a.WB = (a.W + C8_WT - 1) / C8_WT;
    a.IT = a.N * a.HB * a.WB;
    int grid = segnb_num_cus() * 4;          // 32 KiB of LDS per block: four blocks per CU hide each other's latencies
    if (grid > a.IT) grid = a.IT;
    if (u8 != nullptr && a.ep_act >= 0)
        hipLaunchKernelGGL((conv_fprop_c8_kernel<true, true>), dim3(grid), dim3(256), 0, stream, a);
    else if (u8 != nullptr)
        hipLaunchKernelGGL((conv_fprop_c8_kernel<true, false>), dim3(grid), dim3(256), 0, stream, a);
    else if (a.ep_act >= 0)
        hipLaunchKernelGGL((conv_fprop_c8_kernel<false, true>), dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_fprop_c8_kernel<false, false>), dim3(grid), dim3(256), 0, stream, a);
    return 1;
}

// ---- uint8 HWC network input ----------------------------------------------------------------------
namespace {
template <typename T>
__global__ void pack_input_u8_kernel(const C8Norm u, long long npix, T* __restrict__ out, int Cp, int ld) {
    const int CPP = Cp / 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix * CPP;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / CPP;
        const int cc = (int)(i - pix * CPP);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cc * 8 + e;
            v[e] = 0.f;
            if (ch < u.C) v[e] = ((float)u.img[pix * u.C + ch] * u.scale - u.mean[ch]) * u.inv_std[ch];
        }
        store8(out + pix * ld + cc * 8, v);
    }
}

int fill_norm(C8Norm& u, const unsigned char* img, int C, float scale, const float* mean, const float* stdv) {
    if (img == nullptr || mean == nullptr || stdv == nullptr || C < 1 || C > 8) return 1;
    u.img = img;
    u.packed_out = nullptr;
    u.C = C;
    u.ld_p = 0;
    u.scale = scale;
    for (int e = 0; e < 8; ++e) {
        u.mean[e] = e < C ? mean[e] : 0.f;
        u.inv_std[e] = e < C ? 1.0f / stdv[e] : 0.f;
        if (e < C && !(stdv[e] != 0.f)) return 1;
    }
    return 0;
}
}  // namespace

extern "C" int segnb_pack_input_u8(const unsigned char* img, int N, int H, int W, int C, float scale, const float* mean,
                                   const float* stdv, void* out, int dtype, int Cp, int ld_out, segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_pack_input_u8 takes host mean / std arrays");
    C8Norm u;
    SEGNB_CHECK_ARG(fill_norm(u, img, C, scale, mean, stdv) == 0, "bad normalisation (1 <= C <= 8, std != 0)");
    SEGNB_CHECK_ARG(out && N > 0 && H > 0 && W > 0 && Cp % 8 == 0 && Cp >= C && ld_out >= Cp && ld_out % 8 == 0, "bad shape");
    const long long npix = (long long)N * H * W;
    int grid = ceil_div(npix * (Cp / 8), 256);
    if (grid > 8192) grid = 8192;
    if (dtype == SEGNB_BF16)
        hipLaunchKernelGGL(pack_input_u8_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, u, npix,
                           (bf16_t*)out, Cp, ld_out);
    else if (dtype == SEGNB_F32)
        hipLaunchKernelGGL(pack_input_u8_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, u, npix, (float*)out,
                           Cp, ld_out);
    else {
        segnb_set_error("segnb_pack_input_u8: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_conv_fprop_u8_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || !segnb_knob_fprop_dma()) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci != 8 || g->Co > 32 || g->Wo < 12 || g->Hi != g->Ho || g->Wi != g->Wo) return 0;
    for (int t = 0; t < 9; ++t)
        if (g->dh[t] < -1 || g->dh[t] > 1 || g->dw[t] < -1 || g->dw[t] > 1) return 0;
    return 1;
}

extern "C" int segnb_conv_fprop_u8(const segnb_conv_geom* g, const unsigned char* img, int C, float scale, const float* mean,
                                   const float* stdv, const void* wpacked, const float* bias, int bias_n, void* out,
                                   double* stats, void* x_packed, int ld_packed, segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_conv_fprop_u8 takes host mean / std arrays");
    SEGNB_CHECK_ARG(g && wpacked && out, "NULL argument");
    SEGNB_CHECK_ARG(segnb_conv_fprop_u8_ok(g, SEGNB_BF16), "geometry not served by the uint8 first-layer kernel (segnb_conv_fprop_u8_ok)");
    C8Norm u;
    SEGNB_CHECK_ARG(fill_norm(u, img, C, scale, mean, stdv) == 0, "bad normalisation (1 <= C <= 8, std != 0)");
    SEGNB_CHECK_ARG(x_packed == nullptr || (ld_packed >= 8 && ld_packed % 8 == 0), "bad packed-input stride");
    u.packed_out = (bf16_t*)x_packed;
    u.ld_p = ld_packed;
    const int rc = c8_launch(g, nullptr, &u, wpacked, bias, bias_n, out, stats, (hipStream_t)stream);
    if (rc != 1) {
        segnb_set_error("segnb_conv_fprop_u8: launch refused (%d)", rc);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}
