// Stride-1 3x3 convolution with a THIN input and a WIDE output (bf16): the data gradient of a dense layer of FCDenseNet
// (lib/models/tiramisu.py:9-20: norm -> relu -> conv(C -> growth 16); its data gradient reads 16 channels and writes the C <= ~1100
// channels of the concat prefix).  K = 9 taps x 16 channels = 144: the launch is the bytes of its OUTPUT (and of the y tensor its
// BatchNorm-backward epilogue reads), which the general gather kernel writes at 1.8 TB/s (profiles/r05_ab.txt: its block tiles walk
// K through LDS with two barriers per 64-deep step and stage every output tile through LDS again).
//
// Here a block owns 256 consecutive pixels and ALL output channels:
//   * the pixels' MFMA operands -- 5 K steps of 32 = (two taps) x (16 channels), the tenth tap zero -- are loaded ONCE per pixel
//     tile straight from global memory into registers (a lane's 16 bytes of a K step are 8 channels of one tap's pixel: no
//     im2col staging; out-of-image taps are out-of-bounds buffer loads = zeros) and stay there: 80 registers;
//   * the block then walks the 64-channel tiles of the output: the tile's weights [64][160] come through a double-buffered LDS
//     image (fetched into registers during the previous tile's MFMAs, one barrier per tile), 80 MFMAs 16x16x32 per wave and
//     tile with A = weights, B = pixels, so a lane ends up with 4 consecutive channels of one pixel;
//   * v_permlane16_swap merges two 16-channel fragments: every store is 16 bytes of one pixel, a wave's store instruction covers
//     16 pixels x 64 contiguous bytes;
//   * BatchNorm-backward reduction of the producing layer in the same pass (segnb_conv_fprop_bnreduce: dz = round(g * act'(z)),
//     sum dz, sum dz * yhat): the y values of the tile's stores are requested before its MFMAs; the sums of a wave's 64 pixels
//     are reduced in a fixed shuffle order, the four waves' partials added in wave order by the channel's owning thread into an
//     fp64 accumulator in LDS that lives across the block's pixel tiles, one fp64 atomic per channel and block at the end.
#include "fprop_dma.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

struct ThinArgs {
    const bf16_t* x;
    const bf16_t* w;          // [Co][9][Ci]
    bf16_t* out;
    unsigned x_bytes, w_bytes, out_bytes;
    int N, H, W, Ci, Co, ld_x, ld_out;
    int dh[9], dw[9];
    int M, NPT, NCT;
    int NCS, CTS;             // channel splits: block b owns pixel tiles b / NCS, b / NCS + gridDim / NCS, ... and channel tiles
                              // [cs * CTS, min(NCT, (cs + 1) * CTS)) with cs = b % NCS (few pixels, many channels: the 8 x 8 level)
    // BatchNorm-backward reduction epilogue (bn_y == NULL: off)
    const bf16_t* bn_y;
    unsigned bn_y_bytes;
    int bn_ld;
    const float* bn_coef;     // [4][Co]: scale, shift, mean, invstd
    double* bn_sums;          // [REPL][2][Co]
    int bn_act;
    float bn_slope;
};

constexpr int TH_ROW = 336;               // LDS bytes per weight row: 160 K values + 16 bytes pad (conflict-free 16-lane reads)
constexpr int TH_MAXC = 1280;             // channels the block-level fp64 accumulators cover

// NJ: 16-channel fragments per channel tile (4: 64-channel tiles; 2: 32 -- the BatchNorm epilogue's y values and sums need the
// registers of the other half of the accumulators)
template <bool BNR, int NJ>
__global__ __launch_bounds__(256, 2) void conv_thin_kernel(const ThinArgs a) {
    constexpr int TC = 16 * NJ, NJP = NJ / 2, WCH = (TC * 20 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char sW[2][TC * TH_ROW];
    __shared__ float sPart[BNR ? 4 * 2 * 64 : 1];           // [wave][which][channel of the tile]
    __shared__ double sAcc[BNR ? 2 * TH_MAXC : 1];          // [which][channel]: owned by thread (which, channel % 64)
    __shared__ float sCoef[BNR ? 2 * 3 * 64 : 1];           // [tile parity]: mean, scale, shift of the tile's channels

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int cidx = ((g & 1) << 1) | (g >> 1);      // after the swap a lane owns the 16-byte chunk cidx of every 32-channel pair
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(BNR ? a.bn_y : a.x), 0, BNR ? (int)a.bn_y_bytes : 0, 0x00020000);

    if constexpr (BNR) {
        for (int c = tid; c < 2 * TH_MAXC; c += 256) sAcc[c] = 0.0;
    }
    const float bneg = !BNR ? 0.f : (a.bn_act == SEGNB_ACT_RELU ? 0.f : (a.bn_act == SEGNB_ACT_LEAKY ? a.bn_slope : 1.f));
    const bool round_dz = bneg != 0.f && bneg != 1.f;

    // weight image of channel tile ct: thread t copies 16-byte chunks q = t, t + 256, ... of the [64][20] chunk grid; chunk (row, c):
    // tap c / 2, channel half c % 2 -- zero beyond the ninth tap, the layer's input channels or its output channels
    u32x4_t wq[WCH];
    auto w_fetch = [&](int ct) {
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int q = tid + u * 256, row = q / 20, c = q - row * 20;
            const int tap = c >> 1, half = c & 1, co = ct * TC + row;
            const bool ok = row < TC && tap < 9 && half * 8 < a.Ci && co < a.Co;
            const unsigned off = ok ? (unsigned)(((co * 9 + tap) * a.Ci + half * 8) * 2) : OOB;
            wq[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0);
        }
    };
    auto w_publish = [&](int buf) {
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int q = tid + u * 256, row = q / 20, c = q - row * 20;
            if (row < TC) *reinterpret_cast<u32x4_t*>(&sW[buf][row * TH_ROW + c * 16]) = wq[u];
        }
    };
    // BatchNorm constants of channel tile ct, published with its weight image (same barrier)
    auto coef_publish = [&](int ct) {
        if constexpr (BNR) {
            if (tid < TC) {
                const int ch = ct * TC + tid;
                const bool in = ch < a.Co;
                float* d = sCoef + (ct & 1) * 192;
                d[tid] = in ? a.bn_coef[2 * a.Co + ch] : 0.f;
                d[64 + tid] = in ? a.bn_coef[ch] : 0.f;
                d[128 + tid] = in ? a.bn_coef[a.Co + ch] : 0.f;
            }
        }
    };

    const int HW = a.H * a.W;
    const int cs = blockIdx.x % a.NCS, ct0 = cs * a.CTS, ct1 = min(a.NCT, ct0 + a.CTS);
    for (int pt = blockIdx.x / a.NCS; pt < a.NPT; pt += gridDim.x / a.NCS) {
        // ---- the pixels of this tile: four 16-pixel fragments per wave, 5 K steps each, resident for all channel tiles
        bf16x8_t fx[4][5];
        int opix[4];                       // output pixel of fragment i for this lane's column (-1: past the end)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = pt * 256 + wave * 64 + 16 * i + r16;
            const bool live = pix < a.M;
            const int pc = live ? pix : 0;
            const int n = pc / HW, rem = pc - n * HW, h = rem / a.W, w0 = rem - h * a.W;
            opix[i] = live ? pix : -1;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int tap = 2 * s + (g >> 1), half = g & 1;
                const int tc = tap < 9 ? tap : 0;
                const int hh = h + a.dh[tc], ww = w0 + a.dw[tc];
                const bool ok = live && tap < 9 && half * 8 < a.Ci && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)((((n * a.H + hh) * a.W + ww) * a.ld_x + half * 8) * 2) : OOB;
                fx[i][s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
            }
        }
        w_fetch(ct0);
        __syncthreads();                   // (the previous pixel tile's last weight image is consumed)
        w_publish(ct0 & 1);
        coef_publish(ct0);
        __syncthreads();
        for (int ct = ct0; ct < ct1; ++ct) {
            const int buf = ct & 1;
            if (ct + 1 < ct1) w_fetch(ct + 1);
            // ---- y values of this tile's outputs (BatchNorm reduction): this lane stores, for fragment pair jp and pixel fragment
            // i, the 16 bytes of channels ct * 64 + 32 * jp + 8 * cidx .. + 7 (see the store below)
            u32x4_t yv[BNR ? 4 : 1][BNR ? NJP : 1];
            const float* const cf = sCoef + (BNR ? (ct & 1) * 192 : 0);
            if constexpr (BNR) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jp = 0; jp < NJP; ++jp) {
                        const int ch = ct * TC + 32 * jp + 8 * cidx;
                        const bool ok = opix[i] >= 0 && ch < a.Co;
                        const unsigned off = ok ? (unsigned)((opix[i] * a.bn_ld + ch) * 2) : OOB;
                        yv[i][jp] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)off, 0, 0);
                    }
            }
            f32x4_t acc[4][NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const bf16x8_t wf =
                        *reinterpret_cast<const bf16x8_t*>(&sW[buf][(16 * j + r16) * TH_ROW + (32 * s + 8 * g) * 2]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, fx[i][s], acc[i][j], 0, 0, 0);
                }
            // ---- stores: fragments (2 jp, 2 jp + 1) of pixel fragment i merged into 16-byte rows.  Before the swap this lane holds
            // channels 16 j + 4 g .. + 3 of pixel r16; after it, lane group g holds 8 consecutive channels:
            // 32 jp + 8 cidx .. + 7  (the roll kernel's merge, fprop_roll.hip)
#pragma unroll
            for (int jp = 0; jp < NJP; ++jp) {
                float s1[8], s2[8];
                if constexpr (BNR) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4_t A0 = acc[i][2 * jp], A1 = acc[i][2 * jp + 1];
                    const unsigned x0 = pack2bf(A0[0], A0[1]), x1 = pack2bf(A0[2], A0[3]);
                    const unsigned y0 = pack2bf(A1[0], A1[1]), y1 = pack2bf(A1[2], A1[3]);
                    auto r0s = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    auto r1s = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    u32x4_t v;
                    v.x = r0s[0]; v.y = r1s[0]; v.z = r0s[1]; v.w = r1s[1];
                    const int ch = ct * TC + 32 * jp + 8 * cidx;
                    const bool ok = opix[i] >= 0 && ch < a.Co;
                    const unsigned off = ok ? (unsigned)((opix[i] * a.ld_out + ch) * 2) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (int)off, 0, 0);
                    if constexpr (BNR) {
                        // dz = round(g * act'(z)), z = (y - mean) * scale + shift (bn_act_bwd_reduce_kernel's arithmetic on the
                        // rounded gradient this launch stores); the second sum over dz * (y - mean), scaled by invstd at the flush
                        // (pixels past the end and channels past Co carry exact zeros: zero operands, zero weights -- no mask needed;
                        // under ReLU / identity dz is g or 0: nothing to round)
                        const u32x4_t yq = yv[i][jp];
                        const unsigned gw[4] = {v.x, v.y, v.z, v.w}, yw[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gv = __uint_as_float((e & 1) ? (gw[e >> 1] & 0xffff0000u) : (gw[e >> 1] << 16));
                            const float yf = __uint_as_float((e & 1) ? (yw[e >> 1] & 0xffff0000u) : (yw[e >> 1] << 16));
                            const int cl = 32 * jp + 8 * cidx + e;
                            const float yc = yf - cf[cl];
                            const float z = yc * cf[64 + cl] + cf[128 + cl];
                            float dv = z > 0.f ? gv : gv * bneg;
                            if (round_dz) dv = bf16_bits_to_f32(f32_to_bf16_bits(dv));
                            s1[e] += dv;
                            s2[e] += dv * yc;
                        }
                    }
                }
                if constexpr (BNR) {
                    // the wave's 64 pixels: the fragments i were added in the lane above; the 16 pixel lanes of a group in a fixed
                    // shuffle order; lane r16 == 0 of group g then holds the sums of channels 32 jp + 8 cidx + e
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v1 = s1[e], v2 = s2[e];
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) {
                            v1 += __shfl_xor(v1, o);
                            v2 += __shfl_xor(v2, o);
                        }
                        if (r16 == 0) {
                            sPart[(wave * 2 + 0) * 64 + 32 * jp + 8 * cidx + e] = v1;
                            sPart[(wave * 2 + 1) * 64 + 32 * jp + 8 * cidx + e] = v2;
                        }
                    }
                }
            }
            if (ct + 1 < ct1) {
                w_publish(buf ^ 1);
                coef_publish(ct + 1);
            }
            __syncthreads();               // next weight image published; this tile's partial sums visible
            if constexpr (BNR) {
                if (tid < 128) {
                    const int which = tid >> 6, cl = tid & 63, ch = ct * TC + cl;
                    if (cl < TC && ch < a.Co && ch < TH_MAXC) {
                        const float t = ((sPart[(0 * 2 + which) * 64 + cl] + sPart[(1 * 2 + which) * 64 + cl]) +
                                         sPart[(2 * 2 + which) * 64 + cl]) + sPart[(3 * 2 + which) * 64 + cl];
                        sAcc[which * TH_MAXC + ch] += (double)t;
                    }
                }
                __syncthreads();           // (sPart free for the next tile)
            }
        }
    }
    if constexpr (BNR) {
        __syncthreads();
        const int c_lo = ct0 * TC, c_n = min(a.Co, ct1 * TC) - c_lo;
        for (int c = tid; c < 2 * c_n; c += 256) {
            const int which = c / c_n, ch = c_lo + c - which * c_n;
            double v = sAcc[which * TH_MAXC + ch];
            if (which == 1) v *= (double)a.bn_coef[3 * a.Co + ch];       // * invstd: sum dz * yhat
            atomicAdd(&a.bn_sums[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + ch], v);
        }
    }
}

bool thin_geometry(const segnb_conv_geom* g) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Hi != g->Ho || g->Wi != g->Wo) return false;
    if ((g->Ci != 8 && g->Ci != 16) || g->Co % 8 != 0 || g->Co < 48 || g->Co > TH_MAXC) return false;
    // (measured, tools/dense_dgrad_bench.py: from 8 x 32 x 32 pixels up this kernel wins -- 15.6 vs 18.4 us with the reduction at 656
    // channels -- below, the general kernel's channel-parallel tiles do: 9.9 vs 12.8 us at 8 x 16 x 16 x 896)
    if ((long long)g->N * g->Ho * g->Wo < 4096) return false;
    if (g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return false;
    const long long ob = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
    const long long ib = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    return ob < (1ll << 31) && ib < (1ll << 31);
}

}  // namespace

// 1 = launched, 0 = not served
int segnb_fprop_thin_try(const segnb_conv_geom* g, const void* in, const void* wpacked, void* out, hipStream_t stream,
                         const segnb_bn_reduce_epilogue* bn) {
    if (!segnb_knob_fprop_thin() || !thin_geometry(g)) return 0;
    ThinArgs a;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.out = (bf16_t*)out;
    a.x_bytes = (unsigned)((((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2);
    a.w_bytes = (unsigned)((long long)g->Co * 9 * g->Ci * 2);
    a.out_bytes = (unsigned)((((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2);
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
    for (int t = 0; t < 9; ++t) {
        a.dh[t] = g->dh[t];
        a.dw[t] = g->dw[t];
    }
    a.M = g->N * g->Ho * g->Wo;
    a.NPT = (a.M + 255) / 256;
    a.NCT = (g->Co + (bn != nullptr ? 31 : 63)) / (bn != nullptr ? 32 : 64);
    a.bn_y = nullptr;
    if (bn != nullptr) {
        if (bn->coef == nullptr) return 0;
        const long long yb = (((long long)g->N * g->Ho * g->Wo - 1) * bn->ld_y + g->Co) * 2;
        if (yb >= (1ll << 31) || bn->ld_y % 8 != 0) return 0;
        a.bn_y = (const bf16_t*)bn->y;
        a.bn_y_bytes = (unsigned)yb;
        a.bn_ld = bn->ld_y;
        a.bn_coef = bn->coef;
        a.bn_sums = bn->sums;
        a.bn_act = bn->act;
        a.bn_slope = bn->slope;
    }
    const int slots = segnb_knob_conv_cus() * 2;
    a.NCS = 1;
    if (a.NPT < slots) {
        a.NCS = (slots + a.NPT - 1) / a.NPT;
        if (a.NCS > a.NCT) a.NCS = a.NCT;
    }
    a.CTS = (a.NCT + a.NCS - 1) / a.NCS;
    a.NCS = (a.NCT + a.CTS - 1) / a.CTS;
    int grid = slots / a.NCS;
    if (grid > a.NPT) grid = a.NPT;
    if (grid < 1) grid = 1;
    grid *= a.NCS;
    if (bn != nullptr)
        hipLaunchKernelGGL((conv_thin_kernel<true, 2>), dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_thin_kernel<false, 4>), dim3(grid), dim3(256), 0, stream, a);
    return 1;
}

int segnb_fprop_thin_ok(const segnb_conv_geom* g) { return segnb_knob_fprop_thin() && thin_geometry(g) ? 1 : 0; }
