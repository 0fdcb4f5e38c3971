// Weight gradient of the thin stride-1 3x3 layers (bf16) as a ROLLING-WINDOW kernel -- the counterpart of fprop_roll.hip
// for aten::convolution_backward's weight part at the 224 x 224 / 112 x 112 levels of ZF_UNET (lib/models/zf_unet.py:5-32).
//
//   dW[co][t][ci] = sum over pixels (n, r, c) of  dy[n][r][c][co] * x[n][r + dh_t][c + dw_t][ci]
//
//   * one WAVE owns a strip of 32 dy columns and 16 of the output channels (co), and slides down a segment of rows: per dy
//     row 18 MFMAs (16 x 16 x 32, K = the row's 32 pixels) add the 9 taps x 2 input-channel fragments into 72 accumulator
//     registers that live for the whole launch.  The two (or more) waves that share a strip split the dy channels, so the
//     expensive dy operand is loaded -- and, when it is not in memory, recomputed -- exactly once;
//   * both operands are pixel-major in the wave's private LDS rings (x: 4 rows x 34 pixels, dy: 2 rows x 32 pixels) and are
//     read transposed by ds_read_b64_tr_b16.  The K index of the MFMA is mapped to pixels as k = 8 kg + 4 r + q <-> pixel
//     16 r + 4 kg + q (the same for both operands), which makes the two 16-lane groups of a half-wave read pixels 4 apart:
//     with the x image's chunk swizzle (fprop_roll.hip) every transposed read is conflict-free at every column shift;
//   * no block-level synchronisation while rows stream (a wave's LDS operations execute in order); the waves of a block meet
//     once, at the end, to add their accumulators into the block's slab in a FIXED order (bitwise reproducible), and the slabs
//     are summed by slab_reduce_kernel as for conv_wgrad_s1x9_kernel;
//   * the global loads pass through registers: x may be given as the pre-BatchNorm output of the producing layer (TFX: the
//     activated tensor is recomputed on the way in) and dy as (g or dz, y) of this layer (TFD: the BatchNorm-backward apply is
//     recomputed), with the expressions of fprop_roll.hip -- the data gradient and the weight gradient of a layer then read
//     the same three tensors and nothing else.
#include "fprop_dma.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

struct WRollArgs {
    const bf16_t* x;
    const bf16_t* d;
    const bf16_t* d2;
    unsigned x_bytes, d_bytes, d2_bytes;
    int ld_x, ld_d, ld_d2;
    float* dwp;
    long long slab_stride;
    int N, H, W, Ci, Co, Ktot;
    int tap[9];
    int SR, NSEG, NSTRIP, NTASK, NCOH;
    const float* tfx_coef;
    const float* tfx_drop;
    int tfx_Cp, tfx_act;
    float tfx_slope;
    const float* tfd_coef;
    const float* tfd_bcoef;
    int tfd_Cp, tfd_act;
    float tfd_slope;
};

__device__ __forceinline__ void unpack8w(const u32x4_t& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

typedef short s16x2w_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk2bfw(unsigned pk) {
    const s16x2w_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2w_t, pk), z));
}

__device__ __forceinline__ void lds_const8w(const float* table, int idx, float (&v)[8]) {
    asm volatile("" : "+v"(idx));
    const float4 a = *reinterpret_cast<const float4*>(table + idx);
    const float4 b = *reinterpret_cast<const float4*>(table + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

constexpr int WR_WAVES = 8;
constexpr int WR_XPX = 34, WR_XROWB = WR_XPX * 64, WR_DROWB = 32 * 32;
constexpr int WR_WAVE_LDS = 4 * WR_XROWB + 2 * WR_DROWB;
constexpr int WR_SLAB_FLOATS = 32 * 9 * 32;
// LDS: [per-wave rings][block slab fp32 [32][9][32]][constant tables]
constexpr int WR_OFF_SLAB = WR_WAVES * WR_WAVE_LDS;
constexpr int WR_OFF_CST = WR_OFF_SLAB + WR_SLAB_FLOATS * 4;
constexpr int WR_CST_FLOATS = 3 * 32 + WR_WAVES * 32 + 5 * 32;      // tfx (sc, sh, mu) | per-wave dropout row | tfd (5 folded)
constexpr int WR_SMEM = WR_OFF_CST + WR_CST_FLOATS * 4;

// Ci = 32 input channels, Co <= 32 (NCOH = ceil(Co / 16) waves per strip).  TFX: 0 = x is the operand, 1 = x is the
// pre-BatchNorm tensor (SEGNB_TF_ACT).  TFD: 0 = d is dy, 2 = d is g / dz and d2 this layer's y (SEGNB_TF_BNBWD)
template <int TFX, int TFD, int DL>
__global__ __launch_bounds__(WR_WAVES * 64, 2) void conv_wgrad_roll_kernel(const WRollArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const slab = reinterpret_cast<float*>(smem + WR_OFF_SLAB);
    float* const cx = reinterpret_cast<float*>(smem + WR_OFF_CST);
    float* const cdrop = cx + 3 * 32;
    float* const cd = cdrop + WR_WAVES * 32;
    constexpr int UN = 4 % DL == 0 ? 4 : 4 * DL;      // x ring slot (i % 4), dy ring slot (i % 2), register set (i % DL): all static

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const xr = smem + wave * WR_WAVE_LDS;
    unsigned char* const dr = xr + 4 * WR_XROWB;
    // (32-bit LDS addresses for the transposed reads: constant slot / row offsets then fold into the instructions' offset field)
    const unsigned xr_l = (unsigned)(size_t)smem + (unsigned)(wave * WR_WAVE_LDS);
    const unsigned dr_l = xr_l + 4 * WR_XROWB;
    const int n16 = lane & 15, kg = lane >> 4, q4 = n16 >> 2, p4 = n16 & 3;
    const int wps = a.NCOH;                              // waves per strip
    const int coh = wave % wps, pair = wave / wps, npair = WR_WAVES / wps;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.d), 0, (int)a.d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(TFD == 2 ? a.d2 : a.d), 0, TFD == 2 ? (int)a.d2_bytes : 0, 0x00020000);

    for (int i = threadIdx.x; i < WR_SLAB_FLOATS; i += WR_WAVES * 64) slab[i] = 0.f;
    float xneg = 0.f, dneg = 0.f;
    if constexpr (TFX == 1) {
        for (int c = threadIdx.x; c < 32; c += WR_WAVES * 64) {
            cx[c] = a.tfx_coef[c];
            cx[32 + c] = a.tfx_coef[a.tfx_Cp + c];
            cx[64 + c] = a.tfx_coef[2 * a.tfx_Cp + c];
        }
        xneg = a.tfx_act == SEGNB_ACT_RELU ? 0.f : (a.tfx_act == SEGNB_ACT_LEAKY ? a.tfx_slope : 1.f);
    }
    if constexpr (TFD == 2) {
        for (int c = threadIdx.x; c < 32; c += WR_WAVES * 64) {
            const bool in = c < a.Co;
            const float sc = in ? a.tfd_coef[c] : 0.f, sh = in ? a.tfd_coef[a.tfd_Cp + c] : 0.f;
            const float mu = in ? a.tfd_coef[2 * a.tfd_Cp + c] : 0.f, is = in ? a.tfd_coef[3 * a.tfd_Cp + c] : 0.f;
            const float ba = in ? a.tfd_bcoef[c] : 0.f, c1 = in ? a.tfd_bcoef[a.tfd_Cp + c] : 0.f;
            const float c2 = in ? a.tfd_bcoef[2 * a.tfd_Cp + c] : 0.f;
            cd[c] = sc;
            cd[32 + c] = sh - mu * sc;
            cd[64 + c] = ba;
            cd[96 + c] = -ba * c2 * is;
            cd[128 + c] = ba * (c2 * is * mu - c1);
        }
        dneg = a.tfd_act == SEGNB_ACT_RELU ? 0.f : (a.tfd_act == SEGNB_ACT_LEAKY ? a.tfd_slope : 1.f);
    }
    __syncthreads();

    // ---- per-lane constants
    // x row image: 34 pixels x 64 B, pixel j = column c0 - 1 + j, chunk slot q ^ sw(j) (fprop_roll.hip); three loads per row
    int xwoff[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int L = 64 * m + lane, j = L >> 2, c = L & 3;
        xwoff[m] = j < WR_XPX ? j * 64 + ((c ^ (((j >> 2) & 1) << 1)) << 4) : -1;
    }
    const int xc0 = (lane & 3) * 8;                     // channel chunk of this lane in an x load
    // dy row image: 32 pixels x 32 B (this wave's 16 channels), one load per row: lane -> pixel lane / 2, chunk lane & 1
    const int dwoff = lane * 16;
    const int dc0 = coh * 16 + (lane & 1) * 8;          // dy channel chunk of this lane in a dy load
    // transposed reads: lane (kg, q4, p4) supplies row (pixel) 16 r + 4 kg + q4, channels 4 p4 .. + 3 of a 16-channel block
    unsigned xro[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) {
            const int j = dx + 4 * kg + q4;             // (r = 1: + 16 pixels = + 1024 B, same swizzle bit)
            xro[dx][cf] = xr_l + (unsigned)(j * 64 + (((2 * cf + (p4 >> 1)) ^ (((j >> 2) & 1) << 1)) << 4) + 8 * (p4 & 1));
        }
    const unsigned dro = dr_l + (unsigned)((4 * kg + q4) * 32 + (p4 >> 1) * 16 + 8 * (p4 & 1));

    f32x4_t acc[3][3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[i][j][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nstream = gridDim.x * npair;
    for (int task = blockIdx.x * npair + pair; task < a.NTASK; task += nstream) {
        const int strip = task % a.NSTRIP;
        const int t2 = task / a.NSTRIP;
        const int seg = t2 % a.NSEG, n = t2 / a.NSEG;
        const int r0 = seg * a.SR;
        const int rows = min(a.SR, a.H - r0);
        const int c0 = strip * 32;

        unsigned xcoff[3];
        bool xcolv[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int L = 64 * m + lane;
            const int col = c0 - 1 + (L >> 2);
            xcolv[m] = xwoff[m] >= 0 && (unsigned)col < (unsigned)a.W;
            xcoff[m] = xcolv[m] ? (unsigned)(col * a.ld_x * 2 + (L & 3) * 16) : OOB;
        }
        const int dcol = c0 + (lane >> 1);
        const bool dcolv = dcol < a.W && dc0 < a.Co;
        const unsigned dcoff = dcolv ? (unsigned)(dcol * a.ld_d * 2 + dc0 * 2) : OOB;
        const unsigned dcoff2 = (TFD == 2 && dcolv) ? (unsigned)(dcol * a.ld_d2 * 2 + dc0 * 2) : OOB;
        // (folded activation transform: fprop_roll.hip)
        float q1[8], q0[8];
        if constexpr (TFX == 1) {
            float tsc[8], tsh[8], tmu[8];
            lds_const8w(cx, xc0, tsc);
            lds_const8w(cx + 32, xc0, tsh);
            lds_const8w(cx + 64, xc0, tmu);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dm = a.tfx_drop != nullptr ? a.tfx_drop[n * a.tfx_Cp + xc0 + e] : 1.f;
                q1[e] = tsc[e] * dm;
                q0[e] = (tsh[e] - tmu[e] * tsc[e]) * dm;
            }
        }

        // x rows j = -1 .. rows (global row r0 + j), dy rows i = 0 .. rows - 1
        u32x4_t lx[DL][3], ldd[DL], ld2[TFD == 2 ? DL : 1];
        auto issue_x = [&](auto set_c, int j) {
            constexpr int set = decltype(set_c)::value;
            const int gr = r0 + j;
            const bool rv = j <= rows && (unsigned)gr < (unsigned)a.H;
            const unsigned pixrow = (unsigned)((n * a.H + gr) * a.W);
#pragma unroll
            for (int m = 0; m < 3; ++m)
                lx[set][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(rv ? pixrow * (unsigned)(a.ld_x * 2) + xcoff[m] : OOB), 0, 0);
        };
        auto issue_d = [&](auto set_c, int i) {
            constexpr int set = decltype(set_c)::value;
            const bool rv = i < rows;
            const unsigned pixrow = (unsigned)((n * a.H + r0 + i) * a.W);
            ldd[set] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, (int)(rv ? pixrow * (unsigned)(a.ld_d * 2) + dcoff : OOB), 0, 0);
            if constexpr (TFD == 2)
                ld2[set] = __builtin_amdgcn_raw_buffer_load_b128(rs_d2, (int)(rv ? pixrow * (unsigned)(a.ld_d2 * 2) + dcoff2 : OOB), 0, 0);
        };
        auto publish_x = [&](auto set_c, auto slot_c, int j) {
            constexpr int set = decltype(set_c)::value, slot = decltype(slot_c)::value;
            const bool rv = (unsigned)(r0 + j) < (unsigned)a.H;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                u32x4_t v = lx[set][m];
                if constexpr (TFX == 1) {
                    float f[8];
                    unpack8w(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = f[e] * q1[e] + q0[e];
                    const bool ok = rv && xcolv[m];
                    if (xneg == 0.f) {
                        v.x = ok ? relu_pk2bfw(pack2bf(f[0], f[1])) : 0u;
                        v.y = ok ? relu_pk2bfw(pack2bf(f[2], f[3])) : 0u;
                        v.z = ok ? relu_pk2bfw(pack2bf(f[4], f[5])) : 0u;
                        v.w = ok ? relu_pk2bfw(pack2bf(f[6], f[7])) : 0u;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * xneg;
                        v.x = ok ? pack2bf(f[0], f[1]) : 0u;
                        v.y = ok ? pack2bf(f[2], f[3]) : 0u;
                        v.z = ok ? pack2bf(f[4], f[5]) : 0u;
                        v.w = ok ? pack2bf(f[6], f[7]) : 0u;
                    }
                }
                if (xwoff[m] >= 0) *reinterpret_cast<u32x4_t*>(xr + slot * WR_XROWB + xwoff[m]) = v;
            }
        };
        auto publish_d = [&](auto set_c, auto slot_c, int i) {
            constexpr int set = decltype(set_c)::value, slot = decltype(slot_c)::value;
            u32x4_t v = ldd[set];
            if constexpr (TFD == 2) {
                float gq[8], yq[8], tsc[8], tsh[8], tba[8], tB[8], tC[8];
                unpack8w(v, gq);
                unpack8w(ld2[set], yq);
                lds_const8w(cd, dc0, tsc);
                lds_const8w(cd + 32, dc0, tsh);
                lds_const8w(cd + 64, dc0, tba);
                lds_const8w(cd + 96, dc0, tB);
                lds_const8w(cd + 128, dc0, tC);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = yq[e] * tsc[e] + tsh[e];
                    const float dzf = gq[e] * (z > 0.f ? 1.f : dneg);
                    gq[e] = tba[e] * dzf + (tB[e] * yq[e] + tC[e]);
                }
                const bool ok = i < rows && dcolv;
                v.x = ok ? pack2bf(gq[0], gq[1]) : 0u;
                v.y = ok ? pack2bf(gq[2], gq[3]) : 0u;
                v.z = ok ? pack2bf(gq[4], gq[5]) : 0u;
                v.w = ok ? pack2bf(gq[6], gq[7]) : 0u;
            }
            *reinterpret_cast<u32x4_t*>(dr + slot * WR_DROWB + dwoff) = v;
        };

        // prologue: x rows -1, 0, 1 published (slots 0, 1, 2), dy row 0 published (slot 0); DL rows of each in flight behind them
        static_for<DL>([&](auto k_c) { issue_x(k_c, -1 + decltype(k_c)::value); });
        static_for<DL>([&](auto k_c) { issue_d(k_c, decltype(k_c)::value); });
        // (DL >= 2.)  x rows -1 and 0 now; row 1 and everything later inside the steps
        publish_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, -1);
        issue_x(std::integral_constant<int, 0>{}, -1 + DL);
        publish_x(std::integral_constant<int, 1 % DL>{}, std::integral_constant<int, 1>{}, 0);
        issue_x(std::integral_constant<int, 1 % DL>{}, DL);
        publish_d(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        issue_d(std::integral_constant<int, 0>{}, DL);

        // step i (UI = i % UN): dy row i (slot i % 2) against x rows i - 1, i, i + 1 (slots (i + dyt) % 4, x row j in slot (j + 1) % 4).
        // Before its MFMAs it publishes x row i + 1 (slot (i + 2) % 4, register set (i + 2) % DL) -- needed by this step -- and
        // afterwards dy row i + 1 (slot (i + 1) % 2, set (i + 1) % DL).
        auto step = [&](auto u_c, int i) {
            constexpr int UI = decltype(u_c)::value;
            constexpr int XS_NEW = (UI + 2) % 4, XSET = (UI + 2) % DL, DS = UI % 2, DSET1 = (UI + 1) % DL;
            if (i >= rows) return;
            // B operands of kernel row dyt: x row i + dyt - 1, three column shifts x two 16-channel fragments
            auto read_group = [&](auto dyt_c, bf16x8_t (&Bf)[6]) {
                constexpr unsigned xs = ((UI + decltype(dyt_c)::value) % 4) * WR_XROWB;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int cf = 0; cf < 2; ++cf) {
                        const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(xro[dx][cf] + xs));
                        const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(xro[dx][cf] + xs + 16 * 64));
                        Bf[dx * 2 + cf] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
            };
            // A operand: dy^T, 16 channels x 32 pixels; the fragment reads run one kernel row ahead of their MFMAs (the
            // compiler otherwise sinks every read next to its MFMA and the LDS latency is paid 18 times per row)
            const bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(dro + DS * WR_DROWB));
            const bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(dro + DS * WR_DROWB + 16 * 32));
            const bf16x8_t A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            bf16x8_t B0[6], B1[6], B2[6];
            read_group(std::integral_constant<int, 0>{}, B0);
            read_group(std::integral_constant<int, 1>{}, B1);
            __builtin_amdgcn_sched_barrier(0);
            // x row i + 1 enters the ring (read by kernel row 2 below: a wave's LDS operations execute in order)
            publish_x(std::integral_constant<int, XSET>{}, std::integral_constant<int, XS_NEW>{}, i + 1);
            if (i + 1 + DL <= rows) issue_x(std::integral_constant<int, XSET>{}, i + 1 + DL);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[0][k >> 1][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B0[k], acc[0][k >> 1][k & 1], 0, 0, 0);
            read_group(std::integral_constant<int, 2>{}, B2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[1][k >> 1][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B1[k], acc[1][k >> 1][k & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < rows) publish_d(std::integral_constant<int, DSET1>{}, std::integral_constant<int, (UI + 1) % 2>{}, i + 1);
            if (i + 1 + DL < rows) issue_d(std::integral_constant<int, DSET1>{}, i + 1 + DL);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[2][k >> 1][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2[k], acc[2][k >> 1][k & 1], 0, 0, 0);
        };
        for (int ib = 0; ib < rows; ib += UN)
            static_for<UN>([&](auto u_c) { step(u_c, ib + decltype(u_c)::value); });
    }

    // ---- the block's slab: waves add their accumulators in a fixed order (the NCOH waves of a strip own disjoint channels
    // and go together), then the slab leaves with coalesced stores
    for (int turn = 0; turn < npair; ++turn) {
        if (pair == turn) {
#pragma unroll
            for (int dyt = 0; dyt < 3; ++dyt)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int t = a.tap[dyt * 3 + dx];
#pragma unroll
                    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int co = coh * 16 + kg * 4 + e, ci = cf * 16 + n16;
                            slab[(co * 9 + t) * 32 + ci] += acc[dyt][dx][cf][e];
                        }
                }
        }
        __syncthreads();
    }
    float* const out = a.dwp + (long long)blockIdx.x * a.slab_stride;
    for (int i = threadIdx.x; i < a.Co * 9 * 32; i += WR_WAVES * 64) {
        const int co = i / (9 * 32), rem = i - co * 9 * 32;
        out[(long long)co * a.Ktot + rem] = slab[i];
    }
}

template <int TFX, int TFD, int DL>
int launch_wroll_dl(WRollArgs& a, int nslab, hipStream_t stream) {
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<TFX, TFD, DL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WR_SMEM);
        if (e != hipSuccess) segnb_set_error("wgrad_roll hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.NCOH = (a.Co + 15) / 16;
    a.NSTRIP = (a.W + 31) / 32;
    const int npair = WR_WAVES / a.NCOH;
    // nslab blocks, npair strips in flight per block: segments sized for about one task per stream
    // (the most segments per strip that still give every stream at most ONE task: a second round for a few streams would
    // double the launch; when even whole strips outnumber the streams, short segments balance the rounds instead)
    const int streams = nslab * npair;
    int nseg = (int)(streams / ((long long)a.N * a.NSTRIP));
    if (nseg < 1) nseg = (a.H + 15) / 16;
    if (nseg > a.H) nseg = a.H;
    const int sr = (a.H + nseg - 1) / nseg;
    a.SR = sr;
    a.NSEG = (a.H + sr - 1) / sr;
    a.NTASK = a.N * a.NSEG * a.NSTRIP;
    hipLaunchKernelGGL((conv_wgrad_roll_kernel<TFX, TFD, DL>), dim3(nslab), dim3(WR_WAVES * 64), WR_SMEM, stream, a);
    return 0;
}

template <int TFX, int TFD>
int launch_wroll(WRollArgs& a, int nslab, hipStream_t stream) {
    // rows of loads in flight per wave: four where the registers allow it (no transforms), two beside the transforms
    if (TFX != 0 || TFD != 0) return launch_wroll_dl<TFX, TFD, 2>(a, nslab, stream);
    return launch_wroll_dl<TFX, TFD, 4>(a, nslab, stream);
}


// ---------------------------------------------------------------------------------------------------------------------------
// The FIRST layer (8 padded input channels, lib/models/zf_unet.py:37 `double_conv_layer(input_channels, filters)`): the same
// rolling scheme with 16-byte x pixels.  An N block of the MFMA is TWO TAPS x 8 channels (taps 2 nb, 2 nb + 1 in kernel-row
// major order; the tenth half-block is computed and discarded), so a dy row costs 5 MFMAs per 16 output channels instead of
// 18 and the kernel is bound by its operand streams alone.  The transposed read serves this directly: the two 8-lane halves
// of a 16-lane group point at different pixels (and, for the pair that straddles two kernel rows, different ring slots).
// TFD == 3: the dy operand is recomputed from (g, y) with the arithmetic and roundings of bn_bwd_apply_kernel's direct form
// (segnb_conv_wgrad_bnapply: a first layer has no data gradient, so nothing else would read the tensor).
constexpr int W8_XROWB = 608;                          // x row: 34 pixels x 16 B, padded so that a straddling pair's halves use disjoint banks
constexpr int W8_SLAB_FLOATS = 32 * 9 * 8;

template <int NW, int COH>
constexpr int w8_wave_lds() { return 4 * W8_XROWB + 2 * 32 * 32 * COH; }
template <int NW, int COH>
constexpr int w8_smem() { return NW * w8_wave_lds<NW, COH>() + W8_SLAB_FLOATS * 4; }

// COH: 16-channel halves of dy a wave owns (1: Co <= 16, 2: Co <= 32).  A wave loads WHOLE dy pixels -- every 128-byte line it
// touches is used in full by one instruction -- and the x row of a strip is loaded once; splitting the dy channels over two
// waves (as conv_wgrad_roll_kernel does) left both operand streams at half-used lines and the kernel at 3.6 TB/s.
template <int TFD, int DL, int NW, int COH>
__global__ __launch_bounds__(NW * 64, 1) void conv_wgrad_c8roll_kernel(const WRollArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WAVE_LDS = w8_wave_lds<NW, COH>(), DPXB = 32 * COH, DROWB = 32 * DPXB;
    float* const slab = reinterpret_cast<float*>(smem + NW * WAVE_LDS);
    constexpr int UN = 4 % DL == 0 ? 4 : 4 * DL;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const xr = smem + wave * WAVE_LDS;
    unsigned char* const dr = xr + 4 * W8_XROWB;
    const unsigned xr_l = (unsigned)(size_t)smem + (unsigned)(wave * WAVE_LDS);
    const unsigned dr_l = xr_l + 4 * W8_XROWB;
    const int n16 = lane & 15, kg = lane >> 4, q4 = n16 >> 2, p4 = n16 & 3;
    const bool hi = (p4 >> 1) != 0;                      // second tap of a pair

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.d), 0, (int)a.d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(TFD == 3 ? a.d2 : a.d), 0, TFD == 3 ? (int)a.d2_bytes : 0, 0x00020000);

    for (int i = threadIdx.x; i < W8_SLAB_FLOATS; i += NW * 64) slab[i] = 0.f;

    // dy row image: 32 pixels x DPXB bytes, COH loads per row: load m, lane l -> 16-byte chunk 64 m + l of the row (pixel
    // (64 m + l) / (2 COH), channel chunk (64 m + l) % (2 COH): the SAME chunk for every m)
    const int dc0 = (lane & (2 * COH - 1)) * 8;
    // per-channel constants of this lane's (fixed) dy chunk
    float k_sc[8], k_sh[8], k_mu[8], k_is[8], k_a[8], k_c1[8], k_c2[8], k_neg = 0.f;
    bool k_round = false;
    if constexpr (TFD == 3) {
        k_neg = a.tfd_act == SEGNB_ACT_RELU ? 0.f : (a.tfd_act == SEGNB_ACT_LEAKY ? a.tfd_slope : 1.f);
        k_round = k_neg != 0.f && k_neg != 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = dc0 + e < a.tfd_Cp ? dc0 + e : 0;
            k_sc[e] = a.tfd_coef[c];
            k_sh[e] = a.tfd_coef[a.tfd_Cp + c];
            k_mu[e] = a.tfd_coef[2 * a.tfd_Cp + c];
            k_is[e] = a.tfd_coef[3 * a.tfd_Cp + c];
            k_a[e] = a.tfd_bcoef[c];
            k_c1[e] = a.tfd_bcoef[a.tfd_Cp + c];
            k_c2[e] = a.tfd_bcoef[2 * a.tfd_Cp + c];
        }
    }
    __syncthreads();

    // transposed reads of x: block nb, half h -> tap k9 = 2 nb + h (kernel row k9 / 3, column shift k9 % 3); lane (kg, q4, p4)
    // supplies pixel 16 r + 4 kg + q4 (+ the shift), channels 4 (p4 & 1) .. + 3 of its tap
    unsigned xro[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        const int k9 = min(2 * nb + (hi ? 1 : 0), 8);
        xro[nb] = xr_l + (unsigned)((k9 % 3 + 4 * kg + q4) * 16 + 8 * (p4 & 1));
    }
    // ... of dy: pixel 16 r + 4 kg + q4, channels 4 p4 .. + 3 of the 16-channel half
    const unsigned dro = dr_l + (unsigned)((4 * kg + q4) * DPXB + (p4 >> 1) * 16 + 8 * (p4 & 1));

    f32x4_t acc[COH][5];
#pragma unroll
    for (int h = 0; h < COH; ++h)
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[h][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nstream = gridDim.x * NW;
    for (int task = blockIdx.x * NW + wave; task < a.NTASK; task += nstream) {
        const int strip = task % a.NSTRIP;
        const int t2 = task / a.NSTRIP;
        const int seg = t2 % a.NSEG, n = t2 / a.NSEG;
        const int r0 = seg * a.SR;
        const int rows = min(a.SR, a.H - r0);
        const int c0 = strip * 32;

        const int xcol = c0 - 1 + lane;
        const bool xcolv = lane < WR_XPX && (unsigned)xcol < (unsigned)a.W;
        const unsigned xcoff = xcolv ? (unsigned)(xcol * a.ld_x * 2) : OOB;
        unsigned dcoff[COH], dcoff2[COH];
        bool dcolv[COH];
#pragma unroll
        for (int m = 0; m < COH; ++m) {
            const int dcol = c0 + (64 * m + lane) / (2 * COH);
            dcolv[m] = dcol < a.W && dc0 < a.Co;
            dcoff[m] = dcolv[m] ? (unsigned)(dcol * a.ld_d * 2 + dc0 * 2) : OOB;
            dcoff2[m] = (TFD == 3 && dcolv[m]) ? (unsigned)(dcol * a.ld_d2 * 2 + dc0 * 2) : OOB;
        }

        u32x4_t lx[DL], ldd[DL][COH], ld2[TFD == 3 ? DL : 1][COH];
        auto issue_x = [&](auto set_c, int j) {
            constexpr int set = decltype(set_c)::value;
            const int gr = r0 + j;
            const bool rv = j <= rows && (unsigned)gr < (unsigned)a.H;
            const unsigned pixrow = (unsigned)((n * a.H + gr) * a.W);
            lx[set] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(rv ? pixrow * (unsigned)(a.ld_x * 2) + xcoff : OOB), 0, 0);
        };
        auto issue_d = [&](auto set_c, int i) {
            constexpr int set = decltype(set_c)::value;
            const bool rv = i < rows;
            const unsigned pixrow = (unsigned)((n * a.H + r0 + i) * a.W);
#pragma unroll
            for (int m = 0; m < COH; ++m) {
                ldd[set][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, (int)(rv ? pixrow * (unsigned)(a.ld_d * 2) + dcoff[m] : OOB), 0, 0);
                if constexpr (TFD == 3)
                    ld2[set][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_d2, (int)(rv ? pixrow * (unsigned)(a.ld_d2 * 2) + dcoff2[m] : OOB), 0, 0);
            }
        };
        auto publish_x = [&](auto set_c, auto slot_c) {
            constexpr int set = decltype(set_c)::value, slot = decltype(slot_c)::value;
            if (lane < WR_XPX) *reinterpret_cast<u32x4_t*>(xr + slot * W8_XROWB + lane * 16) = lx[set];
        };
        auto publish_d = [&](auto set_c, auto slot_c, int i) {
            constexpr int set = decltype(set_c)::value, slot = decltype(slot_c)::value;
#pragma unroll
            for (int m = 0; m < COH; ++m) {
                u32x4_t v = ldd[set][m];
                if constexpr (TFD == 3) {
                    float gq[8], yq[8];
                    unpack8w(v, gq);
                    unpack8w(ld2[set][m], yq);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = (yq[e] - k_mu[e]) * k_sc[e] + k_sh[e] + 0.f;
                        float d = gq[e] * 1.f * (z > 0.f ? 1.f : k_neg);
                        if (k_round) d = __uint_as_float(pack2bf(d, 0.f) << 16);
                        const float yh = (yq[e] - k_mu[e]) * k_is[e];
                        gq[e] = k_a[e] * (d - k_c1[e] - yh * k_c2[e]);
                    }
                    const bool ok = i < rows && dcolv[m];
                    v.x = ok ? pack2bf(gq[0], gq[1]) : 0u;
                    v.y = ok ? pack2bf(gq[2], gq[3]) : 0u;
                    v.z = ok ? pack2bf(gq[4], gq[5]) : 0u;
                    v.w = ok ? pack2bf(gq[6], gq[7]) : 0u;
                }
                *reinterpret_cast<u32x4_t*>(dr + slot * DROWB + (64 * m + lane) * 16) = v;
            }
        };

        static_for<DL>([&](auto k_c) { issue_x(k_c, -1 + decltype(k_c)::value); });
        static_for<DL>([&](auto k_c) { issue_d(k_c, decltype(k_c)::value); });
        publish_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        issue_x(std::integral_constant<int, 0>{}, -1 + DL);
        publish_x(std::integral_constant<int, 1 % DL>{}, std::integral_constant<int, 1>{});
        issue_x(std::integral_constant<int, 1 % DL>{}, DL);
        publish_d(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        issue_d(std::integral_constant<int, 0>{}, DL);

        // step i: x row i + 1 enters the ring (slot (i + 2) % 4), then dy row i (slot i % 2) meets x rows i - 1 .. i + 1
        // (x row j in slot (j + 1) % 4: kernel row dyt of step i reads slot (i + dyt) % 4); dy row i + 1 is published behind
        auto step = [&](auto u_c, int i) {
            constexpr int UI = decltype(u_c)::value;
            constexpr int XS_NEW = (UI + 2) % 4, XSET = (UI + 2) % DL, DS = UI % 2, DSET1 = (UI + 1) % DL;
            if (i >= rows) return;
            publish_x(std::integral_constant<int, XSET>{}, std::integral_constant<int, XS_NEW>{});
            if (i + 1 + DL <= rows) issue_x(std::integral_constant<int, XSET>{}, i + 1 + DL);
            bf16x8_t A[COH];
#pragma unroll
            for (int h = 0; h < COH; ++h) {
                const bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(dro + DS * DROWB + h * 32));
                const bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(dro + DS * DROWB + h * 32 + 16 * DPXB));
                A[h] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            bf16x8_t B[5];
            static_for<5>([&](auto nb_c) {
                constexpr int nb = decltype(nb_c)::value;
                constexpr int kA = 2 * nb, kB = nb == 4 ? 8 : 2 * nb + 1;
                constexpr unsigned sA = ((UI + kA / 3) % 4) * W8_XROWB, sB = ((UI + kB / 3) % 4) * W8_XROWB;
                const unsigned ad = xro[nb] + (sA == sB ? sA : (hi ? sB : sA));
                const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)ad);
                const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(uintptr_t)(ad + 16 * 16));
                B[nb] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            });
#pragma unroll
            for (int nb = 0; nb < 5; ++nb)
#pragma unroll
                for (int h = 0; h < COH; ++h) acc[h][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[h], B[nb], acc[h][nb], 0, 0, 0);
            if (i + 1 < rows) publish_d(std::integral_constant<int, DSET1>{}, std::integral_constant<int, (UI + 1) % 2>{}, i + 1);
            if (i + 1 + DL < rows) issue_d(std::integral_constant<int, DSET1>{}, i + 1 + DL);
        };
        for (int ib = 0; ib < rows; ib += UN)
            static_for<UN>([&](auto u_c) { step(u_c, ib + decltype(u_c)::value); });
    }

    // the block's slab: the waves add their accumulators in a fixed order (bitwise reproducible)
    for (int turn = 0; turn < NW; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const int k9 = 2 * nb + (n16 >> 3);
                if (k9 < 9) {
                    const int t = a.tap[k9];
#pragma unroll
                    for (int h = 0; h < COH; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) slab[((h * 16 + kg * 4 + e) * 9 + t) * 8 + (n16 & 7)] += acc[h][nb][e];
                }
            }
        }
        __syncthreads();
    }
    float* const out = a.dwp + (long long)blockIdx.x * a.slab_stride;
    for (int i = threadIdx.x; i < a.Co * 72; i += NW * 64) out[i] = slab[i];
}

template <int TFD, int DL, int NW, int COH>
int launch_c8roll_nw(WRollArgs& a, int nslab, hipStream_t stream) {
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_c8roll_kernel<TFD, DL, NW, COH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, w8_smem<NW, COH>());
        if (e != hipSuccess) segnb_set_error("wgrad_c8roll hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.NCOH = 1;
    a.NSTRIP = (a.W + 31) / 32;
    // nslab blocks of NW waves, one (strip, row segment) task per wave where the shape allows it (conv_wgrad_roll_kernel)
    const int streams = nslab * NW;
    int nseg = (int)(streams / ((long long)a.N * a.NSTRIP));
    if (nseg < 1) nseg = (a.H + 15) / 16;
    if (nseg > a.H) nseg = a.H;
    const int sr = (a.H + nseg - 1) / nseg;
    a.SR = sr;
    a.NSEG = (a.H + sr - 1) / sr;
    a.NTASK = a.N * a.NSEG * a.NSTRIP;
    hipLaunchKernelGGL((conv_wgrad_c8roll_kernel<TFD, DL, NW, COH>), dim3(nslab), dim3(NW * 64), (w8_smem<NW, COH>()), stream, a);
    return 0;
}

template <int TFD, int COH>
int launch_c8roll_coh(WRollArgs& a, int nslab, hipStream_t stream) {
    constexpr int nw = 0, dl = 2;          // (block width / rows of loads in flight: the measured defaults; other values below are kept for reference)
    // 8 waves of <= 256 registers (the recomputing variant holds 56 constants and two operand streams: 192); the plain variant
    // takes the same partition, so that both sum in the same order (segnb_conv_wgrad_bnapply == apply pass + segnb_conv_wgrad
    // bit for bit).  Measured (tools/c8_bench.py, us incl. the 6.7 us slab reduction): 8 waves 40.3 / 59.7, 16 waves 50.0 / 273.6;
    // 4-wave blocks on 512 slabs (two per CU, could share a CU with the side stream's kernels): 52.7 alone, +-0 in the step
    const bool w16 = nw == 16;
    if (w16) return dl == 4 ? launch_c8roll_nw<TFD, 4, 16, COH>(a, nslab, stream) : launch_c8roll_nw<TFD, 2, 16, COH>(a, nslab, stream);
    return dl == 4 ? launch_c8roll_nw<TFD, 4, 8, COH>(a, nslab, stream) : launch_c8roll_nw<TFD, 2, 8, COH>(a, nslab, stream);
}

template <int TFD>
int launch_c8roll(WRollArgs& a, int nslab, hipStream_t stream) {
    return a.Co <= 16 ? launch_c8roll_coh<TFD, 1>(a, nslab, stream) : launch_c8roll_coh<TFD, 2>(a, nslab, stream);
}
}  // namespace

bool segnb_wgrad_roll_applies(const segnb_conv_geom* g) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Hi != g->Ho || g->Wi != g->Wo) return false;
    if (g->Ci != 32 || g->Co > 32 || g->Co % 8 != 0 || g->Wo < 32 || g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return false;
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        if (g->dh[t] < -1 || g->dh[t] > 1 || g->dw[t] < -1 || g->dw[t] > 1) return false;
        const int k = (g->dh[t] + 1) * 3 + (g->dw[t] + 1);
        if (seen[k]) return false;
        seen[k] = true;
    }
    return true;
}

// 1 = handled, 0 = not applicable, else error.  nslab: the slab count of the workspace (segnb_conv_wgrad_slabs): one block per
// slab.  tfx / tfd: operand transforms (NULL: the operand is in memory)
int segnb_wgrad_roll_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab, hipStream_t stream,
                         bool partial, const segnb_operand_tf* tfx, const segnb_operand_tf* tfd) {
    if (!segnb_wgrad_roll_applies(g) || nslab < 1) return 0;
    WRollArgs a;
    for (int t = 0; t < 9; ++t) a.tap[(g->dh[t] + 1) * 3 + (g->dw[t] + 1)] = t;
    a.x = (const bf16_t*)in;
    a.d = (const bf16_t*)dout;
    a.d2 = nullptr;
    a.ld_x = g->ld_in;
    a.ld_d = g->ld_out;
    a.ld_d2 = 0;
    const long long xb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    const long long db = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
    if (xb >= (1ll << 31) || db >= (1ll << 31)) return 0;
    a.x_bytes = (unsigned)xb;
    a.d_bytes = (unsigned)db;
    a.d2_bytes = 0;
    a.dwp = dwp;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Ci = g->Ci; a.Co = g->Co;
    a.Ktot = 9 * g->Ci;
    a.slab_stride = (long long)g->Co * a.Ktot;
    a.tfx_coef = a.tfx_drop = a.tfd_coef = a.tfd_bcoef = nullptr;
    a.tfx_Cp = a.tfd_Cp = 0;
    a.tfx_act = a.tfd_act = 0;
    a.tfx_slope = a.tfd_slope = 0.f;
    if (tfx != nullptr) {
        if (tfx->kind != SEGNB_TF_ACT || tfx->coef == nullptr || tfx->Cp < g->Ci) return 0;
        a.tfx_coef = tfx->coef;
        a.tfx_drop = tfx->drop;
        a.tfx_Cp = tfx->Cp;
        a.tfx_act = tfx->act;
        a.tfx_slope = tfx->slope;
    }
    if (tfd != nullptr) {
        if (tfd->kind != SEGNB_TF_BNBWD || tfd->coef == nullptr || tfd->bcoef == nullptr || tfd->y == nullptr || tfd->drop != nullptr ||
            tfd->Cp < g->Co || tfd->ld_y % 8 != 0)
            return 0;
        const long long yb = (((long long)g->N * g->Ho * g->Wo - 1) * tfd->ld_y + g->Co) * 2;
        if (yb >= (1ll << 31)) return 0;
        a.d2 = (const bf16_t*)tfd->y;
        a.d2_bytes = (unsigned)yb;
        a.ld_d2 = tfd->ld_y;
        a.tfd_coef = tfd->coef;
        a.tfd_bcoef = tfd->bcoef;
        a.tfd_Cp = tfd->Cp;
        a.tfd_act = tfd->act;
        a.tfd_slope = tfd->slope;
    }
    int rc;
    if (tfx != nullptr && tfd != nullptr) rc = launch_wroll<1, 2>(a, nslab, stream);
    else if (tfx != nullptr) rc = launch_wroll<1, 0>(a, nslab, stream);
    else if (tfd != nullptr) rc = launch_wroll<0, 2>(a, nslab, stream);
    else rc = launch_wroll<0, 0>(a, nslab, stream);
    if (rc) return rc;
    if (nslab > 1 && !partial) segnb_slab_reduce(dwp, a.slab_stride, nslab, stream);
    return 1;
}

bool segnb_wgrad_c8roll_applies(const segnb_conv_geom* g) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Hi != g->Ho || g->Wi != g->Wo) return false;
    if (g->Ci != 8 || g->Co > 32 || g->Co % 8 != 0 || g->Wo < 32 || g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return false;
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        if (g->dh[t] < -1 || g->dh[t] > 1 || g->dw[t] < -1 || g->dw[t] > 1) return false;
        const int k = (g->dh[t] + 1) * 3 + (g->dw[t] + 1);
        if (seen[k]) return false;
        seen[k] = true;
    }
    return true;
}

// the first layer's weight gradient (conv_wgrad_c8roll_kernel).  bna: dy recomputed from (g, y) (dout is then ignored)
int segnb_wgrad_c8roll_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab, hipStream_t stream,
                           bool partial, const segnb_wgrad_bnapply* bna) {
    if (!segnb_wgrad_c8roll_applies(g) || nslab < 1) return 0;
    WRollArgs a;
    for (int t = 0; t < 9; ++t) a.tap[(g->dh[t] + 1) * 3 + (g->dw[t] + 1)] = t;
    const long long npix = (long long)g->N * g->Ho * g->Wo;
    a.x = (const bf16_t*)in;
    a.ld_x = g->ld_in;
    const long long xb = ((npix - 1) * g->ld_in + g->Ci) * 2;
    long long db, d2b = 0;
    if (bna != nullptr) {
        if (bna->ld_g % 8 != 0 || bna->ld_y % 8 != 0 || bna->Cp < g->Co) return 0;
        a.d = (const bf16_t*)bna->g;
        a.ld_d = bna->ld_g;
        a.d2 = (const bf16_t*)bna->y;
        a.ld_d2 = bna->ld_y;
        db = ((npix - 1) * bna->ld_g + g->Co) * 2;
        d2b = ((npix - 1) * bna->ld_y + g->Co) * 2;
        a.tfd_coef = bna->coef;
        a.tfd_bcoef = bna->bcoef;
        a.tfd_Cp = bna->Cp;
        a.tfd_act = bna->act;
        a.tfd_slope = bna->slope;
    } else {
        a.d = (const bf16_t*)dout;
        a.ld_d = g->ld_out;
        a.d2 = nullptr;
        a.ld_d2 = 0;
        db = ((npix - 1) * g->ld_out + g->Co) * 2;
        a.tfd_coef = a.tfd_bcoef = nullptr;
        a.tfd_Cp = 0;
        a.tfd_act = 0;
        a.tfd_slope = 0.f;
    }
    if (xb >= (1ll << 31) || db >= (1ll << 31) || d2b >= (1ll << 31)) return 0;
    a.x_bytes = (unsigned)xb;
    a.d_bytes = (unsigned)db;
    a.d2_bytes = (unsigned)d2b;
    a.dwp = dwp;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Ci = g->Ci; a.Co = g->Co;
    a.Ktot = 9 * g->Ci;
    a.slab_stride = (long long)g->Co * a.Ktot;
    a.tfx_coef = a.tfx_drop = nullptr;
    a.tfx_Cp = a.tfx_act = 0;
    a.tfx_slope = 0.f;
    const int rc = bna != nullptr ? launch_c8roll<3>(a, nslab, stream) : launch_c8roll<0>(a, nslab, stream);
    if (rc) return rc;
    if (nslab > 1 && !partial) segnb_slab_reduce(dwp, a.slab_stride, nslab, stream);
    return 1;
}
