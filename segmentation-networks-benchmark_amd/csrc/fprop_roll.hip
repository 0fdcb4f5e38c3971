// Stride-1 3x3 convolution forward / data gradient (bf16) for the THIN layers as a ROLLING-WINDOW kernel: the 224x224 /
// 112x112 levels of ZF_UNET (lib/models/zf_unet.py:37-38,56-57) and their data gradients.
//
// Why another kernel (round 4): the wave-specialised conv_fprop_rw_kernel spends 58 of its 71 us on the 32 -> 32 @ 224 x 224
// layer in its block-wide hand-over skeleton (timing build with fetches, MFMAs and stores ALL removed: profiles/r04_rw_dbg.txt),
// not on bytes or MFMAs.  This kernel has no block-level synchronisation at all:
//   * one WAVE owns a strip of 16 * NF output columns and slides down SR rows of it.  The wave's whole weight matrix
//     (9 taps x Co x Ci <= 144 registers per lane) stays in REGISTERS as MFMA A operands, so LDS carries pixels only;
//   * an input row is loaded once (global -> registers -> the wave's private 3-row LDS ring), read back as three
//     column-shifted B fragments, and every fragment feeds the three kernel rows (dy = 0, 1, 2) x all output-channel
//     fragments: 6 * COF MFMAs per ds_read_b128 instead of 1-2 (conv_fprop_rw_kernel: 3 reads per 2 MFMAs = 75 % of the LDS
//     peak).  Three rolling accumulator rows: input row i adds to output rows i + 1, i, i - 1; there is no vertical halo
//     re-read inside a segment;
//   * the ring is private to the wave and a wave's LDS operations execute in order, so a row is published by program order
//     alone: no s_barrier, no flags.  Latency is hidden by eight independent waves per CU, each with three rows of loads
//     in flight;
//   * the global loads pass through registers, which is where the BatchNorm + ReLU of the PRODUCING layer (forward) or the
//     BatchNorm-backward apply (data gradient) can be applied on the way in (RollArgs::tf) -- the activated / dy tensors
//     of those layers then never exist in memory;
//   * accumulators are transposed (MFMA A = weights, B = pixels: a lane holds 4 consecutive channels of one pixel); two
//     16-channel fragments are merged into 16-byte rows by v_permlane16_swap, so an output row leaves as ONE fully coalesced
//     global_store_dwordx4 per 32 channels, straight from registers.
// LDS image of a row: pixel j (column c0 - 1 + j) at j * 2 * CI bytes, its 16-byte channel chunk q at slot q ^ sw(j); sw
// was searched against the ds_read_b128 lane groups of MI355X_MICROARCH.md (LDS table): conflict-free for every column
// shift, and the ds_write_b128 of a loaded row (8 consecutive lanes = 128 contiguous bytes) as well.
#include "fprop_dma.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct RollArgs {
    const bf16_t* x;
    const bf16_t* w;
    const float* bias;
    int bias_n;
    bf16_t* out;
    double* stats;
    unsigned x_bytes, w_bytes, out_bytes;
    int N, H, W, Ci, Co, ld_x, ld_out, Ktot;      // H x W: the OUTPUT grid
    int Hi, Wi;           // input tensor; input row of output row r under window row dy = r + dy + dhmin (dhmin = -1: padding 1,
    int dhmin, dwmin;     // 0: no padding -- lib/models/linknet.py:60 -- -2: its data gradient)
    int tap[9];           // packed-matrix tap index of window position (dy, dx) at [dy * 3 + dx]
    int SR, NSEG, NSTRIP, NTASK;
    // fused BatchNorm-backward reduction of the layer that produced this data gradient's forward input
    // (segnb_conv_fprop_bnreduce): see FdArgs
    const bf16_t* bn_y;
    unsigned bn_y_bytes;
    int bn_ld;
    const float* bn_coef;
    double* bn_sums;
    int bn_act;
    float bn_slope;
    // transform of the input on its way into LDS (template parameter TF):
    //   TF = 1  x holds the PRE-BatchNorm output y of the producing layer; the convolution's operand is
    //           round(drop * act((y - mean) * scale + shift)) -- bn_act_fwd_kernel's expression -- and never exists in memory
    //   TF = 2  x holds the gradient g (or dz) of this layer's activation, x2 its pre-BatchNorm output y; the operand is
    //           dy = round(a * (dz - c1 - yhat * c2)), dz = round(g * drop * act'(z)) -- bn_bwd_apply_kernel's direct form
    const bf16_t* x2;
    unsigned x2_bytes;
    int ld_x2;
    const float* tf_coef;     // [4][tf_Cp]: scale, shift, mean, invstd (segnb_bn_finalize)
    const float* tf_bcoef;    // [3][tf_Cp]: a, c1, c2 (segnb_bn_bwd_finalize); TF = 2 only
    const float* tf_drop;     // [N][tf_Cp] Dropout2d multipliers or NULL
    int tf_Cp, tf_act;
    float tf_slope;
};

template <int CI>
__device__ __forceinline__ int roll_sw(int j) {
    if constexpr (CI == 32) return ((j >> 2) & 1) << 1;
    else return (((j >> 1) & 1) << 1) | (((j >> 2) & 1) << 2);
}

__device__ __forceinline__ void unpack8(const u32x4_t& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// 8 per-channel constants from an LDS table, re-read at every use: the index is laundered through an empty asm so that the
// loop-invariant loads are not hoisted back into (40-56) registers
__device__ __forceinline__ void lds_const8(const float* table, int idx, float (&v)[8]) {
    asm volatile("" : "+v"(idx));
    const float4 a = *reinterpret_cast<const float4*>(table + idx);
    const float4 b = *reinterpret_cast<const float4*>(table + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

typedef short s16x2_t __attribute__((ext_vector_type(2)));
// ReLU of two packed bf16 values: a negative bf16 is a negative int16 (v_pk_max_i16; -0 -> +0)
__device__ __forceinline__ unsigned relu_pk2bf(unsigned pk) {
    const s16x2_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, pk), z));
}

// KS: 32-channel K steps of the input (Ci = 32 * KS); COF: 16-channel output fragments (Co <= 16 * COF, COF even);
// NF: 16-pixel fragments per strip row; EPI: 0 plain, 1 BatchNorm statistics of the output, 2 BatchNorm-backward reduction
// of the producing layer, 3 the activation mask of a producing layer WITHOUT BatchNorm (bn_coef NULL: bn_y is the activated
// tensor, the row is stored as dz = round(g * act'(y)) after the y row has landed -- behind the step's MFMAs instead of before
// them -- and sum dz is accumulated); TF: input transform (RollArgs); DL: rows of global loads in flight per wave (register sets);
// WPS: waves per SIMD the register allocation is held to (2: <= 256 registers, 1: <= 512)
// KSP / CSP: waves per strip that split the INPUT channels (32 each; partial sums reduced through LDS, the epilogue of output row o
// done by wave o % KSP) / the OUTPUT channels (16 * COF each, independent).  With a split a block is ONE strip (KSP * CSP waves)
template <int KS, int COF, int NF, int EPI, int TF, int DL, int WPS, int KSP = 1, int CSP = 1>
__global__ __launch_bounds__((KSP * CSP == 1 ? 4 : KSP * CSP) * 64, WPS) void conv_roll_kernel(const RollArgs a) {
    constexpr int WST = KSP * CSP, NWB = WST == 1 ? 4 : WST, SPB = NWB / WST, NTHR = NWB * 64;
    static_assert(WST == 1 || (KS == 1 && TF == 0 && EPI < 2), "split strips: 32 input channels per wave, plain operands");
    constexpr int CI = 32 * KS, CPP = CI / 8, PXB = CI * 2, RW = 16 * NF + 2, ROWB = RW * PXB;
    constexpr int NLD = (RW * CPP + 63) / 64;
    constexpr int NP = COF / 2;                      // 32-channel output pairs
    constexpr bool STATS = EPI == 1, BNRED = EPI == 2 || EPI == 3, MASKST = EPI == 3;
    constexpr int UN = DL % 3 == 0 ? DL : 3 * DL;    // unroll: ring slot (i % 3) and register set (i % DL) both static
    static_assert(COF % 2 == 0, "output fragments are stored in pairs");
    static_assert(TF == 0 || 64 % CPP == 0, "a lane's channel chunk must not depend on the load instruction");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NWB][3 * ROWB];
    __shared__ double red[2][16 * COF * CSP];
    // K split: partial accumulator rows, [writer wave][row parity][fragment (f, h)][lane] float4
    constexpr int XFR = NF * COF;
    __shared__ __attribute__((aligned(16))) float xch[KSP > 1 ? KSP * 2 * XFR * 64 * 4 : 4];
    // per-channel constants of the input transform / the BatchNorm-reduce epilogue live in LDS, not in registers (the weights
    // hold 72-144 of the 256): a lane re-reads the 8 values of its chunk where it uses them (same address in 16 lanes: broadcast)
    __shared__ __attribute__((aligned(16))) float tfc[TF == 0 ? 1 : 5][TF == 0 ? 4 : CI];
    __shared__ __attribute__((aligned(16))) float bnc[2][BNRED ? 16 * COF : 4];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const my = smem[wave];
    const int n16 = lane & 15, g = lane >> 4;
    const int sub = wave % WST, kpart = sub % KSP, cpart = sub / KSP, sidx = wave / WST;
    const int co0 = cpart * 16 * COF;                   // first output channel of this wave

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(TF == 2 ? a.x2 : a.x), 0, TF == 2 ? (int)a.x2_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(BNRED ? a.bn_y : a.x), 0, BNRED ? (int)a.bn_y_bytes : 0, 0x00020000);

    if (STATS || BNRED) {
        for (int i = threadIdx.x; i < 2 * 16 * COF * CSP; i += NTHR) (&red[0][0])[i] = 0.0;
        __syncthreads();
    }

    // ---- resident weights: A fragment (dy, dx, h, ks): lane (m = n16, kg = g) holds W[h * 16 + m][tap][ks * 32 + 8 kg ..]
    bf16x8_t Wf[9][COF][KS];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int h = 0; h < COF; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int co = co0 + h * 16 + n16;
                const unsigned off = co < a.Co ? (unsigned)(co * a.Ktot + a.tap[t] * a.Ci + kpart * 32 + ks * 32 + g * 8) * 2u : OOB;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0);
                Wf[t][h][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
    // after the swap a lane owns the 16-byte chunk cidx of pixel n16 of every 32-channel pair
    const int cidx = ((g & 1) << 1) | (g >> 1);
    float bs[COF][4];
#pragma unroll
    for (int h = 0; h < COF; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + h * 16 + g * 4 + e;
            bs[h][e] = (!BNRED && a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;      // (a data gradient has no bias)
        }
    // BatchNorm-reduce epilogue: z = y * sc + shh decides act'(z); the second sum is taken over dz * y (raw) and centred at
    // the flush, in fp64: sum dz * (y - mean) = sum dz * y - mean * sum dz (two constants per channel instead of three)
    float bneg = 0.f;
    if constexpr (BNRED) {
        for (int c = threadIdx.x; c < 16 * COF; c += NTHR) {
            const bool in = c < a.Co;
            if constexpr (MASKST) {                    // z = y: the activated value carries the sign
                bnc[0][c] = 1.f;
                bnc[1][c] = 0.f;
            } else {
                bnc[0][c] = in ? a.bn_coef[c] : 0.f;
                bnc[1][c] = in ? a.bn_coef[a.Co + c] - a.bn_coef[2 * a.Co + c] * a.bn_coef[c] : 0.f;
            }
        }
        bneg = a.bn_act == SEGNB_ACT_RELU ? 0.f : (a.bn_act == SEGNB_ACT_LEAKY ? a.bn_slope : 1.f);
    }
    float s1[NP][8], s2[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[p][e] = s2[p][e] = 0.f;

    // ---- input transform: per-channel constants of THIS lane's chunk (lane & (CPP - 1): the same for every load)
    // TF = 1: exactly bn_act_fwd_kernel's expression.  TF = 2 in a folded form (5 constants per channel instead of 7; the
    // registers are the budget here): with z = y * sc + shh (shh = shift - mean * scale), sel = act'(z),
    //   dy = a * (g * drop * sel) + (B * y + C),   B = -a * c2 * invstd,  C = a * (c2 * invstd * mean - c1)
    // = a * (dz - c1 - yhat * c2) of bn_bwd_apply_kernel up to fp32 rounding (dz is not rounded to bf16 in between: it is
    // exact for ReLU without dropout, the timed configuration's direct layers, and for dz inputs with act = NONE).
    float tneg = 0.f;
    const int tc0 = (lane % CPP) * 8;
    if constexpr (TF != 0) {
        for (int c = threadIdx.x; c < CI; c += NTHR) {
            const float sc = a.tf_coef[c], sh = a.tf_coef[a.tf_Cp + c], mu = a.tf_coef[2 * a.tf_Cp + c];
            tfc[0][c] = sc;
            if constexpr (TF == 1) {
                tfc[1][c] = sh;
                tfc[2][c] = mu;
            } else {
                const float is = a.tf_coef[3 * a.tf_Cp + c];
                const float ba = a.tf_bcoef[c], c1 = a.tf_bcoef[a.tf_Cp + c], c2 = a.tf_bcoef[2 * a.tf_Cp + c];
                tfc[1][c] = sh - mu * sc;
                tfc[2][c] = ba;
                tfc[3][c] = -ba * c2 * is;
                tfc[4][c] = ba * (c2 * is * mu - c1);
            }
        }
        tneg = a.tf_act == SEGNB_ACT_RELU ? 0.f : (a.tf_act == SEGNB_ACT_LEAKY ? a.tf_slope : 1.f);
    }
    if (TF != 0 || BNRED) __syncthreads();

    // ---- per-lane constants of the row image
    int woff[NLD];                 // LDS byte offset this lane's chunk of load m lands at (-1: no chunk)
#pragma unroll
    for (int m = 0; m < NLD; ++m) {
        const int L = 64 * m + lane;
        const int j = L / CPP, c = L % CPP;
        woff[m] = j < RW ? j * PXB + ((c ^ roll_sw<CI>(j)) << 4) : -1;
    }
    int roff[3][NF][KS];           // fragment (dx, f, ks): pixel j = 16 f + n16 + dx, chunk q = 4 ks + g
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int j = 16 * f + n16 + dx;
                roff[dx][f][ks] = j * PXB + (((4 * ks + g) ^ roll_sw<CI>(j)) << 4);
            }

    // K split (KSP > 1): wave k of a strip takes input channels [32 k, 32 k + 32), the partial sums meet in LDS
    const int src_ld = a.ld_x;
    const int src_ch = KSP > 1 ? kpart * 32 : 0;
    const int nstrips = gridDim.x * SPB;
    for (int task = blockIdx.x * SPB + sidx; task < a.NTASK; task += nstrips) {
        const int strip = task % a.NSTRIP;
        const int t2 = task / a.NSTRIP;
        const int seg = t2 % a.NSEG, n = t2 / a.NSEG;
        const int r0 = seg * a.SR;
        const int rows = min(a.SR, a.H - r0);
        const int nin = rows + 2;
        const int c0 = strip * 16 * NF;

        unsigned coff[NLD], coff2[NLD];
        bool colv[NLD];
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int L = 64 * m + lane;
            const int col = c0 + a.dwmin + L / CPP;
            colv[m] = woff[m] >= 0 && (unsigned)col < (unsigned)a.Wi;
            coff[m] = colv[m] ? (unsigned)(col * src_ld * 2 + src_ch * 2 + (L % CPP) * 16) : OOB;
            coff2[m] = (TF == 2 && colv[m]) ? (unsigned)(col * a.ld_x2 * 2 + (L % CPP) * 16) : OOB;
        }
        // TF = 1: the Dropout2d multiplier of the image (>= 0) folds into the affine map, drop * act(z) = act(drop * z) for
        // the "negative side scaled" activations: two constants per channel, in registers, and for ReLU the activation is one
        // packed integer maximum on the rounded pair (bn_act_fwd_kernel rounds drop * act(z) computed from (y - mean) * scale
        // + shift: the folded form differs by fp32 rounding before the one bf16 rounding)
        float q1[8], q0[8];
        if constexpr (TF == 1) {
            float tsc[8], tsh[8], tmu[8];
            lds_const8(tfc[0], tc0, tsc);
            lds_const8(tfc[1], tc0, tsh);
            lds_const8(tfc[2], tc0, tmu);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dm = a.tf_drop != nullptr ? a.tf_drop[n * a.tf_Cp + tc0 + e] : 1.f;
                q1[e] = tsc[e] * dm;
                q0[e] = (tsh[e] - tmu[e] * tsc[e]) * dm;
            }
        }
        u32x4_t ld[DL][NLD], ld2[TF == 2 ? DL : 1][NLD];
        auto issue = [&](auto set_c, int i) {
            constexpr int set = decltype(set_c)::value;
            const int gi = r0 + a.dhmin + i;
            const bool rv = i < nin && (unsigned)gi < (unsigned)a.Hi;
            const unsigned pixrow = (unsigned)((n * a.Hi + gi) * a.Wi);
            const unsigned rowoff = pixrow * (unsigned)(a.ld_x * 2);
#pragma unroll
            for (int m = 0; m < NLD; ++m) {
                const unsigned voff = rv ? rowoff + coff[m] : OOB;
                ld[set][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff, 0, 0);
                if constexpr (TF == 2) {
                    const unsigned voff2 = rv ? pixrow * (unsigned)(a.ld_x2 * 2) + coff2[m] : OOB;
                    ld2[set][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (int)voff2, 0, 0);
                }
            }
        };
        // row i held in register set `set` -> ring slot `slot`, transformed on the way
        auto publish = [&](auto set_c, auto slot_c, int i) {
            constexpr int set = decltype(set_c)::value, slot = decltype(slot_c)::value;
            const int gi = r0 + a.dhmin + i;
            const bool rv = (unsigned)gi < (unsigned)a.Hi;     // rows outside the image are zero AFTER the transform
#pragma unroll
            for (int m = 0; m < NLD; ++m) {
                u32x4_t v = ld[set][m];
                if constexpr (TF == 1) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = f[e] * q1[e] + q0[e];
                    const bool ok = rv && colv[m];
                    if (tneg == 0.f) {             // ReLU (wave-uniform)
                        v.x = ok ? relu_pk2bf(pack2bf(f[0], f[1])) : 0u;
                        v.y = ok ? relu_pk2bf(pack2bf(f[2], f[3])) : 0u;
                        v.z = ok ? relu_pk2bf(pack2bf(f[4], f[5])) : 0u;
                        v.w = ok ? relu_pk2bf(pack2bf(f[6], f[7])) : 0u;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * tneg;
                        v.x = ok ? pack2bf(f[0], f[1]) : 0u;
                        v.y = ok ? pack2bf(f[2], f[3]) : 0u;
                        v.z = ok ? pack2bf(f[4], f[5]) : 0u;
                        v.w = ok ? pack2bf(f[6], f[7]) : 0u;
                    }
                } else if constexpr (TF == 2) {
                    float gq[8], yq[8], tsc[8], tsh[8], tmu[8], tB[8], tC[8];
                    unpack8(v, gq);
                    unpack8(ld2[set][m], yq);
                    lds_const8(tfc[0], tc0, tsc);
                    lds_const8(tfc[1], tc0, tsh);
                    lds_const8(tfc[2], tc0, tmu);
                    lds_const8(tfc[3], tc0, tB);
                    lds_const8(tfc[4], tc0, tC);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = yq[e] * tsc[e] + tsh[e];
                        const float dzf = gq[e] * (z > 0.f ? 1.f : tneg);
                        gq[e] = tmu[e] * dzf + (tB[e] * yq[e] + tC[e]);
                    }
                    const bool ok = rv && colv[m];
                    v.x = ok ? pack2bf(gq[0], gq[1]) : 0u;
                    v.y = ok ? pack2bf(gq[2], gq[3]) : 0u;
                    v.z = ok ? pack2bf(gq[4], gq[5]) : 0u;
                    v.w = ok ? pack2bf(gq[6], gq[7]) : 0u;
                }
                if (woff[m] >= 0) *reinterpret_cast<u32x4_t*>(my + slot * ROWB + woff[m]) = v;
            }
        };

        f32x4_t acc[3][NF][COF];
        u32x4_t yv[3][NF][NP];      // BatchNorm-reduce epilogue: y rows of the output rows, requested three steps ahead

        static_for<DL>([&](auto k_c) { issue(k_c, decltype(k_c)::value); });
        publish(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        issue(std::integral_constant<int, 0>{}, DL);

        auto issue_y = [&](auto set_c, int o) {
            constexpr int set = decltype(set_c)::value;
            const bool rvo = o >= 1 && o <= rows;
            const unsigned prow = (unsigned)((n * a.H + r0 - 1 + o) * a.W);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int col = c0 + 16 * f + n16;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int ch = p * 32 + cidx * 8;
                    const bool ok = rvo && col < a.W && ch < a.Co;
                    const unsigned yoff = ok ? ((prow + (unsigned)col) * (unsigned)a.bn_ld + (unsigned)ch) * 2u : OOB;
                    yv[set][f][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)yoff, 0, 0);
                }
            }
        };
        // step i (UI = i % UN): row i is in ring slot i % 3; row i + 1 waits in register set (i + 1) % DL
        auto step = [&](auto u_c, int i) {
            constexpr int UI = decltype(u_c)::value;
            constexpr int U = UI % 3, U1 = (U + 1) % 3, U2 = (U + 2) % 3, S1 = (UI + 1) % DL;
            // -- E1: output row o = i - 2 (accumulator slot U1, final since step i - 1) leaves
            const int o = i - 2;
            const bool rowv = o >= 1 && o <= rows;
            bool ev = rowv;
            if constexpr (KSP > 1) {
                // K split: every wave of the strip holds a partial sum of the row; wave o % KSP gathers the others' and does the
                // row's epilogue (rotating: the epilogue work is spread over the waves).  One LDS barrier per step for the
                // block (= the strip: all its waves run the same steps); slots alternate with the step's parity.
                const bool owner = ((o + 3 * KSP) % KSP) == kpart;
                float* const xw = xch + (size_t)((kpart * 2 + (i & 1)) * XFR) * 256;
                if (rowv && !owner) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int h = 0; h < COF; ++h)
                            *reinterpret_cast<f32x4_t*>(xw + ((f * COF + h) * 64 + lane) * 4) = acc[U1][f][h];
                }
                lds_barrier();
                if (rowv && owner) {
#pragma unroll
                    for (int k2 = 1; k2 < KSP; ++k2) {
                        const int kp = (kpart + k2) % KSP;
                        const float* const xr = xch + (size_t)((kp * 2 + (i & 1)) * XFR) * 256;
#pragma unroll
                        for (int f = 0; f < NF; ++f)
#pragma unroll
                            for (int h = 0; h < COF; ++h)
                                acc[U1][f][h] += *reinterpret_cast<const f32x4_t*>(xr + ((f * COF + h) * 64 + lane) * 4);
                    }
                }
                ev = rowv && owner;
            }
            u32x4_t P[NF][NP];
            if (ev) {
                const int go = r0 - 1 + o;
                const unsigned prow = (unsigned)((n * a.H + go) * a.W);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int col = c0 + 16 * f + n16;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const f32x4_t A0 = acc[U1][f][2 * p], A1 = acc[U1][f][2 * p + 1];
                        unsigned x0 = pack2bf(A0[0] + bs[2 * p][0], A0[1] + bs[2 * p][1]);
                        unsigned x1 = pack2bf(A0[2] + bs[2 * p][2], A0[3] + bs[2 * p][3]);
                        unsigned y0 = pack2bf(A1[0] + bs[2 * p + 1][0], A1[1] + bs[2 * p + 1][1]);
                        unsigned y1 = pack2bf(A1[2] + bs[2 * p + 1][2], A1[3] + bs[2 * p + 1][3]);
                        // rows (16 lanes) 0..3 hold channels 4 g .. 4 g + 3 of each fragment; swap -> 8 consecutive channels
                        auto r0s = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                        auto r1s = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                        u32x4_t v;
                        v.x = r0s[0]; v.y = r1s[0]; v.z = r0s[1]; v.w = r1s[1];
                        P[f][p] = v;
                        const int ch = co0 + p * 32 + cidx * 8;
                        const bool ok = col < a.W && ch < a.Co;
                        const unsigned voff = ok ? ((prow + (unsigned)col) * (unsigned)a.ld_out + (unsigned)ch) * 2u : OOB;
                        if constexpr (!MASKST) __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (int)voff, 0, 0);
                    }
                }
            }
            // -- fragments of row i
            bf16x8_t fr[3][NF][KS];
            if (i < nin) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            fr[dx][f][ks] = *reinterpret_cast<const bf16x8_t*>(my + U * ROWB + roff[dx][f][ks]);
            }
            // -- row i + 1 into the ring, its register set reloads row i + 1 + DL
            if (i + 1 < nin) publish(std::integral_constant<int, S1>{}, std::integral_constant<int, U1>{}, i + 1);
            if (i + 1 + DL < nin) issue(std::integral_constant<int, S1>{}, i + 1 + DL);
            // -- MFMAs: kernel row dy of input row i adds to output row i + 1 - dy
            if (i < nin) {
                if (i + 1 <= rows) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int h = 0; h < COF; ++h) {
                            f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks)
                                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[dx][h][ks], fr[dx][f][ks], c, 0, 0, 0);
                            acc[U1][f][h] = c;
                        }
                }
                if (i >= 1 && i <= rows) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int h = 0; h < COF; ++h)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks)
                                    acc[U][f][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[3 + dx][h][ks], fr[dx][f][ks],
                                                                                             acc[U][f][h], 0, 0, 0);
                }
                if (i >= 2) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int h = 0; h < COF; ++h)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks)
                                    acc[U2][f][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[6 + dx][h][ks], fr[dx][f][ks],
                                                                                              acc[U2][f][h], 0, 0, 0);
                }
            }
            // -- E2: statistics of the row stored in E1 (on the values as stored)
            if ((STATS || BNRED) && ev) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float m = (c0 + 16 * f + n16) < a.W ? 1.f : 0.f;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        float v[8];
                        unpack8(P[f][p], v);
                        if constexpr (STATS) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float fm = v[e] * m;
                                s1[p][e] += fm;
                                s2[p][e] += fm * fm;
                            }
                        } else {
                            // dz = round(g * act'(z)), z = (y - mean) * scale + shift: the arithmetic of bn_act_bwd_reduce_kernel
                            float yq[8], bsc[8], bsh[8];
                            unpack8(yv[U1][f][p], yq);
                            if constexpr (MASKST) {
                                // the row leaves HERE, masked: dz = round(g * act'(y)), y the activated tensor itself
                                unsigned pk[4];
#pragma unroll
                                for (int e = 0; e < 8; e += 2) {
                                    const float d0 = v[e] * (yq[e] > 0.f ? 1.f : bneg), d1 = v[e + 1] * (yq[e + 1] > 0.f ? 1.f : bneg);
                                    pk[e >> 1] = pack2bf(d0, d1);
                                    s1[p][e] += __uint_as_float(pk[e >> 1] << 16) * m;
                                    s1[p][e + 1] += __uint_as_float(pk[e >> 1] & 0xffff0000u) * m;
                                }
                                u32x4_t dzv;
                                dzv.x = pk[0]; dzv.y = pk[1]; dzv.z = pk[2]; dzv.w = pk[3];
                                const int col = c0 + 16 * f + n16, ch = co0 + p * 32 + cidx * 8;
                                const bool ok = col < a.W && ch < a.Co;
                                const unsigned voff =
                                    ok ? (((unsigned)((n * a.H + r0 - 1 + o) * a.W) + (unsigned)col) * (unsigned)a.ld_out + (unsigned)ch) * 2u : OOB;
                                __builtin_amdgcn_raw_buffer_store_b128(dzv, rs_o, (int)voff, 0, 0);
                                continue;
                            }
                            lds_const8(bnc[0], p * 32 + cidx * 8, bsc);
                            lds_const8(bnc[1], p * 32 + cidx * 8, bsh);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float z = yq[e] * bsc[e] + bsh[e];
                                const float dv = bf16_bits_to_f32(f32_to_bf16_bits(v[e] * (z > 0.f ? 1.f : bneg))) * m;
                                s1[p][e] += dv;
                                s2[p][e] += dv * yq[e];
                            }
                        }
                    }
                }
            }
            // -- the y row of output row i + 1 (stored at step i + 3): set (i + 1) % 3 = U1, free since E2 above
            if constexpr (BNRED) issue_y(std::integral_constant<int, U1>{}, i + 1);
        };
        for (int ib = 0; ib < nin + 1; ib += UN)
            static_for<UN>([&](auto u_c) { step(u_c, ib + decltype(u_c)::value); });
    }

    if constexpr (STATS || BNRED) {
        // fp32 per lane over its rows, fp64 from here on: 16 pixel lanes of a row -> LDS -> one atomic per channel and block
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                double v1 = (double)s1[p][e], v2 = (double)s2[p][e];
                if constexpr (MASKST) {
                    v2 = 0.0;
                } else if constexpr (BNRED) {
                    const int c = p * 32 + cidx * 8 + e;
                    const double mu = c < a.Co ? (double)a.bn_coef[2 * a.Co + c] : 0.0;
                    const double is = c < a.Co ? (double)a.bn_coef[3 * a.Co + c] : 0.0;
                    v2 = (v2 - mu * v1) * is;                                  // sum dz * (y - mean) * invstd = sum dz * yhat
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    v1 += __shfl_xor(v1, o);
                    v2 += __shfl_xor(v2, o);
                }
                if (n16 == 0) {
                    atomicAdd(&red[0][co0 + p * 32 + cidx * 8 + e], v1);
                    atomicAdd(&red[1][co0 + p * 32 + cidx * 8 + e], v2);
                }
            }
        __syncthreads();
        double* const acc_out = BNRED ? a.bn_sums : a.stats;
        for (int i = threadIdx.x; i < 2 * 16 * COF * CSP; i += NTHR) {
            const int which = i / (16 * COF * CSP), col = i % (16 * COF * CSP);
            if (col < a.Co)
                atomicAdd(&acc_out[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + col], red[which][col]);
        }
    }
}

template <int KS, int COF, int NF, int TF, int DL, int WPS>
int launch_roll(RollArgs& a, hipStream_t stream) {
    a.NSTRIP = (a.W + 16 * NF - 1) / (16 * NF);
    // segment height: the whole launch should be about one round of the chip's wave slots (WPS blocks x 4 waves per CU)
    // (the most segments per strip that still give every wave at most ONE task; short segments when even whole strips
    // outnumber the wave slots)
    const int slots = segnb_knob_conv_cus() * 4 * WPS;       // (segnb_tune "conv_cu_pct": CUs left to a data-parallel job's collectives)
    int nseg = (int)(slots / ((long long)a.N * a.NSTRIP));
    if (nseg < 1) nseg = (a.H + 15) / 16;
    if (nseg > a.H) nseg = a.H;
    const int sr = (a.H + nseg - 1) / nseg;
    a.SR = sr;
    a.NSEG = (a.H + sr - 1) / sr;
    a.NTASK = a.N * a.NSEG * a.NSTRIP;
    int blocks = (a.NTASK + 3) / 4;
    const int maxb = segnb_knob_conv_cus() * WPS;
    if (blocks > maxb) blocks = maxb;
    if (a.stats != nullptr)
        hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, 1, TF, DL, WPS>), dim3(blocks), dim3(256), 0, stream, a);
    else if (a.bn_y != nullptr && a.bn_coef == nullptr) {
        if constexpr (TF == 0 && NF == 1)
            hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, 3, TF, DL, WPS>), dim3(blocks), dim3(256), 0, stream, a);
        else
            return -1;
    } else if (a.bn_y != nullptr)
        hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, 2, TF, DL, WPS>), dim3(blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, 0, TF, DL, WPS>), dim3(blocks), dim3(256), 0, stream, a);
    return 0;
}

// strips whose waves split the input channels (KSP) or the output channels (CSP): one block per strip
template <int NF, int KSP, int CSP>
int launch_roll_split(RollArgs& a, hipStream_t stream) {
    constexpr int WST = KSP * CSP;
    a.NSTRIP = (a.W + 16 * NF - 1) / (16 * NF);
    const int per_cu = 8 / WST;                                  // strips in flight per CU at two waves per SIMD
    const int slots = segnb_knob_conv_cus() * per_cu;
    int nseg = (int)(slots / ((long long)a.N * a.NSTRIP));
    if (nseg < 1) nseg = (a.H + 15) / 16;
    if (nseg > a.H) nseg = a.H;
    const int sr = (a.H + nseg - 1) / nseg;
    a.SR = sr;
    a.NSEG = (a.H + sr - 1) / sr;
    a.NTASK = a.N * a.NSEG * a.NSTRIP;
    int blocks = a.NTASK;
    if (blocks > slots) blocks = slots;
    if (a.stats != nullptr)
        hipLaunchKernelGGL((conv_roll_kernel<1, 2, NF, 1, 0, 3, 2, KSP, CSP>), dim3(blocks), dim3(WST * 64), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_roll_kernel<1, 2, NF, 0, 0, 3, 2, KSP, CSP>), dim3(blocks), dim3(WST * 64), 0, stream, a);
    return 0;
}

}  // namespace

// 1 = handled, 0 = not applicable, else error
int segnb_fprop_roll_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked, unsigned w_bytes,
                         const float* bias, int bias_n, void* out, double* stats, hipStream_t stream,
                         const segnb_bn_reduce_epilogue* bn, const segnb_operand_tf* tf, const segnb_upcat_src* uc) {
    const int knob = segnb_knob_fprop_roll();
    if (!knob) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo) return 0;
    if (g->Co % 8 != 0 || g->Wo < 32 || g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return 0;
    // same-size (padding 1) everywhere; another padding (the valid 3 x 3 convolution of lib/models/linknet.py:60 and its data
    // gradient: input and output grids differ) only for the plain one-wave form
    const bool same = g->Hi == g->Ho && g->Wi == g->Wo && dhmin == -1 && dwmin == -1;
    // (the epilogues address y by OUTPUT pixel: they do not care about the input grid)
    if (!same && (tf != nullptr || uc != nullptr || g->Ci != 32 || g->Co > 32)) return 0;
    if (bn != nullptr && bn->coef == nullptr && (tf != nullptr || g->Ci != 32 || g->Co > 32)) return 0;      // (activation mask: plain form)
    // 32 -> <= 32: one wave per strip.  Wider inputs (64, 96 -> <= 32: the waves of a strip split K) and wider outputs
    // (32 -> 64, 96: they split the output channels) when nothing else rides on the launch
    // Measured (MI355X, bs=32, profiles/r04_ab.txt): two-way splits pay -- 32 -> 64 @ 112 x 112 38.2 -> 28.5 us, 64 -> 32 32.2 -> 28.3
    // us -- three-way splits do not: 96 -> 32 @ 224 x 224 with the K split 141 -> 208 us (three waves in lock step behind one
    // barrier per row lose what independent waves hide), 32 -> 96 with the channel split 134 -> 167 us (every wave re-loads the
    // whole input and writes a third of each pixel); those stay on conv_fprop_rw_kernel, and so does the virtual concat (uc).
    if (uc != nullptr) return 0;
    const bool plain1 = g->Ci == 32 && g->Co <= 32;
    const bool ksplit = g->Ci == 64 && g->Co <= 32 && bn == nullptr && tf == nullptr && knob >= 2;
    const bool csplit = g->Ci == 32 && g->Co == 64 && bn == nullptr && tf == nullptr && knob >= 2;
    if (!plain1 && !ksplit && !csplit) return 0;
    RollArgs a;
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        const int k = (g->dh[t] - dhmin) * 3 + (g->dw[t] - dwmin);
        if (seen[k]) return 0;
        seen[k] = true;
        a.tap[k] = t;
    }
    a.Hi = g->Hi; a.Wi = g->Wi; a.dhmin = dhmin; a.dwmin = dwmin;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.bias = bias;
    a.bias_n = bias_n;
    a.out = (bf16_t*)out;
    a.stats = stats;
    a.x_bytes = in_bytes;
    a.w_bytes = w_bytes;
    {
        const long long ob = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
        if (ob >= (1ll << 31)) return 0;
        a.out_bytes = (unsigned)ob;
    }
    a.N = g->N; a.H = g->Ho; a.W = g->Wo;
    a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
    a.Ktot = 9 * g->Ci;
    a.x2 = nullptr;
    a.tf_coef = a.tf_bcoef = a.tf_drop = nullptr;
    a.bn_y = nullptr;
    if (ksplit) {
        const int rc = launch_roll_split<2, 2, 1>(a, stream);
        return rc ? rc : 1;
    }
    if (csplit) {
        const int rc = launch_roll_split<2, 1, 2>(a, stream);
        return rc ? rc : 1;
    }
    if (bn != nullptr) {
        if (stats != nullptr) return 0;
        const long long yb = (((long long)g->N * g->Ho * g->Wo - 1) * bn->ld_y + g->Co) * 2;
        if (yb >= (1ll << 31)) return 0;
        a.bn_y = (const bf16_t*)bn->y;
        a.bn_y_bytes = (unsigned)yb;
        a.bn_ld = bn->ld_y;
        a.bn_coef = bn->coef;
        a.bn_sums = bn->sums;
        a.bn_act = bn->act;
        a.bn_slope = bn->slope;
    }
    int rc;
    if (tf != nullptr) {
        if (tf->Cp < g->Ci || tf->coef == nullptr) return 0;
        a.tf_coef = tf->coef;
        a.tf_bcoef = tf->bcoef;
        a.tf_drop = tf->drop;
        a.tf_Cp = tf->Cp;
        a.tf_act = tf->act;
        a.tf_slope = tf->slope;
        if (tf->kind == SEGNB_TF_ACT) {
            if (bn != nullptr) return 0;
            rc = launch_roll<1, 2, 1, 1, 3, 2>(a, stream);
        } else if (tf->kind == SEGNB_TF_BNBWD) {
            if (tf->y == nullptr || tf->bcoef == nullptr || stats != nullptr || tf->ld_y % 8 != 0) return 0;
            const long long yb = (((long long)g->N * g->Hi * g->Wi - 1) * tf->ld_y + g->Ci) * 2;
            if (yb >= (1ll << 31)) return 0;
            a.x2 = (const bf16_t*)tf->y;
            a.x2_bytes = (unsigned)yb;
            a.ld_x2 = tf->ld_y;
            rc = launch_roll<1, 2, 1, 2, 3, 2>(a, stream);
        } else {
            return 0;
        }
        return rc ? rc : 1;
    }
    // 32-column strips (more bytes in flight per wave, 6 % instead of 12 % halo columns) where the registers allow it: the
    // BatchNorm-reduce epilogue's per-channel constants do not fit beside them
    rc = (knob == 2 && bn == nullptr) ? launch_roll<1, 2, 2, 0, 3, 2>(a, stream) : launch_roll<1, 2, 1, 0, 3, 2>(a, stream);
    return rc ? rc : 1;
}

// include/segnb_hip.h: segnb_conv_fprop_bnreduce with coef == NULL (the plain one-wave form of segnb_fprop_roll_try, any padding)
extern "C" int segnb_conv_fprop_actmask_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || !segnb_knob_bnreduce_fused()) return 0;
    return segnb_fprop_roll_actmask_ok(g) || segnb_fprop_dma_actmask_ok(g) ? 1 : 0;
}

// conv_roll_kernel, EPI = 3 (32 -> <= 32 channels)
int segnb_fprop_roll_actmask_ok(const segnb_conv_geom* g) {
    if (!segnb_knob_fprop_roll() || getenv("SEGNB_FPROP_GENERAL") != nullptr) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci != 32 || g->Co > 32 || g->Co % 8 != 0 || g->Wo < 32) return 0;
    if (g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return 0;
    for (int t = 0; t < 9; ++t) {
        const int k = (g->dh[t] - dhmin) * 3 + (g->dw[t] - dwmin);
        if (seen[k]) return 0;
        seen[k] = true;
    }
    return (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2 < (1ll << 31) ? 1 : 0;
}

static bool roll_tf_geom_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || !segnb_knob_fprop_roll() || getenv("SEGNB_FPROP_GENERAL") != nullptr) return false;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Hi != g->Ho || g->Wi != g->Wo) return false;
    if (g->Ci != 32 || g->Co > 32 || g->Co % 8 != 0 || g->Wo < 32 || g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return false;
    for (int t = 0; t < 9; ++t)
        if (g->dh[t] < -1 || g->dh[t] > 1 || g->dw[t] < -1 || g->dw[t] > 1) return false;
    return (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2 < (1ll << 31);
}

// include/segnb_hip.h: convolution whose input operand is recomputed from what the producing layer left in memory
extern "C" int segnb_conv_fprop_tf_ok(const segnb_conv_geom* g, int dtype, int kind) {
    return (roll_tf_geom_ok(g, dtype) && (kind == SEGNB_TF_ACT || kind == SEGNB_TF_BNBWD)) ? 1 : 0;
}

extern "C" int segnb_conv_fprop_tf(const segnb_conv_geom* g, int dtype, const void* in, const segnb_operand_tf* tf,
                                   const void* wpacked, const float* bias, int bias_n, void* out, double* stats,
                                   const segnb_bn_reduce_epilogue* bn, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_tf, g, dtype, in, tf, wpacked, bias, bias_n, out, stats, bn, stream);
    SEGNB_CHECK_ARG(g && in && tf && wpacked && out, "NULL argument");
    SEGNB_CHECK_ARG(segnb_conv_fprop_tf_ok(g, dtype, tf->kind), "geometry not served (segnb_conv_fprop_tf_ok)");
    SEGNB_CHECK_ARG(tf->coef != nullptr && tf->Cp >= g->Ci && tf->Cp % 8 == 0, "bad transform coefficients");
    SEGNB_CHECK_ARG(tf->kind != SEGNB_TF_BNBWD || (tf->y != nullptr && tf->bcoef != nullptr && stats == nullptr && tf->drop == nullptr),
                    "bad BNBWD transform (a dropout layer hands over dz with act = NONE)");
    SEGNB_CHECK_ARG(tf->kind != SEGNB_TF_ACT || bn == nullptr, "a forward launch has no BatchNorm-reduce epilogue");
    SEGNB_CHECK_ARG(bn == nullptr || (bn->y && bn->coef && bn->sums && bn->ld_y >= g->Co && bn->ld_y % 8 == 0), "bad epilogue");
    const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    const long long wb = (long long)g->Co * g->ntaps * g->Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    const int rc = segnb_fprop_roll_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, bias, bias_n, out, stats, (hipStream_t)stream,
                                        bn, tf);
    if (rc != 1) {
        segnb_set_error("segnb_conv_fprop_tf: the kernel refused the launch (%d)", rc);
        return rc > 1 ? rc : SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}
