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
    int N, H, W, Ci, Co, ld_x, ld_out, Ktot;
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

// KS: 32-channel K steps of the input (Ci = 32 * KS); COF: 16-channel output fragments (Co <= 16 * COF, COF even);
// NF: 16-pixel fragments per strip row
template <int KS, int COF, int NF, bool STATS, bool BNRED>
__global__ __launch_bounds__(256, 2) void conv_roll_kernel(const RollArgs a) {
    constexpr int CI = 32 * KS, CPP = CI / 8, PXB = CI * 2, RW = 16 * NF + 2, ROWB = RW * PXB;
    constexpr int NLD = (RW * CPP + 63) / 64;
    constexpr int NP = COF / 2;                      // 32-channel output pairs
    static_assert(COF % 2 == 0, "output fragments are stored in pairs");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][3 * ROWB];
    __shared__ double red[2][16 * COF];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const my = smem[wave];
    const int n16 = lane & 15, g = lane >> 4;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(BNRED ? a.bn_y : a.x), 0, BNRED ? (int)a.bn_y_bytes : 0, 0x00020000);

    if (STATS || BNRED) {
        for (int i = threadIdx.x; i < 2 * 16 * COF; i += 256) (&red[0][0])[i] = 0.0;
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
                const int co = h * 16 + n16;
                const unsigned off = co < a.Co ? (unsigned)(co * a.Ktot + a.tap[t] * a.Ci + ks * 32 + g * 8) * 2u : OOB;
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
            const int co = h * 16 + g * 4 + e;
            bs[h][e] = (a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;
        }
    float bsc[NP][8], bsh[NP][8], bmu[NP][8];
    float bneg = 0.f;
    if constexpr (BNRED) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = p * 32 + cidx * 8 + e;
                const bool in = c < a.Co;
                bsc[p][e] = in ? a.bn_coef[c] : 0.f;
                bsh[p][e] = in ? a.bn_coef[a.Co + c] : 0.f;
                bmu[p][e] = in ? a.bn_coef[2 * a.Co + c] : 0.f;
            }
        bneg = a.bn_act == SEGNB_ACT_RELU ? 0.f : (a.bn_act == SEGNB_ACT_LEAKY ? a.bn_slope : 1.f);
    }
    float s1[NP][8], s2[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[p][e] = s2[p][e] = 0.f;

    // ---- per-lane constants of the row image
    int woff[NLD];                 // LDS byte offset this lane's chunk of load m lands at (-1: no chunk)
    int wj[NLD], wc[NLD];
#pragma unroll
    for (int m = 0; m < NLD; ++m) {
        const int L = 64 * m + lane;
        const int j = L / CPP, c = L % CPP;
        wj[m] = j;
        wc[m] = c;
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

    const int nwaves = gridDim.x * 4;
    for (int task = blockIdx.x * 4 + wave; task < a.NTASK; task += nwaves) {
        const int strip = task % a.NSTRIP;
        const int t2 = task / a.NSTRIP;
        const int seg = t2 % a.NSEG, n = t2 / a.NSEG;
        const int r0 = seg * a.SR;
        const int rows = min(a.SR, a.H - r0);
        const int nin = rows + 2;
        const int c0 = strip * 16 * NF;

        unsigned coff[NLD];
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int col = c0 - 1 + wj[m];
            coff[m] = (woff[m] >= 0 && (unsigned)col < (unsigned)a.W) ? (unsigned)(col * a.ld_x * 2 + wc[m] * 16) : OOB;
        }
        u32x4_t ld[3][NLD];
        auto issue = [&](auto set_c, int i) {
            constexpr int set = decltype(set_c)::value;
            const int gi = r0 - 1 + i;
            const bool rv = i < nin && (unsigned)gi < (unsigned)a.H;
            const unsigned rowbase = (unsigned)((n * a.H + gi) * a.W) * (unsigned)(a.ld_x * 2);
#pragma unroll
            for (int m = 0; m < NLD; ++m) {
                const unsigned voff = rv ? rowbase + coff[m] : OOB;
                ld[set][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff, 0, 0);
            }
        };
        auto publish = [&](auto set_c) {          // row held in register set `set` -> ring slot `set`
            constexpr int set = decltype(set_c)::value;
#pragma unroll
            for (int m = 0; m < NLD; ++m)
                if (woff[m] >= 0) *reinterpret_cast<u32x4_t*>(my + set * ROWB + woff[m]) = ld[set][m];
        };

        f32x4_t acc[3][NF][COF];
        u32x4_t yv[NF][NP];

        issue(std::integral_constant<int, 0>{}, 0);
        issue(std::integral_constant<int, 1>{}, 1);
        issue(std::integral_constant<int, 2>{}, 2);
        publish(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 0>{}, 3);

        // step i: row i is in ring slot U = i % 3 (and register set U is already reloading row i + 3)
        auto step = [&](auto u_c, int i) {
            constexpr int U = decltype(u_c)::value, U1 = (U + 1) % 3, U2 = (U + 2) % 3;
            // -- E1: output row o = i - 2 (accumulator slot U1, final since step i - 1) leaves
            const int o = i - 2;
            const bool ev = o >= 1 && o <= rows;
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
                        const int ch = p * 32 + cidx * 8;
                        const bool ok = col < a.W && ch < a.Co;
                        const unsigned voff = ok ? ((prow + (unsigned)col) * (unsigned)a.ld_out + (unsigned)ch) * 2u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (int)voff, 0, 0);
                        if constexpr (BNRED) {
                            const unsigned yoff = ok ? ((prow + (unsigned)col) * (unsigned)a.bn_ld + (unsigned)ch) * 2u : OOB;
                            yv[f][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)yoff, 0, 0);
                        }
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
            // -- row i + 1 into the ring, its register set reloads row i + 4
            if (i + 1 < nin) publish(std::integral_constant<int, U1>{});
            if (i + 4 < nin) issue(std::integral_constant<int, U1>{}, i + 4);
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
                            float yq[8];
                            unpack8(yv[f][p], yq);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float yc = yq[e] - bmu[p][e];
                                const float z = yc * bsc[p][e] + bsh[p][e];
                                const float dv = bf16_bits_to_f32(f32_to_bf16_bits(v[e] * (z > 0.f ? 1.f : bneg))) * m;
                                s1[p][e] += dv;
                                s2[p][e] += dv * yc;
                            }
                        }
                    }
                }
            }
        };
        for (int ib = 0; ib < nin + 1; ib += 3) {
            step(std::integral_constant<int, 0>{}, ib);
            step(std::integral_constant<int, 1>{}, ib + 1);
            step(std::integral_constant<int, 2>{}, ib + 2);
        }
    }

    if constexpr (STATS || BNRED) {
        // fp32 per lane over its rows, fp64 from here on: 16 pixel lanes of a row -> LDS -> one atomic per channel and block
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v1 = s1[p][e], v2 = s2[p][e];
                if constexpr (BNRED) {
                    const int c = p * 32 + cidx * 8 + e;
                    v2 *= c < a.Co ? a.bn_coef[3 * a.Co + c] : 0.f;           // * invstd: sum dz * yhat
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    v1 += __shfl_xor(v1, o);
                    v2 += __shfl_xor(v2, o);
                }
                if (n16 == 0) {
                    atomicAdd(&red[0][p * 32 + cidx * 8 + e], (double)v1);
                    atomicAdd(&red[1][p * 32 + cidx * 8 + e], (double)v2);
                }
            }
        __syncthreads();
        double* const acc_out = BNRED ? a.bn_sums : a.stats;
        for (int i = threadIdx.x; i < 2 * 16 * COF; i += 256) {
            const int which = i / (16 * COF), col = i % (16 * COF);
            if (col < a.Co)
                atomicAdd(&acc_out[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + col], red[which][col]);
        }
    }
}

template <int KS, int COF, int NF>
int launch_roll(RollArgs& a, hipStream_t stream) {
    a.NSTRIP = (a.W + 16 * NF - 1) / (16 * NF);
    // segment height: the whole launch should be about one round of the chip's wave slots (2 blocks x 4 waves per CU)
    const int slots = segnb_num_cus() * 8;
    int sr = a.H;
    for (int nseg = 1; nseg <= a.H; ++nseg) {
        sr = (a.H + nseg - 1) / nseg;
        if ((long long)a.N * a.NSTRIP * nseg >= slots * 7 / 8 || sr <= 14) break;
    }
    a.SR = sr;
    a.NSEG = (a.H + sr - 1) / sr;
    a.NTASK = a.N * a.NSEG * a.NSTRIP;
    int blocks = (a.NTASK + 3) / 4;
    const int maxb = segnb_num_cus() * 2;
    if (blocks > maxb) blocks = maxb;
    if (a.stats != nullptr)
        hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, true, false>), dim3(blocks), dim3(256), 0, stream, a);
    else if (a.bn_y != nullptr)
        hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, false, true>), dim3(blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_roll_kernel<KS, COF, NF, false, false>), dim3(blocks), dim3(256), 0, stream, a);
    return 0;
}

}  // namespace

// 1 = handled, 0 = not applicable, else error
int segnb_fprop_roll_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked, unsigned w_bytes,
                         const float* bias, int bias_n, void* out, double* stats, hipStream_t stream,
                         const segnb_bn_reduce_epilogue* bn) {
    const int knob = segnb_knob_fprop_roll();
    if (!knob) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Hi != g->Ho || g->Wi != g->Wo) return 0;
    if (g->Ci != 32 || g->Co > 32 || g->Co % 8 != 0 || g->Wo < 32 || g->ld_in % 8 != 0 || g->ld_out % 8 != 0) return 0;
    RollArgs a;
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        if (g->dh[t] < -1 || g->dh[t] > 1 || g->dw[t] < -1 || g->dw[t] > 1) return 0;
        const int k = (g->dh[t] + 1) * 3 + (g->dw[t] + 1);
        if (seen[k]) return 0;
        seen[k] = true;
        a.tap[k] = t;
    }
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
    a.bn_y = nullptr;
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
    // 32-column strips (more bytes in flight per wave, 6 % instead of 12 % halo columns) where the registers allow it: the
    // BatchNorm-reduce epilogue's per-channel constants do not fit beside them
    const int rc = (knob == 2 && bn == nullptr) ? launch_roll<1, 2, 2>(a, stream) : launch_roll<1, 2, 1>(a, stream);
    return rc ? rc : 1;
}
