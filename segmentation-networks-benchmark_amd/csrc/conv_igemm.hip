// Implicit-GEMM gather-convolution on MFMA for gfx950: forward / data-gradient (conv_fprop) and
// weight-gradient (conv_wgrad), NHWC activations, no im2col buffer.
//
// Replaces aten::convolution / aten::convolution_backward as reached from nn.Conv2d /
// nn.ConvTranspose2d in lib/models/zf_unet.py:8, linknet.py:12-21,41, tiramisu.py:14,65,
// unet16.py:17,38 (cuDNN / oneDNN in the reference).
//
// GEMM view (fprop):  M = N*QH*QW pixels, N = Co, K = ntaps*Ci.
//   A[m][k]  gathered on the fly: k -> (tap, ci), row m -> (n, qh, qw) -> input pixel, zero outside
//   B[co][k] packed weights, K contiguous
// Block = 256 threads = 4 waves; tile BM x BN; K step = 128 bytes of K per row (64 bf16 / 32 f32),
// register-staged double-buffered LDS, rows padded to 144 B (conflict-free ds_read_b128).
// Wave tile WM x WN out of v_mfma_f32_32x32x16_bf16 (bf16) / v_mfma_f32_32x32x2_f32 (exact fp32).
// Epilogue: +bias, round to T, per-channel sum / sum^2 of the stored values (BatchNorm batch
// statistics), staged through LDS so global stores are 16-byte, pixel-contiguous.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LDS_ROW = 144;  // 128 B of K + 16 B pad
constexpr int NT = 256;       // threads per block

struct FpropArgs {
    segnb_conv_geom g;
    const void* in;
    const void* w;
    const float* bias;
    int bias_n;
    const float* ep_coef;      // affine + activation epilogue (segnb_conv_fprop_act); ep_act < 0: off
    int ep_act;
    float ep_slope;
    void* out;
    double* stats;
    int M, Ktot, ksteps, MT, NTL, GM;
    unsigned in_bytes, w_bytes;          // extents of the two buffers (buffer-load bounds checks)
    // fused BatchNorm-backward REDUCTION in the store pass of a data-gradient launch (segnb_conv_fprop_bnreduce; template BNR):
    // the output is the gradient g of an activation a = act(BatchNorm(y)); with y the store threads accumulate sum dz and
    // sum dz * yhat, dz = round(g * act'(z)) -- segnb_bn_act_bwd_reduce without its own pass over (g, y).  FCDenseNet's dense
    // layers (tiramisu.py:9-20: norm -> relu -> conv): the data gradient 16 -> C of every layer, C up to ~1100
    const void* bn_y;
    int bn_ld;
    const float* bn_coef;                // [4][Co]: scale, shift, mean, invstd (segnb_bn_finalize)
    double* bn_sums;                     // [SEGNB_STAT_REPLICAS][2][Co]
    int bn_act;
    float bn_slope;
    // BNM 2 (segnb_conv_fprop_bnapply): the SECOND launch of a data gradient that is never stored -- the tile is recomputed (K is
    // 144 deep for a dense layer: the launch is its output's bytes), dz = round(g * act'(z)) again, and what leaves is the
    // BatchNorm-backward result dy = round(a * (dz - c1 - yhat * c2)) [+ the gradient already in bn_acc] with (a, c1, c2) from the
    // sums the first launch (BNM 1: sums only, nothing stored) completed -- segnb_bn_bwd_apply_fused_direct(_acc) without g in memory
    double bn_count;
    const float* bn_gamma;               // [bn_C] or NULL
    int bn_C;                            // real BatchNorm channels (<= Co)
    float* bn_bcoef;                     // [3][Co] out (written by the blocks of pixel tile 0)
    float* bn_dgamma;                    // [bn_C] +=
    float* bn_dbeta;
    void* bn_acc;                        // [N][Ho][Wo][bn_acc_ld] the BatchNorm input's gradient
    int bn_acc_ld, bn_accumulate;
    int bn_mode;                         // the BNM instantiation a BNR launch takes
    // segnb_conv_fprop_drop (conv_fprop_deepk_kernel): out = round(round(acc + bias) * drop[n][co]); statistics rows stats_ld apart
    const float* drop;
    int ld_drop, stats_ld;
};

struct WgradArgs {
    segnb_conv_geom g;
    const void* in;
    const void* dout;
    float* dwp;
    int M, Ktot, MT, NTL, S, steps_per_split, total_steps;
};

// buffer descriptor over [base, base + bytes): raw buffer (stride 0), loads beyond `bytes` return zeros
constexpr unsigned OOB_OFFSET = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ int xcd_remap(int b, int G) {
    // blocks b, b+8, ... share an XCD (round-robin dispatch): give each XCD a contiguous range of
    // logical tiles so neighbours (same pixels, different channel tile) hit one L2.  Bijective for any G.
    const int q = G >> 3, r = G & 7, x = b & 7, j = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// acc[TM][TN] += A_tile(rows am0.., K step) * B_tile(rows bn0..)^T ; tiles are [rows][LDS_ROW] images
template <typename T, int TM, int TN>
__device__ __forceinline__ void mma_step(const unsigned char* __restrict__ sA,
                                         const unsigned char* __restrict__ sB, int r, int h,
                                         f32x16_t (&acc)[TM][TN]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8_t af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8_t*>(sA + (32 * i + r) * LDS_ROW + kk * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8_t*>(sB + (32 * j + r) * LDS_ROW + kk * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    } else {
        // fp32: lane (r,h) feeds k = 16h + 4q + e at sub-step (q,e) -- any K permutation is a valid
        // contraction order as long as A and B agree; this one keeps the LDS reads 16-byte.
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const float4*>(sA + (32 * i + r) * LDS_ROW + h * 64 + q * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const float4*>(sB + (32 * j + r) * LDS_ROW + h * 64 + q * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bfr[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bfr[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bfr[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bfr[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
}

template <int BM, int BN, typename T>
constexpr int fprop_smem_bytes() {
    return 2 * (BM + BN) * LDS_ROW + BM * 16 + 4 * 2 * BN * 4 + SEGNB_MAX_TAPS * 8;
}

// ================================================================================================
// forward / data-gradient
// ================================================================================================
// BNM (with BNR): 0 the gradient is stored and reduced, 1 reduced only (nothing stored), 2 recomputed and APPLIED (FpropArgs::bn_acc)
template <typename T, int BM, int BN, int WM, int WN, bool EP = false, bool BNR = false, int BNM = 0>      // EP: affine + activation epilogue; BNR: FpropArgs::bn_y (separate instantiations)
__global__ __launch_bounds__(NT) void conv_fprop_kernel(const FpropArgs a) {
    static_assert(!BNR || (!EP && sizeof(T) == 2 && NT % (BN / 8) == 0 && NT >= 2 * BN), "BatchNorm-reduce store pass: bf16, one fixed channel chunk per thread");
    static_assert(BNM == 0 || BNR, "BNM selects the form of the BatchNorm store pass");
    __shared__ float sApp[BNM == 2 ? 4 * BN : 1];          // BNM 2: a, c1, c2, invstd of the block's channels
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TILE_BYTES = (BM + BN) * LDS_ROW;
    constexpr int OUT_ROW = BN * (int)sizeof(T) + 16;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves");
    static_assert(BM * OUT_ROW <= 2 * TILE_BYTES, "staging fits");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sTiles = smem;
    int4* sRow = reinterpret_cast<int4*>(smem + 2 * TILE_BYTES);
    float* sStat = reinterpret_cast<float*>(sRow + BM);
    int2* sTap = reinterpret_cast<int2*>(sStat + 4 * 2 * BN);      // sStat: one [2*BN] row per wave row

    const segnb_conv_geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int c = tid & 7, row0 = tid >> 3;

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = L % a.NTL;
    const int gq = L / a.NTL;
    const int n_base = nt * BN;

    T* __restrict__ outT = reinterpret_cast<T*>(a.out);

    if (tid < g.ntaps) sTap[tid] = make_int2(g.dh[tid], g.dw[tid]);

    const int QHW = g.QH * g.QW;
    double st = 0.0;
    const __amdgpu_buffer_rsrc_t rsrc_in = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.w, a.w_bytes);
    unsigned b_off[BI];                 // weight rows of this thread: byte offset of (co, k = c*EPC), constant
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int co = n_base + row0 + 32 * j;
        b_off[j] = co < g.Co ? (unsigned)(co * a.Ktot + c * EPC) * (unsigned)sizeof(T) : OOB_OFFSET;
    }

    // BNR: this thread stores channel chunk tid % OC of every row it stores: its two sums stay in registers, the per-channel
    // constants (mean, scale, shift) of the block's BN channels in LDS (sStat: a BNR launch takes no forward statistics) --
    // 24 registers that cost the 128 x 128 tile its second block per CU
    float bs1[BNR ? 8 : 1], bs2[BNR ? 8 : 1];
    float bneg = 0.f;
    if constexpr (BNR) {
        for (int c2 = tid; c2 < BN; c2 += NT) {
            const int ch = n_base + c2;
            const bool in = ch < g.Co;
            sStat[c2] = in ? a.bn_coef[2 * g.Co + ch] : 0.f;          // mean
            sStat[BN + c2] = in ? a.bn_coef[ch] : 0.f;                // scale
            sStat[2 * BN + c2] = in ? a.bn_coef[g.Co + ch] : 0.f;     // shift
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bs1[e] = bs2[e] = 0.f;
        bneg = a.bn_act == SEGNB_ACT_RELU ? 0.f : (a.bn_act == SEGNB_ACT_LEAKY ? a.bn_slope : 1.f);
        if constexpr (BNM == 2) {
            // the fused finalize of bn_bwd_apply_kernel (norm_act.hip: bn_bwd_coef): every block derives (a, c1, c2) of its
            // channels from the completed sums; the blocks of pixel tile 0 publish them and add dgamma / dbeta
            for (int c2 = tid; c2 < BN; c2 += NT) {
                const int ch = n_base + c2;
                float av = 0.f, c1 = 0.f, c2v = 0.f, is = 0.f;
                if (ch < a.bn_C) {
                    double v1[SEGNB_STAT_REPLICAS], v2[SEGNB_STAT_REPLICAS];
#pragma unroll
                    for (int rp = 0; rp < SEGNB_STAT_REPLICAS; ++rp) {
                        v1[rp] = a.bn_sums[(long long)(rp * 2) * g.Co + ch];
                        v2[rp] = a.bn_sums[(long long)(rp * 2 + 1) * g.Co + ch];
                    }
                    double sdz = 0.0, sdzy = 0.0;
#pragma unroll
                    for (int rp = 0; rp < SEGNB_STAT_REPLICAS; ++rp) {
                        sdz += v1[rp];
                        sdzy += v2[rp];
                    }
                    is = a.bn_coef[3 * g.Co + ch];
                    av = (a.bn_gamma != nullptr ? a.bn_gamma[ch] : 1.f) * is;
                    c1 = (float)(sdz / a.bn_count);
                    c2v = (float)(sdzy / a.bn_count);
                    if (gq == 0) {
                        if (a.bn_dgamma != nullptr) a.bn_dgamma[ch] += (float)sdzy;
                        if (a.bn_dbeta != nullptr) a.bn_dbeta[ch] += (float)sdz;
                    }
                }
                if (gq == 0 && ch < g.Co && a.bn_bcoef != nullptr) {
                    a.bn_bcoef[ch] = av;
                    a.bn_bcoef[g.Co + ch] = c1;
                    a.bn_bcoef[2 * g.Co + ch] = c2v;
                }
                sApp[c2] = av;
                sApp[BN + c2] = c1;
                sApp[2 * BN + c2] = c2v;
                sApp[3 * BN + c2] = is;
            }
        }
    }

    for (int mt = gq; mt < a.MT; mt += a.GM) {
        const int m_base = mt * BM;
        __syncthreads();  // previous tile's staging / row table fully consumed
        for (int rr = tid; rr < BM; rr += NT) {
            const int m = m_base + rr;
            int4 ri;
            if (m < a.M) {
                const int n = m / QHW;
                const int rem = m - n * QHW;
                const int qh = rem / g.QW;
                const int qw = rem - qh * g.QW;
                ri.x = n * (g.Hi * g.Wi);
                ri.y = qh * g.in_step;
                ri.z = qw * g.in_step;
                ri.w = (n * g.Ho + qh * g.out_step + g.oh0) * g.Wo + qw * g.out_step + g.ow0;
            } else {
                ri.x = 0;
                ri.y = -(1 << 28);
                ri.z = -(1 << 28);
                ri.w = -1;
            }
            sRow[rr] = ri;
        }
        __syncthreads();

        int rn[AI], rh[AI], rw[AI];
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int4 ri = sRow[row0 + 32 * i];
            rn[i] = ri.x;
            rh[i] = ri.y;
            rw[i] = ri.z;
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // Loads go through buffer descriptors: a 32-bit byte offset per lane, a wave-uniform scalar offset per K
        // step, and the hardware bounds check -- a row outside the image (or a channel tile outside Co) gets an
        // offset beyond the buffer and reads back zeros, with no branch and no register clearing.  This thread's
        // 16-byte chunk walks K as (tap, channel): tracked incrementally, the gather offsets of the AI rows are
        // recomputed only when the tap changes.  (The previous form -- division of k by Ci, bounds tests, 64-bit
        // addresses and zero fills for every row of every K step -- issued 9 VALU instructions per MFMA.)
        uint4 ra[AI], rb[BI];
        int k_tap = 0, k_ci = c * EPC;                      // this thread's chunk of K step 0
        while (k_ci >= g.Ci) { k_ci -= g.Ci; ++k_tap; }
        unsigned a_off[AI];
        auto set_tap = [&]() {
            const int2 t = sTap[k_tap < g.ntaps ? k_tap : 0];
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int hi = rh[i] + t.x, wi = rw[i] + t.y;
                const bool ok = k_tap < g.ntaps && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                a_off[i] = ok ? (unsigned)((rn[i] + hi * g.Wi + wi) * g.ld_in) * (unsigned)sizeof(T) : OOB_OFFSET;
            }
        };
        set_tap();
        auto gload = [&](int s) {
#pragma unroll
            for (int i = 0; i < AI; ++i) ra[i] = buf_load16(rsrc_in, a_off[i] + (unsigned)k_ci * (unsigned)sizeof(T), 0);
            const unsigned kb = (unsigned)(s * BK) * (unsigned)sizeof(T);          // wave-uniform
            const bool kok = s * BK + c * EPC < a.Ktot;                             // false only in a ragged last step
#pragma unroll
            for (int j = 0; j < BI; ++j) rb[j] = buf_load16(rsrc_w, kok ? b_off[j] : OOB_OFFSET, kb);
            // advance this thread's chunk by one K step
            k_ci += BK;
            if (k_ci >= g.Ci) {
                do { k_ci -= g.Ci; ++k_tap; } while (k_ci >= g.Ci);
                set_tap();
            }
        };
        auto lstore = [&](int buf) {
            unsigned char* base = sTiles + buf * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < AI; ++i)
                *reinterpret_cast<uint4*>(base + (row0 + 32 * i) * LDS_ROW + c * 16) = ra[i];
#pragma unroll
            for (int j = 0; j < BI; ++j)
                *reinterpret_cast<uint4*>(base + (BM + row0 + 32 * j) * LDS_ROW + c * 16) = rb[j];
        };

        gload(0);
        lstore(0);
        __syncthreads();
        for (int s = 0; s < a.ksteps; ++s) {
            const int buf = s & 1;
            if (s + 1 < a.ksteps) gload(s + 1);
            const unsigned char* sA = sTiles + buf * TILE_BYTES + (wr * WM) * LDS_ROW;
            const unsigned char* sB = sTiles + buf * TILE_BYTES + (BM + wc * WN) * LDS_ROW;
            mma_step<T, TM, TN>(sA, sB, r, h, acc);
            if (s + 1 < a.ksteps) lstore(buf ^ 1);
            __syncthreads();
        }

        // BNR: the y rows of the pixels this thread stores are requested NOW -- their latency runs under the staging below
        // (requested inside the store loop they were RPT dependent round trips per tile: the launch took 2.4 x as long)
        constexpr int RPT = BNR ? BM * (BN / 8) / NT : 1;
        uint4 yq[RPT], xq[BNM == 2 ? RPT : 1];
        if constexpr (BNR) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int q = tid + k * NT;
                const int row = q / (BN / 8), cc = q - row * (BN / 8);
                const int opix = sRow[row].w;
                const int co = n_base + cc * 8;
                yq[k] = make_uint4(0u, 0u, 0u, 0u);
                if (opix >= 0 && co < g.Co)
                    yq[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.bn_y) + (long long)opix * a.bn_ld + co);
                if constexpr (BNM == 2) {
                    xq[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (opix >= 0 && co < g.Co && a.bn_accumulate)
                        xq[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.bn_acc) + (long long)opix * a.bn_acc_ld + co);
                }
            }
        }
        // ---- epilogue: bias, round, stats, stage to LDS ------------------------------------------
        unsigned char* sOut = sTiles;
        float cs1[TN], cs2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wc * WN + 32 * j + r;
            const int co = n_base + col;
            float bv = (a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;
            float sv = 1.f, ep_neg = 1.f;
            if constexpr (EP) {
                if (a.ep_coef != nullptr && co < a.g.Co) {       // (acc + bias - mean) * scale + shift
                    sv = a.ep_coef[co];
                    bv = (bv - a.ep_coef[2 * a.g.Co + co]) * sv + a.ep_coef[a.g.Co + co];
                }
                ep_neg = a.ep_act == SEGNB_ACT_RELU ? 0.f : (a.ep_act == SEGNB_ACT_LEAKY ? a.ep_slope : 1.f);
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wr * WM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float ev = acc[i][j][e] + bv;
                    if constexpr (EP) {
                        ev = acc[i][j][e] * sv + bv;
                        if (ev < 0.f) ev = ev * ep_neg + 0.f;
                    }
                    const T tv = Elem<T>::from_f32(ev);
                    *reinterpret_cast<T*>(sOut + row * OUT_ROW + col * (int)sizeof(T)) = tv;
                    if (m_base + row < a.M) {
                        const float vr = Elem<T>::to_f32(tv);
                        s1 += vr;
                        s2 += vr * vr;
                    }
                }
            }
            cs1[j] = s1;
            cs2[j] = s2;
        }
        if (a.stats != nullptr) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float t1 = cs1[j] + __shfl_xor(cs1[j], 32);
                const float t2 = cs2[j] + __shfl_xor(cs2[j], 32);
                if (h == 0) {
                    // one slot per (wave row, column), summed below in a fixed order: reproducible statistics
                    const int col = wc * WN + 32 * j + r;
                    sStat[wr * 2 * BN + col] = t1;
                    sStat[wr * 2 * BN + BN + col] = t2;
                }
            }
        }
        __syncthreads();
        if (a.stats != nullptr && tid < 2 * BN) {
#pragma unroll
            for (int q = 0; q < BM / WM; ++q) st += (double)sStat[q * 2 * BN + tid];
        }
        constexpr int OC = BN / EPC;
        static_assert(!BNR || (BM * OC) % NT == 0, "BatchNorm-reduce: whole store rounds");
#pragma unroll
        for (int k = 0; k < (BNR ? RPT : 1); ++k)
        for (int q = BNR ? tid + k * NT : tid; q < (BNR ? tid + k * NT + 1 : BM * OC); q += NT) {
            const int row = q / OC, cc = q - row * OC;
            const int opix = sRow[row].w;
            const int co = n_base + cc * EPC;
            if (opix >= 0 && co < g.Co) {
                const uint4 gv = *reinterpret_cast<const uint4*>(sOut + row * OUT_ROW + cc * 16);
                if constexpr (BNM == 0) *reinterpret_cast<uint4*>(outT + (long long)opix * g.ld_out + co) = gv;
                if constexpr (BNR) {
                    // dz = round(g * act'(z)), z = (y - mean) * scale + shift: bn_act_bwd_reduce_kernel's arithmetic on the
                    // ROUNDED gradient this launch stores
                    const uint4 yv = yq[k];
                    float gq8[8], yf[8];
                    gq8[0] = __uint_as_float(gv.x << 16); gq8[1] = __uint_as_float(gv.x & 0xffff0000u);
                    gq8[2] = __uint_as_float(gv.y << 16); gq8[3] = __uint_as_float(gv.y & 0xffff0000u);
                    gq8[4] = __uint_as_float(gv.z << 16); gq8[5] = __uint_as_float(gv.z & 0xffff0000u);
                    gq8[6] = __uint_as_float(gv.w << 16); gq8[7] = __uint_as_float(gv.w & 0xffff0000u);
                    yf[0] = __uint_as_float(yv.x << 16); yf[1] = __uint_as_float(yv.x & 0xffff0000u);
                    yf[2] = __uint_as_float(yv.y << 16); yf[3] = __uint_as_float(yv.y & 0xffff0000u);
                    yf[4] = __uint_as_float(yv.z << 16); yf[5] = __uint_as_float(yv.z & 0xffff0000u);
                    yf[6] = __uint_as_float(yv.w << 16); yf[7] = __uint_as_float(yv.w & 0xffff0000u);
                    float bmu[8], bsc[8], bsh[8];
                    load8(sStat + cc * 8, bmu);
                    load8(sStat + BN + cc * 8, bsc);
                    load8(sStat + 2 * BN + cc * 8, bsh);
                    if constexpr (BNM == 2) {
                        // bn_bwd_apply_kernel's arithmetic (norm_act.hip: bn_dz_elem, bn_apply_elem, the ACC form's rounded add)
                        float aa[8], ac1[8], ac2[8], ais[8], old[8], o8[8];
                        load8(sApp + cc * 8, aa);
                        load8(sApp + BN + cc * 8, ac1);
                        load8(sApp + 2 * BN + cc * 8, ac2);
                        load8(sApp + 3 * BN + cc * 8, ais);
                        const uint4 xv = xq[k];
                        old[0] = __uint_as_float(xv.x << 16); old[1] = __uint_as_float(xv.x & 0xffff0000u);
                        old[2] = __uint_as_float(xv.y << 16); old[3] = __uint_as_float(xv.y & 0xffff0000u);
                        old[4] = __uint_as_float(xv.z << 16); old[5] = __uint_as_float(xv.z & 0xffff0000u);
                        old[6] = __uint_as_float(xv.w << 16); old[7] = __uint_as_float(xv.w & 0xffff0000u);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float z = (yf[e] - bmu[e]) * bsc[e] + bsh[e] + 0.f;
                            const float dv = bf16_bits_to_f32(f32_to_bf16_bits(gq8[e] * 1.f * (z > 0.f ? 1.f : bneg)));
                            const float yh = (yf[e] - bmu[e]) * ais[e];
                            const float dy = bf16_bits_to_f32(f32_to_bf16_bits(aa[e] * (dv - ac1[e] - yh * ac2[e])));
                            o8[e] = a.bn_accumulate ? __fadd_rn(old[e], dy) : dy;
                        }
                        uint4 ov;
                        ov.x = pack2bf(o8[0], o8[1]); ov.y = pack2bf(o8[2], o8[3]);
                        ov.z = pack2bf(o8[4], o8[5]); ov.w = pack2bf(o8[6], o8[7]);
                        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.bn_acc) + (long long)opix * a.bn_acc_ld + co) = ov;
                    } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yc = yf[e] - bmu[e];
                        const float z = yc * bsc[e] + bsh[e];
                        const float dv = bf16_bits_to_f32(f32_to_bf16_bits(gq8[e] * (z > 0.f ? 1.f : bneg)));
                        bs1[e] += dv;
                        bs2[e] += dv * yc;
                    }
                    }
                }
            }
        }
    }
    if constexpr (BNR && BNM != 2) {
        // fixed-order block reduction (the threads of a channel chunk are tid = cc + k * OC), one fp64 atomic per channel
        constexpr int OC = BN / 8;
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [NT][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid * 16 + e] = bs1[e];
            red[tid * 16 + 8 + e] = bs2[e];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid - which * BN;
            const int c8 = col >> 3, e = col & 7;
            double sum = 0.0;
            for (int k = 0; k < NT / OC; ++k) sum += (double)red[(k * OC + c8) * 16 + which * 8 + e];
            const int co = n_base + col;
            if (co < g.Co) {
                if (which == 1) sum *= (double)a.bn_coef[3 * g.Co + co];      // * invstd: sum dz * yhat
                atomicAdd(&a.bn_sums[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * g.Co + co], sum);
            }
        }
    }
    if (a.stats != nullptr && tid < 2 * BN) {
        const int which = tid / BN, col = tid - which * BN;
        const int co = n_base + col;
        if (co < g.Co) atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * g.Co + co], st);
    }
}

// ================================================================================================
// forward of FEW pixels x FEW output channels x VERY deep K (tiramisu.py:14 at the bottleneck: 8 x 16 x 16 pixels,
// Ci ~ 1000, Co = 16 -- the block tiles above leave 16 blocks walking 140 K steps each: 175 us for 0.6 GFLOP).
// Both MFMA operands are K-contiguous in HBM as they are (NHWC pixels, packed weights), and a 32 x 32 tile per wave
// shares nothing with its neighbours, so there is no LDS staging: each of the 16 waves of a block owns every 16th
// 16-element K unit of ONE 32-pixel x 32-channel tile, loads its fragments straight into registers (8 units in flight)
// and the 16 partial tiles are summed through LDS in a fixed order.  No statistics / epilogue forms (callers that need
// them take the general kernel).
// ================================================================================================
constexpr int DK_WAVES = 16, DK_DEPTH = 8;
__global__ __launch_bounds__(DK_WAVES * 64) void conv_fprop_deepk_kernel(const FpropArgs a) {
    __shared__ float sAcc[DK_WAVES][16][64];
    const segnb_conv_geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m_base = blockIdx.x * 32, n_base = blockIdx.y * 32;
    const __amdgpu_buffer_rsrc_t rsrc_in = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.w, a.w_bytes);
    const int QHW = g.QH * g.QW;
    // A row of this lane = pixel m_base + r; B row = output channel n_base + r
    const int m = m_base + r;
    const bool m_ok = m < a.M;
    const int n = m_ok ? m / QHW : 0;
    const int rem = m - n * QHW;
    const int qh = rem / g.QW, qw = rem - qh * g.QW;
    const int ph = qh * g.in_step, pw = qw * g.in_step;
    const int co = n_base + r;
    const unsigned b_row = co < g.Co ? (unsigned)(co * a.Ktot + h * 8) * 2u : OOB_OFFSET;
    const int cpt = g.Ci >> 4;                       // 16-element units per tap
    const int units = g.ntaps * cpt;
    int tap = 0, c16 = wave;                         // this wave's next unit
    while (c16 >= cpt) { c16 -= cpt; ++tap; }
    unsigned a_row = OOB_OFFSET;
    auto set_tap = [&]() {
        a_row = OOB_OFFSET;
        if (tap < g.ntaps && m_ok) {
            const int hi = ph + g.dh[tap], wi = pw + g.dw[tap];
            if ((unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi)
                a_row = (unsigned)(((n * g.Hi + hi) * g.Wi + wi) * g.ld_in + h * 8) * 2u;
        }
    };
    set_tap();
    uint4 fa[DK_DEPTH], fb[DK_DEPTH];
    auto issue = [&](int slot) {
        const bool live = tap < g.ntaps;
        fa[slot] = buf_load16(rsrc_in, live && a_row != OOB_OFFSET ? a_row + (unsigned)(c16 * 32) : OOB_OFFSET, 0);
        fb[slot] = buf_load16(rsrc_w, live && b_row != OOB_OFFSET ? b_row + (unsigned)((tap * g.Ci + c16 * 16) * 2) : OOB_OFFSET, 0);
        c16 += DK_WAVES;
        if (c16 >= cpt) {
            do { c16 -= cpt; ++tap; } while (c16 >= cpt);
            set_tap();
        }
    };
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < DK_DEPTH; ++i) issue(i);
    const int rounds = (units + DK_WAVES * DK_DEPTH - 1) / (DK_WAVES * DK_DEPTH);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int i = 0; i < DK_DEPTH; ++i) {
            const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(&fa[i]);
            const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(&fb[i]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
            issue(i);                                 // (past the last unit: out-of-range offsets, zeros)
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) sAcc[wave][e][lane] = acc[e];
    __syncthreads();
    // thread -> accumulator element (e, lane) of the tile: 16 * 64 = the block's 1024 threads
    const int e = tid >> 6;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < DK_WAVES; ++w) v += sAcc[w][e][lane];
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    const int mo = m_base + row;
    float stored = 0.f;
    if (mo < a.M && co < g.Co) {
        if (a.bias != nullptr && co < a.bias_n) v += a.bias[co];
        const int no = mo / QHW;
        const int ro = mo - no * QHW;
        const int qho = ro / g.QW, qwo = ro - qho * g.QW;
        const long long opix = (long long)(no * g.Ho + qho * g.out_step + g.oh0) * g.Wo + qwo * g.out_step + g.ow0;
        bf16_t tv = Elem<bf16_t>::from_f32(v);
        if (a.drop != nullptr)       // Dropout2d multiplier of (image, channel): segnb_conv_fprop_drop
            tv = Elem<bf16_t>::from_f32(Elem<bf16_t>::to_f32(tv) * a.drop[(long long)no * a.ld_drop + co]);
        reinterpret_cast<bf16_t*>(a.out)[opix * g.ld_out + co] = tv;
        stored = Elem<bf16_t>::to_f32(tv);
    }
    if (a.stats != nullptr) {        // statistics of the stored tile: 32 rows per channel through LDS, fixed order, one atomic each
        __syncthreads();             // (every thread has read its partial sums)
        float* red = &sAcc[0][0][0];                     // [32 rows][32 channels]
        red[row * 32 + r] = stored;
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, col = tid & 31;
            double sum = 0.0;
            for (int k = 0; k < 32; ++k) {
                const float x = red[k * 32 + col];
                sum += which ? (double)(x * x) : (double)x;
            }
            const int cc = n_base + col;
            if (cc < g.Co)
                atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.stats_ld + cc], sum);
        }
    }
}

static bool fprop_deepk_shape(const segnb_conv_geom& g) {
    const int M = g.N * g.QH * g.QW, Ktot = g.ntaps * g.Ci;
    if (!segnb_knob_fprop_deepk() || (g.Ci & 15) != 0 || Ktot < 2048) return false;
    // the block tiles of the general kernel would leave more than half of the CUs without a block
    const long long general_blocks = (long long)ceil_div(M, 128) * ceil_div(g.Co, g.Co <= 32 ? 32 : 64);
    return general_blocks * 2 <= segnb_num_cus();
}
// (plain launches: no statistics -- the kernel has them for segnb_conv_fprop_drop, which calls it directly -- and no epilogue forms)
static bool fprop_deepk_applies(const FpropArgs& a) { return a.stats == nullptr && a.ep_act < 0 && fprop_deepk_shape(a.g); }

// ================================================================================================
// weight gradient:  dW[co][k'] += sum_pixels dy[pix][co] * im2col(x)[pix][k']
// GEMM view: M = Co, N = K' = ntaps*Ci, reduction over pixels (split across blocks, fp32 atomics).
// Both operands are pixel-major in HBM but MFMA wants the reduction index contiguous per lane, so
// the loader transposes EPC x EPC (8x8 bf16 / 4x4 f32) blocks in registers on the way into LDS.
// ================================================================================================
__device__ __forceinline__ void transpose_block(const uint4 (&in)[8], uint4 (&out)[8], bf16_t*) {
    const unsigned* d = reinterpret_cast<const unsigned*>(in);  // d[p*4 + w]
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned lo = d[(2 * j) * 4 + (c >> 1)];
            const unsigned hi = d[(2 * j + 1) * 4 + (c >> 1)];
            o[j] = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
        out[c] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
__device__ __forceinline__ void transpose_block(const uint4 (&in)[4], uint4 (&out)[4], float*) {
    out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
    out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
    out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
    out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BKP = 8 * EPC;               // pixels per K step
    constexpr int ITEMS_A = (BM / EPC) * 8;    // (chunk column, pixel group) blocks of the dy tile
    constexpr int ITEMS_B = (BN / EPC) * 8;
    constexpr int ITEMS = ITEMS_A + ITEMS_B;
    constexpr int IPT = (ITEMS + NT - 1) / NT;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TILE_BYTES = (BM + BN) * LDS_ROW;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sTiles = smem;

    const segnb_conv_geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    const int ntiles = a.MT * a.NTL;
    const int split = blockIdx.x / ntiles;
    const int tile = blockIdx.x - split * ntiles;
    const int mtile = tile / a.NTL, ntile = tile - mtile * a.NTL;
    const int mb = mtile * BM;   // co base
    const int nb = ntile * BN;   // k' base

    const T* __restrict__ inT = reinterpret_cast<const T*>(a.in);
    const T* __restrict__ doT = reinterpret_cast<const T*>(a.dout);

    // per-item constants
    int it_rowbase[IPT];   // LDS row of channel 0 of this item (A rows first, then B rows)
    int it_o[IPT];         // pixel group within the K step
    int it_ch[IPT];        // channel / ci element offset in the source tensor, or -1 = nothing to load
    int it_dh[IPT], it_dw[IPT];
    bool it_isA[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int it = tid + u * NT;
        it_rowbase[u] = -1;
        it_ch[u] = -1;
        it_o[u] = 0;
        it_dh[u] = it_dw[u] = 0;
        it_isA[u] = true;
        if (it < ITEMS) {
            const bool isA = it < ITEMS_A;
            const int idx = isA ? it : it - ITEMS_A;
            const int o = idx & 7, cc = idx >> 3;
            it_isA[u] = isA;
            it_o[u] = o;
            if (isA) {
                it_rowbase[u] = cc * EPC;
                const int co = mb + cc * EPC;
                it_ch[u] = (co < g.Co) ? co : -1;
            } else {
                it_rowbase[u] = BM + cc * EPC;
                const int kp = nb + cc * EPC;
                if (kp < a.Ktot) {
                    const int tap = kp / g.Ci;
                    it_ch[u] = kp - tap * g.Ci;
                    it_dh[u] = g.dh[tap];
                    it_dw[u] = g.dw[tap];
                }
            }
        }
    }

    const int QHW = g.QH * g.QW;
    const int step_begin = split * a.steps_per_split;
    int step_end = step_begin + a.steps_per_split;
    if (step_end > a.total_steps) step_end = a.total_steps;
    const int nsteps = step_end - step_begin;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 regs[IPT][EPC];
    auto gload = [&](int s) {
        const int pix0 = (step_begin + s) * BKP;
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            int m = pix0 + it_o[u] * EPC;
            int n = m / QHW;
            int rem = m - n * QHW;
            int qh = rem / g.QW;
            int qw = rem - qh * g.QW;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                // branch-free: an element outside the image / the pixel range loads the tensor's first chunk instead and
                // is zeroed afterwards, so that the EPC loads of an item are all in flight together (under the nested
                // conditions of the first version each load waited for the one before it)
                const T* p = it_isA[u] ? doT : inT;
                bool ok = it_ch[u] >= 0 && m < a.M;
                if (it_isA[u]) {
                    const long long pix = (long long)(n * g.Ho + qh * g.out_step + g.oh0) * g.Wo + qw * g.out_step + g.ow0;
                    if (ok) p = doT + pix * g.ld_out + it_ch[u];
                } else {
                    const int hi = qh * g.in_step + it_dh[u], wi = qw * g.in_step + it_dw[u];
                    ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                    const long long pix = (long long)(n * g.Hi + hi) * g.Wi + wi;
                    if (ok) p = inT + pix * g.ld_in + it_ch[u];
                }
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                regs[u][e] = ok ? v : make_uint4(0, 0, 0, 0);
                ++m;
                if (++qw == g.QW) {
                    qw = 0;
                    if (++qh == g.QH) {
                        qh = 0;
                        ++n;
                    }
                }
            }
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* base = sTiles + buf * TILE_BYTES;
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            if (it_rowbase[u] >= 0) {
                uint4 t[EPC];
                transpose_block(regs[u], t, (T*)nullptr);
#pragma unroll
                for (int cidx = 0; cidx < EPC; ++cidx)
                    *reinterpret_cast<uint4*>(base + (it_rowbase[u] + cidx) * LDS_ROW + it_o[u] * 16) = t[cidx];
            }
        }
    };

    if (nsteps > 0) {
        // ONE LDS tile: a K step is a handful of MFMAs behind a gather of 8 scattered 16-byte loads per thread, so the loop
        // is bound by load latency (rocprofv3: 80-84 % of the wave cycles waiting, 0.1 % MFMA) and what hides it is the
        // number of blocks a CU holds, not a second tile (half the LDS per block: 5 blocks per CU instead of 2-3)
        gload(0);
        lstore(0);
        __syncthreads();
        const unsigned char* sA = sTiles + (wr * WM) * LDS_ROW;
        const unsigned char* sB = sTiles + (BM + wc * WN) * LDS_ROW;
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) gload(s + 1);
            mma_step<T, TM, TN>(sA, sB, r, h, acc);
            __syncthreads();                          // every wave has read the tile
            if (s + 1 < nsteps) {
                lstore(0);
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int kp = nb + wc * WN + 32 * j + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = mb + wr * WM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (co < g.Co && kp < a.Ktot) atomicAdd(&a.dwp[(long long)co * a.Ktot + kp], acc[i][j][e]);
                }
            }
    }
}

// ================================================================================================
// weight gradient of a convolution with AT MOST 8 output channels (one 16-byte group of dy): the heads that are a general
// convolution (linknet.py:62: 2x2, 32 -> 1 at 512 x 512).  The MFMA tiles above are built for wide dW matrices; here dW is
// 8 x (taps * Ci) numbers and the work is one pass over x and dy -- HBM-bound.  Plain FMAs: thread = (tap, 8-channel group
// of x) x pixel lane, 8 x 8 accumulators; a block walks whole output rows (no per-pixel division), x is fetched as
// contiguous 64-byte pixel rows by neighbouring threads, dy as one broadcast 16-byte load.  Partial sums meet in LDS, then
// in the workspace, with fp32 atomics (as in the general kernel).
// ================================================================================================
__global__ __launch_bounds__(NT) void conv_wgrad_co8_kernel(const WgradArgs a, int nslot, int PL) {
    __shared__ float sAcc[NT * 64 / 4];      // [nslot][64] <= 64 slots x 64 when PL >= 4; larger slot counts loop below
    const segnb_conv_geom& g = a.g;
    const int tid = threadIdx.x;
    const int slot = tid % nslot, pl = tid / nslot;
    const bool live = pl < PL;
    const int cpt = g.Ci >> 3;
    const int tap = slot / cpt, chunk = slot - tap * cpt;
    const int dh = g.dh[tap], dw = g.dw[tap];
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.in);
    const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(a.dout);
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    const int rows = g.N * g.QH;
    for (int row = blockIdx.x; row < rows && live; row += gridDim.x) {
        const int n = row / g.QH, qh = row - n * g.QH;
        const int hi = qh * g.in_step + dh;
        if ((unsigned)hi >= (unsigned)g.Hi) continue;
        const bf16_t* xr = x + (long long)(n * g.Hi + hi) * g.Wi * g.ld_in + chunk * 8;
        const bf16_t* dr = dy + ((long long)(n * g.Ho + qh * g.out_step + g.oh0) * g.Wo + g.ow0) * g.ld_out;
        // four pixels per trip, all eight loads issued before the first FMA
        for (int qw0 = pl; qw0 < g.QW; qw0 += 4 * PL) {
            float xv[4][8], dv[4][8];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qw = qw0 + u * PL;
                const int wi = qw * g.in_step + dw;
                ok[u] = qw < g.QW && (unsigned)wi < (unsigned)g.Wi;
                const int qs = ok[u] ? qw : 0, ws = ok[u] ? wi : 0;
                load8(xr + (long long)ws * g.ld_in, xv[u]);
                load8(dr + (long long)qs * g.out_step * g.ld_out, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(dv[u][i], xv[u][j], acc[i][j]);
            }
        }
    }
    // slots in groups that fit the LDS array
    const int per_pass = (NT * 64 / 4) / 64;          // 64 slots per pass
    for (int s0 = 0; s0 < nslot; s0 += per_pass) {
        for (int i = tid; i < per_pass * 64; i += NT) sAcc[i] = 0.f;
        __syncthreads();
        if (live && slot >= s0 && slot < s0 + per_pass) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(&sAcc[(slot - s0) * 64 + i * 8 + j], acc[i][j]);
        }
        __syncthreads();
        const int n_here = nslot - s0 < per_pass ? nslot - s0 : per_pass;
        for (int i = tid; i < n_here * 64; i += NT) {
            const int sl = s0 + i / 64, co = (i >> 3) & 7, ci = i & 7;
            const int t = sl / cpt, ch = sl - t * cpt;
            const float v = sAcc[i];
            if (v != 0.f) atomicAdd(&a.dwp[(long long)co * a.Ktot + t * g.Ci + ch * 8 + ci], v);
        }
        __syncthreads();
    }
}

static bool wgrad_co8_applies(const segnb_conv_geom* g) {
    const int nslot = g->ntaps * (g->Ci / 8);
    return g->Co == 8 && nslot <= NT && (long long)g->N * g->QH * g->QW >= 1024;
}

// ================================================================================================
// weight pack / gradient unpack
// ================================================================================================
struct PackArgs {
    int Mp, Cp, ntaps;
    long long s_m, s_c;
    const int* mmap;
    const int* cmap;
    int tap_off[SEGNB_MAX_TAPS];
};

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp, const PackArgs p) {
    const long long total = (long long)p.Mp * p.ntaps * p.Cp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cp = (int)(i % p.Cp);
        const long long q = i / p.Cp;
        const int t = (int)(q % p.ntaps);
        const int mp = (int)(q / p.ntaps);
        const int m = p.mmap[mp], c = p.cmap[cp];
        float v = 0.f;
        if (m >= 0 && c >= 0) v = w[m * p.s_m + c * p.s_c + p.tap_off[t]];
        wp[i] = Elem<T>::from_f32(v);
    }
}

__global__ void unpack_wgrad_kernel(float* __restrict__ dwp, float* __restrict__ gw, const PackArgs p,
                                    int accumulate) {
    const long long total = (long long)p.Mp * p.ntaps * p.Cp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cp = (int)(i % p.Cp);
        const long long q = i / p.Cp;
        const int t = (int)(q % p.ntaps);
        const int mp = (int)(q / p.ntaps);
        const int m = p.mmap[mp], c = p.cmap[cp];
        if (m >= 0 && c >= 0) {
            float* dst = gw + m * p.s_m + c * p.s_c + p.tap_off[t];
            *dst = accumulate ? (*dst + dwp[i]) : dwp[i];
        }
        dwp[i] = 0.f;   // workspace is consumed: ready for the next step's atomics without a memset
    }
}

// ---- batched forms: ONE launch packs every weight matrix of a model (or unpacks every gradient) --------
// job table in device memory (built once by the host, pointers are stable); block -> job by binary search
struct __attribute__((aligned(8))) PackJob {
    const float* w;          // pack: fp32 parameter (source);  unpack: fp32 gradient (destination)
    void* packed;            // pack: destination (dtype);      unpack: fp32 workspace (source, re-zeroed)
    const int* mmap;
    const int* cmap;
    long long s_m, s_c;
    int Mp, Cp, ntaps, dtype;
    int block_start;         // first block of this job; jobs sorted by it
    int nslab;               // unpack jobs: partial slabs to sum ([nslab][Mp][ntaps][Cp]); 0/1 = one
    // masked != 0: tap_off[t] is a BIT MASK over the (<= 9) kernel positions of the parameter tensor instead of one offset --
    //   pack  : packed(m, t, c) = sum over the set positions k of w(m, c, k)      (rounded once, after the fp32 sum)
    //   unpack: gw(m, c, k)    += sum over the taps t whose mask holds k of dwp(m, t, c)          (taps in order)
    // = the convolution of a nearest-x2 upsampled tensor as the 4x4 / stride-2 transposed convolution it is, on the
    // reference's own 3x3 parameter (lib/models/zf_unet.py:42,78-90: Upsample(scale_factor=2) -> cat -> conv3x3)
    int masked;
    int pad_;
    int tap_off[SEGNB_MAX_TAPS];
};

// the job that owns block b (block_start ascending, job 0 starts at 0): every wave counts the starts <= b, 64 jobs per round trip
// (a binary search was log2(njobs) DEPENDENT loads -- 6 for ZF_UNET's 46 jobs, most of a small tile's life)
template <typename Job>
__device__ __forceinline__ int find_job(const Job* jobs, int njobs, int b) {
    const int lane = threadIdx.x & 63;
    int cnt = 0;
    for (int k = 0; k < njobs; k += 64) {
        const bool le = k + lane < njobs && jobs[k + lane].block_start <= b;
        cnt += __popcll(__ballot(le));
    }
    return __builtin_amdgcn_readfirstlane(cnt - 1);
}

// Tiled through LDS so that BOTH sides are coalesced.  The parameter tensor has the tap index fastest
// (stride 1) and one of the two channel indices at stride T_src = min(s_m, s_c) ("fast" channel), the other at
// a large stride ("slow" channel).  A tile = SLOW x FAST channels x all taps:
//   parameter side : for each slow index a run of FAST*T_src contiguous floats
//   packed side    : [mp][t][cp], contiguous along cp
// fwd pack  (rows = co slow, cols = ci fast): tile 1 x 256   -> packed runs of 256 elements
// dgrad pack (rows = ci fast, cols = co slow): tile 8 x 64   -> packed runs of 64 elements
struct TileGeom {
    int fast_is_c;      // 1: cp indexes the fast channel (fwd layout), 0: mp does (dgrad layout)
    int F, S;           // tile extents along fast / slow packed index
    int nF, nS;         // tiles along each
    int Tsrc;           // taps in the parameter tensor (KH*KW)
};

__device__ __forceinline__ TileGeom tile_geom(const PackJob& j) {
    TileGeom g;
    g.fast_is_c = j.s_c < j.s_m;
    g.Tsrc = (int)(g.fast_is_c ? j.s_c : j.s_m);
    if (g.fast_is_c) {
        g.F = 256; g.S = 1;
        g.nF = (j.Cp + g.F - 1) / g.F; g.nS = j.Mp;
    } else {
        g.F = 8; g.S = 64;
        g.nF = (j.Mp + g.F - 1) / g.F; g.nS = (j.Cp + g.S - 1) / g.S;
    }
    return g;
}

constexpr int PACK_LDS_FLOATS = 256 * 9 > 64 * 8 * 9 ? 256 * 9 : 64 * 8 * 9;   // 4608

// is_pack: parameter -> packed (round to dtype); else packed fp32 workspace -> += gradient, workspace zeroed.
// The channel maps of the tile are staged in LDS once (a map lookup per element made every parameter access the
// tail of a dependent-load chain) and the packed side moves 16 bytes per lane (8 bf16 / 4 fp32 along the
// contiguous channel index; channel counts are padded to multiples of 8).
template <bool IS_PACK>
__device__ __forceinline__ void pack_tile(const PackJob* __restrict__ jobs, int njobs, int bid, float* tile, long long* sFast,
                                          long long* sSlow, int& sCnt) {
    const PackJob& j = jobs[find_job(jobs, njobs, bid)];
    const TileGeom g = tile_geom(j);
    const int tb = bid - j.block_start;
    const int tf = tb % g.nF, ts = tb / g.nF;
    const int f0 = tf * g.F, s0 = ts * g.S;
    const int run = g.F * g.Tsrc;                    // contiguous floats per slow index on the parameter side
    const int Mp = j.Mp, Cp = j.Cp, nt = j.ntaps;
    float* param = const_cast<float*>(j.w);
    const int n_param = g.S * run;
    {
        const int* fmap = g.fast_is_c ? j.cmap : j.mmap;
        const int* smap = g.fast_is_c ? j.mmap : j.cmap;
        const int nfast = g.fast_is_c ? Cp : Mp, nslow = g.fast_is_c ? Mp : Cp;
        const long long fstride = g.fast_is_c ? j.s_c : j.s_m, sstride = g.fast_is_c ? j.s_m : j.s_c;
        for (int i = threadIdx.x; i < g.F; i += 256) {
            const int v = f0 + i < nfast ? fmap[f0 + i] : -1;
            sFast[i] = v >= 0 ? v * fstride : -1;
        }
        for (int i = threadIdx.x; i < g.S; i += 256) {
            const int v = s0 + i < nslow ? smap[s0 + i] : -1;
            sSlow[i] = v >= 0 ? v * sstride : -1;
        }
    }
    __syncthreads();
    // Parameter side in 16-byte accesses when the tile's fast channels are a contiguous run of the tensor (consecutive real
    // channels: everything but a tile that crosses a padding gap) and every row starts 16-byte aligned: dword accesses moved
    // the 0.38 GB of a ZF_UNET pack at 2.5 TB/s
    int nvalid = 0;
    bool vec;
    {
        const int nfast = g.fast_is_c ? Cp : Mp;
        const int lim = min(g.F, nfast - f0);
        bool ok = true;
        for (int i = threadIdx.x; i < g.F; i += 256) {
            const long long v = sFast[i];
            if (i < lim && v >= 0) ok = ok && v == sFast[0] + (long long)i * g.Tsrc;
            if (v >= 0 && i > 0 && sFast[i - 1] < 0) ok = false;               // a gap inside the run
        }
        for (int i = threadIdx.x; i < g.S; i += 256) {
            const long long v = sSlow[i];
            if (v >= 0) ok = ok && ((v + (sFast[0] < 0 ? 0 : sFast[0])) & 3) == 0;
        }
        // count of valid fast channels (they form a prefix when ok)
        int cnt = 0;
        for (int i = threadIdx.x; i < g.F; i += 256) cnt += sFast[i] >= 0 ? 1 : 0;
        if (threadIdx.x == 0) sCnt = 0;
        __syncthreads();
        if (cnt) atomicAdd(&sCnt, cnt);
        vec = __syncthreads_and(ok ? 1 : 0) != 0;
        nvalid = sCnt;
        vec = vec && sFast[0] >= 0 && ((nvalid * g.Tsrc) & 3) == 0 && ((const uintptr_t)param & 15) == 0;
    }
    const int row4 = nvalid * g.Tsrc / 4;               // float4 per slow row on the vector path
    auto param_off = [&](int sl, int fa) -> long long {      // element offset of (slow, fast, tap 0) or -1
        const long long a = sSlow[sl], b = sFast[fa];
        return (a < 0 || b < 0) ? -1 : a + b;
    };
    // packed side: [mp][t][cp]; groups of 8 consecutive cp (one 16-byte bf16 vector / two fp32 vectors)
    const int ncp = g.fast_is_c ? g.F : g.S;             // cp extent of the tile (multiple of 8)
    const int nmp = g.fast_is_c ? g.S : g.F;
    const int groups = nmp * nt * (ncp / 8);
    auto group = [&](int gi, int& mpl, int& t, int& cpl) {       // tile-local (mp, tap, first cp) of group gi
        cpl = (gi % (ncp / 8)) * 8;
        t = (gi / (ncp / 8)) % nt;
        mpl = gi / ((ncp / 8) * nt);
    };
    auto tile_idx = [&](int mpl, int t, int cpl) {               // LDS index of packed element (mp, t, cp)
        const int sl = g.fast_is_c ? mpl : cpl, fa = g.fast_is_c ? cpl : mpl;
        return sl * run + fa * g.Tsrc + j.tap_off[t];
    };
    const int cp_step = g.fast_is_c ? g.Tsrc : run;              // LDS stride between consecutive cp
    auto tile_base = [&](int mpl, int cpl) {                     // LDS index of kernel position 0 of packed (mp, cp)
        const int sl = g.fast_is_c ? mpl : cpl, fa = g.fast_is_c ? cpl : mpl;
        return sl * run + fa * g.Tsrc;
    };

    if (IS_PACK) {
        // parameter -> LDS, in parameter order (coalesced along the fast channel and the taps).  All of a lane's loads
        // are issued before the first LDS store (a rolled loop kept ONE 4-byte load per lane in flight: 2.5 TB/s)
        constexpr int MAXU = PACK_LDS_FLOATS / 256;
        if (vec) {
            constexpr int MAXU4 = (PACK_LDS_FLOATS / 4 + 255) / 256;
            if (nvalid < g.F)                                    // padding channels of the tile read as zero
                for (int i = threadIdx.x; i < n_param; i += 256) tile[i] = 0.f;
            if (nvalid < g.F) __syncthreads();
            float4 pv4[MAXU4];
            const int n4 = g.S * row4;
#pragma unroll
            for (int u = 0; u < MAXU4; ++u) {
                const int i = threadIdx.x + u * 256;
                pv4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n4) {
                    const int sl = i / row4, r = i - sl * row4;
                    if (sSlow[sl] >= 0) pv4[u] = *reinterpret_cast<const float4*>(param + sSlow[sl] + sFast[0] + 4 * r);
                }
            }
#pragma unroll
            for (int u = 0; u < MAXU4; ++u) {
                const int i = threadIdx.x + u * 256;
                if (i < n4) {
                    const int sl = i / row4, r = i - sl * row4;
                    *reinterpret_cast<float4*>(tile + sl * run + 4 * r) = pv4[u];
                }
            }
        } else {
        float pv[MAXU];
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const int i = threadIdx.x + u * 256;
            pv[u] = 0.f;
            if (i < n_param) {
                const int sl = i / run, r = i - sl * run;
                const int fa = r / g.Tsrc, tp = r - fa * g.Tsrc;
                const long long off = param_off(sl, fa);
                if (off >= 0) pv[u] = param[off + tp];
            }
        }
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const int i = threadIdx.x + u * 256;
            if (i < n_param) tile[i] = pv[u];
        }
        }
        __syncthreads();
        // LDS -> packed, 8 consecutive cp per lane
        for (int gi = threadIdx.x; gi < groups; gi += 256) {
            int mpl, t, cpl;
            group(gi, mpl, t, cpl);
            const int mp = (g.fast_is_c ? s0 : f0) + mpl, cp = (g.fast_is_c ? f0 : s0) + cpl;
            if (mp < Mp && cp < Cp) {
                float v[8];
                if (j.masked) {
                    const int base0 = tile_base(mpl, cpl);
                    const unsigned mask = (unsigned)j.tap_off[t];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = 0.f;
                    for (int pos = 0; pos < g.Tsrc; ++pos)
                        if (mask >> pos & 1u) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += tile[base0 + pos + k * cp_step];
                        }
                } else {
                    const int base = tile_idx(mpl, t, cpl);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = tile[base + k * cp_step];
                }
                const long long di = ((long long)mp * nt + t) * Cp + cp;
                if (j.dtype == SEGNB_BF16)
                    store8(reinterpret_cast<bf16_t*>(j.packed) + di, v);
                else
                    store8(reinterpret_cast<float*>(j.packed) + di, v);
            }
        }
    } else {
        float* dwp = reinterpret_cast<float*>(j.packed);
        for (int i = threadIdx.x; i < n_param; i += 256) tile[i] = 0.f;
        __syncthreads();
        if (j.masked) {
            // one thread owns every tap of its (mp, 8 x cp) group: the taps that share a kernel position are summed in
            // registers, in tap order (no collisions in the tile, bitwise reproducible)
            const int pairs = nmp * (ncp / 8);
            const long long sstride = (long long)Mp * nt * Cp;
            for (int gi = threadIdx.x; gi < pairs; gi += 256) {
                const int cpl = (gi % (ncp / 8)) * 8, mpl = gi / (ncp / 8);
                const int mp = (g.fast_is_c ? s0 : f0) + mpl, cp = (g.fast_is_c ? f0 : s0) + cpl;
                if (mp >= Mp || cp >= Cp) continue;
                float acc[9][8];
#pragma unroll
                for (int pos = 0; pos < 9; ++pos)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[pos][k] = 0.f;
                for (int t = 0; t < nt; ++t) {
                    const long long di = ((long long)mp * nt + t) * Cp + cp;
                    float v[8];
                    load8(dwp + di, v);
                    for (int sl = 1; sl < j.nslab; ++sl) {
                        float u[8];
                        load8(dwp + sl * sstride + di, u);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += u[k];
                    }
                    const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    store8(dwp + di, z);
                    const unsigned mask = (unsigned)j.tap_off[t];
#pragma unroll
                    for (int pos = 0; pos < 9; ++pos)
                        if (mask >> pos & 1u) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) acc[pos][k] += v[k];
                        }
                }
                const int base0 = tile_base(mpl, cpl);
#pragma unroll
                for (int pos = 0; pos < 9; ++pos)
                    if (pos < g.Tsrc) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) tile[base0 + pos + k * cp_step] = acc[pos][k];
                    }
            }
        } else
        for (int gi = threadIdx.x; gi < groups; gi += 256) {
            int mpl, t, cpl;
            group(gi, mpl, t, cpl);
            const int mp = (g.fast_is_c ? s0 : f0) + mpl, cp = (g.fast_is_c ? f0 : s0) + cpl;
            if (mp < Mp && cp < Cp) {
                const long long di = ((long long)mp * nt + t) * Cp + cp;
                float v[8];
                load8(dwp + di, v);
                // partial slabs of pixel splits (a job's nslab > 1): summed here, in slab order, instead of by
                // a reduction launch per layer that wrote the sum back for this kernel to read again
                const long long sstride = (long long)Mp * nt * Cp;
                for (int sl = 1; sl < j.nslab; ++sl) {
                    float u[8];
                    load8(dwp + sl * sstride + di, u);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += u[k];
                }
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                store8(dwp + di, z);                                  // workspace consumed
                const int base = tile_idx(mpl, t, cpl);
#pragma unroll
                for (int k = 0; k < 8; ++k) tile[base + k * cp_step] = v[k];      // taps are distinct: no collisions
            }
        }
        __syncthreads();
        constexpr int MAXU = PACK_LDS_FLOATS / 256;
        if (vec) {
            constexpr int MAXU4 = (PACK_LDS_FLOATS / 4 + 255) / 256;
            float4 pv4[MAXU4];
            const int n4 = g.S * row4;
#pragma unroll
            for (int u = 0; u < MAXU4; ++u) {
                const int i = threadIdx.x + u * 256;
                if (i < n4) {
                    const int sl = i / row4, r = i - sl * row4;
                    if (sSlow[sl] >= 0) pv4[u] = *reinterpret_cast<const float4*>(param + sSlow[sl] + sFast[0] + 4 * r);
                }
            }
#pragma unroll
            for (int u = 0; u < MAXU4; ++u) {
                const int i = threadIdx.x + u * 256;
                if (i < n4) {
                    const int sl = i / row4, r = i - sl * row4;
                    if (sSlow[sl] >= 0) {
                        const float4 t4 = *reinterpret_cast<const float4*>(tile + sl * run + 4 * r);
                        float4 o = pv4[u];
                        o.x += t4.x; o.y += t4.y; o.z += t4.z; o.w += t4.w;
                        *reinterpret_cast<float4*>(param + sSlow[sl] + sFast[0] + 4 * r) = o;
                    }
                }
            }
            return;
        }
        float pv[MAXU];
        long long po[MAXU];
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {                       // all gradient loads of the lane first (see the pack side)
            const int i = threadIdx.x + u * 256;
            po[u] = -1;
            if (i < n_param) {
                const int sl = i / run, r = i - sl * run;
                const int fa = r / g.Tsrc, tp = r - fa * g.Tsrc;
                const long long off = param_off(sl, fa);
                if (off >= 0) {
                    po[u] = off + tp;
                    pv[u] = param[off + tp];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MAXU; ++u)
            if (po[u] >= 0) param[po[u]] = pv[u] + tile[threadIdx.x + u * 256];
    }
}

// One block per tile, or (grid < total: segnb_tune "pack_blocks" / SEGNB_PACK_BLOCKS) persistent blocks that walk the tiles: a
// pack launched BESIDE the forward's first levels on few CUs takes its time without taking their HBM bandwidth
template <bool IS_PACK>
__global__ __launch_bounds__(256) void pack_tiled_kernel(const PackJob* __restrict__ jobs, int njobs, int total) {
    __shared__ __attribute__((aligned(16))) float tile[PACK_LDS_FLOATS];
    __shared__ long long sFast[256], sSlow[64];         // element offset of each fast / slow channel of the tile, -1 = pad
    __shared__ int sCnt;
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        pack_tile<IS_PACK>(jobs, njobs, b, tile, sFast, sSlow, sCnt);
        if (b + (int)gridDim.x < total) __syncthreads();
    }
}

// BOTH matrices of a plain 3 x 3 convolution from ONE read of its parameter (bf16): the forward matrix [Cop][9][Cip] and the
// data-gradient matrix [Cip][9][Cop] (kernel positions in each form's own tap order).  At every step's start the two tiled
// packs above read each fp32 parameter twice and their 9-KB tiles are a chain of dependent round trips (ZF_UNET: 88 us for
// 252 MB read + 123 MB written, tools/pack_bench.py); here a block owns 32 output x 64 input channels (72 KB of the
// parameter, 18 float4 loads per lane issued together), rounds once into an LDS tile and writes both forms from it in
// 128-byte / 64-byte runs.  Channels are in place (no maps: real channel c = packed channel c, padding behind them).
struct PackPairJob {
    const float* w;        // [Co][Ci][3][3]
    void* pf;              // bf16 [Cop][9][Cip]
    void* pd;              // bf16 [Cip][9][Cop], or NULL: forward matrix only
    int Ci, Co, Cip, Cop;
    int block_start, pad_;
    int tapf[9], tapd[9];  // kernel position kh * 3 + kw of packed tap t of either form
};
constexpr int PP_CO = 32, PP_CI = 64;
constexpr int PP_RS = PP_CI * 9 + 4;          // LDS row in bf16 elements (rows 8-byte aligned; 8 rows apart = 16 banks apart)
constexpr int PP_LOADS = PP_CO * PP_CI * 9 / 4 / 256;

// SGD (segnb_sgd_pack_pair_multi): the optimizer's update w -= lr * g is applied to the tile on its way through the registers --
// sgd_kernel's expression, so the parameters are bit-identical to segnb_sgd_step's -- and written back; the packed matrices are
// then those of the UPDATED parameters: the next forward's weight pack (a second read of every parameter) is already done.
// g_delta: the gradient of element e of the flat parameter buffer sits g_delta floats behind it (flat_g - flat_p).
template <bool SGD>
__global__ __launch_bounds__(256) void pack_pair_kernel(const PackPairJob* __restrict__ jobs, int njobs, long long g_delta, float lr) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[PP_CO * PP_RS];
    const PackPairJob& j = jobs[find_job(jobs, njobs, (int)blockIdx.x)];
    const int tb = blockIdx.x - j.block_start;
    const int nci = (j.Cip + PP_CI - 1) / PP_CI;
    const int ci0 = (tb % nci) * PP_CI, co0 = (tb / nci) * PP_CO;
    const int Ci = j.Ci, Co = j.Co, Cip = j.Cip, Cop = j.Cop;
    const int nv = min(PP_CI, Ci - ci0);                    // real input channels of the tile (a multiple of 4, maybe <= 0)
    const int run4 = nv > 0 ? nv * 9 / 4 : 0;               // float4 per parameter row
    constexpr int ROW4 = PP_CI * 9 / 4;
    float4 v[PP_LOADS];
    if ((Ci & 3) == 0) {
#pragma unroll
        for (int u = 0; u < PP_LOADS; ++u) {
            const int i = threadIdx.x + 256 * u;
            const int co_l = i / ROW4, r = i - co_l * ROW4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < run4 && co0 + co_l < Co)
                v[u] = *reinterpret_cast<const float4*>(j.w + ((long long)(co0 + co_l) * Ci + ci0) * 9 + 4 * r);
        }
        if constexpr (SGD) {
            float4 gv[PP_LOADS];
#pragma unroll
            for (int u = 0; u < PP_LOADS; ++u) {
                const int i = threadIdx.x + 256 * u;
                const int co_l = i / ROW4, r = i - co_l * ROW4;
                gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < run4 && co0 + co_l < Co)
                    gv[u] = *reinterpret_cast<const float4*>(j.w + g_delta + ((long long)(co0 + co_l) * Ci + ci0) * 9 + 4 * r);
            }
#pragma unroll
            for (int u = 0; u < PP_LOADS; ++u) {
                const int i = threadIdx.x + 256 * u;
                const int co_l = i / ROW4, r = i - co_l * ROW4;
                float4 pv = v[u];
                pv.x -= lr * gv[u].x; pv.y -= lr * gv[u].y; pv.z -= lr * gv[u].z; pv.w -= lr * gv[u].w;
                v[u] = pv;
                if (r < run4 && co0 + co_l < Co)
                    *reinterpret_cast<float4*>(const_cast<float*>(j.w) + ((long long)(co0 + co_l) * Ci + ci0) * 9 + 4 * r) = pv;
            }
        }
    } else {
        // rows that are not 16-byte aligned (the first layer: 3 input channels): element loads, same tile image
        const int runf = nv > 0 ? nv * 9 : 0;
#pragma unroll
        for (int u = 0; u < PP_LOADS; ++u) {
            const int i = threadIdx.x + 256 * u;
            const int co_l = i / ROW4, r = i - co_l * ROW4;
            const float* row = j.w + ((long long)(co0 + co_l) * Ci + ci0) * 9 + 4 * r;
            const bool in = co0 + co_l < Co;
            v[u].x = (in && 4 * r < runf) ? row[0] : 0.f;
            v[u].y = (in && 4 * r + 1 < runf) ? row[1] : 0.f;
            v[u].z = (in && 4 * r + 2 < runf) ? row[2] : 0.f;
            v[u].w = (in && 4 * r + 3 < runf) ? row[3] : 0.f;
            if constexpr (SGD) {
                float* wrow = const_cast<float*>(row);
                const float* grow = row + g_delta;
                if (in && 4 * r < runf) { v[u].x -= lr * grow[0]; wrow[0] = v[u].x; }
                if (in && 4 * r + 1 < runf) { v[u].y -= lr * grow[1]; wrow[1] = v[u].y; }
                if (in && 4 * r + 2 < runf) { v[u].z -= lr * grow[2]; wrow[2] = v[u].z; }
                if (in && 4 * r + 3 < runf) { v[u].w -= lr * grow[3]; wrow[3] = v[u].w; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PP_LOADS; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int co_l = i / ROW4, r = i - co_l * ROW4;
        uint2 pk;
        pk.x = pack2bf(v[u].x, v[u].y);
        pk.y = pack2bf(v[u].z, v[u].w);
        *reinterpret_cast<uint2*>(tile + co_l * PP_RS + 4 * r) = pk;
    }
    __syncthreads();
    unsigned short* const pf = reinterpret_cast<unsigned short*>(j.pf);
    unsigned short* const pd = reinterpret_cast<unsigned short*>(j.pd);
    // forward form: (co, t) rows of 64 input channels, 8 per lane
    for (int gi = threadIdx.x; gi < PP_CO * 9 * (PP_CI / 8); gi += 256) {
        const int c8 = gi % (PP_CI / 8), t = (gi / (PP_CI / 8)) % 9, co_l = gi / (9 * (PP_CI / 8));
        const int co = co0 + co_l, ci = ci0 + c8 * 8;
        if (co >= Cop || ci >= Cip) continue;
        const unsigned short* src = tile + co_l * PP_RS + c8 * 72 + j.tapf[t];
        uint4 o;
        o.x = (unsigned)src[0] | ((unsigned)src[9] << 16);
        o.y = (unsigned)src[18] | ((unsigned)src[27] << 16);
        o.z = (unsigned)src[36] | ((unsigned)src[45] << 16);
        o.w = (unsigned)src[54] | ((unsigned)src[63] << 16);
        *reinterpret_cast<uint4*>(pf + ((long long)co * 9 + t) * Cip + ci) = o;
    }
    // data-gradient form: (ci, t) rows of 32 output channels, 8 per lane (pd == NULL: a layer without a data gradient)
    if (pd == nullptr) return;
    for (int gi = threadIdx.x; gi < PP_CI * 9 * (PP_CO / 8); gi += 256) {
        const int o8 = gi % (PP_CO / 8), t = (gi / (PP_CO / 8)) % 9, ci_l = gi / (9 * (PP_CO / 8));
        const int ci = ci0 + ci_l, co = co0 + o8 * 8;
        if (ci >= Cip || co >= Cop) continue;
        const unsigned short* src = tile + (o8 * 8) * PP_RS + ci_l * 9 + j.tapd[t];
        uint4 o;
        o.x = (unsigned)src[0] | ((unsigned)src[PP_RS] << 16);
        o.y = (unsigned)src[2 * PP_RS] | ((unsigned)src[3 * PP_RS] << 16);
        o.z = (unsigned)src[4 * PP_RS] | ((unsigned)src[5 * PP_RS] << 16);
        o.w = (unsigned)src[6 * PP_RS] | ((unsigned)src[7 * PP_RS] << 16);
        *reinterpret_cast<uint4*>(pd + ((long long)ci * 9 + t) * Cop + co) = o;
    }
}

// Element-wise batched forms for the jobs the tiled kernel does not take (parameter tensors with more than 3 x 3 positions: the
// 7 x 7 stem, the 4 x 4 transposed convolutions): the same job table, PACKW_ITEMS packed elements per thread, block -> job by
// binary search.  One launch instead of one ctypes call + launch per job (LinkNet34: 21 of them at every step's start).
constexpr int PACKW_ITEMS = 4;
template <bool IS_PACK>
__global__ __launch_bounds__(256) void pack_elem_multi_kernel(const PackJob* __restrict__ jobs, int njobs) {
    const PackJob& j = jobs[find_job(jobs, njobs, blockIdx.x)];
    const long long total = (long long)j.Mp * j.ntaps * j.Cp;
    const long long base = (long long)(blockIdx.x - j.block_start) * (256 * PACKW_ITEMS);
    float* param = const_cast<float*>(j.w);
#pragma unroll
    for (int u = 0; u < PACKW_ITEMS; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i >= total) continue;
        const int cp = (int)(i % j.Cp);
        const long long q = i / j.Cp;
        const int t = (int)(q % j.ntaps);
        const int mp = (int)(q / j.ntaps);
        const int m = j.mmap[mp], c = j.cmap[cp];
        if (IS_PACK) {
            float v = 0.f;
            if (m >= 0 && c >= 0) v = param[m * j.s_m + c * j.s_c + j.tap_off[t]];
            if (j.dtype == SEGNB_BF16) reinterpret_cast<bf16_t*>(j.packed)[i] = Elem<bf16_t>::from_f32(v);
            else reinterpret_cast<float*>(j.packed)[i] = v;
        } else {
            float* dwp = reinterpret_cast<float*>(j.packed);
            if (m >= 0 && c >= 0) param[m * j.s_m + c * j.s_c + j.tap_off[t]] += dwp[i];
            dwp[i] = 0.f;   // workspace is consumed: ready for the next step's atomics without a memset
        }
    }
}

template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int H, int W,
                                  int Cp, int ld) {
    // one thread per (pixel, 8-channel chunk); reads are W-contiguous per channel plane
    const long long npix = (long long)N * H * W;
    const int CPP = Cp / 8;
    const long long total = npix * CPP;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / CPP;
        const int cc = (int)(i - pix * CPP);
        const long long n = pix / ((long long)H * W);
        const long long hw = pix - n * (long long)H * W;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cc * 8 + e;
            v[e] = ch < C ? x[(n * C + ch) * (long long)H * W + hw] : 0.f;
        }
        store8(out + pix * ld + cc * 8, v);
    }
}

template <typename K>
int set_smem(K kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        segnb_set_error("hipFuncSetAttribute(%d bytes): %s", bytes, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_fprop(FpropArgs& a, hipStream_t stream) {
    constexpr int smem = fprop_smem_bytes<BM, BN, T>();
    static_assert(smem >= NT * 16 * 4, "BatchNorm-reduce scratch");
    static int attr_rc = [] {
        int rc = set_smem(conv_fprop_kernel<T, BM, BN, WM, WN>, smem);
        if (rc == 0) rc = set_smem(conv_fprop_kernel<T, BM, BN, WM, WN, true>, smem);
        if constexpr (sizeof(T) == 2) {
            if (rc == 0) rc = set_smem(conv_fprop_kernel<T, BM, BN, WM, WN, false, true>, smem);
            if constexpr (BN == 64) {         // (the two-launch form of a dense layer's data gradient: 64-channel tiles only)
                if (rc == 0) rc = set_smem(conv_fprop_kernel<T, BM, BN, WM, WN, false, true, 1>, smem);
                if (rc == 0) rc = set_smem(conv_fprop_kernel<T, BM, BN, WM, WN, false, true, 2>, smem);
            }
        }
        return rc;
    }();
    if (attr_rc) return attr_rc;
    a.MT = ceil_div(a.M, BM);
    a.NTL = ceil_div(a.g.Co, BN);
    const int cus = segnb_num_cus();
    int per_cu = (160 * 1024) / smem;
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    int gm = (cus * per_cu) / a.NTL;
    if (gm < 1) gm = 1;
    if (gm > a.MT) gm = a.MT;
    a.GM = gm;
    const int grid = a.GM * a.NTL;
    if (a.bn_y != nullptr) {
        if constexpr (sizeof(T) == 2) {
            if (a.bn_mode == 0) {
                hipLaunchKernelGGL((conv_fprop_kernel<T, BM, BN, WM, WN, false, true>), dim3(grid), dim3(NT), smem, stream, a);
            } else if constexpr (BN == 64) {
                if (a.bn_mode == 1)
                    hipLaunchKernelGGL((conv_fprop_kernel<T, BM, BN, WM, WN, false, true, 1>), dim3(grid), dim3(NT), smem, stream, a);
                else
                    hipLaunchKernelGGL((conv_fprop_kernel<T, BM, BN, WM, WN, false, true, 2>), dim3(grid), dim3(NT), smem, stream, a);
            } else {
                return SEGNB_E_UNSUPPORTED;
            }
        } else {
            return SEGNB_E_UNSUPPORTED;
        }
    } else if (a.ep_act >= 0)
        hipLaunchKernelGGL((conv_fprop_kernel<T, BM, BN, WM, WN, true>), dim3(grid), dim3(NT), smem, stream, a);
    else
        hipLaunchKernelGGL((conv_fprop_kernel<T, BM, BN, WM, WN>), dim3(grid), dim3(NT), smem, stream, a);
    return 0;
}

template <typename T>
int dispatch_fprop(FpropArgs& a, hipStream_t stream) {
    const int Co = a.g.Co;
    const long long M = a.M;
    const int cus = segnb_num_cus();
    const long long want = (long long)cus * 3 / 2;
    if (Co <= 32) return launch_fprop<T, 128, 32, 32, 32>(a, stream);
    if (Co <= 64) {
        if ((M + 127) / 128 >= want) return launch_fprop<T, 128, 64, 64, 32>(a, stream);
        return launch_fprop<T, 64, 64, 32, 32>(a, stream);
    }
    const long long t128 = ((M + 127) / 128) * ((Co + 127) / 128);
    // (BatchNorm-reduce store pass: eight y rows in flight + the sums would leave the 128 x 128 tile one block per CU -- its K
    // is 144 deep, the second read of the pixel rows by the narrower tile is cheap)
    if (t128 >= want && a.bn_y == nullptr) return launch_fprop<T, 128, 128, 64, 64>(a, stream);
    const long long t64 = ((M + 127) / 128) * ((Co + 63) / 64);
    if (t64 >= want) return launch_fprop<T, 128, 64, 64, 32>(a, stream);
    return launch_fprop<T, 64, 64, 32, 32>(a, stream);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_wgrad(WgradArgs& a, hipStream_t stream) {
    constexpr int smem = (BM + BN) * LDS_ROW;
    static int attr_rc = set_smem(conv_wgrad_kernel<T, BM, BN, WM, WN>, smem);
    if (attr_rc) return attr_rc;
    constexpr int BKP = 8 * Elem<T>::EPC;
    a.MT = ceil_div(a.g.Co, BM);
    a.NTL = ceil_div(a.Ktot, BN);
    a.total_steps = ceil_div(a.M, BKP);
    const int tiles = a.MT * a.NTL;
    const int cus = segnb_num_cus();
    // blocks per CU the pixel range is split for (measured on LinkNet34's 7x7 / transposed / 2x2 layers: 2 -> 5.53 ms of
    // weight gradients per step, 4 -> 5.19 ms, 8 -> 5.10 ms; step 10.7 -> 10.3 ms at 4)
    constexpr int per_cu = 8;
    int S = (cus * per_cu + tiles - 1) / tiles;
    if (S < 1) S = 1;
    // keep at least 4 K steps per split so the pipeline prologue amortises
    const int maxS = a.total_steps / 4 > 0 ? a.total_steps / 4 : 1;
    if (S > maxS) S = maxS;
    a.steps_per_split = ceil_div(a.total_steps, S);
    a.S = ceil_div(a.total_steps, a.steps_per_split);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, BM, BN, WM, WN>), dim3(tiles * a.S), dim3(NT), smem, stream, a);
    return 0;
}

template <typename T>
int dispatch_wgrad(WgradArgs& a, hipStream_t stream) {
    const int Co = a.g.Co;
    if (Co <= 32) return launch_wgrad<T, 32, 128, 32, 32>(a, stream);
    if (Co <= 64) return launch_wgrad<T, 64, 128, 32, 64>(a, stream);
    return launch_wgrad<T, 128, 128, 64, 64>(a, stream);
}

int check_geom(const segnb_conv_geom* g) {
    SEGNB_CHECK_ARG(g != nullptr, "geom is NULL");
    SEGNB_CHECK_ARG(g->ntaps >= 1 && g->ntaps <= SEGNB_MAX_TAPS, "ntaps out of range");
    SEGNB_CHECK_ARG(g->Ci > 0 && g->Ci % 8 == 0 && g->Co > 0 && g->Co % 8 == 0, "Ci/Co must be multiples of 8");
    SEGNB_CHECK_ARG(g->ld_in % 8 == 0 && g->ld_out % 8 == 0 && g->ld_in >= g->Ci && g->ld_out >= g->Co,
                    "ld must be a multiple of 8 and >= channels");
    SEGNB_CHECK_ARG(g->N > 0 && g->QH > 0 && g->QW > 0 && g->Hi > 0 && g->Wi > 0 && g->Ho > 0 && g->Wo > 0,
                    "empty tensor");
    SEGNB_CHECK_ARG(g->in_step >= 1 && g->out_step >= 1, "steps must be >= 1");
    SEGNB_CHECK_ARG((g->QH - 1) * g->out_step + g->oh0 < g->Ho && (g->QW - 1) * g->out_step + g->ow0 < g->Wo &&
                        g->oh0 >= 0 && g->ow0 >= 0,
                    "output positions exceed the output tensor");
    SEGNB_CHECK_ARG((long long)g->N * g->Hi * g->Wi < (1ll << 31) && (long long)g->N * g->Ho * g->Wo < (1ll << 31),
                    "pixel count exceeds int32");
    return 0;
}

}  // namespace

static int conv_fprop_impl(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, const float* bias,
                           int bias_n, void* out, double* stats, segnb_stream_t stream, const segnb_act_epilogue* ep,
                           const segnb_bn_reduce_epilogue* bn = nullptr, const segnb_bn_apply_epilogue* ap = nullptr, int bn_mode = 0);

extern "C" int segnb_conv_fprop(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                                const float* bias, int bias_n, void* out, double* stats,
                                segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop, g, dtype, in, wpacked, bias, bias_n, out, stats, stream);
    return conv_fprop_impl(g, dtype, in, wpacked, bias, bias_n, out, stats, stream, nullptr);
}

extern "C" int segnb_conv_fprop_act(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                                    const float* bias, int bias_n, void* out, const segnb_act_epilogue* ep,
                                    segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_act, g, dtype, in, wpacked, bias, bias_n, out, ep, stream);
    SEGNB_CHECK_ARG(ep != nullptr && ep->act >= SEGNB_ACT_NONE && ep->act <= SEGNB_ACT_LEAKY, "bad epilogue");
    return conv_fprop_impl(g, dtype, in, wpacked, bias, bias_n, out, nullptr, stream, ep);
}

static int conv_fprop_impl(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, const float* bias,
                           int bias_n, void* out, double* stats, segnb_stream_t stream, const segnb_act_epilogue* ep,
                           const segnb_bn_reduce_epilogue* bn, const segnb_bn_apply_epilogue* ap, int bn_mode) {
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(in && wpacked && (out || bn_mode != 0), "NULL tensor");
    FpropArgs a;
    a.bn_y = nullptr;
    a.bn_mode = bn_mode;
    a.bn_acc = nullptr;
    a.bn_accumulate = 0;
    if (bn != nullptr) {          // (the general kernel's BatchNorm-reduce store pass: segnb_conv_fprop_bnreduce)
        a.bn_y = bn->y;
        a.bn_ld = bn->ld_y;
        a.bn_coef = bn->coef;
        a.bn_sums = bn->sums;
        a.bn_act = bn->act;
        a.bn_slope = bn->slope;
    }
    if (ap != nullptr) {          // (segnb_conv_fprop_bnapply)
        a.bn_y = ap->y;
        a.bn_ld = ap->ld_y;
        a.bn_coef = ap->coef;
        a.bn_sums = const_cast<double*>(ap->sums);
        a.bn_act = ap->act;
        a.bn_slope = ap->slope;
        a.bn_count = ap->count;
        a.bn_gamma = ap->gamma;
        a.bn_C = ap->C;
        a.bn_bcoef = ap->bcoef;
        a.bn_dgamma = ap->dgamma;
        a.bn_dbeta = ap->dbeta;
        a.bn_acc = ap->dx;
        a.bn_acc_ld = ap->ld_dx;
        a.bn_accumulate = ap->accumulate;
    }
    a.ep_act = ep != nullptr ? ep->act : -1;
    a.ep_coef = ep != nullptr ? ep->coef : nullptr;
    a.ep_slope = ep != nullptr ? ep->slope : 0.f;
    a.drop = nullptr;
    a.ld_drop = 0;
    a.stats_ld = g->Co;
    a.g = *g;
    a.in = in;
    a.w = wpacked;
    {
        const long long esz = dtype == SEGNB_BF16 ? 2 : 4;
        const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * esz;
        const long long wb = (long long)g->Co * g->ntaps * g->Ci * esz;
        SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
        a.in_bytes = (unsigned)inb;
        a.w_bytes = (unsigned)wb;
    }
    a.bias = bias;
    a.bias_n = bias_n;
    a.out = out;
    a.stats = stats;
    a.M = g->N * g->QH * g->QW;
    a.Ktot = g->ntaps * g->Ci;
    int rc;
    if (dtype == SEGNB_BF16) {
        // stride-1 3x3: image halo tile staged once in LDS, all taps from shifted rows (fprop_s1.hip)
        static const bool general_env = getenv("SEGNB_FPROP_GENERAL") != nullptr;   // A/B testing only
        const bool general_only = general_env || bn != nullptr || ap != nullptr;
        // (the affine + activation epilogue lives in the c8, rw, ws and general kernels: s1 is skipped for it)
        rc = general_only ? 0 : segnb_fprop_c8_try(g, in, wpacked, bias, bias_n, out, stats, (hipStream_t)stream, ep);
        if (rc == 0 && !general_env && ap == nullptr && bn_mode == 0 && out != nullptr && ep == nullptr && stats == nullptr && bias == nullptr)
            rc = segnb_fprop_thin_try(g, in, wpacked, out, (hipStream_t)stream, bn);      // (thin input, wide output; bn or plain)
        if (rc == 0 && !general_only && ep == nullptr)
            rc = segnb_fprop_roll_try(g, in, a.in_bytes, wpacked, a.w_bytes, bias, bias_n, out, stats, (hipStream_t)stream);
        if (rc == 0 && !general_only)
            rc = segnb_fprop_rw_try(g, in, a.in_bytes, wpacked, a.w_bytes, bias, bias_n, out, stats,
                                    (hipStream_t)stream, nullptr, ep);
        if (rc == 0 && !general_only)
            rc = segnb_fprop_dma_try(g, in, a.in_bytes, wpacked, a.w_bytes, bias, bias_n, out, stats,
                                     (hipStream_t)stream, ep);
        if (rc == 0 && !general_only && fprop_deepk_applies(a)) {
            // few pixels x few channels x deep K (before the halo-tile kernel below, whose blocks also walk K serially)
            hipLaunchKernelGGL(conv_fprop_deepk_kernel, dim3(ceil_div(a.M, 32), ceil_div(g->Co, 32)), dim3(DK_WAVES * 64), 0,
                               (hipStream_t)stream, a);
            rc = 1;
        }
        if (rc == 0 && !general_only && ep == nullptr)
            rc = segnb_fprop_s1_try(g, in, wpacked, bias, bias_n, out, stats, (hipStream_t)stream);
        if (rc == 0 && !general_only && ep == nullptr)
            rc = segnb_fprop_sx_try(g, in, wpacked, bias, bias_n, out, stats, (hipStream_t)stream);
        if (rc == 1) {
            SEGNB_LAUNCH_CHECK();
            return 0;
        }
        if (rc != 0) return rc;
        a.ksteps = ceil_div(a.Ktot, 64);
        rc = dispatch_fprop<bf16_t>(a, (hipStream_t)stream);
    } else if (dtype == SEGNB_F32) {
        a.ksteps = ceil_div(a.Ktot, 32);
        rc = dispatch_fprop<float>(a, (hipStream_t)stream);
    } else {
        segnb_set_error("segnb_conv_fprop: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    if (rc) return rc;
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// ---- conv -> Dropout2d -> statistics in one launch (include/segnb_hip.h: segnb_conv_fprop_drop) -------------------------------------
// served where the plain cascade above ends in conv_fprop_deepk_kernel or conv_fprop_s1x9_kernel: <= 32 output channels behind more
// than 96 input channels (below that the rolling / LDS-DMA kernels take the launch), a stride-1 3 x 3 window
static bool drop_s1_shape(const segnb_conv_geom* g) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Wo <= 8) return false;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    return dhmax - dhmin == 2 && dwmax - dwmin == 2;
}

extern "C" int segnb_conv_fprop_drop_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || getenv("SEGNB_FPROP_GENERAL") != nullptr || !segnb_knob_fprop_drop()) return 0;
    if (g->Co > 32 || g->Co % 8 != 0 || g->Ci <= 96 || g->Ci % 8 != 0) return 0;
    const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    const long long wb = (long long)g->Co * g->ntaps * g->Ci * 2;
    if (inb >= (1ll << 31) || wb >= (1ll << 31)) return 0;
    return (fprop_deepk_shape(*g) || drop_s1_shape(g)) ? 1 : 0;
}

extern "C" int segnb_conv_fprop_drop(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, const float* bias,
                                     int bias_n, void* out, const float* dropmul, int ld_drop, double* stats, int stats_ld,
                                     segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_drop, g, dtype, in, wpacked, bias, bias_n, out, dropmul, ld_drop, stats, stats_ld, stream);
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(in && wpacked && out && dropmul, "NULL argument");
    SEGNB_CHECK_ARG(segnb_conv_fprop_drop_ok(g, dtype), "geometry not served (segnb_conv_fprop_drop_ok)");
    SEGNB_CHECK_ARG(ld_drop >= g->Co && (stats == nullptr || stats_ld >= g->Co), "bad strides");
    if (fprop_deepk_shape(*g)) {
        FpropArgs a;
        a.g = *g;
        a.in = in;
        a.w = wpacked;
        a.in_bytes = (unsigned)((((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2);
        a.w_bytes = (unsigned)((long long)g->Co * g->ntaps * g->Ci * 2);
        a.bias = bias;
        a.bias_n = bias_n;
        a.ep_coef = nullptr;
        a.ep_act = -1;
        a.ep_slope = 0.f;
        a.out = out;
        a.stats = stats;
        a.M = g->N * g->QH * g->QW;
        a.Ktot = g->ntaps * g->Ci;
        a.ksteps = a.MT = a.NTL = a.GM = 0;
        a.bn_y = nullptr;
        a.bn_mode = 0;
        a.bn_acc = nullptr;
        a.bn_accumulate = 0;
        a.drop = dropmul;
        a.ld_drop = ld_drop;
        a.stats_ld = stats_ld > 0 ? stats_ld : g->Co;
        hipLaunchKernelGGL(conv_fprop_deepk_kernel, dim3(ceil_div(a.M, 32), ceil_div(g->Co, 32)), dim3(DK_WAVES * 64), 0,
                           (hipStream_t)stream, a);
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    const int rc = segnb_fprop_s1_try(g, in, wpacked, bias, bias_n, out, stats, (hipStream_t)stream, dropmul, ld_drop, stats_ld);
    if (rc != 1) {
        segnb_set_error("segnb_conv_fprop_drop: the kernel refused the launch (%d)", rc);
        return rc > 1 ? rc : SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// ---- forward of an Upsample(x2) -> conv3x3 segment on the low-resolution tensor, accumulating (fprop_dma.hip, WsCfg UP_ = 2)
int segnb_fprop_upf_try(int N, int H, int W, int Ci, int ld_in, const void* in, unsigned in_bytes, const void* wpacked,
                        unsigned w_bytes, int Co, int CoW, void* out, int ld_out, double* stats, hipStream_t stream,
                        const float* bias = nullptr, int bias_n = 0, int no_prev = 0, int ep_act = -1, float ep_slope = 0.f);

extern "C" int segnb_upconv_fprop_acc_ok(int N, int H, int W, int Ci, int Co, int ld_out, int dtype) {
    if (dtype != SEGNB_BF16 || getenv("SEGNB_FPROP_GENERAL") != nullptr || !segnb_knob_fprop_dma() || !segnb_knob_fprop_upd()) return 0;
    if (N <= 0 || H <= 0 || W <= 0 || Ci % 64 != 0 || Ci < 128 || Co <= 32 || Co % 8 != 0 || W < 12) return 0;
    return (((long long)N * 4 * H * W - 1) * ld_out + Co) * 2 < (1ll << 31) ? 1 : 0;
}

extern "C" int segnb_upconv_fprop_acc(int dtype, int N, int H, int W, int Ci, int ld_in, const void* in, const void* wpacked,
                                      int Co, int CoW, void* out, int ld_out, double* stats, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_upconv_fprop_acc, dtype, N, H, W, Ci, ld_in, in, wpacked, Co, CoW, out, ld_out, stats, stream);
    SEGNB_CHECK_ARG(in && wpacked && out, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_upconv_fprop_acc_ok(N, H, W, Ci, Co, ld_out, dtype), "shape not served (segnb_upconv_fprop_acc_ok)");
    SEGNB_CHECK_ARG(CoW >= Co && ld_in >= Ci && ld_out >= Co, "bad strides");
    const long long inb = (((long long)N * H * W - 1) * ld_in + Ci) * 2, wb = 4ll * CoW * 4 * Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    const int rc = segnb_fprop_upf_try(N, H, W, Ci, ld_in, in, (unsigned)inb, wpacked, (unsigned)wb, Co, CoW, out, ld_out, stats,
                                       (hipStream_t)stream);
    if (rc == 1) {
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_upconv_fprop_acc: no kernel for this shape");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

// ---- forward of a ConvTranspose2d(4, 2, 1) (unet16.py:30) on the same kernel: the four output phases in one launch, nothing to
// accumulate into, bias in the accumulator staging
extern "C" int segnb_upconv_fprop_ok(int N, int H, int W, int Ci, int Co, int ld_out, int dtype) {
    if (dtype != SEGNB_BF16 || getenv("SEGNB_FPROP_GENERAL") != nullptr || !segnb_knob_fprop_dma() || !segnb_knob_fprop_upd()) return 0;
    if (N <= 0 || H <= 0 || W <= 0 || Ci % 64 != 0 || Ci < 128 || Co % 8 != 0 || Co <= 0 || W < 12) return 0;
    return (((long long)N * 4 * H * W - 1) * ld_out + Co) * 2 < (1ll << 31) ? 1 : 0;
}

extern "C" int segnb_upconv_fprop(int dtype, int N, int H, int W, int Ci, int ld_in, const void* in, const void* wpacked, int Co,
                                  int CoW, const float* bias, int bias_n, void* out, int ld_out, double* stats,
                                  segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_upconv_fprop, dtype, N, H, W, Ci, ld_in, in, wpacked, Co, CoW, bias, bias_n, out, ld_out, stats, stream);
    SEGNB_CHECK_ARG(in && wpacked && out, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_upconv_fprop_ok(N, H, W, Ci, Co, ld_out, dtype), "shape not served (segnb_upconv_fprop_ok)");
    SEGNB_CHECK_ARG(CoW >= Co && ld_in >= Ci && ld_out >= Co && bias_n >= 0 && bias_n <= Co, "bad strides");
    const long long inb = (((long long)N * H * W - 1) * ld_in + Ci) * 2, wb = 4ll * CoW * 4 * Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    const int rc = segnb_fprop_upf_try(N, H, W, Ci, ld_in, in, (unsigned)inb, wpacked, (unsigned)wb, Co, CoW, out, ld_out, stats,
                                       (hipStream_t)stream, bias_n > 0 ? bias : nullptr, bias_n, 1);
    if (rc == 1) {
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_upconv_fprop: no kernel for this shape");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

// the same with the activation of unet16.py:38-40 (ConvTranspose2d -> ReLU) in the accumulator staging: no pass over the output
extern "C" int segnb_upconv_fprop_act(int dtype, int N, int H, int W, int Ci, int ld_in, const void* in, const void* wpacked, int Co,
                                      int CoW, const float* bias, int bias_n, void* out, int ld_out, const segnb_act_epilogue* ep,
                                      segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_upconv_fprop_act, dtype, N, H, W, Ci, ld_in, in, wpacked, Co, CoW, bias, bias_n, out, ld_out, ep, stream);
    SEGNB_CHECK_ARG(in && wpacked && out && ep, "NULL tensor");
    SEGNB_CHECK_ARG(ep->coef == nullptr && (ep->act == SEGNB_ACT_NONE || ep->act == SEGNB_ACT_RELU || ep->act == SEGNB_ACT_LEAKY),
                    "activation only (no folded BatchNorm on this kernel)");
    SEGNB_CHECK_ARG(ep->act != SEGNB_ACT_LEAKY || (ep->slope >= 0.f && ep->slope <= 1.f), "leaky slope outside [0, 1]");
    SEGNB_CHECK_ARG(segnb_upconv_fprop_ok(N, H, W, Ci, Co, ld_out, dtype), "shape not served (segnb_upconv_fprop_ok)");
    SEGNB_CHECK_ARG(CoW >= Co && ld_in >= Ci && ld_out >= Co && bias_n >= 0 && bias_n <= Co, "bad strides");
    const long long inb = (((long long)N * H * W - 1) * ld_in + Ci) * 2, wb = 4ll * CoW * 4 * Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    const int rc = segnb_fprop_upf_try(N, H, W, Ci, ld_in, in, (unsigned)inb, wpacked, (unsigned)wb, Co, CoW, out, ld_out, nullptr,
                                       (hipStream_t)stream, bias_n > 0 ? bias : nullptr, bias_n, 1, ep->act, ep->slope);
    if (rc == 1) {
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_upconv_fprop_act: no kernel for this shape");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

// ---- virtual concat: cat([Upsample x2(u), skip]) read from the two tensors (include/segnb_hip.h)
static bool wgrad_general_only();
static bool upcat_geom_ok(const segnb_conv_geom* g, int dtype, int Cu) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || getenv("SEGNB_FPROP_GENERAL") != nullptr || wgrad_general_only()) return false;
    if (!segnb_knob_fprop_dma() || g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || (g->Hi & 1) || (g->Wi & 1) || g->Hi != g->Ho || g->Wi != g->Wo || g->Wo < 12) return false;
    if (Cu <= 0 || Cu >= g->Ci) return false;
    const bool thin = g->Ci % 32 == 0 && g->Ci <= 96 && g->Co <= 64 && Cu % 32 == 0 && segnb_knob_fprop_rw();      // fprop_rw.hip
    const bool wide = g->Ci % 64 == 0 && Cu % 64 == 0 && g->Co > 32;                                                  // fprop_dma.hip
    if (!thin && !wide) return false;
    return segnb_wgrad_s1_slabs(g) > 0;
}

extern "C" int segnb_conv_upcat_ok(const segnb_conv_geom* g, int dtype, int Cu) { return upcat_geom_ok(g, dtype, Cu) ? 1 : 0; }

extern "C" int segnb_conv_fprop_upcat(const segnb_conv_geom* g, int dtype, const void* in, const segnb_upcat_src* src,
                                      const void* wpacked, const float* bias, int bias_n, void* out, double* stats,
                                      segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_upcat, g, dtype, in, src, wpacked, bias, bias_n, out, stats, stream);
    SEGNB_CHECK_ARG(in && src && src->u && wpacked && out, "NULL tensor");
    SEGNB_CHECK_ARG(upcat_geom_ok(g, dtype, src->Cu), "geometry not served (segnb_conv_upcat_ok)");
    const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + (g->Ci - src->Cu)) * 2;
    const long long wb = (long long)g->Co * g->ntaps * g->Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    int rc = segnb_fprop_roll_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, bias, bias_n, out, stats, (hipStream_t)stream,
                                  nullptr, nullptr, src);
    if (rc == 0)
        rc = segnb_fprop_rw_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, bias, bias_n, out, stats, (hipStream_t)stream,
                                nullptr, nullptr, src);
    if (rc == 0)
        rc = segnb_fprop_dma_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, bias, bias_n, out, stats, (hipStream_t)stream,
                                 nullptr, src);
    if (rc == 1) {
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_conv_fprop_upcat: no kernel for this geometry");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

static bool upsum_geom_ok(const segnb_conv_geom* g, int dtype, int Cu) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || getenv("SEGNB_FPROP_GENERAL") != nullptr) return false;
    if (!segnb_knob_fprop_dma() || !segnb_knob_fprop_rw() || g->ntaps != 9 || g->in_step != 1 || g->out_step != 1) return false;
    if (g->oh0 != 0 || g->ow0 != 0 || g->QH != g->Ho || g->QW != g->Wo || (g->Ho & 1) || (g->Wo & 1) || g->Wo < 12) return false;
    return g->Ci % 32 == 0 && g->Ci <= 96 && g->Co <= 96 && g->Co % 8 == 0 && Cu % 8 == 0 && Cu > 0 && Cu < g->Co;
}

extern "C" int segnb_conv_fprop_upsum_ok(const segnb_conv_geom* g, int dtype, int Cu) { return upsum_geom_ok(g, dtype, Cu) ? 1 : 0; }

extern "C" int segnb_conv_fprop_upsum(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, void* out,
                                      const segnb_upcat_src* dst, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_upsum, g, dtype, in, wpacked, out, dst, stream);
    SEGNB_CHECK_ARG(in && wpacked && out && dst && dst->u, "NULL tensor");
    SEGNB_CHECK_ARG(upsum_geom_ok(g, dtype, dst->Cu), "geometry not served (segnb_conv_fprop_upsum_ok)");
    const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    const long long wb = (long long)g->Co * g->ntaps * g->Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    const int rc = segnb_fprop_rw_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, nullptr, 0, out, nullptr, (hipStream_t)stream,
                                      nullptr, nullptr, nullptr, dst);
    if (rc == 1) {
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_conv_fprop_upsum: no kernel for this geometry");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

extern "C" int segnb_conv_wgrad_upcat(const segnb_conv_geom* g, int dtype, const void* in, const segnb_upcat_src* src,
                                      const void* dout, float* dwp, int nslab, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_wgrad_upcat, g, dtype, in, src, dout, dwp, nslab, stream);
    const segnb_wgrad_target* const tgt = segnb_take_wgrad_target();
    SEGNB_CHECK_ARG(in && src && src->u && dout && dwp, "NULL tensor");
    SEGNB_CHECK_ARG(tgt == nullptr || (tgt->ntaps == g->ntaps && tgt->Co <= g->Co && g->Co - tgt->Co < 8 && tgt->Ci <= g->Ci),
                    "the armed segnb_wgrad_target does not belong to this geometry (taps / channel counts)");
    SEGNB_CHECK_ARG(upcat_geom_ok(g, dtype, src->Cu), "geometry not served (segnb_conv_upcat_ok)");
    SEGNB_CHECK_ARG(nslab == segnb_conv_wgrad_slabs(g, dtype), "nslab differs from segnb_conv_wgrad_slabs()");
    const int rc = segnb_wgrad_s1_try(g, in, dout, dwp, nslab, (hipStream_t)stream, false, nullptr, src, tgt);
    if (rc == 1 || rc == 2) {
        if (tgt != nullptr && rc == 1) segnb_wgrad_to_param(dwp, g->Co, g->ntaps, g->Ci, nslab, tgt, false, (hipStream_t)stream);
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_conv_wgrad_upcat: no kernel for this geometry");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

// few input channels -> many output channels: the data gradient of a dense layer (tiramisu.py:9-20, growth 16 -> the prefix);
// the general gather kernel serves it, with the reduction in its store pass (conv_fprop_kernel<..., BNR>)
static bool bnreduce_general(const segnb_conv_geom* g) { return g->Ci <= 24 && g->Co >= 32 && g->Co % 8 == 0; }

extern "C" int segnb_conv_fprop_bnreduce_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || !segnb_knob_bnreduce_fused()) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Co % 8 != 0) return 0;
    if (bnreduce_general(g)) return 1;      // (the general gather kernel: any width, any tap offsets)
    if (getenv("SEGNB_FPROP_GENERAL") != nullptr || !segnb_knob_fprop_dma() || !segnb_knob_fprop_rw() || g->Wo < 12) return 0;
    for (int t = 0; t < 9; ++t)
        if (g->dh[t] < -1 || g->dh[t] > 1 || g->dw[t] < -1 || g->dw[t] > 1) return 0;
    if (g->Ci % 32 != 0 || g->Ci > 96 || g->Co > 64) return 0;
    if (g->Co > 32 && g->Ci > 32) return 0;      // (the 64-wide tile keeps one 32-channel chunk of weights resident)
    return 1;
}

extern "C" int segnb_conv_fprop_bnreduce(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                                         void* out, const segnb_bn_reduce_epilogue* ep, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_bnreduce, g, dtype, in, wpacked, out, ep, stream);
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(in && wpacked && out && ep && ep->y && ep->sums, "NULL argument");
    SEGNB_CHECK_ARG(ep->ld_y >= g->Co && ep->ld_y % 8 == 0, "bad y stride");
    const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    const long long wb = (long long)g->Co * g->ntaps * g->Ci * 2;
    SEGNB_CHECK_ARG(inb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB (32-bit buffer offsets)");
    int rc = 0;
    if (ep->coef == nullptr) {
        // activation mask of a producing layer without BatchNorm: out = dz (conv_roll_kernel, EPI = 3)
        SEGNB_CHECK_ARG(segnb_conv_fprop_actmask_ok(g, dtype), "geometry not served by a fused kernel (segnb_conv_fprop_actmask_ok)");
        if (segnb_fprop_roll_actmask_ok(g))
            rc = segnb_fprop_roll_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, nullptr, 0, out, nullptr, (hipStream_t)stream, ep);
        else      // 64-channel-chunk inputs: the MASK instantiation of conv_fprop_ws_kernel
            rc = segnb_fprop_dma_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, nullptr, 0, out, nullptr, (hipStream_t)stream, nullptr,
                                     nullptr, ep);
        if (rc != 1) {
            segnb_set_error("segnb_conv_fprop_bnreduce: the fused kernel refused the launch (%d)", rc);
            return rc > 1 ? rc : SEGNB_E_BADARG;
        }
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    SEGNB_CHECK_ARG(segnb_conv_fprop_bnreduce_ok(g, dtype), "geometry not served by a fused kernel (segnb_conv_fprop_bnreduce_ok)");
    if (bnreduce_general(g)) {
        SEGNB_CHECK_ARG(ep->ld_y % 8 == 0 && (((long long)g->N * g->Ho * g->Wo - 1) * ep->ld_y + g->Co) * 2 < (1ll << 31), "bad y");
        return conv_fprop_impl(g, dtype, in, wpacked, nullptr, 0, out, nullptr, stream, nullptr, ep);
    }
    if (g->Ci <= 96) {
        rc = segnb_fprop_roll_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, nullptr, 0, out, nullptr, (hipStream_t)stream, ep);
        if (rc == 0)
            rc = segnb_fprop_rw_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, nullptr, 0, out, nullptr, (hipStream_t)stream, ep);
    } else {
        rc = segnb_fprop_dma_try(g, in, (unsigned)inb, wpacked, (unsigned)wb, nullptr, 0, out, nullptr, (hipStream_t)stream, nullptr,
                                 nullptr, ep);
    }
    if (rc != 1) {
        segnb_set_error("segnb_conv_fprop_bnreduce: the fused kernel refused the launch (%d)", rc);
        return rc > 1 ? rc : SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// ---- a dense layer's data gradient that is never stored (include/segnb_hip.h): two launches of the general kernel
extern "C" int segnb_conv_fprop_bnapply_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || !segnb_knob_bnreduce_fused()) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Co % 8 != 0 || g->Co <= 32) return 0;      // (64-channel tiles)
    return bnreduce_general(g) ? 1 : 0;
}

extern "C" int segnb_conv_fprop_bnsums(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                                       const segnb_bn_reduce_epilogue* ep, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_bnsums, g, dtype, in, wpacked, ep, stream);
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(in && wpacked && ep && ep->y && ep->coef && ep->sums, "NULL argument");
    SEGNB_CHECK_ARG(segnb_conv_fprop_bnapply_ok(g, dtype), "geometry not served (segnb_conv_fprop_bnapply_ok)");
    SEGNB_CHECK_ARG(ep->ld_y >= g->Co && ep->ld_y % 8 == 0 && (((long long)g->N * g->Ho * g->Wo - 1) * ep->ld_y + g->Co) * 2 < (1ll << 31),
                    "bad y");
    return conv_fprop_impl(g, dtype, in, wpacked, nullptr, 0, nullptr, nullptr, stream, nullptr, ep, nullptr, 1);
}

extern "C" int segnb_conv_fprop_bnapply(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                                        const segnb_bn_apply_epilogue* ep, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_fprop_bnapply, g, dtype, in, wpacked, ep, stream);
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(in && wpacked && ep && ep->y && ep->coef && ep->sums && ep->dx, "NULL argument");
    SEGNB_CHECK_ARG(segnb_conv_fprop_bnapply_ok(g, dtype), "geometry not served (segnb_conv_fprop_bnapply_ok)");
    SEGNB_CHECK_ARG(ep->C > 0 && ep->C <= g->Co && ep->count > 0.0, "bad channel count / pixel count");
    SEGNB_CHECK_ARG(ep->ld_y >= g->Co && ep->ld_y % 8 == 0 && (((long long)g->N * g->Ho * g->Wo - 1) * ep->ld_y + g->Co) * 2 < (1ll << 31),
                    "bad y");
    SEGNB_CHECK_ARG(ep->ld_dx >= g->Co && ep->ld_dx % 8 == 0 && (((long long)g->N * g->Ho * g->Wo - 1) * ep->ld_dx + g->Co) * 2 < (1ll << 31),
                    "bad dx");
    return conv_fprop_impl(g, dtype, in, wpacked, nullptr, 0, nullptr, nullptr, stream, nullptr, nullptr, ep, 2);
}

static bool wgrad_general_only() {
    static const bool v = getenv("SEGNB_WGRAD_GENERAL") != nullptr;   // A/B testing only
    return v;
}

extern "C" int segnb_conv_wgrad_slabs(const segnb_conv_geom* g, int dtype) {
    if (check_geom(g)) return -1;
    if (dtype == SEGNB_BF16 && !wgrad_general_only()) {
        const int s = segnb_wgrad_s1_slabs(g);
        if (s > 0) return s;
        const int sx = segnb_wgrad_sx_slabs(g);
        if (sx > 0) return sx;
    }
    return 1;
}

extern "C" int segnb_conv_wgrad(const segnb_conv_geom* g, int dtype, const void* in, const void* dout,
                                float* dwp, int nslab, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_wgrad, g, dtype, in, dout, dwp, nslab, stream);
    const segnb_wgrad_target* const tgt = segnb_take_wgrad_target();
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(tgt == nullptr || (tgt->ntaps == g->ntaps && tgt->Co <= g->Co && g->Co - tgt->Co < 8 && tgt->Ci <= g->Ci),
                    "the armed segnb_wgrad_target does not belong to this geometry (taps / channel counts)");
    SEGNB_CHECK_ARG(in && dout && dwp, "NULL tensor");
    SEGNB_CHECK_ARG(nslab == segnb_conv_wgrad_slabs(g, dtype), "nslab differs from segnb_conv_wgrad_slabs()");
    WgradArgs a;
    a.g = *g;
    a.in = in;
    a.dout = dout;
    a.dwp = dwp;
    a.M = g->N * g->QH * g->QW;
    a.Ktot = g->ntaps * g->Ci;
    int rc;
    if (dtype == SEGNB_BF16) {
        // stride-1 3x3: pixel-major LDS tiles + transposing LDS reads, all taps per block (wgrad_s1.hip)
        // (with a target the fast kernels leave their slabs unreduced: segnb_wgrad_to_param sums them into the parameter's gradient)
        const bool part = tgt != nullptr;
        rc = (wgrad_general_only() || !segnb_knob_wgrad_roll()) ? 0 : segnb_wgrad_roll_try(g, in, dout, dwp, nslab, (hipStream_t)stream, part);
        if (rc == 0 && !wgrad_general_only() && segnb_knob_wgrad_c8roll() && segnb_wgrad_s1_slabs(g) > 0)
            rc = segnb_wgrad_c8roll_try(g, in, dout, dwp, nslab, (hipStream_t)stream, part);
        if (rc == 0)
            rc = wgrad_general_only() ? 0 : segnb_wgrad_s1_try(g, in, dout, dwp, nslab, (hipStream_t)stream, false, nullptr, nullptr, tgt);
        if (rc == 1 || rc == 2) {
            if (tgt != nullptr && rc == 1) segnb_wgrad_to_param(dwp, g->Co, g->ntaps, g->Ci, nslab, tgt, false, (hipStream_t)stream);
            SEGNB_LAUNCH_CHECK();
            return 0;
        }
        if (rc != 0) return rc;
        rc = wgrad_general_only() ? 0 : segnb_wgrad_sx_try(g, in, dout, dwp, nslab, (hipStream_t)stream, part);
        if (rc == 1) {
            if (tgt != nullptr) segnb_wgrad_to_param(dwp, g->Co, g->ntaps, g->Ci, nslab, tgt, false, (hipStream_t)stream);
            SEGNB_LAUNCH_CHECK();
            return 0;
        }
        if (rc != 0) return rc;
        if (wgrad_co8_applies(g)) {
            const int nslot = g->ntaps * (g->Ci / 8), PL = NT / nslot;
            int grid = segnb_num_cus() * 8;
            if (grid > g->N * g->QH) grid = g->N * g->QH;
            hipLaunchKernelGGL(conv_wgrad_co8_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a, nslot, PL);
        } else
            rc = dispatch_wgrad<bf16_t>(a, (hipStream_t)stream);
    } else if (dtype == SEGNB_F32)
        rc = dispatch_wgrad<float>(a, (hipStream_t)stream);
    else {
        segnb_set_error("segnb_conv_wgrad: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    if (rc) return rc;
    // general kernels: one slab, accumulated with atomics into the zeroed workspace -- delivered and re-zeroed in one pass
    if (tgt != nullptr) segnb_wgrad_to_param(dwp, g->Co, g->ntaps, g->Ci, 1, tgt, true, (hipStream_t)stream);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// weight gradient whose dy operand is the BatchNorm-backward apply of (g, y), recomputed in the kernel (first layer of a
// network: nothing else reads that dy -- no data gradient -- so the apply pass and its tensor disappear)
extern "C" int segnb_conv_wgrad_bnapply_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || wgrad_general_only()) return 0;
    // SEGNB_WGRAD_BNAPPLY: unset = only where the rolling first-layer kernel serves (conv_wgrad_c8roll_kernel: -0.85 % of the
    // timed step), 1 = the thin tile kernel's variant too (measured neutral: profiles/r04_ab.txt), 0 = never
    const char* e = getenv("SEGNB_WGRAD_BNAPPLY");
    if ((e != nullptr && e[0] == '0') || g->Co > 32 || g->Co % 8 != 0 || segnb_wgrad_s1_slabs(g) <= 0) return 0;
    if (segnb_knob_wgrad_c8roll() && segnb_wgrad_c8roll_applies(g)) return 1;
    return e != nullptr ? 1 : 0;
}

extern "C" int segnb_conv_wgrad_bnapply(const segnb_conv_geom* g, int dtype, const void* in, const void* gsrc, int ld_g,
                                        const void* y, int ld_y, const float* coef, const float* bcoef, int Cp, int act,
                                        float slope, float* dwp, int nslab, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_wgrad_bnapply, g, dtype, in, gsrc, ld_g, y, ld_y, coef, bcoef, Cp, act, slope, dwp, nslab, stream);
    const segnb_wgrad_target* const tgt = segnb_take_wgrad_target();
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(tgt == nullptr || (tgt->ntaps == g->ntaps && tgt->Co <= g->Co && g->Co - tgt->Co < 8 && tgt->Ci <= g->Ci),
                    "the armed segnb_wgrad_target does not belong to this geometry (taps / channel counts)");
    SEGNB_CHECK_ARG(in && gsrc && y && coef && bcoef && dwp, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_conv_wgrad_bnapply_ok(g, dtype), "geometry not served (segnb_conv_wgrad_bnapply_ok)");
    SEGNB_CHECK_ARG(nslab == segnb_conv_wgrad_slabs(g, dtype), "nslab differs from segnb_conv_wgrad_slabs()");
    SEGNB_CHECK_ARG(Cp >= g->Co && ld_g >= g->Co && ld_y >= g->Co, "bad strides");
    const segnb_wgrad_bnapply bna = {gsrc, ld_g, y, ld_y, coef, bcoef, Cp, act, slope};
    int rc = segnb_knob_wgrad_c8roll() ? segnb_wgrad_c8roll_try(g, in, nullptr, dwp, nslab, (hipStream_t)stream, tgt != nullptr, &bna) : 0;
    if (rc == 0) rc = segnb_wgrad_s1_try(g, in, nullptr, dwp, nslab, (hipStream_t)stream, false, &bna, nullptr, tgt);
    if (rc == 1 || rc == 2) {
        if (tgt != nullptr && rc == 1) segnb_wgrad_to_param(dwp, g->Co, g->ntaps, g->Ci, nslab, tgt, false, (hipStream_t)stream);
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_conv_wgrad_bnapply: no kernel for this geometry");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

// weight gradient whose operands are recomputed on load (include/segnb_hip.h: segnb_operand_tf)
extern "C" int segnb_conv_wgrad_tf_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16 || check_geom(g) || wgrad_general_only()) return 0;
    return (segnb_wgrad_roll_applies(g) && segnb_wgrad_s1_slabs(g) > 0) ? 1 : 0;
}

extern "C" int segnb_conv_wgrad_tf(const segnb_conv_geom* g, int dtype, const void* in, const segnb_operand_tf* tf_in,
                                   const void* dout, const segnb_operand_tf* tf_dout, float* dwp, int nslab,
                                   segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_conv_wgrad_tf, g, dtype, in, tf_in, dout, tf_dout, dwp, nslab, stream);
    const segnb_wgrad_target* const tgt = segnb_take_wgrad_target();
    if (int rc = check_geom(g)) return rc;
    SEGNB_CHECK_ARG(tgt == nullptr || (tgt->ntaps == g->ntaps && tgt->Co <= g->Co && g->Co - tgt->Co < 8 && tgt->Ci <= g->Ci),
                    "the armed segnb_wgrad_target does not belong to this geometry (taps / channel counts)");
    SEGNB_CHECK_ARG(in && dout && dwp, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_conv_wgrad_tf_ok(g, dtype), "geometry not served (segnb_conv_wgrad_tf_ok)");
    SEGNB_CHECK_ARG(nslab == segnb_conv_wgrad_slabs(g, dtype), "nslab differs from segnb_conv_wgrad_slabs()");
    SEGNB_CHECK_ARG(tf_in == nullptr || (tf_in->kind == SEGNB_TF_ACT && tf_in->coef != nullptr && tf_in->Cp >= g->Ci), "bad input transform");
    SEGNB_CHECK_ARG(tf_dout == nullptr || (tf_dout->kind == SEGNB_TF_BNBWD && tf_dout->coef && tf_dout->bcoef && tf_dout->y &&
                                           tf_dout->drop == nullptr && tf_dout->Cp >= g->Co), "bad dout transform");
    const int rc = segnb_wgrad_roll_try(g, in, dout, dwp, nslab, (hipStream_t)stream, tgt != nullptr, tf_in, tf_dout);
    if (rc == 1) {
        if (tgt != nullptr) segnb_wgrad_to_param(dwp, g->Co, g->ntaps, g->Ci, nslab, tgt, false, (hipStream_t)stream);
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (rc == 0) {
        segnb_set_error("segnb_conv_wgrad_tf: no kernel for this geometry");
        return SEGNB_E_UNSUPPORTED;
    }
    return rc;
}

static int fill_pack_args(PackArgs& p, int Mp, int Cp, int ntaps, long long s_m, long long s_c,
                          const int* tap_off_host, const int* mmap, const int* cmap) {
    SEGNB_CHECK_ARG(Mp > 0 && Cp > 0 && ntaps >= 1 && ntaps <= SEGNB_MAX_TAPS, "bad packed shape");
    SEGNB_CHECK_ARG(tap_off_host && mmap && cmap, "NULL map");
    p.Mp = Mp;
    p.Cp = Cp;
    p.ntaps = ntaps;
    p.s_m = s_m;
    p.s_c = s_c;
    p.mmap = mmap;
    p.cmap = cmap;
    for (int t = 0; t < ntaps; ++t) p.tap_off[t] = tap_off_host[t];
    return 0;
}

extern "C" int segnb_pack_weight(const float* w, void* wpacked, int dtype, int Mp, int Cp, int ntaps,
                                 long long s_m, long long s_c, const int* tap_off_host, const int* mmap,
                                 const int* cmap, segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_pack_weight takes a host tap table");
    SEGNB_CHECK_ARG(w && wpacked, "NULL tensor");
    PackArgs p;
    if (int rc = fill_pack_args(p, Mp, Cp, ntaps, s_m, s_c, tap_off_host, mmap, cmap)) return rc;
    const long long total = (long long)Mp * ntaps * Cp;
    int grid = ceil_div(total, 256);
    if (grid > 4096) grid = 4096;
    if (dtype == SEGNB_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                           (bf16_t*)wpacked, p);
    else if (dtype == SEGNB_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                           (float*)wpacked, p);
    else {
        segnb_set_error("segnb_pack_weight: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_unpack_wgrad(float* dwp, float* gw, int Mp, int Cp, int ntaps, long long s_m,
                                  long long s_c, const int* tap_off_host, const int* mmap, const int* cmap,
                                  int accumulate, segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_unpack_wgrad takes a host tap table");
    SEGNB_CHECK_ARG(dwp && gw, "NULL tensor");
    PackArgs p;
    if (int rc = fill_pack_args(p, Mp, Cp, ntaps, s_m, s_c, tap_off_host, mmap, cmap)) return rc;
    const long long total = (long long)Mp * ntaps * Cp;
    int grid = ceil_div(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dwp, gw, p, accumulate);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_pack_job_bytes(void) { return (int)sizeof(PackJob); }

// number of blocks (tiles) a job occupies -- the host needs it to fill PackJob.block_start
extern "C" int segnb_pack_job_blocks(int Mp, int Cp, int ntaps, long long s_m, long long s_c) {
    PackJob j;
    j.Mp = Mp; j.Cp = Cp; j.ntaps = ntaps; j.s_m = s_m; j.s_c = s_c;
    const bool fast_is_c = s_c < s_m;
    const long long tsrc = fast_is_c ? s_c : s_m;
    // tiled kernel sized for parameter tensors of <= 3x3 positions.  An unmasked job has at most one packed tap per kernel
    // position (ntaps <= tsrc <= 9); more packed taps than positions means a MASKED job (several packed taps per position,
    // PackJob.masked: the 4 x 4 sub-pixel kernels, 16 taps) -- anything beyond that is refused
    if (tsrc > 9 || ntaps > 16) return -1;
    if (fast_is_c) return ((Cp + 255) / 256) * Mp;
    return ((Mp + 7) / 8) * ((Cp + 63) / 64);
}

extern "C" int segnb_pack_weight_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_pack_weight_multi, jobs, njobs, total_blocks, stream);
    SEGNB_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "bad job table");
    int grid = total_blocks;
    const int cap = segnb_knob_pack_blocks();
    if (cap > 0 && grid > cap) grid = cap;
    hipLaunchKernelGGL(pack_tiled_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob*)jobs, njobs, total_blocks);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_unpack_wgrad_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_unpack_wgrad_multi, jobs, njobs, total_blocks, stream);
    SEGNB_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "bad job table");
    hipLaunchKernelGGL(pack_tiled_kernel<false>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob*)jobs, njobs, total_blocks);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_pack_pair_job_bytes(void) { return (int)sizeof(PackPairJob); }

// blocks of one pair job, or -1 when the shape is not served (the two single-form jobs of segnb_pack_weight_multi then)
extern "C" int segnb_pack_pair_job_blocks(int Co, int Ci, int Cop, int Cip) {
    if (Co <= 0 || Ci <= 0 || Cop < Co || Cip < Ci || Cop % 8 != 0 || Cip % 8 != 0) return -1;
    return ((Cop + PP_CO - 1) / PP_CO) * ((Cip + PP_CI - 1) / PP_CI);
}

extern "C" int segnb_pack_weight_pair_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_pack_weight_pair_multi, jobs, njobs, total_blocks, stream);
    SEGNB_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "bad job table");
    hipLaunchKernelGGL(pack_pair_kernel<false>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackPairJob*)jobs, njobs,
                       0ll, 0.f);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// torch.optim.SGD.step() (torch_train.py:71,190: plain SGD) on the parameters of the job table AND their weight pack in one pass:
// flat_p / flat_g are the flat parameter / gradient buffers every job's `w` points into (segnb.engine.FlatParams); the parameters the
// table does not cover take segnb_sgd_ranges
extern "C" int segnb_sgd_pack_pair_multi(const void* jobs, int njobs, int total_blocks, float* flat_p, const float* flat_g, float lr,
                                         segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_sgd_pack_pair_multi belongs to optimizer.step(), outside the recorded lists");
    SEGNB_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0 && flat_p && flat_g, "bad job table / buffers");
    SEGNB_CHECK_ARG((((uintptr_t)flat_p | (uintptr_t)flat_g) & 15) == 0, "buffers must be 16-byte aligned");
    hipLaunchKernelGGL(pack_pair_kernel<true>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackPairJob*)jobs, njobs,
                       (long long)(flat_g - flat_p), lr);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// p[i] -= lr * g[i] over nranges element ranges of two parallel fp32 buffers: ranges = device int64 [nranges][3] (start, length,
// first index of the range in the concatenation of all ranges), total = sum of the lengths.  What segnb_sgd_pack_pair_multi leaves
// of a model's flat parameter buffer: convolution biases, BatchNorm parameters, the classifier (a few thousand elements)
__global__ __launch_bounds__(256) void sgd_ranges_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         const long long* __restrict__ ranges, int nranges, long long total, float lr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = nranges - 1;                    // last range whose first index is <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ranges[3 * mid + 2] <= i) lo = mid; else hi = mid - 1;
    }
    const long long e = ranges[3 * lo] + (i - ranges[3 * lo + 2]);
    p[e] -= lr * g[e];
}

extern "C" int segnb_sgd_ranges(float* p, const float* g, const long long* ranges, int nranges, long long total, float lr,
                                segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_sgd_ranges belongs to optimizer.step(), outside the recorded lists");
    SEGNB_CHECK_ARG(p && g && ranges && nranges > 0 && total > 0, "bad arguments");
    hipLaunchKernelGGL(sgd_ranges_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, ranges,
                       nranges, total, lr);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_pack_elem_job_blocks(int Mp, int Cp, int ntaps) {
    if (Mp <= 0 || Cp <= 0 || ntaps < 1 || ntaps > SEGNB_MAX_TAPS) return -1;
    return (int)(((long long)Mp * ntaps * Cp + 256 * PACKW_ITEMS - 1) / (256 * PACKW_ITEMS));
}

extern "C" int segnb_pack_weight_elem_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_pack_weight_elem_multi, jobs, njobs, total_blocks, stream);
    SEGNB_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "bad job table");
    hipLaunchKernelGGL(pack_elem_multi_kernel<true>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob*)jobs, njobs);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_unpack_wgrad_elem_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_unpack_wgrad_elem_multi, jobs, njobs, total_blocks, stream);
    SEGNB_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "bad job table");
    hipLaunchKernelGGL(pack_elem_multi_kernel<false>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob*)jobs, njobs);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_pack_input_nchw(const float* x, int N, int C, int H, int W, void* out, int dtype, int Cp,
                                     int ld_out, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_pack_input_nchw, x, N, C, H, W, out, dtype, Cp, ld_out, stream);
    SEGNB_CHECK_ARG(x && out, "NULL tensor");
    SEGNB_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && Cp >= C && ld_out >= Cp && ld_out % 8 == 0,
                    "bad shape");
    const long long total = (long long)N * H * W * (Cp / 8);
    int grid = ceil_div(total, 256);
    if (grid > 8192) grid = 8192;
    if (dtype == SEGNB_BF16)
        hipLaunchKernelGGL(pack_input_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                           (bf16_t*)out, N, C, H, W, Cp, ld_out);
    else if (dtype == SEGNB_F32)
        hipLaunchKernelGGL(pack_input_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (float*)out,
                           N, C, H, W, Cp, ld_out);
    else {
        segnb_set_error("segnb_pack_input_nchw: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}
