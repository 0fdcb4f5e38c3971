// Stride-1 3x3 convolution forward / data-gradient (bf16), gfx950 -- the fast path behind segnb_conv_fprop.
//
// im2col-free implicit GEMM with the IMAGE TILE staged in LDS: for a block of R x WT output pixels the
// (R+2) x (WT+2) halo tile of the input is loaded ONCE per 32/64-channel chunk, pixel-major, and all nine
// taps take their A fragments from shifted pixel rows of that one tile (a tap (dh,dw) is a constant LDS
// offset).  The general gather kernel (conv_igemm.hip) fetches every tap's A tile from global memory again:
// 9x the L2 traffic and 9x the address arithmetic for the same FLOPs.
//
// GEMM view: M = pixels of the tile (BM = R*WT), N = output channels (BN), K = 9 taps x Ci.
//   step = (channel chunk c, tap t): B tile = W[co][t][c*BKC ..] (BN x BKC, double-buffered, register prefetch)
//   A fragment of MFMA row-tile i for tap t: LDS row (pixel(i, lane) shifted by the tap), 16 bytes per lane
// Rows are padded by 16 B (conflict-free ds_read_b128).  Epilogue as in conv_igemm.hip: +bias, round to bf16,
// per-channel sum / sum^2 of the stored values (BatchNorm statistics), LDS-staged 16-byte stores.
// Blocks are persistent over pixel tiles for a fixed channel tile so the statistics stay in registers.
#include "common.h"

namespace {

struct FpS1Args {
    const bf16_t* x;
    const bf16_t* w;
    const float* bias;
    int bias_n;
    bf16_t* out;
    double* stats;
    int N, H, W;          // output grid
    int Hi, Wi;           // input tensor
    int Ci, Co, ld_x, ld_out, Ktot;
    int dhmin, dwmin;
    int dh[9], dw[9];     // tap offsets minus (dhmin, dwmin): 0..2
    int HB, WB, IT, NTL, GM, NCH;
    // segnb_conv_fprop_drop: out = round(round(acc + bias) * drop[n][co]) (Dropout2d multipliers, [N][ld_drop] fp32; NULL: off) and
    // the statistics of THAT tensor go to a table whose rows are stats_ld doubles apart (a concat buffer's table: segnb_bn_stats_ld)
    const float* drop;
    int ld_drop, stats_ld;
};

__device__ __forceinline__ int xcd_remap_s1(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7, j = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

template <int BN, int R, int WT, int WAVES_M, int BKC, int TPS>
struct FpS1Cfg {
    static constexpr int BM = R * WT;
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int XR = R + 2, XC = WT + 2;
    static constexpr int SX = BKC * 2 + 16;          // x tile row (one pixel) in bytes
    static constexpr int SB = BKC * 2 + 16;          // weight tile row (one output channel)
    static constexpr int X_BYTES = XR * XC * SX;
    static constexpr int B_BYTES = TPS * BN * SB;      // weight tile of one step = TPS taps
    static constexpr int OUT_ROW = BN * 2 + 16;
    static constexpr int TILE_BYTES = X_BYTES + 2 * B_BYTES;
    static constexpr int STAGE_BYTES = BM * OUT_ROW;
    static constexpr int MAIN_BYTES = TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES;
    static constexpr int SMEM = MAIN_BYTES + BM * 4 + WAVES_M * 2 * BN * 4;   // + one stats slot row per wave row
    static_assert(WM % 32 == 0 && WN % 32 == 0 && WAVES_M * WAVES_N == 4, "wave tiling");
};

// TPS = taps per barrier step (1 or 3): the 32-channel tiles (4 MFMAs per wave and tap) take a whole kernel row of
// taps per step, three barriers per channel chunk instead of nine (-10 % on the 224x224 layers; with 64-channel
// tiles the tripled weight buffers cost the second resident block and lose 40 %)
template <int BN, int R, int WT, int WAVES_M, int BKC, int TPS>
__global__ __launch_bounds__(256) void conv_fprop_s1x9_kernel(const FpS1Args a) {
    using C = FpS1Cfg<BN, R, WT, WAVES_M, BKC, TPS>;
    constexpr int SPC = 9 / TPS;                        // steps per channel chunk
    static_assert(TPS == 1 || TPS == 3, "taps per step");
    constexpr int BM = C::BM, TM = C::TM, TN = C::TN, XC = C::XC, SX = C::SX, SB = C::SB;
    constexpr int XCH = C::XR * XC * (BKC / 8);         // 16-byte chunks of the x tile
    constexpr int XPT = (XCH + 255) / 256;
    constexpr int BCH = TPS * BN * (BKC / 8);
    constexpr int BPT = (BCH + 255) / 256;
    constexpr int KK = BKC / 16;
    constexpr int OUT_ROW = C::OUT_ROW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sB = smem + C::X_BYTES;
    unsigned char* sOut = smem;
    int* sPix = reinterpret_cast<int*>(smem + C::MAIN_BYTES);        // output pixel index per tile row, -1 = outside
    float* sStat = reinterpret_cast<float*>(sPix + BM);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

    const int L = xcd_remap_s1(blockIdx.x, gridDim.x);
    const int nt = L % a.NTL, gq = L / a.NTL;
    const int n_base = nt * BN;

    // lane -> LDS byte offset of its pixel row inside the x tile, per MFMA row-tile
    int a_base[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + r;
        a_base[i] = ((m / WT) * XC + (m % WT)) * SX + h * 16;
    }
    const int b_base = (wn * C::WN + r) * SB + h * 16;

    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = (a.dh[t] * XC + a.dw[t]) * SX;

    double st = 0.0;
    const int nsteps = a.NCH * SPC;
    float bias_r[TN];                   // this lane's output channels are fixed for the whole block
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n_base + wn * C::WN + 32 * j + r;
        bias_r[j] = (a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;
    }

    // coordinates of the tile whose operands are being LOADED (one tile ahead of the one being computed at the end
    // of an iteration: the first x chunk and weight step of the next tile are requested before the epilogue, so
    // their latency -- an HBM round trip per tile, 12-25 tiles per block on the 224x224 layers -- hides behind it)
    int n = 0, h0 = 0, w0 = 0;
    auto set_tile = [&](int it) {
        n = it / (a.HB * a.WB);
        const int rem = it - n * (a.HB * a.WB);
        const int hb = rem / a.WB, wb = rem - hb * a.WB;
        h0 = hb * R;
        w0 = wb * WT;
    };
    uint4 rx[XPT], rb[BPT];
    if (gq < a.IT) set_tile(gq);
        auto gload_x = [&](int c) {
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                const int q = tid + u * 256;
                const int pix = q / (BKC / 8), cc = q - pix * (BKC / 8);
                const int xr = pix / XC, xc = pix - xr * XC;
                const int hi = h0 + a.dhmin + xr, wi = w0 + a.dwmin + xc;
                const int ch = c * BKC + cc * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (q < XCH && ch < a.Ci && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
                    v = *reinterpret_cast<const uint4*>(a.x + ((long long)(n * a.Hi + hi) * a.Wi + wi) * a.ld_x + ch);
                rx[u] = v;
            }
        };
        auto lstore_x = [&]() {
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                const int q = tid + u * 256;
                if (q < XCH) {
                    const int pix = q / (BKC / 8), cc = q - pix * (BKC / 8);
                    *reinterpret_cast<uint4*>(sX + pix * SX + cc * 16) = rx[u];
                }
            }
        };
        auto gload_b = [&](int step) {
            const int c = step / SPC, t0 = (step - c * SPC) * TPS;
#pragma unroll
            for (int u = 0; u < BPT; ++u) {
                const int q = tid + u * 256;
                const int krow = q / (BKC / 8), cc = q - krow * (BKC / 8);      // krow = tap-in-step * BN + channel row
                const int k = krow / BN, row = krow - k * BN;
                const int co = n_base + row, ch = c * BKC + cc * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (q < BCH && co < a.Co && ch < a.Ci)
                    v = *reinterpret_cast<const uint4*>(a.w + (long long)co * a.Ktot + (t0 + k) * a.Ci + ch);
                rb[u] = v;
            }
        };
        auto lstore_b = [&](int buf) {
#pragma unroll
            for (int u = 0; u < BPT; ++u) {
                const int q = tid + u * 256;
                if (q < BCH) {
                    const int krow = q / (BKC / 8), cc = q - krow * (BKC / 8);
                    *reinterpret_cast<uint4*>(sB + buf * C::B_BYTES + krow * SB + cc * 16) = rb[u];
                }
            }
        };

    if (gq < a.IT) {
        gload_x(0);
        gload_b(0);
    }
    for (int it = gq; it < a.IT; it += a.GM) {
        __syncthreads();        // previous tile's staging area / pixel table consumed
        for (int rr = tid; rr < BM; rr += 256) {
            const int ho = h0 + rr / WT, wo = w0 + rr % WT;
            sPix[rr] = (ho < a.H && wo < a.W) ? (n * a.H + ho) * a.W + wo : -1;
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        lstore_x();             // requested before the previous tile's epilogue (or above, for the first tile)
        lstore_b(0);
        __syncthreads();
        for (int c = 0; c < a.NCH; ++c) {
#pragma unroll
            for (int ts = 0; ts < SPC; ++ts) {     // unrolled: tap offsets are registers, not per-step kernarg loads
                const int step = c * SPC + ts;
                const int buf = (c + ts) & 1;      // == step & 1 (SPC is odd)
                const bool more = step + 1 < nsteps;
                const bool new_chunk = (ts == SPC - 1) && more;
                if (more) gload_b(step + 1);
                if (new_chunk) gload_x(c + 1);
#pragma unroll
                for (int k = 0; k < TPS; ++k) {
                    const unsigned char* pa = sX + toff[ts * TPS + k];
                    const unsigned char* pb = sB + buf * C::B_BYTES + k * BN * SB + b_base;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        bf16x8_t af[TM], bfr[TN];
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            af[i] = *reinterpret_cast<const bf16x8_t*>(pa + a_base[i] + kk * 32);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bfr[j] = *reinterpret_cast<const bf16x8_t*>(pb + j * 32 * SB + kk * 32);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    }
                }
                if (more) lstore_b(buf ^ 1);
                if (new_chunk) {
                    __syncthreads();            // every wave is done with the current x tile
                    lstore_x();
                }
                __syncthreads();
            }
        }

        if (it + a.GM < a.IT) {          // next tile's first operands: in flight during the epilogue
            set_tile(it + a.GM);
            gload_x(0);
            gload_b(0);
        }

        // ---- epilogue ---------------------------------------------------------------------------------
        float cs1[TN], cs2[TN];
        const int n_cur = it / (a.HB * a.WB);          // (n already belongs to the tile being loaded)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * C::WN + 32 * j + r;
            const int co = n_base + col;
            const float bv = bias_r[j];
            // Dropout2d multiplier of (image, channel); 1 without: round(x * 1) == x, the plain launches store the same bits
            const float mv = (a.drop != nullptr && co < a.Co) ? a.drop[(long long)n_cur * a.ld_drop + co] : 1.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * C::WM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bf16_t tv = Elem<bf16_t>::from_f32(Elem<bf16_t>::to_f32(Elem<bf16_t>::from_f32(acc[i][j][e] + bv)) * mv);
                    *reinterpret_cast<bf16_t*>(sOut + row * OUT_ROW + col * 2) = tv;
                    if (sPix[row] >= 0) {
                        const float vr = Elem<bf16_t>::to_f32(tv);
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
                    // one slot per (wave row, column): plain stores, summed below in a fixed order, so the
                    // statistics (and everything after them) are reproducible run to run
                    const int col = wn * C::WN + 32 * j + r;
                    sStat[wm * 2 * BN + col] = t1;
                    sStat[wm * 2 * BN + BN + col] = t2;
                }
            }
        }
        __syncthreads();
        if (a.stats != nullptr && tid < 2 * BN) {
#pragma unroll
            for (int q = 0; q < WAVES_M; ++q) st += (double)sStat[q * 2 * BN + tid];
        }
        constexpr int OC = BN / 8;
        for (int q = tid; q < BM * OC; q += 256) {
            const int row = q / OC, cc = q - row * OC;
            const int opix = sPix[row];
            const int co = n_base + cc * 8;
            if (opix >= 0 && co < a.Co)
                *reinterpret_cast<uint4*>(a.out + (long long)opix * a.ld_out + co) =
                    *reinterpret_cast<const uint4*>(sOut + row * OUT_ROW + cc * 16);
        }
    }
    if (a.stats != nullptr && tid < 2 * BN) {
        const int which = tid / BN, col = tid - which * BN;
        const int co = n_base + col;
        if (co < a.Co) atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.stats_ld + co], st);
    }
}

template <int BN, int R, int WT, int WAVES_M, int BKC, int TPS = 1>
int launch_fs1(FpS1Args& a, hipStream_t stream) {
    using C = FpS1Cfg<BN, R, WT, WAVES_M, BKC, TPS>;
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_s1x9_kernel<BN, R, WT, WAVES_M, BKC, TPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) segnb_set_error("fprop_s1 hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + R - 1) / R;
    a.WB = (a.W + WT - 1) / WT;
    a.IT = a.N * a.HB * a.WB;
    a.NTL = (a.Co + BN - 1) / BN;
    a.NCH = (a.Ci + BKC - 1) / BKC;
    int per_cu = (160 * 1024) / C::SMEM;
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    int gm = (segnb_num_cus() * per_cu) / a.NTL;
    if (gm < 1) gm = 1;
    if (gm > a.IT) gm = a.IT;
    a.GM = gm;
    hipLaunchKernelGGL((conv_fprop_s1x9_kernel<BN, R, WT, WAVES_M, BKC, TPS>), dim3(a.GM * a.NTL), dim3(256), C::SMEM,
                       stream, a);
    return 0;
}

constexpr int NOT_HANDLED = -12345;

template <int BKC>
int dispatch_fs1(FpS1Args& a, hipStream_t stream) {
    const int cus = segnb_num_cus();
    if (a.W > 16) {
        // thin inputs (the 224x224 / 112x112 levels and their concat data gradients): measured per layer, the image
        // tile wins whatever the output width -- 32-channel tiles for Ci <= 32 (32 -> 96: 224 -> 202 us), 64-channel
        // tiles for Ci <= 64 on >= 112-pixel rows (64 -> 192: 176 -> 150 us); 96-channel tiles were slower than both
        if (a.Co <= 32 || a.Ci <= 32) return launch_fs1<32, 8, 32, 4, BKC, 3>(a, stream);
        const long long its = (long long)a.N * ((a.H + 3) / 4) * ((a.W + 31) / 32);
        if (a.Co <= 64 || its * ((a.Co + 127) / 128) < cus || (a.Ci <= 64 && a.W >= 112))
            return launch_fs1<64, 4, 32, 2, BKC>(a, stream);
        // wide layers on >= 28-pixel rows: the general gather kernel's flattened-pixel 128x128 tiles (no partial
        // row segments, one barrier per step) measure 20-25 % faster than the image-tile form here
        return NOT_HANDLED;
    }
    // measured per layer on MI355X (tools/layer_bench.py, after the general kernel got buffer-descriptor gathers):
    //   8 < W <= 16 : 64-channel tiles here (2 blocks / CU) beat both the 128-channel tiles (-20 %) and the general
    //                 kernel, except for very wide outputs (Co > 1024: the data gradient of a concat layer)
    //   W <= 8      : the general kernel's flattened 128x128 tiles are 25 % faster than 8x8 image tiles
    if (a.W > 8 && a.Co <= 1024) return launch_fs1<64, 8, 16, 2, BKC>(a, stream);
    return NOT_HANDLED;
}

}  // namespace

// 1 = handled, 0 = not a stride-1 3x3 case (caller uses the general gather kernel), else error
int segnb_fprop_s1_try(const segnb_conv_geom* g, const void* in, const void* wpacked, const float* bias, int bias_n,
                       void* out, double* stats, hipStream_t stream, const float* drop, int ld_drop, int stats_ld) {
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci < 32) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return 0;
    FpS1Args a;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.bias = bias;
    a.bias_n = bias_n;
    a.out = (bf16_t*)out;
    a.stats = stats;
    a.drop = drop;
    a.ld_drop = ld_drop;
    a.stats_ld = stats_ld > 0 ? stats_ld : g->Co;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Hi = g->Hi; a.Wi = g->Wi;
    a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
    a.Ktot = 9 * g->Ci;
    a.dhmin = dhmin; a.dwmin = dwmin;
    for (int t = 0; t < 9; ++t) {
        a.dh[t] = g->dh[t] - dhmin;
        a.dw[t] = g->dw[t] - dwmin;
    }
    const int rc = (g->Ci % 64 == 0) ? dispatch_fs1<64>(a, stream) : dispatch_fs1<32>(a, stream);
    if (rc == NOT_HANDLED) return 0;
    return rc ? rc : 1;
}
