// Stride-1 3x3 convolution forward / data-gradient (bf16) for the THIN layers -- Ci in {32, 64, 96}, Co <= 96: the
// 224x224 / 112x112 levels of ZF_UNET (lib/models/zf_unet.py:37-38,56-57) and their data gradients.  These layers
// are HBM-bound (32 -> 32 at 224x224, bs=32: 206 MB of activations against 30 GFLOP), so the design goal is bytes
// in flight, not MFMA issue:
//   * the WHOLE weight matrix of the block's output channels (<= 72 KiB: 9 taps x Ci x BN) is fetched once into LDS
//     and stays there -- no weight ring, no per-tap barrier;
//   * one barrier step = one (pixel tile, 32-channel chunk): all nine taps, 18 K slices, 36-108 MFMAs per matrix wave;
//   * the 18 x 18 pixel x 32 channel halo tile (21 KiB) runs through a 4-stage ring fetched THREE steps ahead by the
//     fetch waves (LDS-DMA, 16 rows x 64 B per wave-instruction): ~63 KiB per CU in flight;
//   * rows are 64 B: slot q of row p holds channel chunk q ^ ((p >> 2) & 3) (source-side swizzle, conflict-free
//     ds_read_b128 for rows distinct mod 16);
//   * same wave specialisation, transposed accumulators, LDS-staged 16-byte stores and store-pass statistics as
//     fprop_dma.hip.
#include "fprop_dma.h"

namespace {

// NSW_ store waves: the store pass (table + staging reads, 16-byte stores, BatchNorm statistics) of a tile must finish
// within the next tile's matrix time -- 0.5 us of MFMAs on the 32-channel layers -- or it paces the block: with two store
// waves the forward launch (statistics in the pass) took 62 us against 52 us for the data gradient (none).
template <int BN_, int NS_, int NSW_ = 2, int R_ = 16, int WT_ = 16>
struct RwCfg {
    static constexpr int BN = BN_, NS = NS_;
    static constexpr int R = R_, WT = WT_, BM = 256;
    static_assert(R_ * WT_ == 256, "tile pixels");
    static constexpr int NCW = 4, NLW = 2, NSW = NSW_, LT = NSW * 64;     // matrix / fetch / store waves
    static constexpr int NT = (NCW + NLW + NSW) * 64;
    static constexpr int TM = 2, TN = BN / 32, NF = TM + TN;
    static constexpr int XR = R + 2, XC = WT + 2, NPIX = XR * XC;
    static constexpr int APIECES = (NPIX + 15) / 16;      // 1-KiB pieces of 16 halo pixels x 64 B
    static constexpr int A_STAGE = APIECES * 1024;
    static constexpr int APW = (APIECES + NLW - 1) / NLW;
    static constexpr int OUT_ROW = BN * 2 + 16;
    static constexpr int OC = BN / 8;
    static constexpr int RG = LT / OC;                   // store threads: OC chunks x RG row groups (RG * OC <= LT active)
    static constexpr int RPT = (BM + RG - 1) / RG;       // staged rows per store thread
    static constexpr bool STATS_OK = LT >= 2 * BN;       // the final statistics reduction needs 2 * BN threads
    // LDS: [halo ring NS stages][output staging: whole tile][pixel tables x2][bias][dummy piece][weights ...]
    static constexpr int OFF_STG = NS * A_STAGE;
    static constexpr int STG_BYTES = BM * OUT_ROW;
    static constexpr int OFF_PIX = OFF_STG + STG_BYTES;
    static constexpr int OFF_BIAS = OFF_PIX + 2 * BM * 4;
    static constexpr int OFF_SCALE = OFF_BIAS + BN * 4;      // per-channel factor of the affine epilogue (segnb_conv_fprop_act)
    static constexpr int OFF_DUMMY = ((OFF_SCALE + BN * 4 + 1023) / 1024) * 1024;
    static constexpr int OFF_W = OFF_DUMMY + 1024;
    static constexpr int W_MAX = 160 * 1024 - OFF_W;      // bytes left for the resident weights
    static constexpr int SMEM = 160 * 1024;
    static_assert(BN % 32 == 0 && NF <= 6, "tile");
    static_assert(LT * 16 * 8 <= OFF_STG, "statistics reduction scratch");
    static_assert(W_MAX >= 9 * BN * 64, "at least one 32-channel chunk of weights");
};

template <class C, bool DBG, bool EP = false>      // EP: see conv_fprop_ws_kernel
__global__ __launch_bounds__(C::NT) void conv_fprop_rw_kernel(const FdArgs a) {
    constexpr int BN = C::BN, NS = C::NS, BM = C::BM, TM = C::TM, TN = C::TN, NF = C::NF, XC = C::XC, R = C::R, WT = C::WT;
    constexpr int APW = C::APW, OC = C::OC, NLW = C::NLW, LT = C::LT, OUT_ROW = C::OUT_ROW;
    constexpr int LA = NS - 1;                      // fetch look-ahead in steps

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    int* sPix = reinterpret_cast<int*>(smem + C::OFF_PIX);
    float* sBias = reinterpret_cast<float*>(smem + C::OFF_BIAS);
    const unsigned lds0 = (unsigned)(size_t)smem;
    unsigned char* const sOut = smem + C::OFF_STG;      // output staging of one whole tile

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool matrix = wave < C::NCW;
    const bool fetcher = !matrix && wave < C::NCW + NLW;
    const int lw = wave - C::NCW;                  // fetch wave index
    const int gq = xcd_remap_fd(blockIdx.x, gridDim.x);      // one channel tile: block = pixel-tile queue

    const i32x4_t rs_x = make_rsrc4(a.x, a.x_bytes);
    const i32x4_t rs_w = make_rsrc4(a.w, a.w_bytes);
    const i32x4_t rs_u = make_rsrc4(a.u != nullptr ? a.u : a.x, a.u != nullptr ? a.u_bytes : a.x_bytes);

    float* sScale = reinterpret_cast<float*>(smem + C::OFF_SCALE);
    for (int c = tid; c < BN; c += C::NT) {
        const float bv = (a.bias != nullptr && c < a.bias_n) ? a.bias[c] : 0.f;
        float sc = 1.f, sh = bv;
        if constexpr (EP) {
            if (a.ep_coef != nullptr && c < a.Co) {      // (acc + bias - mean) * scale + shift
                sc = a.ep_coef[c];
                sh = (bv - a.ep_coef[2 * a.Co + c]) * sc + a.ep_coef[a.Co + c];
            }
            sScale[c] = sc;
        }
        sBias[c] = sh;
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

    if (fetcher) {
        // ================= fetch stream =================
        // resident weights: LDS row (c*9 + t)*BN + co, 64 B each
        {
            const int wrows = a.NCH * 9 * BN;
            for (int piece = lw; piece * 16 < wrows; piece += NLW) {
                const int wrow = piece * 16 + (lane >> 2);
                const int q = lane & 3;
                const int ct = wrow / BN, co = wrow - ct * BN;
                const int c = ct / 9, t = ct - c * 9;
                const unsigned voff = (wrow < wrows && co < a.Co)
                                          ? (unsigned)(co * a.Ktot + t * a.Ci + c * 32) * 2u + (unsigned)((q ^ ((wrow >> 2) & 3)) * 16)
                                          : OOB;
                dma16(lds0 + C::OFF_W + piece * 1024, voff, rs_w, 0);
            }
        }
        // halo pieces of this wave: piece = lw + NLW * pa.  Per lane the pixel of a piece is fixed, so its offset
        // relative to the tile origin and its halo coordinates are computed ONCE; per tile only the origin (scalar) and
        // the four border compares remain (recomputing pix / XC etc. per tile cost the fetch waves ~2000 cycles per
        // tile: more than the tile's MFMAs on the 32-channel layers).
        unsigned a_rel[APW], a_xy[APW], a_voff[APW];
        unsigned a_uoff[APW];             // virtual concat (FdArgs::u): the same halo pixels in the low-resolution tensor
        const bool vcat = a.u != nullptr;
#pragma unroll
        for (int pa = 0; pa < APW; ++pa) {
            const int pix = (lw + NLW * pa) * 16 + (lane >> 2);
            const int q = lane & 3;
            const int xr = pix / XC, xc = pix - xr * XC;
            a_rel[pa] = (unsigned)(xr * a.Wi + xc) * (unsigned)a.ld_x * 2u + (unsigned)((q ^ ((pix >> 2) & 3)) * 16);
            a_xy[pa] = pix < C::NPIX ? (unsigned)xr | ((unsigned)xc << 16) : 0x7fff7fffu;      // never inside the image
        }
        auto set_fetch_tile = [&](int it) {
            const bool live = it < a.IT;
            const int n = it / (a.HB * a.WB);
            const int rem = it - n * (a.HB * a.WB);
            const int hb = rem / a.WB, wb = rem - hb * a.WB;
            const int h0 = hb * R + a.dhmin, w0 = wb * WT + a.dwmin;
            const unsigned base = (unsigned)(((n * a.Hi + h0) * a.Wi + w0) * a.ld_x * 2);
            const unsigned hlim = live ? (unsigned)a.Hi : 0u;
#pragma unroll
            for (int pa = 0; pa < APW; ++pa) {
                const int hi = h0 + (int)(a_xy[pa] & 0xffffu), wi = w0 + (int)(a_xy[pa] >> 16);
                const bool ok = (unsigned)hi < hlim && (unsigned)wi < (unsigned)a.Wi;
                a_voff[pa] = ok ? base + a_rel[pa] : OOB;
                if (vcat) {          // nearest-x2 upsample = pixel (hi >> 1, wi >> 1) of u, same channel slot
                    const int pixl = (lw + NLW * pa) * 16 + (lane >> 2);
                    a_uoff[pa] = ok ? (unsigned)((n * a.Hu + (hi >> 1)) * a.Wu + (wi >> 1)) * (unsigned)a.ld_u * 2u +
                                          (unsigned)(((lane & 3) ^ ((pixl >> 2) & 3)) * 16)
                                    : OOB;
                }
            }
        };
        auto fetch_a = [&](int c, int stage) {
#pragma unroll
            for (int pa = 0; pa < APW; ++pa) {
                const int piece = lw + NLW * pa;
                const unsigned dst = piece < C::APIECES ? lds0 + stage * C::A_STAGE + piece * 1024 : lds0 + C::OFF_DUMMY;
                if (vcat) {          // chunks 0 .. NCHU-1 from u (upsampled on the way), the others from x = the skip tensor
                    if (c < a.NCHU) dma16(dst, a_uoff[pa], rs_u, (unsigned)c * 64u);
                    else dma16(dst, a_voff[pa], rs_x, (unsigned)(c - a.NCHU) * 64u);
                    continue;
                }
                if (!(DBG && (a.dbg & 2))) dma16(dst, a_voff[pa], rs_x, (unsigned)c * 64u);
            }
        };
        // fetch cursor: (tile, chunk) of the step whose halo tile is requested next
        int f_it = gq, f_c = 0, f_g = 0;
        set_fetch_tile(f_it);
        auto fetch_next = [&]() {
            fetch_a(f_c, f_g % NS);
            ++f_g;
            if (++f_c == a.NCH) {
                f_c = 0;
                f_it += a.GM;
                set_fetch_tile(f_it);
            }
        };
#pragma unroll
        for (int k = 0; k < LA; ++k) fetch_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();

        int g = 0;
        for (int it = gq; it < a.IT; it += a.GM) {
            for (int c = 0; c < a.NCH; ++c, ++g) {
                fetch_next();                              // halo tile of step g + LA into the stage read in step g - 1
                if (c + 1 == a.NCH) raw_barrier();         // M (staging hand-over of the other waves)
                step_sync<(LA - 1) * APW>();               // B: step g + 1 (fetched in step g + 1 - LA) has landed; the
                                                           // fetches of the LA - 1 steps after it stay in flight
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // look-ahead fetches past the last tile
        raw_barrier();
        raw_barrier();
    } else if (!matrix) {
        // ================= store pass =================
        // On waves of its own: global stores stay counted in vmcnt until they are acknowledged (microseconds under
        // load), and a fetch wave's counted wait would sit behind them -- measured as 25 us of a 72 us layer.
        const int ltid = tid - (C::NCW + NLW) * 64;
        lds_barrier();                                     // (the prologue barrier)
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        // coalesced store pass of one staged tile + statistics of the stored values (pixel table `tab`).  Branch-free:
        // all table and staging reads of a thread go out first, pixels outside the image are dropped by the output
        // descriptor's range check (a per-row `if` serialised two LDS round trips per row: ~2100 cycles per tile).
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
        // fused BatchNorm-backward reduction (a.bn_y != NULL): per-thread constants of its channel chunk
        const bool bnred = C::STATS_OK && a.bn_y != nullptr;
        const __amdgpu_buffer_rsrc_t rs_y =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bnred ? a.bn_y : a.out), 0, bnred ? (int)a.bn_y_bytes : 0, 0x00020000);
        float bsc[8], bsh[8], bmu[8];
        float bneg = 0.f;
        if (bnred) {
            const int c0 = (ltid % OC) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = c0 + e < a.Co;
                bsc[e] = in ? a.bn_coef[c0 + e] : 0.f;
                bsh[e] = in ? a.bn_coef[a.Co + c0 + e] : 0.f;
                bmu[e] = in ? a.bn_coef[2 * a.Co + c0 + e] : 0.f;
            }
            bneg = a.bn_act == SEGNB_ACT_RELU ? 0.f : (a.bn_act == SEGNB_ACT_LEAKY ? a.bn_slope : 1.f);
        }
        auto store_pass = [&](const int* tab) {
            if (DBG && (a.dbg & 8)) return;
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            const int cc = ltid % OC, row0 = ltid / OC;
            const int ocu = a.up_out != nullptr ? a.up_C / 8 : 0;      // leading chunks that leave through the 2 x 2 sums below
            const bool cok = cc * 8 < a.Co && row0 < C::RG && cc >= ocu;      // (12 chunks per row: 120 of the 128 threads)
            if (a.up_out != nullptr) {
                // fused Upsample(x2) backward: one (2 x 2 window, 8-channel chunk) per thread and trip.  Tile origins and the
                // image size are even (checked on the host): a window lies inside the image or outside it as a whole.
                const int items = (BM / 4) * ocu;
                for (int i = ltid; i < items; i += LT) {
                    const int cu = i % ocu, q = i / ocu;
                    const int m00 = (2 * (q / (WT / 2))) * WT + 2 * (q % (WT / 2));
                    const int opix = tab[m00];
                    if (opix < 0) continue;
                    float acc8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc8[e] = 0.f;
#pragma unroll
                    for (int wq = 0; wq < 4; ++wq) {
                        const int row = m00 + (wq >> 1) * WT + (wq & 1);
                        const u32x4_t v4 = *reinterpret_cast<const u32x4_t*>(sOut + row * OUT_ROW + cu * 16);
                        acc8[0] += __uint_as_float(v4.x << 16); acc8[1] += __uint_as_float(v4.x & 0xffff0000u);
                        acc8[2] += __uint_as_float(v4.y << 16); acc8[3] += __uint_as_float(v4.y & 0xffff0000u);
                        acc8[4] += __uint_as_float(v4.z << 16); acc8[5] += __uint_as_float(v4.z & 0xffff0000u);
                        acc8[6] += __uint_as_float(v4.w << 16); acc8[7] += __uint_as_float(v4.w & 0xffff0000u);
                    }
                    const int hw = a.H * a.W;
                    const int n = opix / hw, rem = opix - n * hw;
                    const int ho = rem / a.W, wo = rem - ho * a.W;
                    const long long lp = ((long long)n * (a.H >> 1) + (ho >> 1)) * (a.W >> 1) + (wo >> 1);
                    store8(a.up_out + lp * a.up_ld + cu * 8, acc8);
                }
            }
#pragma unroll
            for (int base = 0; base < C::RPT; base += 8) {
                int opix[8];
                u32x4_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = row0 + (base + k) * C::RG;
                    opix[k] = (base + k < C::RPT && row < BM) ? tab[row] : -1;
                }
                u32x4_t yv[8];
                if (bnred) {
                    // the y rows of these pixels: requested before anything else of the batch (HBM latency)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool ok = cok && opix[k] >= 0;
                        const unsigned yoff = ok ? (unsigned)opix[k] * (unsigned)a.bn_ld * 2u + (unsigned)cc * 16u : OOB;
                        yv[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)yoff, 0, 0);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = row0 + (base + k) * C::RG;
                    v[k] = *reinterpret_cast<const u32x4_t*>(sOut + (row < BM ? row : 0) * OUT_ROW + cc * 16);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool ok = cok && opix[k] >= 0;
                    const unsigned voff = ok ? (unsigned)opix[k] * (unsigned)a.ld_out * 2u + (unsigned)cc * 16u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v[k], rs_out, (int)voff, 0, 0);
                    if (C::STATS_OK && a.stats != nullptr) {
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
                    } else if (bnred) {
                        // dz = round(g * act'(z)), z = (y - mean) * scale + shift: the arithmetic of bn_act_bwd_reduce_kernel
                        const float m = ok ? 1.f : 0.f;
                        float gq[8], yq[8];
                        gq[0] = __uint_as_float(v[k].x << 16); gq[1] = __uint_as_float(v[k].x & 0xffff0000u);
                        gq[2] = __uint_as_float(v[k].y << 16); gq[3] = __uint_as_float(v[k].y & 0xffff0000u);
                        gq[4] = __uint_as_float(v[k].z << 16); gq[5] = __uint_as_float(v[k].z & 0xffff0000u);
                        gq[6] = __uint_as_float(v[k].w << 16); gq[7] = __uint_as_float(v[k].w & 0xffff0000u);
                        yq[0] = __uint_as_float(yv[k].x << 16); yq[1] = __uint_as_float(yv[k].x & 0xffff0000u);
                        yq[2] = __uint_as_float(yv[k].y << 16); yq[3] = __uint_as_float(yv[k].y & 0xffff0000u);
                        yq[4] = __uint_as_float(yv[k].z << 16); yq[5] = __uint_as_float(yv[k].z & 0xffff0000u);
                        yq[6] = __uint_as_float(yv[k].w << 16); yq[7] = __uint_as_float(yv[k].w & 0xffff0000u);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float yc = yq[e] - bmu[e];
                            const float z = yc * bsc[e] + bsh[e];
                            const float dv = bf16_bits_to_f32(f32_to_bf16_bits(gq[e] * (z > 0.f ? 1.f : bneg))) * m;
                            s1[e] += dv;
                            s2[e] += dv * yc;
                        }
                    }
                }
            }
        };
        // The store pass of tile i runs during the first step of tile i+1, under the matrix waves' MFMAs: the staging
        // buffer is handed over by two barriers at the end of every tile (M: staging free / B: staging written).
        int par = 0;
        bool pending = false;
        for (int it = gq; it < a.IT; it += a.GM, par ^= 1) {
            {
                const int n = it / (a.HB * a.WB);
                const int rem = it - n * (a.HB * a.WB);
                const int hb = rem / a.WB, wb = rem - hb * a.WB;
                for (int rr = ltid; rr < BM; rr += LT) {
                    const int ho = hb * R + rr / WT, wo = wb * WT + rr % WT;
                    sPix[par * BM + rr] = (ho < a.H && wo < a.W) ? (n * a.H + ho) * a.W + wo : -1;
                }
            }
            for (int c = 0; c < a.NCH; ++c) {
                if (c == 0 && pending) store_pass(sPix + (par ^ 1) * BM);      // previous tile
                if (c + 1 == a.NCH) lds_barrier();         // M: the staging buffer is free (and the pixel table written)
                raw_barrier();                             // B
            }
            pending = true;
        }
        if (pending) store_pass(sPix + (par ^ 1) * BM);
        lds_barrier();
        if (bnred) {
            const int c0 = (ltid % OC) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) s2[e] *= c0 + e < a.Co ? a.bn_coef[3 * a.Co + c0 + e] : 0.f;    // * invstd: sum dz * yhat
        }
        double* const acc_out = bnred ? a.bn_sums : a.stats;
        if (C::STATS_OK && acc_out != nullptr) {
            double* red = reinterpret_cast<double*>(smem);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[ltid * 16 + e] = (double)s1[e];
                red[ltid * 16 + 8 + e] = (double)s2[e];
            }
        }
        lds_barrier();
        {
            if (C::STATS_OK && acc_out != nullptr && ltid < 2 * BN) {
                const double* red = reinterpret_cast<const double*>(smem);
                const int which = ltid / BN, col = ltid - which * BN;
                const int cc = col >> 3, e = col & 7;
                double s = 0.0;
                for (int k = 0; k < C::RG; ++k) s += red[(k * OC + cc) * 16 + which * 8 + e];
                if (col < a.Co)
                    atomicAdd(&acc_out[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + col], s);
            }
        }
    } else {
        // ================= matrix stream =================
        const int wm = wave;                               // 4 x 1 waves: rows wm*64 .. +63, all BN channels
        int b_rd[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = 32 * j + r;
            b_rd[j] = C::OFF_W + row * 64 + (((h ^ (row >> 2)) & 1) << 4) + (((row >> 3) & 1) << 5);
        }
        int a_rd[9][TM];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = wm * 64 + 32 * i + r;
                const int p = (m / WT) * XC + (m % WT) + a.dh[t] * XC + a.dw[t];
                int v = (p << 6) + (((h ^ (p >> 2)) & 1) << 4) + (((p >> 3) & 1) << 5);
                asm volatile("" : "+v"(v));
                a_rd[t][i] = v;
            }
        lds_barrier();                                     // weights and the first LA halo tiles have landed

        int g = 0;
        for (int it = gq; it < a.IT; it += a.GM) {
            f32x16_t acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int c = 0; c < a.NCH; ++c, ++g) {
                const int sbase = (g % NS) * C::A_STAGE;
                const int wbase = c * (9 * BN * 64);
                if (!(DBG && (a.dbg & 4))) {
                    // Fragment reads run AHEAD slices in front of the MFMAs through AHEAD + 1 register sets: one slice carries
                    // only TM * TN MFMAs (64 cycles on the 32-channel layers) -- less than an LDS read takes under load, so with
                    // one slice of look-ahead every slice stalled on its fragments (~1 us of the 2.1 us a tile took).
                    constexpr int AHEAD = NF <= 3 ? 4 : 2, SETS = AHEAD + 1;      // (64-channel tiles: 168 registers at 10 waves)
                    static_assert(AHEAD * NF <= 15, "lgkmcnt range");
                    bf16x8_t fr[SETS][NF];
                    int bk[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bk[j] = b_rd[j] + wbase;
                    // slice sl = 2*t + kk; fragment set sl % SETS
                    auto read_slice = [&](auto sl_c) {
                        constexpr int sl = decltype(sl_c)::value;
                        constexpr int t = sl >> 1, kk = sl & 1;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int ad = bk[j] ^ (kk << 5);
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[sl % SETS][j]) : "v"(ad), "n"(t * BN * 64));
                        }
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int ad = (a_rd[t][i] + sbase) ^ (kk << 5);
                            FD_READ(fr[sl % SETS][TN + i], ad);
                        }
                    };
                    static_for<AHEAD>([&](auto k_c) { read_slice(k_c); });
                    __builtin_amdgcn_s_setprio(1);
                    static_for<18>([&](auto sl_c) {
                        constexpr int sl = decltype(sl_c)::value;
                        auto& accr = acc;               // (asm operands alone do not make a generic lambda capture it)
                        if constexpr (sl + AHEAD < 18) read_slice(std::integral_constant<int, sl + AHEAD>{});
                        constexpr int younger = (sl + AHEAD < 18 ? AHEAD : 17 - sl) * NF;      // reads issued after this slice's
                        ws_wait<younger>(fr[sl % SETS]);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) FD_MFMA(accr[i][j], fr[sl % SETS][j], fr[sl % SETS][TN + i]);
                    });
                    __builtin_amdgcn_s_setprio(0);
                }
                if (c + 1 < a.NCH) raw_barrier();
            }
            // (the chunk loop above leaves out the barrier of the tile's last step: it is B below)
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::);      // last MFMA results land before they are read
            raw_barrier();                                 // M: the fetch waves are done with the previous tile's staging
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * 64 + 32 * i + r;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = 32 * j + 8 * q + 4 * h;
                        const float4 bv4 = *reinterpret_cast<const float4*>(sBias + col);
                        float4 sv4 = make_float4(1.f, 1.f, 1.f, 1.f);
                        if constexpr (EP) sv4 = *reinterpret_cast<const float4*>(sScale + col);
                        uint2 pk;
                        pk.x = pack2bf(ep(acc[i][j][4 * q + 0], sv4.x, bv4.x), ep(acc[i][j][4 * q + 1], sv4.y, bv4.y));
                        pk.y = pack2bf(ep(acc[i][j][4 * q + 2], sv4.z, bv4.z), ep(acc[i][j][4 * q + 3], sv4.w, bv4.w));
                        *reinterpret_cast<uint2*>(sOut + row * OUT_ROW + col * 2) = pk;
                    }
                }
            }
            lds_barrier();                                 // B: staged (also the step barrier of the last chunk)
        }
        raw_barrier();
        raw_barrier();
    }
}

template <class C>
int launch_rw(FdArgs& a, hipStream_t stream) {
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_rw_kernel<C, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_rw_kernel<C, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_rw_kernel<C, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) segnb_set_error("fprop_rw hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    if (a.NCH * 9 * C::BN * 64 > C::W_MAX) return -12345;
    a.HB = (a.H + C::R - 1) / C::R;
    a.WB = (a.W + C::WT - 1) / C::WT;
    a.IT = a.N * a.HB * a.WB;
    a.NTL = 1;
    int gm = segnb_knob_conv_cus();
    if (gm > a.IT) gm = a.IT;
    a.GM = gm;
    if (a.ep_act >= 0)
        hipLaunchKernelGGL((conv_fprop_rw_kernel<C, false, true>), dim3(a.GM), dim3(C::NT), C::SMEM, stream, a);
    else if (a.dbg)       // timing builds: separately instantiated, the production kernel carries no run-time checks
        hipLaunchKernelGGL((conv_fprop_rw_kernel<C, true>), dim3(a.GM), dim3(C::NT), C::SMEM, stream, a);
    else
        hipLaunchKernelGGL((conv_fprop_rw_kernel<C, false>), dim3(a.GM), dim3(C::NT), C::SMEM, stream, a);
    return 0;
}

}  // namespace

// 1 = handled, 0 = not applicable, else error
int segnb_fprop_rw_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked,
                       unsigned w_bytes, const float* bias, int bias_n, void* out, double* stats,
                       hipStream_t stream, const segnb_bn_reduce_epilogue* bn, const segnb_act_epilogue* ep,
                       const segnb_upcat_src* uc, const segnb_upcat_src* upsum) {
    if (!segnb_knob_fprop_dma() || !segnb_knob_fprop_rw()) return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci % 32 != 0 || g->Ci > 96 || g->Co > 96 || g->Wo < 12) return 0;
    int dhmin = g->dh[0], dhmax = g->dh[0], dwmin = g->dw[0], dwmax = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dhmax = g->dh[t] > dhmax ? g->dh[t] : dhmax;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
        dwmax = g->dw[t] > dwmax ? g->dw[t] : dwmax;
    }
    if (dhmax - dhmin != 2 || dwmax - dwmin != 2) return 0;
    FdArgs a;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.x_bytes = in_bytes;
    a.w_bytes = w_bytes;
    a.bias = bias;
    a.bias_n = bias_n;
    a.out = (bf16_t*)out;
    a.stats = stats;
    a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Hi = g->Hi; a.Wi = g->Wi;
    a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
    a.Ktot = 9 * g->Ci;
    a.NCH = g->Ci / 32;
    {
        const long long ob = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
        if (ob >= (1ll << 31)) return 0;
        a.out_bytes = (unsigned)ob;
    }
    a.dhmin = dhmin; a.dwmin = dwmin;
    for (int t = 0; t < 9; ++t) {
        a.dh[t] = g->dh[t] - dhmin;
        a.dw[t] = g->dw[t] - dwmin;
    }
    a.dbg = segnb_knob_fprop_dma_dbg();
    a.up_out = nullptr;
    a.no_prev = 0;
    if (upsum != nullptr) {
        // (segnb_upcat_src reused as the destination: u = the low-resolution gradient, Cu = leading channels that go there)
        if (upsum->Cu % 8 != 0 || upsum->Cu <= 0 || upsum->Cu >= g->Co || (g->Ho & 1) || (g->Wo & 1) || stats != nullptr ||
            bn != nullptr || ep != nullptr || a.dbg)
            return 0;
        a.up_out = (bf16_t*)const_cast<void*>(upsum->u);
        a.up_C = upsum->Cu;
        a.up_ld = upsum->ld_u;
    }
    a.u = nullptr;
    if (uc != nullptr) {
        // virtual concat: `in` = the skip tensor (logical channels Cu..), uc->u = the tensor the first Cu channels are upsampled from
        if (uc->Cu % 32 != 0 || uc->Cu <= 0 || uc->Cu >= g->Ci || (g->Hi & 1) || (g->Wi & 1) || a.dbg) return 0;
        a.u = (const bf16_t*)uc->u;
        a.ld_u = uc->ld_u;
        a.NCHU = uc->Cu / 32;
        a.Hu = g->Hi / 2;
        a.Wu = g->Wi / 2;
        const long long ub = (((long long)g->N * a.Hu * a.Wu - 1) * uc->ld_u + uc->Cu) * 2;
        if (ub >= (1ll << 31)) return 0;
        a.u_bytes = (unsigned)ub;
    }
    a.bn_y = nullptr;
    a.ep_act = ep != nullptr ? ep->act : -1;
    a.ep_coef = ep != nullptr ? ep->coef : nullptr;
    a.ep_slope = ep != nullptr ? ep->slope : 0.f;
    if (bn != nullptr) {
        if (stats != nullptr || g->Co > 64) return 0;                 // one accumulator set per launch; 96-wide: no statistics threads
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
    // ring depth by what the resident weights leave of the 160 KiB.  96-channel outputs (the data gradient of the
    // 96 -> 32 concat layer: 53 KiB of staging beside 55 KiB of weights) run on a two-stage ring and without the
    // BatchNorm statistics (twelve chunks per row do not divide the store threads; a data gradient has none)
    const int nsw = segnb_knob_rw_store_waves();
    // 8 x 32 pixel tiles on wide images: longer contiguous rows per fetch / store (224 x 224: 65 vs 69 us forward, 54 vs 58
    // data gradient, same box); 16 x 16 stays better at 112 x 112 (40 vs 45 us)
    if (g->Co <= 32 && (nsw == 8 || (nsw == 4 && g->Wo >= 192)))
        rc = a.NCH == 1 ? launch_rw<RwCfg<32, 5, 4, 8, 32>>(a, stream)
             : a.NCH == 2 ? launch_rw<RwCfg<32, 4, 4, 8, 32>>(a, stream) : launch_rw<RwCfg<32, 3, 4, 8, 32>>(a, stream);
    else if (g->Co <= 32 && nsw == 4)
        rc = a.NCH == 1 ? launch_rw<RwCfg<32, 5, 4>>(a, stream)
             : a.NCH == 2 ? launch_rw<RwCfg<32, 4, 4>>(a, stream) : launch_rw<RwCfg<32, 3, 4>>(a, stream);
    else if (g->Co <= 32)
        rc = a.NCH == 1 ? launch_rw<RwCfg<32, 5>>(a, stream)
             : a.NCH == 2 ? launch_rw<RwCfg<32, 4>>(a, stream) : launch_rw<RwCfg<32, 3>>(a, stream);
    else if (g->Co <= 64)
        rc = nsw == 4 ? launch_rw<RwCfg<64, 3, 4>>(a, stream) : launch_rw<RwCfg<64, 3>>(a, stream);
    else if (stats == nullptr)
        rc = launch_rw<RwCfg<96, 2>>(a, stream);
    else
        return 0;
    if (rc == -12345) return 0;
    return rc ? rc : 1;
}
