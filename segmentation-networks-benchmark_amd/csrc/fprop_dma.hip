// Stride-1 3x3 convolution forward / data-gradient (bf16), gfx950 -- the direct-to-LDS, wave-specialised pipeline
// behind segnb_conv_fprop for layers with Ci % 64 == 0 (aten::convolution / convolution_backward(input) of
// lib/models/zf_unet.py:8 and the other 3x3 stride-1 convolutions of lib/models/*).
//
// fprop_s1.hip stages both operands through registers with one or two barriers per tap and exposes a global-load
// round trip at every one of them (PMC: 35-55 % of wave time parked in s_waitcnt / s_barrier, MFMA pipe 12-25 % busy).
// Here NOTHING passes through registers on its way to LDS:
//   * the input halo tile (R+2) x (WT+2) pixels x 64 channels and the weight tile BN x 64 channels of ONE tap are
//     fetched by `buffer_load_dwordx4 ... lds` (LDS-DMA): one wave-instruction moves 8 rows x 128 B = whole cache
//     lines, out-of-image pixels / out-of-range channels are the descriptor's range check (zeros land in LDS --
//     tools/probe_dma.hip);
//   * rows are 128 B with no padding (the DMA destination is lane-linear), bank conflicts are removed by an XOR
//     swizzle applied to the SOURCE address: the 16-byte slot q of row p holds channel chunk q ^ ((p >> 1) & 7), so
//     the sixteen lanes of a ds_read_b128 group (rows distinct mod 16) hit sixteen distinct bank quads;
//   * weights run through a ring of four one-tap stages fetched three taps ahead, the halo tile is double buffered
//     and fetched one 64-channel chunk ahead (across tile boundaries: the block is persistent and the stream of
//     (tile, chunk, tap) steps never drains); every tap ends with a COUNTED s_waitcnt vmcnt(N) -- this tap's fetches
//     stay in flight -- and one raw s_barrier;
//   * accumulators are TRANSPOSED (MFMA A operand = weights, B operand = pixels): a lane ends up with four
//     consecutive channels of one pixel per register quad, so the epilogue stages bf16 quads with ds_write_b64, and
//     the BatchNorm statistics are taken by the threads of the coalesced store pass (fixed channel chunk per thread).
//
// WAVE SPECIALISATION.  Waves 0..3 (one per SIMD) only read fragments and issue MFMAs; waves 4..7 (their SIMD
// partners) only issue the fetches, wait for them, and run the epilogue's store pass.  What was measured on the way
// (uniform 8-wave forms of the same pipeline, timing builds with fetches / MFMAs / reads / epilogue removed):
//   * an LDS-DMA instruction costs its wave 60-180 issue cycles, ~2.7 of them per tap and wave: 10-20 % of the kernel
//     when every wave pays that between its MFMAs;
//   * a ping-pong split of each tap (fetch + read phase vs MFMA phase, the two waves of a SIMD one phase apart) did
//     not help: the memory phase (fetches, ~50 address VALU, 16 reads and their latency) is longer than 16 MFMAs;
//   * pinning the issue order (reads of K slice kk+1 before the MFMAs of slice kk) by volatile asm is necessary --
//     the compiler sinks every read next to its MFMA -- but was not sufficient on its own.
// Here the matrix pipe of a SIMD belongs to ONE wave with a 128 x 64 or 64 x 64 accumulator tile: 32 / 16 MFMAs per
// tap against 24 / 16 fragment reads issued one K slice ahead (the first slice of a tap during the last slice of the
// tap before it, across the barrier), tap addresses precomputed per lane, while the partner's fetches issue in the gaps.
#include "common.h"

#include "fprop_dma.h"

#include <mutex>
#include <vector>

#ifndef SEGNB_EXP
#define SEGNB_EXP 0      // experiments on the production kernel (whole-file compile-time switches; never set in the build)
#endif

namespace {

// in-kernel time stamps of block 0 (timing builds, segnb_tune("fprop_dma_dbg", 32)): [role][step][4] shader clocks.
// The instrumented kernels are SEPARATE instantiations (template flag DBG): as run-time checks the stamp / timing-build
// tests cost every tap of the production kernel ~100 cycles (an empty tap measured 330 cycles).
__device__ unsigned long long g_stamps[3 * 256 * 4];
#define FD_STAMP(role, step, k)                                                                      \
    do {                                                                                            \
        if (DBG && ((a.dbg & 32) || ((a.dbg & 64) && (role) == 0 && (k) == 0)) && blockIdx.x == 0 && lane == 0 &&  \
            (step) < 256)                                                                            \
            g_stamps[((role) * 256 + (step)) * 4 + (k)] = __builtin_amdgcn_s_memtime();             \
    } while (0)

// TALL_: the batch is tiled as ONE image of N * (H + 1) rows -- every image followed by one virtual zero row that is
// the bottom padding of the image above it and the top padding of the one below -- and WT = W columns: tiles of R x W
// virtual pixels instead of one 16 x 16 tile per image.  For 7 x 7 images: 8 row tiles of 252 pixels instead of 32 tiles
// of 49 (81 % padding); the rows m >= R * WT of a tile are dummies.
// NTAP_ = 4, UPD_: the data gradient of an Upsample(x2) -> conv3x3 segment on the LOW-resolution grid, i.e. the 4x4 /
// stride-2 gather (ntaps 16, in_step 2) of segnb.engine.UpConvOp.  Per output pixel (Y, X) the sixteen taps split by the
// parity plane (py, px) of the high-resolution gradient dz they read: plane pixel (Y - py + ta, X - px + tb), ta, tb in
// {0, 1}.  A K chunk = (plane, 64 channels) with a 2 x 2 tap window whose halo origin is shifted by (-py, -px): the
// matrix waves see ONE tap geometry (window offsets {0,1}^2) for every chunk, the halo waves keep one set of per-lane source
// offsets per plane (dz pixel (2 Yp + py, 2 Xp + px): a strided LDS-DMA gather, whole 128-byte channel rows).
//
// UP_ = 2 (UPF): the FORWARD of such a segment, out[2 Y + py, 2 X + px] += sum_{a, b < 2} u[Y + py - 1 + a, X + px - 1 + b] . W[py,px][a,b]
// (the four phase launches of the 4x4 / stride-2 transposed convolution as ONE launch): a block owns one (phase, 64-channel
// tile) -- its weights and its halo origin (py - 1, px - 1) are fixed, the tile grid is the LOW-resolution one, the output
// pixel table maps a tile row to the high-resolution pixel of its phase -- and ADDS to what the skip segment's launch left
// in `out` (read at the start of a tile into 32 registers, added when the accumulators are staged; statistics on the sum).
template <int BN_, int R_, int WT_, int CW_M_, bool TALL_ = false, bool MF16_ = true, int NTAP_ = 9, int UP_ = 0>
struct WsCfg {
    static constexpr bool UPD_ = UP_ == 1;
    static constexpr bool UPF = UP_ == 2;
    static_assert(UP_ == 0 || (NTAP_ == 4 && !TALL_ && MF16_), "up-sampled segment forms: 2 x 2 window, 16x16x32 form");
    static constexpr int BN = BN_, R = R_, WT = WT_;
    static constexpr int NTAP = NTAP_, KW = NTAP_ == 9 ? 3 : 2, KH = KW;
    static constexpr bool UPD = UPD_;
    static constexpr int UP = UP_;
    static constexpr int NPL = UPD_ ? 4 : 1;             // source planes (sets of per-lane halo offsets)
    static_assert(NTAP_ == 9 || NTAP_ == 4, "3 x 3 or 2 x 2 tap window");
    static_assert(!UPD_ || (NTAP_ == 4 && !TALL_ && MF16_), "plane gather: 2 x 2 window, 16x16x32 form");
    static constexpr bool TALL = TALL_;
    static constexpr bool MF16 = MF16_;                  // v_mfma_f32_16x16x32_bf16 (else 32x32x16)
    static constexpr int TM16 = (TALL_ ? 256 : R_ * WT_) / CW_M_ / 16, TN16 = BN_ / (4 / CW_M_) / 16;
    static constexpr int NB = 4;
    static constexpr int NT = 512, NCW = 4, NLW = 4;
    static constexpr int BM = TALL_ ? 256 : R * WT;
    static_assert(R * WT <= BM, "tile pixels");
    static constexpr int WAVES_M = CW_M_, WAVES_N = NCW / CW_M_;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int XR = R + KH - 1, XC = WT + 2, NPIX = XR * XC;       // (even halo pitch kept for the 2 x 2 window)
    static constexpr int APIECES = (NPIX + 7) / 8;
    static constexpr int A_BYTES = APIECES * 1024;
    // fetch waves: NBW stream the weight ring, NAW the halo tiles -- separate vmcnt queues (a counted wait is in issue
    // order: behind a halo piece coming from HBM the weight ring's wait stalled every tap)
    static constexpr int NBW = 2, NAW = 2;
    static constexpr int APW = (APIECES + NAW - 1) / NAW;
    static constexpr int A_STEPS = NTAP - 2;             // halo pieces go out during taps 0..NTAP-3, all waited for in tap NTAP-2
    static constexpr int APS = (APW + A_STEPS - 1) / A_STEPS;
    static constexpr int BPIECES = BN / 8;
    static constexpr int B_STAGE = BN * 128;
    static constexpr int BPW = BPIECES / NBW;
    static constexpr int OUT_ROW = BN * 2 + 16;
    static constexpr int OC = BN / 8;
    static constexpr int MT = NCW * 64;                  // matrix threads: they also store (one channel chunk each)
    static constexpr int RPT = BM / (MT / OC);           // staged rows per thread = taps that carry one row's store
    // LDS: [halo x2][weight ring][output staging: whole tile][dummy piece][pixel tables x2][bias]
    static constexpr int OFF_B = 2 * A_BYTES;
    static constexpr int OFF_STG = OFF_B + NB * B_STAGE;
    static constexpr int OFF_DUMMY = OFF_STG + BM * OUT_ROW;
    static constexpr int OFF_PIX = OFF_DUMMY + 1024;
    static constexpr int OFF_BIAS = OFF_PIX + 4 * BM * 4;      // pixel tables of tiles k-1 (being stored), k, k+1, k+2 (set up)
    static constexpr int OFF_SCALE = OFF_BIAS + BN * 4;       // per-channel factor of the affine epilogue (1 when off)
    static constexpr int OFF_TICKET = OFF_SCALE + BN * 4;     // split K: the ticket the block drew (one word)
    static constexpr int SMEM = OFF_TICKET + 16;
    static constexpr int RED_BYTES = MT * 16 * 8;
    static_assert(WM % 32 == 0 && WN % 32 == 0 && WAVES_M * WAVES_N == NCW, "wave tiling");
    static_assert(BPIECES % NBW == 0 && NBW + NAW == NLW, "fetch wave roles");
    static_assert(MT % OC == 0 && MT >= 2 * BN && RPT <= 8 && BM == MT, "store pass / statistics threads");
    static_assert(RED_BYTES <= OFF_STG, "statistics reduction scratch");
    static_assert(SMEM <= 160 * 1024, "LDS");
    static_assert(TM + TN <= 6, "fragment wait statement");
};

// EP: the affine + activation epilogue (segnb_conv_fprop_act) is a SEPARATE instantiation: as run-time checks in the staging
// code of the training kernels it cost 0.22 ms per step (5.62 vs 5.40 ms, same box)
// STATS: BatchNorm statistics in the store rows (forward launches) -- data gradients take none and run the instantiation
// without the 16 accumulator registers and ~32 VALU per stored row
// (A BatchNorm-backward REDUCTION variant of the data gradient -- the two halo waves taking the sums -- was built in round 4 and
// measured slower in the step, ZF_UNET +1.6 %, LinkNet34 +2.8 %: DESIGN 11.10; removed in round 5.  fprop_rw.hip / fprop_roll.hip
// keep their fused reductions.)
// SPLITK: FdArgs::KS blocks per (pixel tile, channel tile), every block owns ONE tile slice (grid = tiles x KS); see FdArgs::KS
// MASK: a DATA GRADIENT whose output tile is the gradient g of a convolution + activation WITHOUT BatchNorm (FdArgs::bn_y = that
// layer's ACTIVATED output, bn_coef NULL): the tile is stored as dz = round(round(g) * act'(y)) -- the layer's own pass over (g, y, dz)
// never runs -- and the store pass's sums (the STATS machinery: sum of the stored values per channel) are its bias gradient.  The
// activations of a tile are read by the two HALO waves (16 bytes x 16 rows per thread, requested behind the tile's first halo pieces
// and in flight for the whole tile), folded into one bit per element and published as 8 bytes per tile row behind the LDS layout
// (C::SMEM .. + 2 KB) during the tile's last tap; the matrix waves read their rows' bits when they stage the accumulators.
// (In the matrix waves themselves -- 32 registers per tile in the accumulator layout, or 2 x 8 in flight three taps ahead of a
// fold -- the loads either spilled or stalled the MFMA stream: 64 -> 64 @ 1024 x 1024 x 4 took 673 us against 486 plain.)
template <class C, bool DBG, bool EP = false, bool STATS = true, bool SPLITK = false, bool MASK = false>
__global__ __launch_bounds__(512) void conv_fprop_ws_kernel(const FdArgs a) {
    static_assert(!SPLITK || (!EP && !DBG && C::UP == 0 && C::NTAP == 9), "split K: plain 3 x 3 forward / data gradient");
    static_assert(!MASK || (STATS && !EP && !DBG && !SPLITK && C::MF16 && C::UP == 0 && C::NTAP == 9 && !C::TALL && C::BM == 256 &&
                            C::BN == 64 && C::WAVES_M == 4 && C::SMEM + 2048 <= 160 * 1024),
                  "activation mask: plain 3 x 3 data gradient, 16x16x32 form, 256 x 64 tiles, 2 KB of mask bytes behind the LDS layout");
    constexpr int BN = C::BN, R = C::R, WT = C::WT, BM = C::BM, TM = C::TM, TN = C::TN, XC = C::XC;
    constexpr int NT = C::NT, NB = C::NB, APW = C::APW, APS = C::APS, BPW = C::BPW, OC = C::OC, NLW = C::NLW;
    constexpr int OUT_ROW = C::OUT_ROW, NF = TM + TN;
    constexpr int NTAP = C::NTAP, KW = C::KW;
    constexpr int SM = NTAP & 3;                         // weight-ring stage of step (chunk cg, tap t) = (cg * SM + t) & 3

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    int* sPix = reinterpret_cast<int*>(smem + C::OFF_PIX);
    float* sBias = reinterpret_cast<float*>(smem + C::OFF_BIAS);
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool loader = wave >= C::NCW;
    const int lw = wave - C::NCW;                                   // loader index 0..3
    const int wm = (wave & 3) / C::WAVES_N, wn = (wave & 3) % C::WAVES_N;

    // consecutive L share an XCD (and its L2).  Default order: the NTL channel tiles of one pixel tile are neighbours (they
    // read the same halo tiles).  ntmajor (tall tiles of the 7 x 7 level, where the weights are 6 x the activations and every
    // XCD would pull ALL of them through the fabric -- 151 MB per 1024 -> 1024 launch, which bounded it): the GM pixel tiles
    // of one (channel tile, slice) are neighbours, and so are the slices of a tile
    int L_ = xcd_remap_fd(blockIdx.x, gridDim.x);
    int ks_ = 0, nt_, gq_;
    if (C::TALL && a.ntmajor) {
        gq_ = L_ % a.GM;
        L_ /= a.GM;
        if constexpr (SPLITK) {
            ks_ = L_ % a.KS;
            L_ /= a.KS;
        }
        nt_ = L_;
    } else {
        if constexpr (SPLITK) {
            ks_ = L_ % a.KS;
            L_ /= a.KS;
        }
        nt_ = L_ % a.NTL;
        gq_ = L_ / a.NTL;
    }
    const int ks = ks_;
    const int c_off = SPLITK ? ks * a.NCH : 0;      // first input-channel chunk of this block
    const int nt = nt_, gq = gq_;
    // n_base: first output channel of the block (output, bias, statistics); w_row0: its first row of the weight matrix
    int n_base_ = nt * BN, w_row0_ = nt * BN, py_ = 0, px_ = 0;
    if constexpr (C::UPF) {
        const int ph = nt / a.NTLR, ct = nt - ph * a.NTLR;      // (phase, channel tile)
        n_base_ = ct * BN;
        w_row0_ = ph * a.CoW + ct * BN;
        py_ = ph >> 1;
        px_ = ph & 1;
    }
    const int n_base = n_base_, w_row0 = w_row0_, py = py_, px = px_;

    const i32x4_t rs_x = make_rsrc4(a.x, a.x_bytes);
    const i32x4_t rs_w = make_rsrc4(a.w, a.w_bytes);
    const i32x4_t rs_u = make_rsrc4(a.u != nullptr ? a.u : a.x, a.u != nullptr ? a.u_bytes : a.x_bytes);

    float* sScale = reinterpret_cast<float*>(smem + C::OFF_SCALE);
    for (int c = tid; c < BN; c += NT) {
        const int co = n_base + c;
        const float bv = (a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;
        float sc = 1.f, sh = bv;
        if constexpr (EP) {
            if (a.ep_coef != nullptr && co < a.Co) {      // (acc + bias - mean) * scale + shift
                sc = a.ep_coef[co];
                sh = (bv - a.ep_coef[2 * a.Co + co]) * sc + a.ep_coef[a.Co + co];
            }
            sScale[c] = sc;
        }
        sBias[c] = sh;
    }
    const float ep_neg = a.ep_act == SEGNB_ACT_RELU ? 0.f : (a.ep_act == SEGNB_ACT_LEAKY ? a.ep_slope : 1.f);
    const float upf_neg = ep_neg;
    const float mask_neg = a.bn_act == SEGNB_ACT_RELU ? 0.f : (a.bn_act == SEGNB_ACT_LEAKY ? a.bn_slope : 1.f);
    auto ep = [&](float acc, float sc, float sh) {
        if constexpr (EP) {
            const float v = acc * sc + sh;
            return v < 0.f ? v * ep_neg + 0.f : v;
        } else {
            return acc + sh;
        }
    };

    // ---- three programs (weight waves / halo waves / matrix waves) with the same barrier sequence: one after the
    // pipeline prologue, one per tap, three after the last tile.  Per-role constants are computed inside the role's
    // branch so that they do not occupy registers of the others.
    int it = gq;
    const bool wfetch = loader && lw < C::NBW;          // weight-ring wave (else: halo wave)
    int cg = 0;
    // Two programs with the same barrier sequence (one barrier per tap, three after the last tile).
    if (loader) {
        // ================= fetch streams =================
        // Nothing but fetches (a wave that also stored would sit behind its own stores' acknowledgements at the next
        // counted wait), and ONE kind of fetch per wave: the weight waves wait tap by tap for L2 hits, the halo waves
        // once per chunk for HBM.
        if (wfetch) {
            // ---- loader-side per-lane constants --------------------------------------------------------------------
            unsigned b_voff[BPW];
            unsigned b_sec = 0;      // (P32) bit pb: this lane's slot of piece pb holds the chunk's SECOND plane
        #pragma unroll
            for (int pb = 0; pb < BPW; ++pb) {
                const int row = ((lw & 1) * BPW + pb) * 8 + (lane >> 3);
                const int q = lane & 7;
                b_sec |= (unsigned)(((q ^ ((row >> 1) & 7)) >> 2) & 1) << pb;
                const int co = n_base + row;
                b_voff[pb] = co < a.Co ? (unsigned)(w_row0 + row) * (unsigned)a.Ktot * 2u + (unsigned)((q ^ ((row >> 1) & 7)) * 16) : OOB;
                if constexpr (C::UPD)
                    if (a.P32 && co < a.Co)      // 32-channel planes: the row's K slots 0..3 / 4..7 belong to two planes (below)
                        b_voff[pb] = (unsigned)(w_row0 + row) * (unsigned)a.Ktot * 2u + (unsigned)(((q ^ ((row >> 1) & 7)) & 3) * 16);
            }

            auto fetch_b = [&](int c, int t, int stage) {
                unsigned soff;
                if constexpr (C::UPD) {
                    // chunk c = (plane, 64-channel slice); tap (ta, tb) of plane (py, px) is tap (2 ta - py + 1, 2 tb - px + 1)
                    // of the 4 x 4 kernel (row-major tap list of segnb.convplan.convt_dgrad)
                    if (a.P32) {
                        // a chunk = planes 2 c and 2 c + 1, 32 channels each: the two halves of a weight row come from two taps of
                        // the 4 x 4 kernel -- a per-lane offset instead of the scalar one
                        const int pa_ = 2 * c, pb_ = 2 * c + 1;
                        const int ta = (2 * (t >> 1) - (pa_ >> 1) + 1) * 4 + (2 * (t & 1) - (pa_ & 1) + 1);
                        const int tb = (2 * (t >> 1) - (pb_ >> 1) + 1) * 4 + (2 * (t & 1) - (pb_ & 1) + 1);
        #pragma unroll
                        for (int pb = 0; pb < BPW; ++pb) {
                            const unsigned off = (unsigned)(((b_sec >> pb & 1u) ? tb : ta) * 32) * 2u;
                            dma16(lds0 + C::OFF_B + stage * C::B_STAGE + ((lw & 1) * BPW + pb) * 1024, b_voff[pb] + off, rs_w, 0u);
                        }
                        return;
                    }
                    const int pl = c / a.NCHP, sl = c - pl * a.NCHP;
                    const int t16 = (2 * (t >> 1) - (pl >> 1) + 1) * 4 + (2 * (t & 1) - (pl & 1) + 1);
                    soff = (unsigned)(t16 * a.Ci + sl * 64) * 2u;
                } else if constexpr (C::UPF) {
                    // window tap (ta, tb) = tap (1 - ta, 1 - tb) of the phase's packed list (segnb.convplan._scatter_phases
                    // lists the kernel rows of a phase in ascending order = descending input offset)
                    soff = (unsigned)((3 - t) * a.Ci + c * 64) * 2u;
                } else {
                    soff = (unsigned)(t * a.Ci + (c + c_off) * 64) * 2u;
                }
        #pragma unroll
                for (int pb = 0; pb < BPW; ++pb)
                    dma16(lds0 + C::OFF_B + stage * C::B_STAGE + ((lw & 1) * BPW + pb) * 1024, b_voff[pb], rs_w, soff);
            };

            fetch_b(0, 0, 0);
            fetch_b(0, 1, 1);
            fetch_b(0, 2, 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            for (; it < a.IT; it += a.GM) {
                for (int c = 0; c < a.NCH; ++c, ++cg) {
                    const int cn = c + 1 == a.NCH ? 0 : c + 1;
                    static_for<NTAP>([&](auto t_c) {
                        constexpr int t = decltype(t_c)::value;
                        constexpr int tf = t + 3 < NTAP ? t + 3 : t + 3 - NTAP;
                        const int cf = t + 3 < NTAP ? c : cn;
                        if (lw == 0) FD_STAMP(1, cg * 9 + t, 0);
                        if (!(SEGNB_EXP & 1) && !(DBG && (a.dbg & 1))) fetch_b(cf, tf, (cg * SM + t + 3) & (NB - 1));
                        if (lw == 0) FD_STAMP(1, cg * 9 + t, 1);
                        // the weights of tap t+2 (fetched during tap t-1) have landed: only this tap's fetch stays in flight
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW) : "memory");
                        if (lw == 0) FD_STAMP(1, cg * 9 + t, 2);
                        if (!(SEGNB_EXP & 64) || t % 3 == 2) raw_barrier();      // (experiment 64: one barrier per kernel row -- WRONG results, timing only)
                        if (lw == 0) FD_STAMP(1, cg * 9 + t, 3);
                    });
                }
            }
        } else {
            // Per lane the pixel of a halo piece is fixed: its offset relative to the tile origin and its halo coordinates are
            // computed ONCE; per tile only the origin (scalar) and the border compares remain (recomputing pix / XC etc. per
            // tile stalled every wave of the block ~1800 cycles at the tile's last chunk: in-kernel stamps, tools/stamps.py).
            // The same call writes the tile's output pixel table (read by the store rows one tile later).
            unsigned a_rel[APW], a_xy[APW], a_voff[C::NPL][APW];
            unsigned a_uoff[APW];         // virtual concat: the same halo pixels in the low-resolution tensor u
            const bool vcat = a.u != nullptr;
        #pragma unroll
            for (int pa = 0; pa < APW; ++pa) {
                const int pix = ((lw & 1) + C::NAW * pa) * 8 + (lane >> 3);
                const int q = lane & 7;
                const int xr = pix / XC, xc = pix - xr * XC;
                const int key = C::XC % 2 == 0 ? xc : pix;        // swizzle key (see a_rd below)
                a_rel[pa] = (unsigned)((xr * a.Wi + xc) * (C::UPD ? 2 : 1)) * (unsigned)a.ld_x * 2u +
                            (unsigned)((q ^ ((key >> 1) & 7)) * 16);
                if constexpr (C::UPD)
                    if (a.P32)       // 32-channel planes: slots 0..3 / 4..7 of a halo row come from the chunk's first / second plane
                        a_rel[pa] = (unsigned)((xr * a.Wi + xc) * 2) * (unsigned)a.ld_x * 2u + (unsigned)(((q ^ ((key >> 1) & 7)) & 3) * 16);
                a_xy[pa] = pix < C::NPIX ? (unsigned)xr | ((unsigned)xc << 16) : 0x7fff7fffu;      // never inside the image
            }
            auto set_fetch_tile = [&](int it, int table) {
                const bool live = it < a.IT;
                if constexpr (C::TALL) {
                    // virtual row v of the tall image -> (image v / (H+1), row v % (H+1)); row H of an image is the shared
                    // zero row.  (float reciprocal: exact for v < 2^20)
                    const int HV = a.H + 1, VT = a.N * HV;
                    const float inv = 1.0f / (float)HV;
                    const int v0 = it * R + a.dhmin;
        #pragma unroll
                    for (int pa = 0; pa < APW; ++pa) {
                        const int v = v0 + (int)(a_xy[pa] & 0xffffu), wi = a.dwmin + (int)(a_xy[pa] >> 16);
                        const int n = (int)(((float)v + 0.5f) * inv);
                        const int hh = v - n * HV;
                        const bool ok = live && (unsigned)v < (unsigned)VT && hh < a.H && (unsigned)wi < (unsigned)a.Wi;
                        const int pix = ((lw & 1) + C::NAW * pa) * 8 + (lane >> 3);
                        a_voff[0][pa] = ok ? (unsigned)((n * a.Hi + hh) * a.Wi + wi) * (unsigned)a.ld_x * 2u +
                                              (unsigned)(((lane & 7) ^ ((pix >> 1) & 7)) * 16)      // (XC odd: key = pix)
                                        : OOB;
                    }
                    const int t128 = (lw & 1) * 64 + lane;
        #pragma unroll
                    for (int rr = t128; rr < BM; rr += C::NAW * 64) {
                        const int v = it * R + rr / WT, wo = rr % WT;
                        const int n = (int)(((float)v + 0.5f) * inv);
                        const int hh = v - n * HV;
                        sPix[table * BM + rr] =
                            (live && rr < R * WT && v < VT && hh < a.H && wo < a.W) ? (n * a.H + hh) * a.W + wo : -1;
                    }
                    return;
                }
                const int n = it / (a.HB * a.WB);
                const int rem = it - n * (a.HB * a.WB);
                const int hb = rem / a.WB, wb = rem - hb * a.WB;
                if constexpr (C::UPD) {
                    // plane (py, px): halo position (xr, xc) = plane pixel (hb R - py + xr, wb WT - px + xc), valid inside
                    // the H x W plane; its source is dz pixel (2 Yp + py, 2 Xp + px).  (32-bit wrap-around arithmetic: the
                    // tile origin may lie one plane pixel outside the tensor, a VALID lane's offset never does)
                    const unsigned hlim = live ? (unsigned)a.H : 0u;
                    static_for<4>([&](auto pl_c) {
                        constexpr int pl = decltype(pl_c)::value, py = pl >> 1, px = pl & 1;
                        const int y0 = hb * R - py, x0 = wb * WT - px;
                        const unsigned base = (unsigned)(((n * a.Hi + 2 * y0 + py) * a.Wi + 2 * x0 + px) * a.ld_x * 2);
        #pragma unroll
                        for (int pa = 0; pa < APW; ++pa) {
                            const int yp = y0 + (int)(a_xy[pa] & 0xffffu), xp = x0 + (int)(a_xy[pa] >> 16);
                            const bool ok = (unsigned)yp < hlim && (unsigned)xp < (unsigned)a.W;
                            a_voff[pl][pa] = ok ? base + a_rel[pa] : OOB;
                        }
                    });
                    if (a.P32) {
                        // chunk c reads plane 2 c in its first four slots and plane 2 c + 1 in the others: fold the four
                        // plane sets into two per-chunk sets, lane by lane
        #pragma unroll
                        for (int pa = 0; pa < APW; ++pa) {
                            const int xc = (int)(a_xy[pa] >> 16);
                            const bool second = ((((int)lane & 7) ^ ((xc >> 1) & 7)) >> 2) != 0;
                            a_voff[0][pa] = second ? a_voff[1][pa] : a_voff[0][pa];
                            a_voff[1][pa] = second ? a_voff[3][pa] : a_voff[2][pa];
                        }
                    }
                } else {
                const int h0 = hb * R + (C::UPF ? py - 1 : a.dhmin), w0 = wb * WT + (C::UPF ? px - 1 : a.dwmin);
                const unsigned base = (unsigned)(((n * a.Hi + h0) * a.Wi + w0) * a.ld_x * 2);
                const unsigned hlim = live ? (unsigned)a.Hi : 0u;
        #pragma unroll
                for (int pa = 0; pa < APW; ++pa) {
                    const int hi = h0 + (int)(a_xy[pa] & 0xffffu), wi = w0 + (int)(a_xy[pa] >> 16);
                    const bool ok = (unsigned)hi < hlim && (unsigned)wi < (unsigned)a.Wi;
                    a_voff[0][pa] = ok ? base + a_rel[pa] : OOB;
                    if (vcat) {      // nearest-x2 upsample = the pixel (hi >> 1, wi >> 1) of u, same channel slot
                        const int pixl = ((lw & 1) + C::NAW * pa) * 8 + (lane >> 3);
                        const int keyl = C::XC % 2 == 0 ? pixl % XC : pixl;
                        a_uoff[pa] = ok ? (unsigned)((n * a.Hu + (hi >> 1)) * a.Wu + (wi >> 1)) * (unsigned)a.ld_u * 2u +
                                              (unsigned)(((lane & 7) ^ ((keyl >> 1) & 7)) * 16)
                                        : OOB;
                    }
                }
                }
                // output pixel of tile rows (-1 = outside the image): 128 halo-wave threads, BM / 128 rows each
                const int t128 = (lw & 1) * 64 + lane;
        #pragma unroll
                for (int rr = t128; rr < BM; rr += C::NAW * 64) {
                    const int ho = hb * R + rr / WT, wo = wb * WT + rr % WT;
                    if constexpr (C::UPF)       // the tile row's pixel of this block's phase in the high-resolution output
                        sPix[table * BM + rr] = (live && ho < a.H && wo < a.W) ? (n * 2 * a.H + 2 * ho + py) * (2 * a.W) + 2 * wo + px : -1;
                    else
                        sPix[table * BM + rr] = (live && ho < a.H && wo < a.W) ? (n * a.H + ho) * a.W + wo : -1;
                }
            };
            auto fetch_a = [&](int p0, int p1, int c_, int buf) {
                const int c = c_ + c_off;
                int pl = 0;
                unsigned soff = (unsigned)c * 128u;
                if constexpr (C::UPD) {
                    if (a.P32) {
                        pl = c;                  // (folded per-chunk sets, whole 32-channel planes: no slice offset)
                        soff = 0u;
                    } else {
                        pl = c / a.NCHP;
                        soff = (unsigned)(c - pl * a.NCHP) * 128u;
                    }
                }
                if constexpr (!C::UPD && !C::UPF && !C::TALL) {
                    if (vcat) {
                        if (c < a.NCHU) {           // a chunk of the (virtually) upsampled segment: fetched from u
        #pragma unroll
                            for (int pa = 0; pa < APW; ++pa) {
                                if (pa >= p0 && pa < p1) {
                                    const int piece = (lw & 1) + C::NAW * pa;
                                    const unsigned dst = piece < C::APIECES ? lds0 + buf * C::A_BYTES + piece * 1024 : lds0 + C::OFF_DUMMY;
                                    dma16(dst, a_uoff[pa], rs_u, (unsigned)c * 128u);
                                }
                            }
                            return;
                        }
                        soff = (unsigned)(c - a.NCHU) * 128u;      // skip segment: x holds its channels from 0
                    }
                }
                auto go = [&](auto pl_c) {
                    constexpr int PL = decltype(pl_c)::value;
        #pragma unroll
                    for (int pa = 0; pa < APW; ++pa) {
                        if (pa >= p0 && pa < p1) {
                            const int piece = (lw & 1) + C::NAW * pa;
                            const unsigned dst = piece < C::APIECES ? lds0 + buf * C::A_BYTES + piece * 1024 : lds0 + C::OFF_DUMMY;
                            dma16(dst, a_voff[PL][pa], rs_x, soff);
                        }
                    }
                };
                if constexpr (C::UPD) {
                    if (pl == 0) go(std::integral_constant<int, 0>{});
                    else if (pl == 1) go(std::integral_constant<int, 1>{});
                    else if (pl == 2) go(std::integral_constant<int, 2>{});
                    else go(std::integral_constant<int, 3>{});
                } else {
                    go(std::integral_constant<int, 0>{});
                }
            };
            // MASK: thread t128 of the 128 halo-wave threads owns channel chunk t128 & 7 of the staged rows (t128 >> 3) + 16 k
            typedef __attribute__((ext_vector_type(4))) unsigned mu32x4_t;
            mu32x4_t mreg[MASK ? 16 : 1];
            const __amdgpu_buffer_rsrc_t rs_y =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(MASK ? a.bn_y : a.x), 0, MASK ? (int)a.bn_y_bytes : 0, 0x00020000);
            auto mask_request = [&](int table) {
                const int t128 = (lw & 1) * 64 + lane;
                const int ch = n_base + (t128 & 7) * 8;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int opix = sPix[table * BM + (t128 >> 3) + 16 * k];
                    const unsigned voff = (opix >= 0 && ch < a.Co) ? (unsigned)opix * (unsigned)a.bn_ld * 2u + (unsigned)ch * 2u : OOB;
                    mreg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)voff, 0, 0);
                }
            };
            auto mask_publish = [&]() {
                const int t128 = (lw & 1) * 64 + lane;
                unsigned char* const dst = smem + C::SMEM + (t128 >> 3) * 8 + (t128 & 7);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const mu32x4_t v = mreg[k];
                    unsigned b = 0u;
                    b |= __uint_as_float(v.x << 16) > 0.f ? 1u : 0u;
                    b |= __uint_as_float(v.x & 0xffff0000u) > 0.f ? 2u : 0u;
                    b |= __uint_as_float(v.y << 16) > 0.f ? 4u : 0u;
                    b |= __uint_as_float(v.y & 0xffff0000u) > 0.f ? 8u : 0u;
                    b |= __uint_as_float(v.z << 16) > 0.f ? 16u : 0u;
                    b |= __uint_as_float(v.z & 0xffff0000u) > 0.f ? 32u : 0u;
                    b |= __uint_as_float(v.w << 16) > 0.f ? 64u : 0u;
                    b |= __uint_as_float(v.w & 0xffff0000u) > 0.f ? 128u : 0u;
                    dst[16 * k * 8] = (unsigned char)b;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // published by this tap's barrier
            };
            set_fetch_tile(it, 0);
            fetch_a(0, APW, 0, 0);
            // During a tile's LAST chunk the first chunk of the next tile is fetched, so the per-lane offsets must
            // describe that tile by then: they are switched during tap 7 of the chunk before (taps 7 and 8 issue nothing
            // and leave ~1.5 k cycles of slack; done at the last chunk's start it held every wave at a barrier).
            if (a.NCH == 1) set_fetch_tile(it + a.GM, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            int tile_no = 0;                                   // tiles of this block; pixel table = tile_no & 3
            for (; it < a.IT; it += a.GM, ++tile_no) {
                for (int c = 0; c < a.NCH; ++c, ++cg) {
                    const bool last = c + 1 == a.NCH;
                    const int cn = last ? 0 : c + 1;
                    // the chunk after this one: is it the last of its tile?  then set up the tile after that one
                    const bool next_last = last ? a.NCH == 1 : c + 2 == a.NCH;
                    const int setup_it = last ? it + 2 * a.GM : it + a.GM;
                    const int setup_tab = (tile_no + (last ? 2 : 1)) & 3;
                    const int abuf = cg & 1;
                    static_for<NTAP>([&](auto t_c) {
                        constexpr int t = decltype(t_c)::value;
                        if (lw == C::NBW) FD_STAMP(2, cg * 9 + t, 0);
                        if constexpr (t < C::A_STEPS)
                            if (!(SEGNB_EXP & 2) && !(DBG && (a.dbg & 2))) fetch_a(t * APS, (t + 1) * APS, cn, abuf ^ 1);
                        if constexpr (MASK) {
                            // the producing layer's activated output at THIS tile's pixels: 16 bytes (8 channels) x 16 rows per thread,
                            // requested behind the first chunk's first halo pieces (in flight for the whole tile: the counted wait of
                            // its last chunk's tap NTAP - 2 covers them), folded into one mask byte each during the tile's last tap
                            if constexpr (t == 0)
                                if (c == 0) mask_request(tile_no & 3);
                            if constexpr (t == NTAP - 1)
                                if (last) mask_publish();
                        }
                        if constexpr (t == NTAP - 2)
                            if (next_last) set_fetch_tile(setup_it, setup_tab);
                        if (lw == C::NBW) FD_STAMP(2, cg * 9 + t, 1);
                        // the next chunk's halo tile is first read during the last tap (look-ahead slices of its tap 0)
                        if constexpr (t == NTAP - 2) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        if (lw == C::NBW) FD_STAMP(2, cg * 9 + t, 2);
                        if (!(SEGNB_EXP & 64) || t % 3 == 2) raw_barrier();      // (experiment 64: one barrier per kernel row -- WRONG results, timing only)
                        if (lw == C::NBW) FD_STAMP(2, cg * 9 + t, 3);
                    });
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // look-ahead fetches of the (absent) next tile
        if constexpr (SPLITK) {       // the matrix waves' slab hand-over: slabs drained / ticket published
            raw_barrier();
            raw_barrier();
        }
        raw_barrier();
        raw_barrier();
        raw_barrier();
    } else {
        // ================= matrix stream (+ the previous tile's stores in its shadow) =================
        // Fragment reads run TWO K slices ahead of the MFMAs through three register sets (under load an LDS read takes
        // longer than the four MFMAs of one slice).  The first two slices of a tap are requested during the last two
        // slices of the tap before it (the fetch waves guarantee a tap's operands one barrier early) and stay in flight
        // across the barrier.  36 slices per chunk = 0 mod 3: the set of slice (t, kk) is static.
        //
        // EPILOGUE IN THE MFMA SHADOW.  At the end of a tile the accumulators are staged (bias, bf16) into a buffer of
        // their own -- no barrier: the first tap barrier of the next tile publishes it -- and the tile's coalesced
        // stores + BatchNorm statistics are issued one staged row per thread and tap during taps 1..RPT of the NEXT
        // tile, between its MFMAs (a matrix wave needs 8 of every 32 issue cycles).  Serialised, the epilogue was
        // 2-2.5 k cycles per tile: 30 % of a 64 -> 64 tile, 7 % of a 256 -> 256 one.
        // ---- compute-side per-lane constants: fragment offsets (see the uniform kernel for the swizzle) -----------
        constexpr bool MF16 = C::MF16;
        int tile_no_out = 0;
        bool pending_out = false;
        constexpr int TM16 = C::TM16, TN16 = C::TN16;
        int b_rd[MF16 ? 1 : TN];
        int a_rd[MF16 ? 1 : 9][MF16 ? 1 : TM];
        // 16x16x32 form: lane = (row r16 of the 16-row fragment, 8-channel group g4 of the 32-channel K slice); the 16-byte
        // slot of (row, slice kk, group g4) is (kk * 4 + g4) ^ ((key >> 1) & 7) -- the same source swizzle, read 32 channels deep
        // per fragment BOTH K-slice variants (kk = 0 / 1: the slot index differs in bit 2, i.e. address bit 6 XOR-ed) are kept,
        // and the tap enters only through its column shift dw (the swizzle key of a pixel is its halo column) plus a
        // wave-uniform row term: a read address is ONE vector add of a scalar (matrix waves have 8 free issue cycles per
        // 16-cycle MFMA).  Requires XC even and taps ordered t = 3 * (row) + (column) (checked on the host).
        static_assert(!MF16 || XC % 2 == 0, "16x16x32 form: the swizzle key must be the halo column");
        int b_rd16[MF16 ? 2 : 1][MF16 ? TN16 : 1];
        int a_rd16[MF16 ? KW : 1][MF16 ? TM16 : 1][MF16 ? 2 : 1];
        if constexpr (MF16) {
            const int r16 = lane & 15, g4 = lane >> 4;
    #pragma unroll
            for (int j = 0; j < TN16; ++j) {
                const int row = wn * C::WN + 16 * j + r16;
                const int v = C::OFF_B + row * 128 + ((g4 ^ ((row >> 1) & 7)) << 4);
                b_rd16[0][j] = v;
                b_rd16[1][j] = v ^ 64;
            }
    #pragma unroll
            for (int dwi = 0; dwi < KW; ++dwi)
    #pragma unroll
                for (int i = 0; i < TM16; ++i) {
                    const int m = wm * C::WM + 16 * i + r16;
                    const int p = (m / WT) * XC + (m % WT) + a.dw[dwi];          // column-shifted pixel, tap row 0
                    const int key = p % XC;
                    int v = (p << 7) + ((g4 ^ ((key >> 1) & 7)) << 4);
                    asm volatile("" : "+v"(v));
                    a_rd16[dwi][i][0] = v;
                    a_rd16[dwi][i][1] = v ^ 64;
                }
        } else {
        #pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * C::WN + 32 * j + r;
                b_rd[j] = C::OFF_B + row * 128 + ((row & 12) << 3) + (((h ^ (row >> 1)) & 1) << 4);
            }
        #pragma unroll
            for (int t = 0; t < 9; ++t)
        #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = wm * C::WM + 32 * i + r;
                    const int p = (m / WT) * XC + (m % WT) + a.dh[t] * XC + a.dw[t];
                    // swizzle key: the halo COLUMN when the row pitch XC is even -- the sixteen lanes of a ds_read_b128 group then
                // cover sixteen consecutive columns mod 16 whether they lie in one tile row (32-wide tiles) or in two (16-wide
                // tiles: keyed by the row index p, two of sixteen lanes collide; measured neutral on the same box, kept for the
                // cleaner bank picture)
                const int key = XC % 2 == 0 ? p % XC : p;
                int v = (p << 7) + ((key & 12) << 3) + (((h ^ (key >> 1)) & 1) << 4);
                    asm volatile("" : "+v"(v));
                    a_rd[t][i] = v;
                }

        }
        lds_barrier();                                         // pipeline prologue: first operands have landed
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        unsigned char* const sOut = smem + C::OFF_STG;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
        const int cc = tid % OC, row0 = tid / OC;              // store pass: channel chunk, first staged row
        const bool cok = n_base + cc * 8 < a.Co;
        float s1[8], s2[8];                                    // statistics of the stored values of chunk cc
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        // A store row in two halves: its two LDS reads (pixel table entry, staged 16 bytes) are asm reads issued at the
        // START of the tap, older than the tap's first fragment wait, so they cost no wait of their own (as compiler
        // loads after the MFMAs they stalled on the look-ahead fragment reads in flight); the store and the statistics
        // run on registers after the MFMAs.
        int row_pix = -1, row_pix2 = -1;
        u32x4_t row_v = {0, 0, 0, 0}, row_v2 = {0, 0, 0, 0};     // (second set: two store rows per step, short 2 x 2-window tiles)
        auto row_load = [&](int k, const int* tab) {           // staged row row0 + k * (MT / OC) of the previous tile
            const int row = row0 + k * (C::MT / OC);
            const unsigned a_pix = (unsigned)(size_t)(tab + row), a_v = (unsigned)(size_t)(sOut + row * OUT_ROW + cc * 16);
            asm volatile("ds_read_b32 %0, %1" : "=v"(row_pix) : "v"(a_pix));
            asm volatile("ds_read_b128 %0, %1" : "=v"(row_v) : "v"(a_v));
        };
        auto row_load2 = [&](int k, const int* tab) {
            const int row = row0 + k * (C::MT / OC);
            const unsigned a_pix = (unsigned)(size_t)(tab + row), a_v = (unsigned)(size_t)(sOut + row * OUT_ROW + cc * 16);
            asm volatile("ds_read_b32 %0, %1" : "=v"(row_pix2) : "v"(a_pix));
            asm volatile("ds_read_b128 %0, %1" : "=v"(row_v2) : "v"(a_v));
        };
        auto row_store_of = [&](int opix, const u32x4_t v) {
            const bool ok = cok && opix >= 0;
            const unsigned voff = ok ? (unsigned)opix * (unsigned)a.ld_out * 2u + (unsigned)(n_base + cc * 8) * 2u : OOB;
            if (!(DBG && (a.dbg & 8))) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (int)voff, 0, 0);
            if constexpr (STATS) {
                const float m = ok ? 1.f : 0.f;
                float f[8];
                f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
                f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
                f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
                f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float fm = f[e] * m;
                    s1[e] += fm;
                    if constexpr (!MASK) s2[e] += fm * fm;      // (MASK: the sums are a bias gradient -- slot 1 stays zero)
                }
            }
        };
        auto row_store = [&]() { row_store_of(row_pix, row_v); };
        // ---- split K: slab hand-over of the one tile slice this block owns (MI355X_MICROARCH.md, inter-workgroup visibility: write-
        // through stores, every storing wave drained, a workgroup barrier, ONE agent-scope add per block; the block whose add
        // came last reads every slab with sc1 loads).  The sum runs over the slabs in slice order, the reducer's own included,
        // so the result does not depend on which block arrives last.
        bool ks_last = true;
        const __amdgpu_buffer_rsrc_t rs_slab =
            __builtin_amdgcn_make_buffer_rsrc(SPLITK ? a.ks_slab : nullptr, 0, SPLITK ? (int)a.ks_slab_bytes : 0, 0x00020000);
        const unsigned ks_tile = (unsigned)(gq * a.NTL + nt);
        const unsigned slab0 = ((ks_tile * (unsigned)a.KS) * 4u + (unsigned)(wave & 3)) * 16384u + (unsigned)lane * 16u;   // slice 0, piece 0
        auto ks_publish = [&](auto&& piece) {              // piece(p): accumulator registers 4 p .. 4 p + 3 of this lane
            unsigned own = slab0 + (unsigned)ks * 65536u;
            asm volatile("" : "+v"(own));                  // (computed here: hoisted to the kernel's start the 16 offsets were spilled)
            static_for<16>([&](auto p_c) {
                constexpr int p = decltype(p_c)::value;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, piece(p_c)), rs_slab, (int)(own + p * 1024u), 0,
                                                       16 /* sc1 */);
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            raw_barrier();                                 // every storing wave has drained
            int* const sTicket = reinterpret_cast<int*>(smem + C::OFF_TICKET);
            if (tid == 0) {
                const int tk = __hip_atomic_fetch_add(a.ks_cnt + ks_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk == a.KS - 1) {
                    __hip_atomic_store(a.ks_cnt + ks_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                sTicket[0] = tk;
            }
            lds_barrier();
            return sTicket[0] == a.KS - 1;
        };
        // The reducer's sum: its own slice from registers, the others by sc1 loads, all of them in flight together (a slab read is a
        // fabric round trip of ~2 us: four dependent batches cost the launch more than the split gained).  Fixed association
        // whoever reduces: KS = 2: p0 + p1; KS = 4: (p0 + p1) + (p2 + p3) -- IEEE addition commutes, so which operand came from
        // registers does not show.  get(p) / set(p, v): accumulator registers 4 p .. 4 p + 3 of this lane.
        auto ks_load = [&](unsigned off) {
            return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_slab, (int)off, 0, 16 /* sc1 */));
        };
        auto ks_combine = [&](auto&& get, auto&& set) {
            unsigned base = slab0;
            asm volatile("" : "+v"(base));
            if (a.KS == 2) {
                const unsigned other = base + (unsigned)(ks ^ 1) * 65536u;
                static_for<2>([&](auto h_c) {               // (two batches of eight: 32 registers -- sixteen at once spilled)
                    constexpr int h8 = decltype(h_c)::value * 8;
                    f32x4_t o[8];
                    static_for<8>([&](auto q_c) { constexpr int q = decltype(q_c)::value; o[q] = ks_load(other + (h8 + q) * 1024u); });
                    static_for<8>([&](auto q_c) {
                        constexpr int q = decltype(q_c)::value;
                        set(std::integral_constant<int, h8 + q>{}, get(std::integral_constant<int, h8 + q>{}) + o[q]);
                    });
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                });
            } else {
                static_for<4>([&](auto h_c) {               // (four pieces x four slices per batch: 64 registers)
                    constexpr int h8 = decltype(h_c)::value * 4;
                    f32x4_t v[4][4];
                    static_for<4>([&](auto q_c) {
                        constexpr int q = decltype(q_c)::value;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            v[q][k] = ks_load(k != ks ? base + (unsigned)k * 65536u + (h8 + q) * 1024u : OOB);
                    });
                    static_for<4>([&](auto q_c) {
                        constexpr int q = decltype(q_c)::value;
                        const f32x4_t own = get(std::integral_constant<int, h8 + q>{});
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[q][k] = k == ks ? own : v[q][k];
                        set(std::integral_constant<int, h8 + q>{}, (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]));
                    });
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the next batch's loads start after this batch's sums)
                });
            }
        };
        if constexpr (MF16) {
            // ---- 16x16x32 matrix stream: 2 K slices of 32 channels per tap; wave tile = TM16 x TN16 tiles of 16 pixels x
            // 16 channels (4 x 4: 16 MFMAs of 16 cycles per slice).  Two fragment sets (set = slice parity): the 8 reads of
            // the NEXT slice go out in the first 8 MFMA gaps of the current one, in consumption order (x0 w0 w1 w2 w3 x1 x2
            // x3), and every use waits only for what it consumes (counted lgkmcnt).
            static_assert(TM16 == 4 && TN16 == 4, "16x16x32 wave tile is 64 pixels x 64 channels");
            bf16x8_t fw[2][TN16], fx[2][TM16];
            // q-th read of a fragment set, in consumption order; t = tap of the slice being fetched, k = its K slice,
            // wbase / xbase = wave-uniform LDS bases of its weight stage / halo buffer (+ the tap's row term)
            auto issue = [&](int q, int set, int wbase, int xbase, int t, int k) {
                if (q == 0) { const int ad = a_rd16[t % KW][0][k] + xbase; FD_READ(fx[set][0], ad); }
                else if (q <= 4) { const int ad = b_rd16[k][q - 1] + wbase; FD_READ(fw[set][q - 1], ad); }
                else { const int ad = a_rd16[t % KW][q - 4][k] + xbase; FD_READ(fx[set][q - 4], ad); }
            };
            int arow[KW];                                          // row term of the tap rows (bytes)
#pragma unroll
            for (int k = 0; k < KW; ++k) arow[k] = __builtin_amdgcn_readfirstlane(a.dh[KW * k] * XC * 128);
            if (!(DBG && (a.dbg & 4))) {
#pragma unroll
                for (int q = 0; q < 8; ++q) issue(q, 0, 0, arow[0], 0, 0);
            }
            int tile_no = 0;
            bool pending = false;
            for (; it < a.IT; it += a.GM, ++tile_no) {
                f32x4_t acc[TM16][TN16];
                // UPF: what the skip segment's launch left at this tile's output pixels (4 bf16 per accumulator quad),
                // requested now, added when the accumulators are staged -- a whole tile later
                uint2 prev[C::UPF ? TM16 : 1][C::UPF ? TN16 : 1];
                if constexpr (C::UPF) {
                    const int r16 = lane & 15, g4 = lane >> 4;
                    const int* tab = sPix + (tile_no & 3) * BM;
#pragma unroll
                    for (int i = 0; i < TM16; ++i) {
                        const int opix = tab[wm * C::WM + 16 * i + r16];
#pragma unroll
                        for (int j = 0; j < TN16; ++j) {
                            const int ch = n_base + wn * C::WN + 16 * j + 4 * g4;
                            uint2 v = make_uint2(0u, 0u);
                            if (opix >= 0 && ch < a.Co && !a.no_prev)
                                v = *reinterpret_cast<const uint2*>(a.out + (long long)opix * a.ld_out + ch);
                            prev[i][j] = v;
                        }
                    }
                }
                for (int c = 0; c < a.NCH; ++c, ++cg) {
                    const int a_base = (cg & 1) * C::A_BYTES;
                    const bool drain = pending && c == 0;
                    static_for<NTAP>([&](auto t_c) {
                        constexpr int t = decltype(t_c)::value;
                        constexpr int tn = t == NTAP - 1 ? 0 : t + 1;
                        auto& accr = acc;
                        const int bstage = ((cg * SM + t) & (NB - 1)) * C::B_STAGE;
                        const int bnext = ((cg * SM + t + 1) & (NB - 1)) * C::B_STAGE;
                        const int anext = t == NTAP - 1 ? ((cg + 1) & 1) * C::A_BYTES : a_base;
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 0);
                        // the previous tile's store rows: one per tap from the tile's second step on -- taps 1..RPT of the first
                        // chunk (3 x 3 window, compile-time) or steps 1..RPT across the first chunks (2 x 2 window: 4 taps each)
                        constexpr bool row_tap9 = NTAP == 9 && t >= 1 && t <= C::RPT;
                        bool do_row = false;
                        int krow = t - 1;
                        if constexpr (NTAP == 9) {
                            do_row = drain;
                        } else {
                            // (a.RPS = 1: rows 0..RPT-1 on steps 1..RPT;  2: rows 2 k, 2 k + 1 on step k + 1 -- tiles of fewer
                            // than RPT + 1 steps: 128 input channels)
                            const int si = c * NTAP + t;
                            krow = (si - 1) * a.RPS;
                            do_row = pending && si >= 1 && krow < C::RPT;
                        }
                        if constexpr (row_tap9 || NTAP != 9)
                            if (do_row) {
                                row_load(krow, sPix + ((tile_no + 3) & 3) * BM);
                                if constexpr (NTAP != 9)
                                    if (a.RPS == 2) row_load2(krow + 1, sPix + ((tile_no + 3) & 3) * BM);
                            }
                        if (!(SEGNB_EXP & 4) && !(DBG && (a.dbg & 4))) {
                            __builtin_amdgcn_s_setprio(1);
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) {
                                const int cur = kk, nxt = kk ^ 1;
                                // the slice fetched during this one: (t, 1) in this tap's stage, or (t + 1, 0) in the next tap's
                                const int wbase = kk == 0 ? bstage : bnext;
                                const int xbase = kk == 0 ? a_base + arow[t / KW] : anext + arow[tn / KW];
                                // the current slice's x0, w0..w3 have landed (its x1..x3 may still be in flight; the
                                // two row-store reads of a row tap are younger: the count is then conservative)
                                ws_wait5<3>(fx[cur][0], fw[cur][0], fw[cur][1], fw[cur][2], fw[cur][3]);
#pragma unroll
                                for (int i = 0; i < TM16; ++i) {
                                    // one read of the next slice every second MFMA gap: at the waits below 2 i of the 8 are out
                                    if (i == 1) ws_wait1<4>(fx[cur][1]);      // older x2, x3 + 2 new reads may be outstanding
                                    if (i == 2) ws_wait1<5>(fx[cur][2]);      // older x3 + 4 new
                                    if (i == 3) ws_wait1<6>(fx[cur][3]);      // 6 new
#pragma unroll
                                    for (int j = 0; j < TN16; ++j) {
                                        const int q = i * TN16 + j;
                                        if (q % 2 == 0 && !(SEGNB_EXP & 16) && !(DBG && (a.dbg & 16))) {
                                            if (kk == 0) issue(q / 2, nxt, wbase, xbase, t, 1);
                                            else issue(q / 2, nxt, wbase, xbase, tn, 0);
                                        }
                                        if (t == 0 && kk == 0 && c == 0)
                                            FD_MFMA16_0(accr[i][j], fw[cur][j], fx[cur][i]);
                                        else
                                            FD_MFMA16(accr[i][j], fw[cur][j], fx[cur][i]);
                                    }
                                }
                            }
                            __builtin_amdgcn_s_setprio(0);
                        }
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 1);
                        if constexpr (row_tap9 || NTAP != 9)
                            if (do_row) {
                                if (DBG && (a.dbg & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                if constexpr (NTAP != 9)
                                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(row_pix), "+v"(row_v), "+v"(row_pix2), "+v"(row_v2));
                                else
                                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(row_pix), "+v"(row_v));     // older than the 8 look-ahead reads
                                row_store();
                                if constexpr (NTAP != 9)
                                    if (a.RPS == 2) row_store_of(row_pix2, row_v2);
                            }
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 2);
                        raw_barrier();
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 3);
                    });
                }
                // look-ahead reads of the next tile's first slice must have LANDED before compiler-scheduled code runs
                ws_wait5<0>(fx[0][0], fw[0][0], fw[0][1], fw[0][2], fw[0][3]);
                ws_wait1<0>(fx[0][1]);
                ws_wait1<0>(fx[0][2]);
                ws_wait1<0>(fx[0][3]);
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::);
                if constexpr (SPLITK) {
                    ks_last = ks_publish([&](auto p_c) { constexpr int p = decltype(p_c)::value; return acc[p / TN16][p % TN16]; });
                    if (ks_last)
                        ks_combine([&](auto p_c) { constexpr int p = decltype(p_c)::value; return acc[p / TN16][p % TN16]; },
                                   [&](auto p_c, const f32x4_t v) { constexpr int p = decltype(p_c)::value; acc[p / TN16][p % TN16] = v; });
                }
                if (!SPLITK || ks_last) {
                // accumulator tile (i, j): this lane holds channels 16 j + 4 g4 + {0..3} of pixel 16 i + r16
                const int r16 = lane & 15, g4 = lane >> 4;
                float4 bv[TN16], sv[TN16];
                uint2 mrow[MASK ? TM16 : 1];      // MASK: the 64 mask bits of this lane's pixel rows (written by the halo waves)
                if constexpr (MASK) {
#pragma unroll
                    for (int i = 0; i < TM16; ++i)
                        mrow[i] = *reinterpret_cast<const uint2*>(smem + C::SMEM + (wm * C::WM + 16 * i + r16) * 8);
                }
#pragma unroll
                for (int j = 0; j < TN16; ++j) {
                    if constexpr (!MASK) bv[j] = *reinterpret_cast<const float4*>(sBias + wn * C::WN + 16 * j + 4 * g4);
                    if constexpr (EP) sv[j] = *reinterpret_cast<const float4*>(sScale + wn * C::WN + 16 * j + 4 * g4);
                    else sv[j] = make_float4(1.f, 1.f, 1.f, 1.f);
                }
#pragma unroll
                for (int i = 0; i < TM16; ++i) {
                    const int row = wm * C::WM + 16 * i + r16;
#pragma unroll
                    for (int j = 0; j < TN16; ++j) {
                        const int col = wn * C::WN + 16 * j + 4 * g4;
                        uint2 pk;
                        if constexpr (C::UPF) {
                            // (activation of a transposed convolution's own forward, segnb_upconv_fprop_act: max(v, v * neg) with
                            // neg = 0 / slope / 1 -- the identity when no activation rides on the launch)
                            const uint2 pv = prev[i][j];
                            const float v0 = acc[i][j][0] + __uint_as_float(pv.x << 16) + bv[j].x;
                            const float v1 = acc[i][j][1] + __uint_as_float(pv.x & 0xffff0000u) + bv[j].y;
                            const float v2 = acc[i][j][2] + __uint_as_float(pv.y << 16) + bv[j].z;
                            const float v3 = acc[i][j][3] + __uint_as_float(pv.y & 0xffff0000u) + bv[j].w;
                            pk.x = pack2bf(fmaxf(v0, v0 * upf_neg), fmaxf(v1, v1 * upf_neg));
                            pk.y = pack2bf(fmaxf(v2, v2 * upf_neg), fmaxf(v3, v3 * upf_neg));
                        } else if constexpr (MASK) {
                            // dz = round(round(g) * act'(y)), act' from the sign of the activated value (1 / mask_neg)
                            // channel 16 j + 4 g4 + e: byte 2 j + (g4 >> 1) of the row's eight, bit 4 (g4 & 1) + e
                            const unsigned mb = (j < 2 ? mrow[i].x : mrow[i].y) >> (8 * ((2 * j + (g4 >> 1)) & 3) + 4 * (g4 & 1));
                            const unsigned g01 = pack2bf(acc[i][j][0], acc[i][j][1]);      // (a data gradient: no bias)
                            const unsigned g23 = pack2bf(acc[i][j][2], acc[i][j][3]);
                            const float g0 = __uint_as_float(g01 << 16), g1 = __uint_as_float(g01 & 0xffff0000u);
                            const float g2 = __uint_as_float(g23 << 16), g3 = __uint_as_float(g23 & 0xffff0000u);
                            pk.x = pack2bf((mb & 1u) ? g0 : g0 * mask_neg, (mb & 2u) ? g1 : g1 * mask_neg);
                            pk.y = pack2bf((mb & 4u) ? g2 : g2 * mask_neg, (mb & 8u) ? g3 : g3 * mask_neg);
                        } else {
                        pk.x = pack2bf(ep(acc[i][j][0], sv[j].x, bv[j].x), ep(acc[i][j][1], sv[j].y, bv[j].y));
                        pk.y = pack2bf(ep(acc[i][j][2], sv[j].z, bv[j].z), ep(acc[i][j][3], sv[j].w, bv[j].w));
                        }
                        *reinterpret_cast<uint2*>(sOut + row * OUT_ROW + col * 2) = pk;
                    }
                }
                }
                pending = !SPLITK || ks_last;
            }
            tile_no_out = tile_no;
            pending_out = pending;
        } else {
            static_assert(MF16 || NTAP == 9, "the 32x32x16 form serves the 3 x 3 window only (fragment sets cycle with 36 slices)");

            bf16x8_t fr[3][NF];
            if (!(DBG && (a.dbg & 4))) {
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int ad = b_rd[j] ^ (kk << 5);
                        FD_READ(fr[kk][j], ad);
                    }
    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int ad = a_rd[0][i] ^ (kk << 5);
                        FD_READ(fr[kk][TN + i], ad);
                    }
                }
            }
            int tile_no = 0;                                       // pixel table of tile k = k & 3 (written by the halo waves)
            bool pending = false;
            for (; it < a.IT; it += a.GM, ++tile_no) {
                f32x16_t acc[TM][TN];                              // (zeroed by the first MFMAs of the tile: C operand 0)
                for (int c = 0; c < a.NCH; ++c, ++cg) {
                    const int a_base = (cg & 1) * C::A_BYTES;
                    const bool drain = pending && c == 0;          // this chunk carries the previous tile's stores
                    static_for<9>([&](auto t_c) {
                        constexpr int t = decltype(t_c)::value;
                        constexpr int tn = t == 8 ? 0 : t + 1;
                        auto& accr = acc;
                        const int bstage = ((cg + t) & (NB - 1)) * C::B_STAGE;
                        const int bnext = ((cg + t + 1) & (NB - 1)) * C::B_STAGE;
                        const int anext = t == 8 ? ((cg + 1) & 1) * C::A_BYTES : a_base;
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 0);
                        constexpr bool row_tap = t >= 1 && t <= C::RPT;
                        if constexpr (row_tap)
                            if (drain) row_load(t - 1, sPix + ((tile_no + 3) & 3) * BM);
                        if (!(SEGNB_EXP & 4) && !(DBG && (a.dbg & 4))) {
                            __builtin_amdgcn_s_setprio(1);
    #pragma unroll
                            for (int kk = 0; kk < 4; ++kk) {
                                const int set_cur = (4 * t + kk) % 3, set_new = (4 * t + kk + 2) % 3;
                                int ad[NF];
                                if (kk < 2) {                   // slice kk + 2 of this tap
    #pragma unroll
                                    for (int j = 0; j < TN; ++j) ad[j] = (b_rd[j] + bstage) ^ ((kk + 2) << 5);
    #pragma unroll
                                    for (int i = 0; i < TM; ++i) ad[TN + i] = (a_rd[t][i] + a_base) ^ ((kk + 2) << 5);
                                } else {                        // slice kk - 2 of the next tap
    #pragma unroll
                                    for (int j = 0; j < TN; ++j) ad[j] = (b_rd[j] + bnext) ^ ((kk - 2) << 5);
    #pragma unroll
                                    for (int i = 0; i < TM; ++i) ad[TN + i] = (a_rd[tn][i] + anext) ^ ((kk - 2) << 5);
                                }
                                // one fragment read of slice +2 in each MFMA gap (issued as a block of four before the MFMAs, the
                                // reads and their address VALU did not fit the shadow of the previous slice's last MFMA)
                                ws_wait<NF>(fr[set_cur]);
                                static_assert(TM * TN == NF, "one read per MFMA gap");
    #pragma unroll
                                for (int i = 0; i < TM; ++i)
    #pragma unroll
                                    for (int j = 0; j < TN; ++j) {
                                        const int q = i * TN + j;
                                        if (!(SEGNB_EXP & 16) && !(DBG && (a.dbg & 16))) FD_READ(fr[set_new][q], ad[q]);
                                        if ((SEGNB_EXP & 128) && i > 0) continue;      // (experiment 128: half the MFMAs -- timing only)
                                        if (t == 0 && kk == 0 && c == 0)
                                            FD_MFMA0(accr[i][j], fr[set_cur][j], fr[set_cur][TN + i]);
                                        else
                                            FD_MFMA(accr[i][j], fr[set_cur][j], fr[set_cur][TN + i]);
                                    }
                            }
                            __builtin_amdgcn_s_setprio(0);
                        }
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 1);
                        if constexpr (row_tap)
                            if (drain) {
                                if (DBG && (a.dbg & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                asm volatile("" : "+v"(row_pix), "+v"(row_v));      // landed: older than the tap's fragment waits
                                row_store();
                            }
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 2);
                        if (!(SEGNB_EXP & 64) || t % 3 == 2) raw_barrier();      // (experiment 64: one barrier per kernel row -- WRONG results, timing only)
                        if (wave == 0) FD_STAMP(0, cg * 9 + t, 3);
                    });
                }
                // The slices requested for the next tile's first tap must have LANDED before compiler-scheduled code runs:
                // their destination registers count as written, and a copy taken before the data arrives is a stale
                // register (seen as run-to-run differences at bs=32).
                ws_wait<0>(fr[0]);
                ws_wait<0>(fr[1]);
                ws_wait<0>(fr[2]);
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::);      // last MFMA results land before they are read
                if constexpr (SPLITK) {
                    static_assert(MF16 || TM * TN == 4, "split K: 16 accumulator quads per lane");
                    auto quad = [&](auto p_c) {
                        constexpr int p = decltype(p_c)::value, t = p >> 2, g = p & 3;
                        const f32x16_t& s16 = acc[t / TN][t % TN];
                        return f32x4_t{s16[4 * g], s16[4 * g + 1], s16[4 * g + 2], s16[4 * g + 3]};
                    };
                    ks_last = ks_publish(quad);
                    if (ks_last)
                        ks_combine(quad, [&](auto p_c, const f32x4_t v) {
                            constexpr int p = decltype(p_c)::value, t = p >> 2, g = p & 3;
                            f32x16_t& d16 = acc[t / TN][t % TN];
                            d16[4 * g] = v[0]; d16[4 * g + 1] = v[1]; d16[4 * g + 2] = v[2]; d16[4 * g + 3] = v[3];
                        });
                }
                if (!SPLITK || ks_last) {
                float4 bv[TN][4];                                  // (one batch of LDS reads, not one round trip per quad)
    #pragma unroll
                for (int j = 0; j < TN; ++j)
    #pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bv[j][g] = *reinterpret_cast<const float4*>(sBias + wn * C::WN + 32 * j + 8 * g + 4 * h);
    #pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * C::WM + 32 * i + r;
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
    #pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = wn * C::WN + 32 * j + 8 * g + 4 * h;
                            const float4 bv4 = bv[j][g];
                            uint2 pk;
                            pk.x = pack2bf(acc[i][j][4 * g + 0] + bv4.x, acc[i][j][4 * g + 1] + bv4.y);
                            pk.y = pack2bf(acc[i][j][4 * g + 2] + bv4.z, acc[i][j][4 * g + 3] + bv4.w);
                            *reinterpret_cast<uint2*>(sOut + row * OUT_ROW + col * 2) = pk;
                        }
                    }
                }
                }
                pending = !SPLITK || ks_last;
            }
            tile_no_out = tile_no;
            pending_out = pending;
        }
        lds_barrier();                                         // the last tile is staged
        if (pending_out) {
#pragma unroll
            for (int k = 0; k < C::RPT; ++k) {
                row_load(k, sPix + ((tile_no_out + 3) & 3) * BM);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(row_pix), "+v"(row_v));
                row_store();
            }
        }
        // ---- statistics: fixed-order block reduction, one fp64 atomic per channel and block ----------------------
        lds_barrier();                                         // (fetch waves: everything has landed; staging consumed)
        if constexpr (STATS) {
            double* red = reinterpret_cast<double*>(smem);     // [MT][16]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[tid * 16 + e] = (double)s1[e];
                red[tid * 16 + 8 + e] = MASK ? 0.0 : (double)s2[e];
            }
        }
        lds_barrier();
        if (STATS && a.stats != nullptr && tid < (MASK ? 1 : 2) * BN) {
            const double* red = reinterpret_cast<const double*>(smem);
            const int which = tid / BN, col = tid - which * BN;
            const int c8 = col >> 3, e = col & 7;
            double sum = 0.0;
            for (int k = 0; k < C::MT / OC; ++k) sum += red[(k * OC + c8) * 16 + which * 8 + e];
            const int co = n_base + col;
            if (co < a.Co)
                atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + co], sum);
        }
    }
}

// split-K workspace: slabs + ticket counters per (device, stream), grown on demand (launches on one stream are ordered by the
// stream; the counters are zeroed when allocated and every launch leaves them zero).  Allocation happens in the eager /
// recording step of a geometry, replayed launch lists find the same pointers.
int ksplit_workspace(hipStream_t stream, size_t slab_bytes, int ntiles, float** slab, int** cnt) {
    struct Ent {
        int dev;
        hipStream_t stream;
        float* slab;
        size_t cap;
        int* cnt;
        int ncnt;
    };
    static std::mutex mu;
    static std::vector<Ent> ents;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(mu);
    Ent* e = nullptr;
    for (auto& it : ents)
        if (it.dev == dev && it.stream == stream) e = &it;
    const bool grow = e == nullptr || slab_bytes > e->cap || ntiles > e->ncnt;
    if (grow) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return -1;   // no allocation under capture
        if (e == nullptr) {
            ents.push_back(Ent{dev, stream, nullptr, 0, nullptr, 0});
            e = &ents.back();
        }
        // (old buffers may still be in use by launches in flight: left allocated -- a handful of times per process)
        if (slab_bytes > e->cap) {
            float* p = nullptr;
            if (hipMalloc(&p, slab_bytes) != hipSuccess) return -1;
            e->slab = p;
            e->cap = slab_bytes;
        }
        if (ntiles > e->ncnt) {
            const int want = ntiles < 1024 ? 1024 : ntiles;
            int* c = nullptr;
            if (hipMalloc(&c, (size_t)want * sizeof(int)) != hipSuccess) return -1;
            // cleared IN the launching stream (a null-stream memset is not ordered with a non-blocking stream: ADVICE r5)
            if (hipMemsetAsync(c, 0, (size_t)want * sizeof(int), stream) != hipSuccess) return -1;
            e->cnt = c;
            e->ncnt = want;
        }
    }
    *slab = e->slab;
    *cnt = e->cnt;
    return 0;
}

// split K pays where a launch has fewer (pixel tile, channel tile) pairs than half the CUs it may use: the 7 x 7 level of
// lib/models/zf_unet.py:44-56 (8 tall row tiles x 8..16 channel tiles at bs = 32).  Returns the number of slices (1 = no split)
template <class C>
int ksplit_factor(const FdArgs& a, int it_total, int ntl, int nch) {
    if (!C::TALL || !segnb_knob_fprop_ksplit()) return 1;
    if (a.ep_act >= 0 || a.dbg || a.up_out != nullptr) return 1;
    const long long tiles = (long long)it_total * ntl;
    int cus = segnb_knob_conv_cus();
    // A data gradient (no statistics) of a two-stream backward runs beside the weight-gradient stream, which sizes its launches
    // for a share of the CUs (wgrad_s1.hip: s1_slabs): slices beyond the CUs that are left would queue behind their partners
    // and only add the hand-over
    // (measured, same box, alternating runs: 4.92 ms per step with half or all of the CUs budgeted for the data gradients)
    if (a.stats == nullptr) cus /= 2;
    int ks = 1;
    if (tiles * 2 <= cus && nch % 2 == 0 && nch / 2 >= 2) ks = 2;
    if (tiles * 4 <= cus && nch % 4 == 0 && nch / 4 >= 2) ks = 4;
    const int forced = segnb_knob_fprop_ksplit();
    if (forced > 1 && nch % forced == 0 && nch / forced >= 2 && (forced == 2 || forced == 4)) ks = forced;
    if (tiles * ks * 65536ll >= (1ll << 31)) return 1;
    return ks;
}

constexpr int NOT_HANDLED = -12345;

template <class C>
int launch_ws(FdArgs& a, hipStream_t stream) {
    if (a.bn_y != nullptr && !(C::MF16 && !C::TALL && C::SMEM + 2048 <= 160 * 1024)) return NOT_HANDLED;      // (activation mask: 16 x 16 tiles, 16x16x32 form)
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, true, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, false, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if constexpr (C::MF16 && !C::TALL && C::SMEM + 2048 <= 160 * 1024) {
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, false, true, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM + 2048);
        }
        if constexpr (C::TALL) {
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, false, true, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, false, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        }
        if (e != hipSuccess) segnb_set_error("fprop_ws hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + C::R - 1) / C::R;
    a.WB = (a.W + C::WT - 1) / C::WT;
    a.IT = a.N * a.HB * a.WB;
    if (C::TALL) {
        if (a.W > C::WT || a.Hi != a.H || a.Wi != a.W) return -12345;
        a.IT = (a.N * (a.H + 1) + C::R - 1) / C::R;
    }
    a.NTL = (a.Co + C::BN - 1) / C::BN;
    a.NCH = a.Ci / 64;
    a.KS = 1;
    a.ntmajor = C::TALL ? 1 : 0;
    if constexpr (C::TALL) {
        const int ks = ksplit_factor<C>(a, a.IT, a.NTL, a.NCH);
        if (ks > 1 && ksplit_workspace(stream, (size_t)a.IT * a.NTL * ks * 65536, a.IT * a.NTL, &a.ks_slab, &a.ks_cnt) == 0) {
            // one tile slice per block: grid = tiles x KS, every block walks NCH / KS chunks
            a.KS = ks;
            a.NCH /= ks;
            a.GM = a.IT;
            a.ks_slab_bytes = (unsigned)((size_t)a.IT * a.NTL * ks * 65536);
            const dim3 grid(a.IT * a.NTL * ks);
            if (a.stats != nullptr || !segnb_knob_fprop_nostats())
                hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, false, true, true>), grid, dim3(C::NT), C::SMEM, stream, a);
            else
                hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, false, false, true>), grid, dim3(C::NT), C::SMEM, stream, a);
            return 0;
        }
    }
    int gm = segnb_knob_conv_cus() / a.NTL;
    if (gm < 1) gm = 1;
    if (gm > a.IT) gm = a.IT;
    a.GM = gm;
    if (a.bn_y != nullptr) {      // activation mask of the producing layer (MASK instantiation: 16x16x32 form only)
        if constexpr (C::MF16 && !C::TALL && C::SMEM + 2048 <= 160 * 1024) {
            hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, false, true, false, true>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM + 2048,
                               stream, a);
            return 0;
        } else {
            return NOT_HANDLED;
        }
    }
    if (a.ep_act >= 0)            // (never with statistics: segnb_conv_fprop_act takes none)
        hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, true, false>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    else if (a.stats == nullptr && !a.dbg && segnb_knob_fprop_nostats())
        hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, false, false>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    else if (a.dbg)
        hipLaunchKernelGGL((conv_fprop_ws_kernel<C, true>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    else
        hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    return 0;
}

// the plane gather (data gradient of an upsampled segment): one instantiation (no statistics, no epilogue)
template <class C>
int launch_ws_upd(FdArgs& a, hipStream_t stream) {
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, false, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) segnb_set_error("fprop_ws hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + C::R - 1) / C::R;
    a.WB = (a.W + C::WT - 1) / C::WT;
    a.IT = a.N * a.HB * a.WB;
    a.NTL = (a.Co + C::BN - 1) / C::BN;
    a.P32 = a.Ci == 32;
    a.NCHP = a.P32 ? 1 : a.Ci / 64;
    a.NCH = a.P32 ? 2 : 4 * a.NCHP;
    a.RPS = a.NCH * C::NTAP - 1 < C::RPT ? 2 : 1;               // the previous tile's store rows ride on steps 1.., one or two each
    if ((a.NCH * C::NTAP - 1) * a.RPS < C::RPT) return NOT_HANDLED;
    int gm = segnb_knob_conv_cus() / a.NTL;
    if (gm < 1) gm = 1;
    if (gm > a.IT) gm = a.IT;
    a.GM = gm;
    hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, false, false>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    return 0;
}

// the forward of an up-sampled segment: four phases in one launch, accumulating (WsCfg UP_ = 2)
template <class C>
int launch_ws_upf(FdArgs& a, hipStream_t stream) {
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_ws_kernel<C, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) segnb_set_error("fprop_ws hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + C::R - 1) / C::R;
    a.WB = (a.W + C::WT - 1) / C::WT;
    a.IT = a.N * a.HB * a.WB;
    a.NTLR = (a.Co + C::BN - 1) / C::BN;
    a.NTL = 4 * a.NTLR;
    a.NCH = a.Ci / 64;
    a.NCHP = a.NCH;
    a.P32 = 0;
    a.RPS = a.NCH * C::NTAP - 1 < C::RPT ? 2 : 1;               // store rows per step
    if ((a.NCH * C::NTAP - 1) * a.RPS < C::RPT) return NOT_HANDLED;
    int gm = segnb_knob_conv_cus() / a.NTL;
    if (gm < 1) gm = 1;
    if (gm > a.IT) gm = a.IT;
    a.GM = gm;
    hipLaunchKernelGGL((conv_fprop_ws_kernel<C, false, false, true>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    return 0;
}

// ntaps 16, in_step 2: out[Y, X] = sum_{a, b < 4} in[2 Y + a - 1, 2 X + b - 1] . W[a * 4 + b]  (segnb.convplan.convt_dgrad of a
// 4 x 4 / stride-2 / pad-1 transposed convolution -- UpConvOp, unet16.py:38, linknet.py:16)
bool upd_geometry(const segnb_conv_geom* g) {
    if (g->ntaps != 16 || g->in_step != 2 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return false;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Hi != 2 * g->Ho || g->Wi != 2 * g->Wo) return false;
    if ((g->Ci % 64 != 0 && g->Ci != 32) || g->Co <= 32 || g->Wo < 12) return false;
    for (int t = 0; t < 16; ++t)
        if (g->dh[t] != t / 4 - 1 || g->dw[t] != t % 4 - 1) return false;
    return true;
}

int try_upd(const segnb_conv_geom* g, FdArgs& a, hipStream_t stream) {
    if (!upd_geometry(g)) return NOT_HANDLED;
    a.Ktot = 16 * g->Ci;
    a.dhmin = a.dwmin = 0;
    for (int t = 0; t < 9; ++t) a.dh[t] = a.dw[t] = 0;
    a.dh[2] = a.dh[3] = 1;          // 2 x 2 window, row-major: (0,0) (0,1) (1,0) (1,1)
    a.dw[1] = a.dw[3] = 1;
    // 8 x 32 or 16 x 16 output tiles: fewer rounds of (equal) tiles on the persistent blocks wins, ties go to 16 x 16
    const int ntl = (g->Co + 63) / 64;
    int gm = segnb_knob_conv_cus() / ntl;
    if (gm < 1) gm = 1;
    const long long it0 = (long long)g->N * ((g->Ho + 7) / 8) * ((g->Wo + 31) / 32);
    const long long it1 = (long long)g->N * ((g->Ho + 15) / 16) * ((g->Wo + 15) / 16);
    int cfg = segnb_knob_fprop_dma_cfg();
    if (cfg < 0) cfg = (it0 + gm - 1) / gm < (it1 + gm - 1) / gm ? 0 : 1;
    if (cfg == 0) return launch_ws_upd<WsCfg<64, 8, 32, 4, false, true, 4, 1>>(a, stream);
    return launch_ws_upd<WsCfg<64, 16, 16, 4, false, true, 4, 1>>(a, stream);
}

int dispatch_fd(FdArgs& a, hipStream_t stream) {
    // A/B testing (segnb_tune / environment): "fprop_dma" = 0 disables this path, "fprop_dma_cfg" forces a configuration
    int cfg = segnb_knob_fprop_dma_cfg();
    if (cfg < 0) {
        // measured per ZF_UNET layer at bs=32 (tools/layer_bench.py --cfg): 64-channel output tiles win on every level
        // from 14x14 to 112x112 -- twice the tiles of a 128-channel form (measured, since removed) on layers that have
        // only 200-800 of them;
        // 16 x 16 pixel tiles tie or beat 8 x 32 except for the widest data gradients.  Outputs of <= 32 channels
        // (half of every tile padding) and 7x7 images stay with fprop_s1 / the general kernel.
        if (a.Co <= 32) return NOT_HANDLED;
        // 7 x 7 images (the deepest ZF_UNET level): tall-image tiles of 36 x 7 virtual pixels -- 8 row tiles x 16 channel
        // tiles = 128 blocks of 144 taps each beat the general kernel's 400 tiles of 64 x 64 (76 -> ~45 us for 1024 -> 1024)
        if (a.W <= 8) {
            if (a.W != 7 || a.H != 7) return NOT_HANDLED;
            return launch_ws<WsCfg<64, 36, 7, 4, true, false>>(a, stream);      // odd halo pitch: 32x32x16 form
        }
        // 8 x 32 or 16 x 16 pixel tiles: whichever needs fewer rounds of (equal) tiles on the persistent blocks --
        // re-measured with the final kernel, the round count decides every case (e.g. 128 -> 384 @56x56: 11 vs 12
        // rounds, 104 vs 122 us; 64 -> 192 @112x112: 21 vs 19 rounds, 133 vs 116 us); ties go to 16 x 16
        const int ntl = (a.Co + 63) / 64;
        int gm = segnb_knob_conv_cus() / ntl;
        if (gm < 1) gm = 1;
        const long long it0 = (long long)a.N * ((a.H + 7) / 8) * ((a.W + 31) / 32);
        const long long it1 = (long long)a.N * ((a.H + 15) / 16) * ((a.W + 15) / 16);
        cfg = (it0 + gm - 1) / gm < (it1 + gm - 1) / gm ? 0 : 1;
        if (a.bn_y != nullptr) cfg = 1;      // (activation mask: the 16 x 16 tile form has the 2 KB of LDS left for the mask bytes)
    }
    bool row_major_taps = true;          // the 16x16x32 form indexes its address tables by (t / 3, t % 3)
    for (int t = 0; t < 9; ++t) row_major_taps = row_major_taps && a.dw[t] == a.dw[t % 3] && a.dh[t] == a.dh[3 * (t / 3)];
    if (segnb_knob_fprop_mf16() && row_major_taps) {
        switch (cfg) {
            case 0: return launch_ws<WsCfg<64, 8, 32, 4, false, true>>(a, stream);
            case 1: return launch_ws<WsCfg<64, 16, 16, 4, false, true>>(a, stream);
            case 2: return launch_ws<WsCfg<64, 36, 7, 4, true, false>>(a, stream);
            default: return NOT_HANDLED;
        }
    }
    switch (cfg) {
        case 0: return launch_ws<WsCfg<64, 8, 32, 4, false, false>>(a, stream);
        case 1: return launch_ws<WsCfg<64, 16, 16, 4, false, false>>(a, stream);
        case 2: return launch_ws<WsCfg<64, 36, 7, 4, true, false>>(a, stream);
        default: return NOT_HANDLED;
    }
}

}  // namespace

// out[n, 2 Y + py, 2 X + px, :] += sum_{a, b < 2} in[n, Y + py - 1 + a, X + px - 1 + b, :] . W[phase][:, tap, :]  for the four
// phases (py, px); wpacked = [4][CoW][4][Ci] (phase-major, the tap lists of segnb.convplan.convt_fwd(4, 2, 1)); statistics of
// the sums.  1 = launched, 0 = not served
int segnb_fprop_upf_try(int N, int H, int W, int Ci, int ld_in, const void* in, unsigned in_bytes, const void* wpacked,
                        unsigned w_bytes, int Co, int CoW, void* out, int ld_out, double* stats, hipStream_t stream,
                        const float* bias, int bias_n, int no_prev, int ep_act, float ep_slope) {
    if (!segnb_knob_fprop_dma() || !segnb_knob_fprop_upd()) return 0;
    if (Ci % 64 != 0 || Ci < 128 || (Co <= 32 && !no_prev) || W < 12) return 0;
    FdArgs a;
    a.x = (const bf16_t*)in;
    a.w = (const bf16_t*)wpacked;
    a.x_bytes = in_bytes;
    a.w_bytes = w_bytes;
    a.bias = bias;
    a.bias_n = bias_n;
    a.no_prev = no_prev;
    a.out = (bf16_t*)out;
    a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.Hi = H; a.Wi = W;
    a.Ci = Ci; a.Co = Co; a.ld_x = ld_in; a.ld_out = ld_out;
    a.CoW = CoW;
    a.Ktot = 4 * Ci;
    const long long ob = (((long long)N * 4 * H * W - 1) * ld_out + Co) * 2;
    if (ob >= (1ll << 31)) return 0;
    a.out_bytes = (unsigned)ob;
    a.dhmin = a.dwmin = 0;
    for (int t = 0; t < 9; ++t) a.dh[t] = a.dw[t] = 0;
    a.dh[2] = a.dh[3] = 1;
    a.dw[1] = a.dw[3] = 1;
    a.dbg = 0;
    a.u = nullptr;
    a.up_out = nullptr;
    a.bn_y = nullptr;
    a.ep_act = ep_act;          // (UPF: applied in the accumulator staging of this instantiation, not the EP one)
    a.ep_coef = nullptr;
    a.ep_slope = ep_slope;
    const int ntl = 4 * ((Co + 63) / 64);
    int gm = segnb_knob_conv_cus() / ntl;
    if (gm < 1) gm = 1;
    const long long it0 = (long long)N * ((H + 7) / 8) * ((W + 31) / 32);
    const long long it1 = (long long)N * ((H + 15) / 16) * ((W + 15) / 16);
    int cfg = segnb_knob_fprop_dma_cfg();
    if (cfg < 0) cfg = (it0 + gm - 1) / gm < (it1 + gm - 1) / gm ? 0 : 1;
    const int rc = cfg == 0 ? launch_ws_upf<WsCfg<64, 8, 32, 4, false, true, 4, 2>>(a, stream)
                            : launch_ws_upf<WsCfg<64, 16, 16, 4, false, true, 4, 2>>(a, stream);
    if (rc == NOT_HANDLED) return 0;
    return rc ? rc : 1;
}

// 1 when the MASK instantiation serves the data gradient g (segnb_conv_fprop_actmask_ok): what segnb_fprop_dma_try + dispatch_fd accept
int segnb_fprop_dma_actmask_ok(const segnb_conv_geom* g) {
    if (!segnb_knob_fprop_dma() || !segnb_knob_fprop_mask() || !segnb_knob_fprop_mf16() || segnb_knob_fprop_dma_cfg() >= 2 ||
        getenv("SEGNB_FPROP_GENERAL") != nullptr)
        return 0;
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci % 64 != 0 || g->Co <= 32 || g->Co % 8 != 0 || g->Wo <= 8) return 0;
    int dhmin = g->dh[0], dwmin = g->dw[0];
    for (int t = 1; t < 9; ++t) {
        dhmin = g->dh[t] < dhmin ? g->dh[t] : dhmin;
        dwmin = g->dw[t] < dwmin ? g->dw[t] : dwmin;
    }
    for (int t = 0; t < 9; ++t) {        // row-major 3 x 3 window (either direction): the 16x16x32 form's address tables
        if (g->dh[t] - dhmin != g->dh[3 * (t / 3)] - dhmin || g->dw[t] - dwmin != g->dw[t % 3] - dwmin) return 0;
        if (g->dh[t] - dhmin > 2 || g->dw[t] - dwmin > 2) return 0;
    }
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        const int k = (g->dh[t] - dhmin) * 3 + (g->dw[t] - dwmin);
        if (seen[k]) return 0;
        seen[k] = true;
    }
    const long long inb = (((long long)g->N * g->Hi * g->Wi - 1) * g->ld_in + g->Ci) * 2;
    const long long ob = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
    return inb < (1ll << 31) && ob < (1ll << 31) ? 1 : 0;
}

extern "C" int segnb_conv_fprop_upd_ok(const segnb_conv_geom* g, int dtype) {
    if (g == nullptr || dtype != SEGNB_BF16) return 0;
    if (!segnb_knob_fprop_dma() || !segnb_knob_fprop_upd() || getenv("SEGNB_FPROP_GENERAL") != nullptr) return 0;
    const long long ob = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
    return upd_geometry(g) && ob < (1ll << 31) ? 1 : 0;
}

// timing builds: copy the stamps of the last stamped launch to the host (3 roles x 256 steps x 4 clocks)
int segnb_fprop_dma_read_stamps(unsigned long long* host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 3 * 256 * 4);
}

// 1 = handled, 0 = not applicable (caller falls through to fprop_s1 / the general gather kernel), else error
int segnb_fprop_dma_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked,
                        unsigned w_bytes, const float* bias, int bias_n, void* out, double* stats,
                        hipStream_t stream, const segnb_act_epilogue* ep, const segnb_upcat_src* uc,
                        const segnb_bn_reduce_epilogue* bn) {
    if (!segnb_knob_fprop_dma()) return 0;
    // (fused BatchNorm-backward REDUCTIONS: fprop_rw.hip / fprop_roll.hip; here only the activation mask of a layer without
    // BatchNorm -- coef NULL -- on a plain data gradient: the MASK instantiation)
    if (bn != nullptr && (bn->coef != nullptr || bn->y == nullptr || bn->sums == nullptr || ep != nullptr || uc != nullptr ||
                          stats != nullptr || bias != nullptr || g->ntaps != 9 || !segnb_knob_fprop_mask()))
        return 0;
    if (g->ntaps == 16 && ep == nullptr && stats == nullptr && bias == nullptr && segnb_knob_fprop_upd()) {
        FdArgs a;
        a.x = (const bf16_t*)in;
        a.w = (const bf16_t*)wpacked;
        a.x_bytes = in_bytes;
        a.w_bytes = w_bytes;
        a.bias = nullptr;
        a.bias_n = 0;
        a.out = (bf16_t*)out;
        a.stats = nullptr;
        a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.Hi = g->Hi; a.Wi = g->Wi;
        a.Ci = g->Ci; a.Co = g->Co; a.ld_x = g->ld_in; a.ld_out = g->ld_out;
        const long long ob = (((long long)g->N * g->Ho * g->Wo - 1) * g->ld_out + g->Co) * 2;
        if (ob >= (1ll << 31)) return 0;
        a.out_bytes = (unsigned)ob;
        a.dbg = 0;
        a.u = nullptr;
        a.up_out = nullptr;
    a.no_prev = 0;
        a.bn_y = nullptr;
        a.ep_act = -1;
        a.ep_coef = nullptr;
        a.ep_slope = 0.f;
        const int rc = try_upd(g, a, stream);
        if (rc == NOT_HANDLED) return 0;
        return rc ? rc : 1;
    }
    if (g->ntaps != 9 || g->in_step != 1 || g->out_step != 1 || g->oh0 != 0 || g->ow0 != 0) return 0;
    if (g->QH != g->Ho || g->QW != g->Wo || g->Ci % 64 != 0) return 0;
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
    a.u = nullptr;
    if (uc != nullptr) {
        // virtual concat: in = the skip tensor (channels Cu.. of the logical input), uc->u = the low-resolution tensor
        if (uc->Cu % 64 != 0 || uc->Cu <= 0 || uc->Cu >= g->Ci || (g->Hi & 1) || (g->Wi & 1) || g->Wo <= 8) return 0;
        a.u = (const bf16_t*)uc->u;
        a.ld_u = uc->ld_u;
        a.NCHU = uc->Cu / 64;
        a.Hu = g->Hi / 2;
        a.Wu = g->Wi / 2;
        const long long ub = (((long long)g->N * a.Hu * a.Wu - 1) * uc->ld_u + uc->Cu) * 2;
        if (ub >= (1ll << 31)) return 0;
        a.u_bytes = (unsigned)ub;
    }
    a.bn_y = nullptr;
    a.bn_act = SEGNB_ACT_NONE;
    a.bn_slope = 0.f;
    if (bn != nullptr) {
        const long long yb = (((long long)g->N * g->Ho * g->Wo - 1) * bn->ld_y + g->Co) * 2;
        if (bn->ld_y % 8 != 0 || yb >= (1ll << 31)) return 0;      // (16-byte loads of 8-channel chunks)
        a.bn_y = (const bf16_t*)bn->y;
        a.bn_y_bytes = (unsigned)yb;
        a.bn_ld = bn->ld_y;
        a.bn_coef = nullptr;
        a.bn_sums = bn->sums;
        a.bn_act = bn->act;
        a.bn_slope = bn->slope;
        a.stats = bn->sums;          // [REPL][2][Co]: slot 0 = sum dz (the bias gradient's source), slot 1 = sum dz^2 (cleared with it)
    }
    a.ep_act = ep != nullptr ? ep->act : -1;
    a.ep_coef = ep != nullptr ? ep->coef : nullptr;
    a.ep_slope = ep != nullptr ? ep->slope : 0.f;
    const int rc = dispatch_fd(a, stream);
    if (rc == NOT_HANDLED) return 0;
    return rc ? rc : 1;
}
