// Stride-1 3x3 convolution forward / data-gradient (bf16), gfx950 -- the direct-to-LDS pipeline behind
// segnb_conv_fprop for layers with Ci % 64 == 0 (aten::convolution / convolution_backward(input) of
// lib/models/zf_unet.py:8 and the other 3x3 stride-1 convolutions of lib/models/*).
//
// fprop_s1.hip stages both operands through registers with one or two barriers per tap and exposes a global-load
// round trip at every one of them (PMC: 35-55 % of wave time parked in s_waitcnt / s_barrier, MFMA pipe 12-25 % busy).
// Here NOTHING passes through registers on its way to LDS:
//   * the input halo tile (R+2) x (WT+2) pixels x 64 channels and the weight tile BN x 64 channels of ONE tap are
//     fetched by `buffer_load_dwordx4 ... lds` (LDS-DMA): one wave-instruction moves 8 rows x 128 B = whole cache
//     lines, out-of-image pixels / out-of-range channels are the descriptor's range check (zeros land in LDS);
//   * rows are 128 B with no padding (the DMA destination is lane-linear), bank conflicts are removed by an XOR
//     swizzle applied to the SOURCE address: the 16-byte slot q of row p holds channel chunk q ^ ((p >> 1) & 7), so
//     the sixteen lanes of a ds_read_b128 group (rows distinct mod 16) hit sixteen distinct bank quads;
//   * weights run through a ring of NB one-tap stages fetched three taps ahead, the halo tile is double buffered
//     and fetched one 64-channel chunk ahead (across tile boundaries: the block is persistent and the stream of
//     (tile, chunk, tap) steps never drains); every step ends with a COUNTED s_waitcnt vmcnt(N) -- the fetches of
//     the last two steps stay in flight -- and one raw s_barrier;
//   * accumulators are TRANSPOSED (MFMA A operand = weights, B operand = pixels): a lane ends up with four
//     consecutive channels of one pixel per register quad, so the epilogue stages bf16 quads with ds_write_b64, and
//     the BatchNorm statistics are taken by the threads of the coalesced store pass (fixed channel chunk per thread).
#include "common.h"

#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct FdArgs {
    const bf16_t* x;
    const bf16_t* w;
    unsigned x_bytes, w_bytes;
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
};

constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ int xcd_remap_fd(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7, j = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

__device__ __forceinline__ i32x4_t make_rsrc4(const void* base, unsigned bytes) {
    const unsigned long long pa = (unsigned long long)base;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// one LDS-DMA piece: lane l's 16 bytes at (descriptor base + voff + soff) land at LDS byte lds_dst + 16*l.
// hipcc does not count these (no s_waitcnt of its own for them): completion is the counted vmcnt of step_sync.
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, const i32x4_t& rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// LDS-only barrier: __syncthreads() would also wait for the global stores of the epilogue and for every DMA in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void step_sync() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BN_, int R_, int WT_, int NW_, int WAVES_M_>
struct FdCfg {
    static constexpr int BN = BN_, R = R_, WT = WT_, NW = NW_, WAVES_M = WAVES_M_;
    static constexpr int NB = 4;                         // weight ring stages (one tap each)
    static constexpr int NT = NW * 64;
    static constexpr int BM = R * WT;
    static constexpr int WAVES_N = NW / WAVES_M;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int XR = R + 2, XC = WT + 2, NPIX = XR * XC;
    static constexpr int APIECES = (NPIX + 7) / 8;       // 1-KiB pieces of 8 halo pixels x 128 B
    static constexpr int A_BYTES = APIECES * 1024;
    static constexpr int APW = (APIECES + NW - 1) / NW;  // pieces per wave and chunk
    static constexpr int A_STEPS = 6;                    // issued during taps 0..5 of the previous chunk
    static constexpr int APS = (APW + A_STEPS - 1) / A_STEPS;
    static constexpr int BPIECES = BN / 8;
    static constexpr int B_STAGE = BN * 128;
    static constexpr int BPW = BPIECES / NW;
    static constexpr int OUT_ROW = BN * 2 + 16;
    static constexpr int NPASS = (BM * OUT_ROW + A_BYTES - 1) / A_BYTES;     // epilogue passes through one A buffer
    static constexpr int EPR = ((BM / 32 + NPASS - 1) / NPASS) * 32;         // rows per pass (whole MFMA row tiles)
    static constexpr int OC = BN / 8;
    static constexpr int OFF_B = 2 * A_BYTES;
    static constexpr int OFF_DUMMY = OFF_B + NB * B_STAGE;
    static constexpr int OFF_PIX = OFF_DUMMY + 1024;
    static constexpr int OFF_BIAS = OFF_PIX + BM * 4;
    static constexpr int SMEM = OFF_BIAS + BN * 4;
    static constexpr int RED_BYTES = NT * 16 * 8;        // final statistics reduction (after the pipeline drained)
    static_assert(WM % 32 == 0 && WN % 32 == 0 && WAVES_M * WAVES_N == NW, "wave tiling");
    static_assert(BPIECES % NW == 0, "weight pieces per wave");
    static_assert(EPR * OUT_ROW <= A_BYTES, "epilogue staging fits one halo buffer");
    static_assert(NT % OC == 0, "fixed channel chunk per store thread");
    static_assert(RED_BYTES <= OFF_DUMMY, "statistics reduction scratch");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <class C>
__global__ __launch_bounds__(C::NT) void conv_fprop_dma_kernel(const FdArgs a) {
    constexpr int BN = C::BN, R = C::R, WT = C::WT, NW = C::NW, BM = C::BM, TM = C::TM, TN = C::TN, XC = C::XC;
    constexpr int NT = C::NT, NB = C::NB, APW = C::APW, APS = C::APS, BPW = C::BPW, OC = C::OC;
    constexpr int OUT_ROW = C::OUT_ROW;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    int* sPix = reinterpret_cast<int*>(smem + C::OFF_PIX);
    float* sBias = reinterpret_cast<float*>(smem + C::OFF_BIAS);
    const unsigned lds0 = (unsigned)(size_t)smem;       // LDS byte address of the array (DMA destinations)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

    const int L = xcd_remap_fd(blockIdx.x, gridDim.x);
    const int nt = L % a.NTL, gq = L / a.NTL;
    const int n_base = nt * BN;

    const i32x4_t rs_x = make_rsrc4(a.x, a.x_bytes);
    const i32x4_t rs_w = make_rsrc4(a.w, a.w_bytes);

    for (int c = tid; c < BN; c += NT) {
        const int co = n_base + c;
        sBias[c] = (a.bias != nullptr && co < a.bias_n) ? a.bias[co] : 0.f;
    }

    // ---- per-lane constants ---------------------------------------------------------------------------------
    // weight pieces of this wave: piece = wave * BPW + pb covers rows 8*piece .. +7 of the BN x 128 B stage
    unsigned b_voff[BPW];
#pragma unroll
    for (int pb = 0; pb < BPW; ++pb) {
        const int row = (wave * BPW + pb) * 8 + (lane >> 3);
        const int q = lane & 7;
        const int co = n_base + row;
        b_voff[pb] = co < a.Co ? (unsigned)co * (unsigned)a.Ktot * 2u + (unsigned)((q ^ ((row >> 1) & 7)) * 16) : OOB;
    }
    // fragment read offsets.  Row p of a tile, chunk k (16 B) lives at p*128 + ((k ^ ((p>>1)&7)) * 16); with
    // k = 2*kk + h:  (p*128 | ((p & 12) << 3) | (((h ^ (p >> 1)) & 1) << 4)) ^ (kk << 5)
    int b_rd[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * C::WN + 32 * j + r;
        b_rd[j] = C::OFF_B + row * 128 + ((row & 12) << 3) + (((h ^ (row >> 1)) & 1) << 4);
    }
    int a_pix[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm * C::WM + 32 * i + r;
        a_pix[i] = (m / WT) * XC + (m % WT);
    }
    int tsh[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tsh[t] = a.dh[t] * XC + a.dw[t];

    // halo pieces of this wave: piece = wave + NW * pa; per-lane source offset of the tile being FETCHED
    unsigned a_voff[APW];
    auto set_fetch_tile = [&](int it) {
        const bool live = it < a.IT;
        const int n = it / (a.HB * a.WB);
        const int rem = it - n * (a.HB * a.WB);
        const int hb = rem / a.WB, wb = rem - hb * a.WB;
        const int h0 = hb * R + a.dhmin, w0 = wb * WT + a.dwmin;
#pragma unroll
        for (int pa = 0; pa < APW; ++pa) {
            const int pix = (wave + NW * pa) * 8 + (lane >> 3);
            const int q = lane & 7;
            const int xr = pix / XC, xc = pix - xr * XC;
            const int hi = h0 + xr, wi = w0 + xc;
            const bool ok = live && pix < C::NPIX && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
            a_voff[pa] = ok ? (unsigned)((n * a.Hi + hi) * a.Wi + wi) * (unsigned)a.ld_x * 2u +
                                  (unsigned)((q ^ ((pix >> 1) & 7)) * 16)
                            : OOB;
        }
    };
    // issue this wave's halo pieces [p0, p1) of channel chunk c into buffer `buf`
    auto fetch_a = [&](int p0, int p1, int c, int buf) {
#pragma unroll
        for (int pa = 0; pa < APW; ++pa) {
            if (pa >= p0 && pa < p1) {
                const int piece = wave + NW * pa;
                const unsigned dst = piece < C::APIECES ? lds0 + buf * C::A_BYTES + piece * 1024 : lds0 + C::OFF_DUMMY;
                dma16(dst, a_voff[pa], rs_x, (unsigned)c * 128u);
            }
        }
    };
    // issue this wave's pieces of the weight tile (chunk c, tap t) into ring stage `stage`
    auto fetch_b = [&](int c, int t, int stage) {
        const unsigned soff = (unsigned)(t * a.Ci + c * 64) * 2u;
#pragma unroll
        for (int pb = 0; pb < BPW; ++pb)
            dma16(lds0 + C::OFF_B + stage * C::B_STAGE + (wave * BPW + pb) * 1024, b_voff[pb], rs_w, soff);
    };

    // store-pass threads keep one 8-channel chunk: statistics of the stored values (fp32 per tile, fp64 across tiles)
    double d1[8], d2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d1[e] = d2[e] = 0.0;

    // ---- pipeline prologue: halo chunk 0 of the first tile, weight taps 0..2 --------------------------------
    int it = gq;
    set_fetch_tile(it);
    fetch_a(0, APW, 0, 0);
    {
        // taps 0..2 of chunk 0 (NCH >= 1, 9 taps per chunk)
        fetch_b(0, 0, 0);
        fetch_b(0, 1, 1);
        fetch_b(0, 2, 2);
    }
    int cg = 0;                         // chunks done by this block: halo buffer = cg & 1, ring stage of tap t = (cg + t) & 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (; it < a.IT; it += a.GM) {
        // output pixel index per tile row (-1 = outside)
        {
            const int n = it / (a.HB * a.WB);
            const int rem = it - n * (a.HB * a.WB);
            const int hb = rem / a.WB, wb = rem - hb * a.WB;
            for (int rr = tid; rr < BM; rr += NT) {
                const int ho = hb * R + rr / WT, wo = wb * WT + rr % WT;
                sPix[rr] = (ho < a.H && wo < a.W) ? (n * a.H + ho) * a.W + wo : -1;
            }
        }
        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int c = 0; c < a.NCH; ++c, ++cg) {
            const bool last = c + 1 == a.NCH;
            const int cn = last ? 0 : c + 1;                    // chunk whose halo tile is fetched during this one
            if (last) set_fetch_tile(it + a.GM);
            const int abuf = cg & 1;
            const int a_base = abuf * C::A_BYTES;
            static_for<9>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                // ---- fetches of this step: weights three taps ahead, a slice of the next halo chunk
                {
                    constexpr int tf = t + 3 < 9 ? t + 3 : t + 3 - 9;
                    const int cf = t + 3 < 9 ? c : cn;
                    fetch_b(cf, tf, (cg + t + 3) & (NB - 1));
                    if constexpr (t < C::A_STEPS) fetch_a(t * APS, (t + 1) * APS, cn, abuf ^ 1);
                }
                // ---- tap t of chunk c
                const int bstage = ((cg + t) & (NB - 1)) * C::B_STAGE;
                int av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int p = a_pix[i] + tsh[t];
                    av[i] = a_base + (p << 7) + ((p & 12) << 3) + (((h ^ (p >> 1)) & 1) << 4);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = b_rd[j] + bstage;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    bf16x8_t af[TM], bfr[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(smem + (av[i] ^ (kk << 5)));
#pragma unroll
                    for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(smem + (bv[j] ^ (kk << 5)));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                // everything but the fetches of the last two steps has landed; then every wave is past its reads
                constexpr int NA_T = (t < C::A_STEPS ? APS : 0) + ((t >= 1 && t - 1 < C::A_STEPS) ? APS : 0);
                step_sync<2 * BPW + NA_T>();
            });
        }

        // ---- epilogue: the halo buffer of the last chunk is free (the other one is being filled) -------------
        unsigned char* sOut = smem + ((cg - 1) & 1) * C::A_BYTES;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int pass = 0; pass < C::NPASS; ++pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * C::WM + 32 * i + r;
                if ((wm * C::WM + 32 * i) / C::EPR == pass) {             // wave-uniform: EPR is a multiple of 32
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = wn * C::WN + 32 * j + 8 * g + 4 * h;
                            const float4 bv4 = *reinterpret_cast<const float4*>(sBias + col);
                            uint2 pk;
                            pk.x = pack2bf(acc[i][j][4 * g + 0] + bv4.x, acc[i][j][4 * g + 1] + bv4.y);
                            pk.y = pack2bf(acc[i][j][4 * g + 2] + bv4.z, acc[i][j][4 * g + 3] + bv4.w);
                            *reinterpret_cast<uint2*>(sOut + (row - pass * C::EPR) * OUT_ROW + col * 2) = pk;
                        }
                    }
                }
            }
            lds_barrier();
            const int cc = tid % OC;
            const int co = n_base + cc * 8;
            for (int rl = tid / OC; rl < C::EPR; rl += NT / OC) {
                const int row = pass * C::EPR + rl;
                if (row < BM) {
                    const int opix = sPix[row];
                    if (opix >= 0 && co < a.Co) {
                        const uint4 v = *reinterpret_cast<const uint4*>(sOut + rl * OUT_ROW + cc * 16);
                        *reinterpret_cast<uint4*>(a.out + (long long)opix * a.ld_out + co) = v;
                        if (a.stats != nullptr) {
                            float f[8];
                            f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
                            f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
                            f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
                            f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                s1[e] += f[e];
                                s2[e] += f[e] * f[e];
                            }
                        }
                    }
                }
            }
            lds_barrier();
        }
        if (a.stats != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                d1[e] += (double)s1[e];
                d2[e] += (double)s2[e];
            }
        }
    }

    // ---- statistics: fixed-order block reduction, one fp64 atomic per channel and block -----------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // look-ahead fetches of the (absent) next tile
    __syncthreads();
    if (a.stats != nullptr) {
        double* red = reinterpret_cast<double*>(smem);    // [NT][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid * 16 + e] = d1[e];
            red[tid * 16 + 8 + e] = d2[e];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid - which * BN;
            const int cc = col >> 3, e = col & 7;
            double s = 0.0;
            for (int k = 0; k < NT / OC; ++k) s += red[(k * OC + cc) * 16 + which * 8 + e];
            const int co = n_base + col;
            if (co < a.Co)
                atomicAdd(&a.stats[((long long)(blockIdx.x % SEGNB_STAT_REPLICAS) * 2 + which) * a.Co + co], s);
        }
    }
}

template <class C>
int launch_fd(FdArgs& a, hipStream_t stream) {
    static int attr_rc = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fprop_dma_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) segnb_set_error("fprop_dma hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }();
    if (attr_rc) return attr_rc;
    a.HB = (a.H + C::R - 1) / C::R;
    a.WB = (a.W + C::WT - 1) / C::WT;
    a.IT = a.N * a.HB * a.WB;
    a.NTL = (a.Co + C::BN - 1) / C::BN;
    a.NCH = a.Ci / 64;
    int per_cu = (160 * 1024) / C::SMEM;
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    int gm = (segnb_num_cus() * per_cu) / a.NTL;
    if (gm < 1) gm = 1;
    if (gm > a.IT) gm = a.IT;
    a.GM = gm;
    hipLaunchKernelGGL((conv_fprop_dma_kernel<C>), dim3(a.GM * a.NTL), dim3(C::NT), C::SMEM, stream, a);
    return 0;
}

constexpr int NOT_HANDLED = -12345;

int dispatch_fd(FdArgs& a, hipStream_t stream) {
    // A/B testing (segnb_tune / environment): "fprop_dma" = 0 disables this path, "fprop_dma_cfg" forces a configuration
    int cfg = segnb_knob_fprop_dma_cfg();
    if (cfg < 0) {
        if (a.W >= 24)
            cfg = a.Co > 64 ? 0 : 2;
        else if (a.W > 8)
            cfg = a.Co > 64 ? 4 : 6;
        else
            return NOT_HANDLED;
    }
    switch (cfg) {
        case 0: return launch_fd<FdCfg<128, 8, 32, 8, 4>>(a, stream);
        case 1: return launch_fd<FdCfg<128, 4, 32, 4, 2>>(a, stream);
        case 2: return launch_fd<FdCfg<64, 8, 32, 8, 4>>(a, stream);
        case 3: return launch_fd<FdCfg<64, 4, 32, 4, 2>>(a, stream);
        case 4: return launch_fd<FdCfg<128, 16, 16, 8, 4>>(a, stream);
        case 5: return launch_fd<FdCfg<128, 8, 16, 4, 2>>(a, stream);
        case 6: return launch_fd<FdCfg<64, 16, 16, 8, 4>>(a, stream);
        case 7: return launch_fd<FdCfg<64, 8, 16, 4, 2>>(a, stream);
        default: return NOT_HANDLED;
    }
}

}  // namespace

// 1 = handled, 0 = not applicable (caller falls through to fprop_s1 / the general gather kernel), else error
int segnb_fprop_dma_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked,
                        unsigned w_bytes, const float* bias, int bias_n, void* out, double* stats,
                        hipStream_t stream) {
    if (!segnb_knob_fprop_dma()) return 0;
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
    a.dhmin = dhmin; a.dwmin = dwmin;
    for (int t = 0; t < 9; ++t) {
        a.dh[t] = g->dh[t] - dhmin;
        a.dw[t] = g->dw[t] - dwmin;
    }
    const int rc = dispatch_fd(a, stream);
    if (rc == NOT_HANDLED) return 0;
    return rc ? rc : 1;
}
