// BatchNorm / activation / Dropout2d / MaxPool2d(2) / nearest-x2 Upsample, fused, NHWC, gfx950.
//
// Replaces nn.BatchNorm2d + nn.ReLU (lib/models/zf_unet.py:9-10,15-16), nn.Dropout2d (:25,31),
// nn.MaxPool2d(2) (:41), nn.Upsample(scale_factor=2) (:42) and torch.cat (:78-90, via `ld` slices)
// and their autograd backward; the 4-phase split mirrors inplace_abn's
// mean_var / forward / edz_eydz / backward (lib/modules/abn/functions.py:81,94,112,118).
//
// All kernels are HBM-bound streams: 16-byte vectors (8 channels per thread), one thread column per
// 8-channel chunk so per-channel sums stay in registers, wave/LDS reduction, fp64 global atomics.
#include "common.h"

namespace {

constexpr int NTHR = 256;
constexpr int REPL = SEGNB_STAT_REPLICAS;

struct EwShape {
    int N, H, W, Cp;
    int CPP;  // 8-channel chunks per pixel
    int CT;   // chunk columns per block (power of two <= 32)
    int PY;   // pixel rows per block = 256 / CT
};

static EwShape make_shape(int N, int H, int W, int Cp) {
    EwShape s;
    s.N = N; s.H = H; s.W = W; s.Cp = Cp;
    s.CPP = Cp / 8;
    int ct = 1;
    while (ct < s.CPP && ct < 32) ct <<= 1;
    s.CT = ct;
    s.PY = NTHR / ct;
    return s;
}

static dim3 make_grid(const EwShape& s, long long items, int max_blocks = 2048) {
    // few, fat blocks: every block ends with one atomic per channel into the same addresses, and
    // same-address atomics serialise (MI355X_MICROARCH: ~14x slower than spread ones)
    const int gy = ceil_div(s.CPP, s.CT);
    long long gx = (items + s.PY - 1) / s.PY;
    long long cap = max_blocks / gy;
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    const long long want = (items + 4ll * s.PY - 1) / (4ll * s.PY);     // >= 4 items per thread
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)gy);
}

// Branch-free activations: none / ReLU / LeakyReLU are all "negative side scaled by s" with s = 1 / 0 / slope (a
// switch on `act` per element compiled to two scalar branches per element -- 153 branches in the pooled backward
// kernel -- with the loads fenced between them).  "+ 0.f" turns ReLU's -0.0 into torch's +0.0.
__device__ __forceinline__ float act_neg_scale(int act, float slope) {
    return act == SEGNB_ACT_RELU ? 0.f : (act == SEGNB_ACT_LEAKY ? slope : 1.f);
}
__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
    return (z > 0.f ? z : z * act_neg_scale(act, slope)) + 0.f;
}
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
    return z > 0.f ? 1.f : act_neg_scale(act, slope);
}

// ------------------------------------------------------------------------------------------------
// per-channel forward coefficients from the replicated statistics (training) or the running statistics (eval);
// `update` = this thread owns the channel's side effects (running statistics)
struct BnFwdParams {
    const double* stats;        // [REPL][2][Cp]; read-only in the fused kernel
    double count;
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* running_mean;
    float* running_var;
    long long* nbt;
    int C, training;
    float* coef;                // [4][Cp] out
    double* zero_buf;           // fused only: the BACKWARD accumulators of this layer, cleared for this step
    int keep_stats;             // segnb_bn_finalize_keep: the statistics are left for the backward to clear (fused protocol)
    int stats_ld;               // channel stride between the [REPL][2] rows of `stats` (0: Cp) -- a channel RANGE of a wider table
};

__device__ __forceinline__ void bn_fwd_coef(const BnFwdParams& p, int Cp, int c, bool update, float& scale,
                                            float& shift, float& mean, float& invstd) {
    scale = shift = mean = invstd = 0.f;
    if (c >= p.C) return;
    double mu, var;
    if (p.training) {
        double v1[REPL], v2[REPL];
        const int SL = p.stats_ld > 0 ? p.stats_ld : Cp;
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            v1[rp] = p.stats[(rp * 2) * SL + c];
            v2[rp] = p.stats[(rp * 2 + 1) * SL + c];
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            s1 += v1[rp];
            s2 += v2[rp];
        }
        mu = s1 / p.count;
        var = s2 / p.count - mu * mu;
        if (var < 0.0) var = 0.0;
        if (update && p.running_mean != nullptr) {
            const double unb = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
            p.running_mean[c] = (float)((1.0 - (double)p.momentum) * (double)p.running_mean[c] + (double)p.momentum * mu);
            p.running_var[c] = (float)((1.0 - (double)p.momentum) * (double)p.running_var[c] + (double)p.momentum * unb);
        }
    } else {
        mu = (double)p.running_mean[c];
        var = (double)p.running_var[c];
    }
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float g = p.gamma != nullptr ? p.gamma[c] : 1.f;
    const float b = p.beta != nullptr ? p.beta[c] : 0.f;
    scale = g * invstd;
    shift = b;   // z = (y - mean) * scale + beta : no cancellation between mean*scale and beta
}

__global__ void bn_finalize_kernel(const BnFwdParams p, int Cp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && p.training && p.nbt != nullptr) p.nbt[0] += 1;
    if (c >= Cp) return;
    float scale, shift, mean, invstd;
    bn_fwd_coef(p, Cp, c, true, scale, shift, mean, invstd);
    if (p.zero_buf != nullptr) {            // (segnb_bn_finalize_keep: this layer's backward accumulators)
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            p.zero_buf[(rp * 2) * Cp + c] = 0.0;
            p.zero_buf[(rp * 2 + 1) * Cp + c] = 0.0;
        }
    }
    if (p.training && c < p.C && !p.keep_stats) {            // the stand-alone finalize CONSUMES the statistics
        double* st = const_cast<double*>(p.stats);
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            st[(rp * 2) * Cp + c] = 0.0;
            st[(rp * 2 + 1) * Cp + c] = 0.0;
        }
    }
    p.coef[c] = scale;
    p.coef[Cp + c] = shift;
    p.coef[2 * Cp + c] = mean;
    p.coef[3 * Cp + c] = invstd;
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// block-level per-channel reduction helper: thread (tx,ty) holds v[8] for chunk tx; result summed over
// ty and atomically added (fp64) to dst[chan] for chan < Cp.  Deterministic: butterfly over the lanes of a
// wave that share tx, one LDS slot per wave, fixed-order fp64 sum over the waves (only the final fp64 atomics
// across blocks are unordered: 1e-16 relative).
constexpr int SRED_FLOATS = (NTHR / 64) * 2 * 32 * 8;
__device__ __forceinline__ void block_channel_sum2(float (&a)[8], float (&b)[8], const EwShape& s, int tx,
                                                   int chunk_base, double* __restrict__ dst0,
                                                   double* __restrict__ dst1, float* sred) {
    // sred: [NTHR/64][2][CT*8] floats in LDS
    for (int off = s.CT; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] += __shfl_xor(a[e], off);
            b[e] += __shfl_xor(b[e], off);
        }
    }
    const int nch = s.CT * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < s.CT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sred[wave * 2 * nch + tx * 8 + e] = a[e];
            sred[wave * 2 * nch + nch + tx * 8 + e] = b[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nch; i += NTHR) {
        const int which = i / nch, idx = i - which * nch;
        const int ch = chunk_base * 8 + idx;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NTHR / 64; ++w) t += (double)sred[w * 2 * nch + i];
        if (ch < s.Cp) atomicAdd((which == 0 ? dst0 : dst1) + ch, t);
    }
}

// raw 8-channel chunk as loaded (16 bytes of bf16 / 32 bytes of fp32): the conversion to fp32 is deferred to the first use,
// so that a thread can keep the loads of several pixels -- and its coefficient prologue -- in flight at once.  A pass over a
// small tensor is ONE dependent round trip long instead of one per pixel (4 pixels per thread: 8.5 -> ~6 us at 7x7 .. 28x28)
template <typename T>
struct Raw8;
template <>
struct Raw8<bf16_t> {
    uint4 u;
};
template <>
struct Raw8<float> {
    float4 a, b;
};
__device__ __forceinline__ void load_raw(const bf16_t* p, Raw8<bf16_t>& r) { r.u = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void load_raw(const float* p, Raw8<float>& r) {
    r.a = *reinterpret_cast<const float4*>(p);
    r.b = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void unpack_raw(const Raw8<bf16_t>& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.u.x << 16); v[1] = __uint_as_float(r.u.x & 0xffff0000u);
    v[2] = __uint_as_float(r.u.y << 16); v[3] = __uint_as_float(r.u.y & 0xffff0000u);
    v[4] = __uint_as_float(r.u.z << 16); v[5] = __uint_as_float(r.u.z & 0xffff0000u);
    v[6] = __uint_as_float(r.u.w << 16); v[7] = __uint_as_float(r.u.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack_raw(const Raw8<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w;
    v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
// v rounded to the storage type in place and stored (p may be NULL: rounding only).  bf16: one v_cvt_pk_bf16_f32 per channel
// pair gives the stored dword, the rounded values are its two halves
__device__ __forceinline__ void round_store8(float* p, float (&v)[8]) {
    if (p != nullptr) store8(p, v);
}
__device__ __forceinline__ void round_store8(bf16_t* p, float (&v)[8]) {
    uint4 u;
    u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
    u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    if (p != nullptr) *reinterpret_cast<uint4*>(p) = u;
}

// exact a / d for 0 <= a < 2^31 through a double reciprocal (8 instructions; the integer division sequence is ~35 and
// the pooled kernels are VALU-bound): the truncated product is off by at most one, the remainder says which way
struct FastDiv {
    int d;
    double inv;
    __device__ __forceinline__ explicit FastDiv(int d_) : d(d_), inv(1.0 / (double)d_) {}
    __device__ __forceinline__ int div(int a) const {
        const int q = (int)((double)a * inv);
        const int r = a - q * d;
        return q + (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    }
};

// the 1x1 classifier on the activated values of the pass (segnb_bn_fwd_fused_head): fp32 NCHW logits
struct HeadFwd {
    const float* w;     // [K][C]
    const float* b;     // [K] or NULL
    float* logits;      // [N][K][H][W]
    int K, C;
};

// HK > 0: the network's LAST activation pass also evaluates the 1x1 classifier (zf_unet.py:58,93) on the values it produces --
// the lanes that hold a pixel's channel chunks (CT consecutive lanes; the launcher requires CT == Cp / 8) add their partial dot
// products with a shuffle tree -- and `out` may be NULL: the activated tensor then never exists in memory
// SOUT: the pass also accumulates the per-channel sum / sum of squares of the values it WRITES into a statistics table (a channel
// range of a wider [REPL][2][ld] table): the BatchNorm layers that will read this tensor as part of a concat prefix find its
// statistics there and never pass over it again (FCDenseNet's dense blocks, tiramisu.py:9-44: every layer normalises the whole prefix)
struct OutStats {
    double* p;
    int ld;
};

template <typename T, bool POOL, bool RES, int HK = 0, bool SOUT = false>
__global__ __launch_bounds__(NTHR) void bn_act_fwd_kernel(const T* __restrict__ y, int ld_y, EwShape s,
                                                          const float* __restrict__ coef, int act, float slope,
                                                          const float* __restrict__ dropmul, T* __restrict__ out,
                                                          int ld_out, T* __restrict__ pool_out, int ld_pool,
                                                          T* __restrict__ up_out, int ld_up,
                                                          const T* __restrict__ res, int ld_res, const BnFwdParams fp,
                                                          const HeadFwd hd = HeadFwd{}, const OutStats so = OutStats{}) {
    static_assert(HK == 0 || !POOL, "the classifier reads the un-pooled activation");
    static_assert(!SOUT || !POOL, "output statistics: the plain walk only");
    __shared__ float sred_o[SOUT ? SRED_FLOATS : 1];
    float os1[8], os2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) os1[e] = os2[e] = 0.f;
    constexpr int U = 2;            // pixels per trip (no pooling)
    constexpr int UP = 2;           // row-pair items per trip (pooling)
    const int tx = threadIdx.x % s.CT, ty = threadIdx.x / s.CT;
    const int cc = blockIdx.y * s.CT + tx;
    const bool active = cc < s.CPP;
    const int c0 = active ? cc * 8 : 0;
    constexpr bool pooling = POOL;
    const int stride = gridDim.x * s.PY;
    // (check_ew guarantees 4*N*H*W < 2^31: 32-bit pixel indices)
    const int npix = s.N * s.H * s.W, hw = s.H * s.W;
    // pooling: a thread owns one pixel COLUMN of a row pair (2 pixels), so the loads / stores of a wave are
    // contiguous runs of a row; the 2x2 window maximum meets its horizontal partner (lane ^ CT: the row is walked
    // over an even padded width) through one cross-lane exchange.  (A thread per 2x2 window touched half of every
    // 128-byte line per instruction.)
    const int H2 = (s.H + 1) >> 1, We = 2 * ((s.W + 1) >> 1);
    const int Hp = s.H >> 1, Wp = s.W >> 1;
    const int nitems = s.N * H2 * We;
    const FastDiv d_hw(hw), d_w(s.W), d_img(H2 * We), d_we(We);
    int it0 = blockIdx.x * s.PY + ty;

    // ---- loads of one trip: everything that does not depend on the coefficients
    constexpr int NRAW = POOL ? 2 * UP : U;
    Raw8<T> ry[NRAW], rr[RES ? NRAW : 1];
    Raw8<float> rdm[POOL ? UP : U];
    int nn[U];
    auto issue_plain = [&](int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = i0 + u * stride;
            const int pc = pix < npix ? pix : i0;
            load_raw(y + (long long)pc * ld_y + c0, ry[u]);
            if constexpr (RES) load_raw(res + (long long)pc * ld_res + c0, rr[u]);
            nn[u] = (dropmul != nullptr || up_out != nullptr || HK > 0) ? d_hw.div(pc) : 0;
            if (dropmul != nullptr) load_raw(dropmul + nn[u] * s.Cp + c0, rdm[u]);
        }
    };
    // pooling: ry / rr [2 * i + r] = row r of item i; rdm[i]
    int pn[UP], ph2[UP], pw[UP];
    auto issue_pool = [&](int i0) {
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const int it = i0 + i * stride;
            const int ic = it < nitems ? it : i0;
            pn[i] = d_img.div(ic);
            const int rem = ic - pn[i] * (H2 * We);
            ph2[i] = d_we.div(rem);
            pw[i] = rem - ph2[i] * We;
            if (dropmul != nullptr) load_raw(dropmul + pn[i] * s.Cp + c0, rdm[i]);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int hh = 2 * ph2[i] + r;
                if (hh < s.H && pw[i] < s.W) {
                    const long long pix = ((long long)pn[i] * s.H + hh) * s.W + pw[i];
                    load_raw(y + pix * ld_y + c0, ry[2 * i + r]);
                    if constexpr (RES) load_raw(res + pix * ld_res + c0, rr[2 * i + r]);
                }
            }
        }
    };
    // the first trip's loads go out BEFORE the coefficient prologue (statistics -> fp64 arithmetic -> LDS -> barrier):
    // two latency chains overlapped instead of run back to back
    if (active) {
        if constexpr (!pooling) {
            if (it0 < npix) issue_plain(it0);
        } else {
            if (it0 < nitems) issue_pool(it0);
        }
    }

    float sc[8], sh[8], mu[8];
    if (fp.coef != nullptr) {
        // fused finalize: every block derives the coefficients of its CT*8 channels from the statistics (one
        // channel per thread, through LDS); the blocks of column 0 also publish them for the backward pass, update
        // the running statistics and clear this layer's BACKWARD accumulators.  The forward statistics are left
        // intact -- other blocks are still reading them -- and are cleared by segnb_bn_bwd_apply_fused.
        __shared__ float scoef[3][32 * 8];
        const int nch = s.CT * 8;
        if ((int)threadIdx.x < nch) {
            const int c = blockIdx.y * nch + threadIdx.x;
            float scale = 0.f, shift = 0.f, mean = 0.f, invstd = 0.f;
            if (c < s.Cp) {
                const bool owner = blockIdx.x == 0;
                bn_fwd_coef(fp, s.Cp, c, owner, scale, shift, mean, invstd);
                if (owner) {
                    fp.coef[c] = scale;
                    fp.coef[s.Cp + c] = shift;
                    fp.coef[2 * s.Cp + c] = mean;
                    fp.coef[3 * s.Cp + c] = invstd;
                    if (fp.zero_buf != nullptr) {
#pragma unroll
                        for (int rp = 0; rp < REPL; ++rp) {
                            fp.zero_buf[(rp * 2) * s.Cp + c] = 0.0;
                            fp.zero_buf[(rp * 2 + 1) * s.Cp + c] = 0.0;
                        }
                    }
                    if (c == 0 && fp.training && fp.nbt != nullptr) fp.nbt[0] += 1;
                }
            }
            scoef[0][threadIdx.x] = scale;
            scoef[1][threadIdx.x] = shift;
            scoef[2][threadIdx.x] = mean;
        }
        __syncthreads();
        if (!active && !SOUT) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = scoef[0][tx * 8 + e];
            sh[e] = scoef[1][tx * 8 + e];
            mu[e] = scoef[2][tx * 8 + e];
        }
    } else if (!active && !SOUT) {
        return;
    } else if (coef != nullptr) {   // six 16-byte loads in flight at once (element-wise selects serialise 24 dword loads)
        load8(coef + c0, sc);
        load8(coef + s.Cp + c0, sh);
        load8(coef + 2 * s.Cp + c0, mu);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = 1.f;
            sh[e] = 0.f;
            mu[e] = 0.f;
        }
    }

    float hw8[HK > 0 ? HK : 1][8];
    if constexpr (HK > 0) {
#pragma unroll
        for (int k = 0; k < HK; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) hw8[k][e] = (k < hd.K && c0 + e < hd.C) ? hd.w[k * hd.C + c0 + e] : 0.f;
    }
    if constexpr (!pooling) {
        // no pooling: consecutive threads take consecutive pixels, so every load / store instruction of a wave
        // covers one contiguous run
        if (SOUT && !active) it0 = npix;
        while (it0 < npix) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = it0 + u * stride;
                if (pix >= npix) break;
                float dm[8], v[8], rv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dm[e] = 1.f;
                if (dropmul != nullptr) unpack_raw(rdm[u], dm);
                unpack_raw(ry[u], v);
                if constexpr (RES) {
                    unpack_raw(rr[u], rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] = dm[e] * act_fwd((v[e] - mu[e]) * sc[e] + sh[e] + rv[e], act, slope);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = dm[e] * act_fwd((v[e] - mu[e]) * sc[e] + sh[e], act, slope);
                }
                round_store8(out != nullptr ? out + (long long)pix * ld_out + c0 : (T*)nullptr, v);
                if constexpr (SOUT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        os1[e] += v[e];
                        os2[e] += v[e] * v[e];
                    }
                }
                if constexpr (HK > 0) {
                    float pk[HK];
#pragma unroll
                    for (int k = 0; k < HK; ++k) {
                        pk[k] = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pk[k] = fmaf(v[e], hw8[k][e], pk[k]);
                        for (int off = 1; off < s.CT; off <<= 1) pk[k] += __shfl_xor(pk[k], off);
                    }
                    if (tx == 0) {
                        const int rem = pix - nn[u] * hw;
#pragma unroll
                        for (int k = 0; k < HK; ++k)
                            if (k < hd.K)
                                hd.logits[((long long)nn[u] * hd.K + k) * hw + rem] = pk[k] + (hd.b != nullptr ? hd.b[k] : 0.f);
                    }
                }
                if (up_out != nullptr) {
                    const int n = nn[u];
                    const int rem = pix - n * hw;
                    const int hh = d_w.div(rem), ww = rem - hh * s.W;
                    const long long W2x = 2ll * s.W;
                    const long long p00 = ((long long)n * 2 * s.H + 2 * hh) * W2x + 2 * ww;
                    store8(up_out + p00 * ld_up + c0, v);
                    store8(up_out + (p00 + 1) * ld_up + c0, v);
                    store8(up_out + (p00 + W2x) * ld_up + c0, v);
                    store8(up_out + (p00 + W2x + 1) * ld_up + c0, v);
                }
            }
            it0 += U * stride;
            if (it0 < npix) issue_plain(it0);
        }
        if constexpr (SOUT) {
            double* rep = so.p + (long long)(blockIdx.x % REPL) * 2 * so.ld;
            block_channel_sum2(os1, os2, s, tx, blockIdx.y * s.CT, rep, rep + so.ld, sred_o);
        }
        return;
    } else {
    while (it0 < nitems) {
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            // (no early exit: the partner lane of the exchange below walks the same items -- nitems and the stride are
            // even -- and an item past the end only skips its loads and stores)
            const bool live = it0 + i * stride < nitems;
            const int n = pn[i], h2 = ph2[i], w = pw[i];
            float dm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) dm[e] = 1.f;
            if (dropmul != nullptr) unpack_raw(rdm[i], dm);
            float mx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = -INFINITY;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int hh = 2 * h2 + r;
                if (live && hh < s.H && w < s.W) {
                    const long long pix = ((long long)n * s.H + hh) * s.W + w;
                    float v[8], rv[8];
                    unpack_raw(ry[2 * i + r], v);
                    if constexpr (RES) {
                        unpack_raw(rr[2 * i + r], rv);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = dm[e] * act_fwd((v[e] - mu[e]) * sc[e] + sh[e] + rv[e], act, slope);
                    round_store8(out != nullptr ? out + pix * ld_out + c0 : (T*)nullptr, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], v[e]);
                    if (up_out != nullptr) {
                        const long long W2x = 2ll * s.W;
                        const long long p00 = ((long long)n * 2 * s.H + 2 * hh) * W2x + 2 * w;
                        store8(up_out + p00 * ld_up + c0, v);
                        store8(up_out + (p00 + 1) * ld_up + c0, v);
                        store8(up_out + (p00 + W2x) * ld_up + c0, v);
                        store8(up_out + (p00 + W2x + 1) * ld_up + c0, v);
                    }
                }
            }
            // horizontal partner: columns w and w^1 are lanes l and l^CT of one wave (items are walked in pairs)
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], s.CT));
            if (live && (w & 1) == 0 && h2 < Hp && (w >> 1) < Wp) {
                const long long pp = ((long long)n * Hp + h2) * Wp + (w >> 1);
                store8(pool_out + pp * ld_pool + c0, mx);
            }
        }
        it0 += UP * stride;
        if (it0 < nitems) issue_pool(it0);
    }
    }
}

struct BnBwdParams {
    const double* sums;         // [REPL][2][Cp]: sum dz, sum dz*yhat; read-only in the fused kernel
    double count;
    const float* gamma;
    float* dgamma;
    float* dbeta;
    int C, accumulate;
    float* bcoef;               // [3][Cp] out
    double* zero_buf;           // fused only: the FORWARD statistics of this layer, cleared for the next step
};

// bcoef = (gamma*invstd, mean(dz), mean(dz*yhat)); `update` = this thread owns dgamma / dbeta of the channel
__device__ __forceinline__ void bn_bwd_coef(const BnBwdParams& p, const float* __restrict__ coef, int Cp, int c,
                                            bool update, float& a, float& c1, float& c2) {
    a = c1 = c2 = 0.f;
    if (c >= p.C) return;
    double v1[REPL], v2[REPL];          // loads first (see bn_fwd_coef)
#pragma unroll
    for (int rp = 0; rp < REPL; ++rp) {
        v1[rp] = p.sums[(rp * 2) * Cp + c];
        v2[rp] = p.sums[(rp * 2 + 1) * Cp + c];
    }
    double sdz = 0.0, sdzy = 0.0;
#pragma unroll
    for (int rp = 0; rp < REPL; ++rp) {
        sdz += v1[rp];
        sdzy += v2[rp];
    }
    const float g = p.gamma != nullptr ? p.gamma[c] : 1.f;
    a = g * coef[3 * Cp + c];
    c1 = (float)(sdz / p.count);
    c2 = (float)(sdzy / p.count);
    if (update) {
        if (p.dgamma != nullptr) p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + (float)sdzy;
        if (p.dbeta != nullptr) p.dbeta[c] = (p.accumulate ? p.dbeta[c] : 0.f) + (float)sdz;
    }
}


// APPLY mode of bn_act_bwd_reduce_kernel: (a, c1, c2) of the block's channels from the sums (fused finalize, exactly as
// bn_bwd_apply_kernel does it: the blocks of column 0 publish them, write dgamma / dbeta and clear the forward statistics)
// + invstd, per thread for its 8 channels.  Ends with a __syncthreads().
__device__ __forceinline__ void apply_src_coefs(const BnBwdParams& bp, const float* __restrict__ coef, const EwShape& s, int tx,
                                                int c0, float* sb3, float (&ap)[4][8]) {
    const int nch = s.CT * 8;                   // (sb3: >= 3 * 32 * 8 floats of LDS)
    if ((int)threadIdx.x < nch) {
        const int c = blockIdx.y * nch + threadIdx.x;
        float a = 0.f, c1 = 0.f, c2 = 0.f;
        if (c < s.Cp) {
            const bool owner = blockIdx.x == 0;
            bn_bwd_coef(bp, coef, s.Cp, c, owner, a, c1, c2);
            if (owner) {
                bp.bcoef[c] = a;
                bp.bcoef[s.Cp + c] = c1;
                bp.bcoef[2 * s.Cp + c] = c2;
                if (bp.zero_buf != nullptr) {
#pragma unroll
                    for (int rp = 0; rp < REPL; ++rp) {
                        bp.zero_buf[(rp * 2) * s.Cp + c] = 0.0;
                        bp.zero_buf[(rp * 2 + 1) * s.Cp + c] = 0.0;
                    }
                }
            }
        }
        sb3[threadIdx.x] = a;
        sb3[256 + threadIdx.x] = c1;
        sb3[512 + threadIdx.x] = c2;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ap[0][e] = sb3[tx * 8 + e];
        ap[1][e] = sb3[256 + tx * 8 + e];
        ap[2][e] = sb3[512 + tx * 8 + e];
        ap[3][e] = coef[3 * s.Cp + c0 + e];
    }
    __syncthreads();
}

// one pixel of the backward once its operands are in registers: summed gradient g -> dz, channel sums.  s2 accumulates
// dz*(y-mean); the caller multiplies by invstd once at the end.
// APPLY: the second pass of a layer whose dz was never stored (segnb_bn_bwd_apply_fused_src): dz is recomputed from the
// gradient sources exactly as the reduction pass computed it and leaves as dy = round(a * (dz - c1 - yhat * c2)); ap[0..3] =
// the per-channel (a, c1, c2, invstd) constants then, nothing is summed
template <typename T, bool APPLY>
__device__ __forceinline__ void bwd_pixel_math(const float (&yv)[8], const float (&sc)[8], const float (&sh)[8],
                                               const float (&mu)[8], const float (&dm)[8], int act, float slope,
                                               const float (&g)[8], const float (&rv)[8], T* __restrict__ dz_at,
                                               float (&s1)[8], float (&s2)[8], const float (*ap)[8]) {
    float d[8], yc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        yc[e] = yv[e] - mu[e];
        const float z = yc[e] * sc[e] + sh[e] + rv[e];
        d[e] = g[e] * dm[e] * act_grad(z, act, slope);
    }
    if constexpr (APPLY) {
        round_store8((T*)nullptr, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float yh = yc[e] * ap[3][e];
            d[e] = ap[0][e] * (d[e] - ap[1][e] - yh * ap[2][e]);
        }
        round_store8(dz_at, d);
    } else {
        round_store8(dz_at, d);          // NULL: sums only (the apply pass recomputes dz)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] += d[e];
            s2[e] += d[e] * yc[e];
        }
    }
}

// order-preserving 16-bit key of a bf16 value (larger value <-> larger unsigned key)
__device__ __forceinline__ unsigned bf16_order_key(unsigned b, bool nonneg) {
    if (nonneg) return b;
    return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}

// HAS_A (segnb_bn_act_bwd_reduce_add; with HAS_D, without pooling / upsampling): a SECOND same-size gradient source, passed in the
// g_up / ld_gu arguments: g = round(g_direct + g_add) -- what segnb_add would have stored -- without that pass (a tensor with two
// consumers: the residual connections of linknet.py:41-62's encoder)
template <typename T, bool HAS_D, bool HAS_P, bool HAS_U, bool APPLY = false, bool RES = false, bool HAS_A = false>
__global__ __launch_bounds__(NTHR) void bn_act_bwd_reduce_kernel(
    const T* __restrict__ y, int ld_y, EwShape s, const float* __restrict__ coef, int act, float slope,
    const float* __restrict__ dropmul, const T* __restrict__ g_direct, int ld_gd, const T* __restrict__ g_pool,
    int ld_gp, const T* __restrict__ g_up, int ld_gu, T* __restrict__ dz, int ld_dz, double* __restrict__ sums,
    const T* __restrict__ res, int ld_res, const BnBwdParams bp) {
    __shared__ float sred[SRED_FLOATS];
    constexpr int U = HAS_P ? (HAS_U ? 1 : 2) : 2;                    // items per trip
    constexpr int NR = HAS_P ? 2 : 1;                                  // pixels (rows) per item
    constexpr int NU = HAS_U ? 4 : 1;
    const int tx = threadIdx.x % s.CT, ty = threadIdx.x / s.CT;
    const int cc = blockIdx.y * s.CT + tx;
    const bool active = cc < s.CPP;
    const int c0 = active ? cc * 8 : 0;
    const int stride = gridDim.x * s.PY;
    const int npix = s.N * s.H * s.W, hw = s.H * s.W;      // 32-bit: check_ew bounds 4*N*H*W < 2^31
    // 2x2 windows: MaxPool2d backward routes the pooled gradient to the FIRST maximum (scan order).  A thread
    // owns one pixel column of a row pair (contiguous loads / stores per wave); the activations of the other
    // column of its window come from lane ^ CT (items are walked over an even padded width, in pairs)
    const int H2 = (s.H + 1) >> 1, We = 2 * ((s.W + 1) >> 1);
    const int Hp = s.H >> 1, Wp = s.W >> 1;
    const int total = HAS_P ? s.N * H2 * We : npix;
    const FastDiv d_hw(hw), d_w(s.W), d_img(H2 * We), d_we(We);
    const long long up_row = 2ll * s.W;
    const bool need_n = dropmul != nullptr || HAS_U;
    int it0 = blockIdx.x * s.PY + ty;

    // ---- the loads of one trip, all requested before the first use: [item][row]
    static_assert(!(RES && HAS_P), "residual input and pooled gradient cannot be combined");
    static_assert(!HAS_A || (HAS_D && !HAS_P && !HAS_U && !APPLY), "second source: the plain direct form");
    Raw8<T> ry[U][NR], rg[U][NR], ru[U][NR][NU], rres[RES ? U : 1], rgp[U], ra[HAS_A ? U : 1][NR];
    Raw8<float> rdm[U];
    int pn[U], ph[U], pw[U];            // image, row (row pair when pooling), column of the item
    bool okr[U][NR];
    auto issue = [&](int i0) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int it = i0 + i * stride;
            const int ic = it < total ? it : i0;
            if constexpr (HAS_P) {
                pn[i] = d_img.div(ic);
                const int rem = ic - pn[i] * (H2 * We);
                ph[i] = d_we.div(rem);
                pw[i] = rem - ph[i] * We;
            } else {
                pn[i] = need_n ? d_hw.div(ic) : 0;
                if constexpr (HAS_U) {
                    const int rem = ic - pn[i] * hw;
                    ph[i] = d_w.div(rem);
                    pw[i] = rem - ph[i] * s.W;
                } else {
                    ph[i] = pw[i] = 0;
                }
            }
            if (dropmul != nullptr) load_raw(dropmul + pn[i] * s.Cp + c0, rdm[i]);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                long long pix;
                if constexpr (HAS_P) {
                    const int hh = 2 * ph[i] + r;
                    okr[i][r] = hh < s.H && pw[i] < s.W;
                    pix = ((long long)pn[i] * s.H + (okr[i][r] ? hh : 0)) * s.W + (okr[i][r] ? pw[i] : 0);
                } else {
                    okr[i][r] = true;
                    pix = ic;
                }
                if (okr[i][r]) {
                    load_raw(y + pix * ld_y + c0, ry[i][r]);
                    if constexpr (HAS_D) load_raw(g_direct + pix * ld_gd + c0, rg[i][r]);
                    if constexpr (HAS_A) load_raw(g_up + pix * ld_gu + c0, ra[i][r]);
                    if constexpr (HAS_U) {
                        const int hh = HAS_P ? 2 * ph[i] + r : ph[i];
                        const long long up00 = ((long long)pn[i] * 2 * s.H + 2 * hh) * up_row + 2 * pw[i];
                        load_raw(g_up + up00 * ld_gu + c0, ru[i][r][0]);
                        load_raw(g_up + (up00 + 1) * ld_gu + c0, ru[i][r][1]);
                        load_raw(g_up + (up00 + up_row) * ld_gu + c0, ru[i][r][2]);
                        load_raw(g_up + (up00 + up_row + 1) * ld_gu + c0, ru[i][r][3]);
                    }
                    if constexpr (RES) load_raw(res + pix * ld_res + c0, rres[i]);
                }
            }
            if constexpr (HAS_P) {
                if (ph[i] < Hp && (pw[i] >> 1) < Wp)
                    load_raw(g_pool + (((long long)pn[i] * Hp + ph[i]) * Wp + (pw[i] >> 1)) * ld_gp + c0, rgp[i]);
            }
        }
    };
    if (active && it0 < total) issue(it0);       // (before the APPLY prologue: two latency chains overlapped)

    float ap[4][8];
    if constexpr (APPLY) apply_src_coefs(bp, coef, s, tx, c0, sred, ap);
    float sc[8], sh[8], mu[8];
    if (coef != nullptr) {          // six 16-byte loads in flight at once (element-wise selects serialise 24 dword loads)
        load8(coef + c0, sc);
        load8(coef + s.Cp + c0, sh);
        load8(coef + 2 * s.Cp + c0, mu);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = 1.f;
            sh[e] = 0.f;
            mu[e] = 0.f;
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    const bool nonneg = act == SEGNB_ACT_RELU;       // activations >= +0: the bf16 bit pattern orders them as it is

    if (active) {
        while (it0 < total) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                // (pooling: no early exit -- the partner lane of the exchange walks the same items, `total` and the
                // stride are even; an item past the end only skips its stores and sums)
                const bool live = it0 + i * stride < total;
                if (!HAS_P && !live) break;
                float dm[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dm[e] = 1.f;
                if (dropmul != nullptr) unpack_raw(rdm[i], dm);
                float yv[NR][8], g[NR][8];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) yv[r][e] = g[r][e] = 0.f;
                    if (okr[i][r]) {
                        unpack_raw(ry[i][r], yv[r]);
                        if constexpr (HAS_D) unpack_raw(rg[i][r], g[r]);
                        if constexpr (HAS_A) {
                            float t[8];
                            unpack_raw(ra[i][r], t);
#pragma unroll
                            for (int e = 0; e < 8; ++e) g[r][e] = round_as(g[r][e] + t[e], (const T*)nullptr);
                        }
                    }
                }
                if constexpr (HAS_P) {
                    const int dx = pw[i] & 1;
                    const bool pooled = ph[i] < Hp && (pw[i] >> 1) < Wp;
                    float gp[8], a[2][8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) gp[e] = 0.f;
                    if (pooled) unpack_raw(rgp[i], gp);
                    // the activations the forward pooled over, as it stored them
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            a[r][e] = dm[e] * act_fwd((yv[r][e] - mu[e]) * sc[e] + sh[e], act, slope);
                    if constexpr (sizeof(T) == 2) {
                        // bf16: the rounded activations compare as integers.  Window position k (scan order) gets the key
                        // (ordered bits << 2) | (3 - k): all four keys differ, the largest is the FIRST maximum.  This lane
                        // holds positions dx (top) and 2 + dx (bottom), the partner lane the other two -- with ITS keys
                        const unsigned prio_t = 3u - (unsigned)dx, prio_b = 1u - (unsigned)dx;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const unsigned pt = pack2bf(a[0][e], a[0][e + 1]), pb = pack2bf(a[1][e], a[1][e + 1]);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const unsigned bt = h ? pt >> 16 : pt & 0xffffu, bb = h ? pb >> 16 : pb & 0xffffu;
                                const unsigned kt = (bf16_order_key(bt, nonneg) << 2) | prio_t;
                                const unsigned kb = (bf16_order_key(bb, nonneg) << 2) | prio_b;
                                const unsigned kpt = __shfl_xor(kt, s.CT), kpb = __shfl_xor(kb, s.CT);
                                const unsigned mo = max(kpt, kpb);
                                if (pooled && kt > max(mo, kb)) g[0][e + h] += gp[e + h];
                                if (pooled && kb > max(mo, kt)) g[1][e + h] += gp[e + h];
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float ptop = __shfl_xor(a[0][e], s.CT), pbot = __shfl_xor(a[1][e], s.CT);
                            // window in scan order: (0,0) (0,1) (1,0) (1,1); this lane's pixels are positions dx and 2+dx
                            const float a0 = dx ? ptop : a[0][e], a1 = dx ? a[0][e] : ptop;
                            const float a2 = dx ? pbot : a[1][e], a3 = dx ? a[1][e] : pbot;
                            int am = 0;
                            float m = a0;
                            if (a1 > m) { m = a1; am = 1; }
                            if (a2 > m) { m = a2; am = 2; }
                            if (a3 > m) { m = a3; am = 3; }
                            if (pooled && am == dx) g[0][e] += gp[e];
                            if (pooled && am == 2 + dx) g[1][e] += gp[e];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if constexpr (HAS_U) {
                        if (okr[i][r]) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float t[8];
                                unpack_raw(ru[i][r][q], t);
#pragma unroll
                                for (int e = 0; e < 8; ++e) g[r][e] += t[e];
                            }
                        }
                    }
                    if (!(live && okr[i][r])) continue;
                    float rv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = 0.f;
                    long long pix;
                    if constexpr (HAS_P) {
                        pix = ((long long)pn[i] * s.H + 2 * ph[i] + r) * s.W + pw[i];
                    } else {
                        pix = it0 + i * stride;
                        if constexpr (RES) unpack_raw(rres[i], rv);
                    }
                    bwd_pixel_math<T, APPLY>(yv[r], sc, sh, mu, dm, act, slope, g[r], rv,
                                             dz != nullptr ? dz + pix * ld_dz + c0 : (T*)nullptr, s1, s2, ap);
                }
            }
            it0 += U * stride;
            if (it0 < total) issue(it0);
        }
    }
    if (!APPLY && sums != nullptr) {
        if (coef != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s2[e] *= coef[3 * s.Cp + c0 + e];    // sum dz*(y-mean) * invstd = sum dz*yhat
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) s2[e] = 0.f;
        }
        double* rep = sums + (long long)(blockIdx.x % REPL) * 2 * s.Cp;
        block_channel_sum2(s1, s2, s, tx, blockIdx.y * s.CT, rep, rep + s.Cp, sred);
    }
}

// Head backward THROUGH the last layer's activation, with that layer's BatchNorm-backward reduction, in one pass over its
// pre-BatchNorm output y (segnb_head_bn_bwd): per pixel da = sum_k dlogits_k w_k (rounded as segnb_head_bwd stores it), dz =
// round(da * drop * act'(z)) -> stored, (sum dz, sum dz * yhat) -> sums, and the classifier's own gradients dw_k += dlogits_k * a,
// db_k += dlogits_k with a = round(drop * act(z)) RECOMPUTED from y -- the activated tensor a and its gradient da never exist in
// memory (segnb_head_bwd reads a and writes da, segnb_bn_act_bwd_reduce reads da and y and writes dz: 5 tensor passes -> 2).
// Thread mapping and the reproducible dw / db protocol are head_bwd_kernel's (head_loss.hip).
template <typename T, int KM>
__global__ __launch_bounds__(NTHR) void head_bn_bwd_kernel(const T* __restrict__ y, int ld_y, EwShape s,
                                                           const float* __restrict__ coef, int act, float slope,
                                                           const float* __restrict__ dropmul, const float* __restrict__ w, int C,
                                                           int K, const float* __restrict__ dl, T* __restrict__ dz, int ld_dz,
                                                           double* __restrict__ sums, float* __restrict__ part) {
    __shared__ float sred[SRED_FLOATS];
    constexpr int U = 2;
    const int tx = threadIdx.x % s.CT, ty = threadIdx.x / s.CT;
    const int cc = blockIdx.y * s.CT + tx;
    const bool active = cc < s.CPP;
    const int c0 = active ? cc * 8 : 0;
    const int stride = gridDim.x * s.PY;
    const int npix = s.N * s.H * s.W, hw = s.H * s.W;
    const FastDiv d_hw(hw);
    float wv[KM][8], gw[KM][8], gb[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        gb[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wv[k][e] = (k < K && c0 + e < C) ? w[k * C + c0 + e] : 0.f;
            gw[k][e] = 0.f;
        }
    }
    float sc[8], sh[8], mu[8], s1[8], s2[8];
    load8(coef + c0, sc);
    load8(coef + s.Cp + c0, sh);
    load8(coef + 2 * s.Cp + c0, mu);
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (active) {
        for (int it0 = blockIdx.x * s.PY + ty; it0 < npix; it0 += U * stride) {
            Raw8<T> ry[U];
            Raw8<float> rdm[U];
            float g[U][KM];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = it0 + u * stride;
                const int pc = pix < npix ? pix : it0;
                load_raw(y + (long long)pc * ld_y + c0, ry[u]);
                const int n = (dropmul != nullptr || KM > 1) ? d_hw.div(pc) : 0;
                if (dropmul != nullptr) load_raw(dropmul + n * s.Cp + c0, rdm[u]);
                if (KM == 1) {
                    g[u][0] = dl[pc];
                } else {
                    const int r = pc - n * hw;
#pragma unroll
                    for (int k = 0; k < KM; ++k) g[u][k] = k < K ? dl[((long long)n * K + k) * hw + r] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = it0 + u * stride;
                if (pix >= npix) break;
                float yv[8], dm[8], a[8], d[8], yc[8], z[8];
                unpack_raw(ry[u], yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) dm[e] = 1.f;
                if (dropmul != nullptr) unpack_raw(rdm[u], dm);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    yc[e] = yv[e] - mu[e];
                    z[e] = yc[e] * sc[e] + sh[e];
                    a[e] = dm[e] * act_fwd(z[e], act, slope);
                    d[e] = 0.f;
                }
                round_store8((T*)nullptr, a);
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        const float gk = g[u][k];
                        gb[k] += gk;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            d[e] = fmaf(gk, wv[k][e], d[e]);
                            gw[k][e] = fmaf(gk, a[e], gw[k][e]);
                        }
                    }
                round_store8((T*)nullptr, d);                    // da as segnb_head_bwd would have stored it
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = d[e] * dm[e] * act_grad(z[e] + 0.f, act, slope);
                // (dz == NULL: sums only -- segnb_head_bn_bwd_apply recomputes dz from the same operands with these expressions)
                round_store8(dz != nullptr ? dz + (long long)pix * ld_dz + c0 : (T*)nullptr, d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += d[e];
                    s2[e] += d[e] * yc[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s2[e] *= coef[3 * s.Cp + c0 + e];        // sum dz*(y-mean) * invstd = sum dz*yhat
    double* rep = sums + (long long)(blockIdx.x % REPL) * 2 * s.Cp;
    block_channel_sum2(s1, s2, s, tx, blockIdx.y * s.CT, rep, rep + s.Cp, sred);
    // ---- dw / db: the PY pixel lanes of a channel chunk summed in lane order through LDS, class by class; every block writes
    // its partial sums to its own row of `part`, segnb_head_bwd_finish adds the rows in block order (bitwise reproducible)
    const int CT = s.CT, PY = s.PY;
    float* sg = sred;
    float* prow = part + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * (K * (CT * 8 + 1));
#pragma unroll
    for (int k = 0; k < KM; ++k)
        if (k < K) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sg[(ty * CT + tx) * 8 + e] = gw[k][e];
            __syncthreads();
            if ((int)threadIdx.x < CT * 8) {
                float sum = 0.f;
                for (int j = 0; j < PY; ++j) sum += sg[j * CT * 8 + threadIdx.x];
                prow[k * (CT * 8 + 1) + threadIdx.x] = sum;
            }
            __syncthreads();
            if (tx == 0) sg[ty] = gb[k];
            __syncthreads();
            if (threadIdx.x == 0) {
                float sum = 0.f;
                for (int j = 0; j < PY; ++j) sum += sg[j];
                prow[k * (CT * 8 + 1) + CT * 8] = sum;
            }
        }
}


// Second pass of the LAST layer when segnb_head_bn_bwd did not store dz (dz = NULL there): dz is a function of the d(logits) map
// (4 bytes per pixel and class) and y alone -- da = round(sum_k dlogits_k w_k), dz = round(da * drop * act'(z)), the expressions of
// head_bn_bwd_kernel above, bit for bit -- so the 2 x Cp-bytes-per-pixel tensor in between need not exist: this pass reads y and
// d(logits) again and leaves dy = round(a * (dz - c1 - yhat * c2)); (a, c1, c2) from the sums inside the launch (fused finalize,
// as segnb_bn_bwd_apply_fused).  ZF_UNET's last layer at 224 x 224, bs 32: one 103 MB write and one 103 MB read less per step.
template <typename T, int KM>
__global__ __launch_bounds__(NTHR) void head_bn_apply_kernel(const T* __restrict__ y, int ld_y, EwShape s,
                                                             const float* __restrict__ coef, int act, float slope,
                                                             const float* __restrict__ dropmul, const float* __restrict__ w, int C,
                                                             int K, const float* __restrict__ dl, T* __restrict__ dy, int ld_dy,
                                                             const BnBwdParams bp) {
    __shared__ float sred[SRED_FLOATS];
    constexpr int U = 2;
    const int tx = threadIdx.x % s.CT, ty = threadIdx.x / s.CT;
    const int cc = blockIdx.y * s.CT + tx;
    const bool active = cc < s.CPP;
    const int c0 = active ? cc * 8 : 0;
    const int stride = gridDim.x * s.PY;
    const int npix = s.N * s.H * s.W, hw = s.H * s.W;
    const FastDiv d_hw(hw);
    Raw8<T> ry[U];
    Raw8<float> rdm[U];
    float g[U][KM];
    auto issue = [&](int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = it0 + u * stride;
            const int pc = pix < npix ? pix : it0;
            load_raw(y + (long long)pc * ld_y + c0, ry[u]);
            const int n = (dropmul != nullptr || KM > 1) ? d_hw.div(pc) : 0;
            if (dropmul != nullptr) load_raw(dropmul + n * s.Cp + c0, rdm[u]);
            if (KM == 1) {
                g[u][0] = dl[pc];
            } else {
                const int r = pc - n * hw;
#pragma unroll
                for (int k = 0; k < KM; ++k) g[u][k] = k < K ? dl[((long long)n * K + k) * hw + r] : 0.f;
            }
        }
    };
    int it0 = blockIdx.x * s.PY + ty;
    if (active && it0 < npix) issue(it0);            // (before the coefficient prologue: two latency chains overlapped)
    float ap[4][8];
    apply_src_coefs(bp, coef, s, tx, c0, sred, ap);
    float wv[KM][8];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[k][e] = (k < K && c0 + e < C) ? w[k * C + c0 + e] : 0.f;
    float sc[8], sh[8], mu[8];
    load8(coef + c0, sc);
    load8(coef + s.Cp + c0, sh);
    load8(coef + 2 * s.Cp + c0, mu);
    if (!active) return;
    while (it0 < npix) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = it0 + u * stride;
            if (pix >= npix) break;
            float yv[8], dm[8], d[8], yc[8], z[8];
            unpack_raw(ry[u], yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) dm[e] = 1.f;
            if (dropmul != nullptr) unpack_raw(rdm[u], dm);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yc[e] = yv[e] - mu[e];
                z[e] = yc[e] * sc[e] + sh[e];
                d[e] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    const float gk = g[u][k];
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] = fmaf(gk, wv[k][e], d[e]);
                }
            round_store8((T*)nullptr, d);                    // da as segnb_head_bwd would have stored it
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = d[e] * dm[e] * act_grad(z[e] + 0.f, act, slope);
            round_store8((T*)nullptr, d);                    // dz as segnb_head_bn_bwd would have stored it
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float yh = yc[e] * ap[3][e];
                d[e] = ap[0][e] * (d[e] - ap[1][e] - yh * ap[2][e]);
            }
            round_store8(dy + (long long)pix * ld_dy + c0, d);
        }
        it0 += U * stride;
        if (it0 < npix) issue(it0);
    }
}


__global__ void bn_bwd_finalize_kernel(const BnBwdParams p, const float* __restrict__ coef, int Cp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cp) return;
    float a, c1, c2;
    bn_bwd_coef(p, coef, Cp, c, true, a, c1, c2);
    if (c < p.C) {                      // the stand-alone finalize CONSUMES the sums
        double* sm = const_cast<double*>(p.sums);
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            sm[(rp * 2) * Cp + c] = 0.0;
            sm[(rp * 2 + 1) * Cp + c] = 0.0;
        }
    }
    p.bcoef[c] = a;
    p.bcoef[Cp + c] = c1;
    p.bcoef[2 * Cp + c] = c2;
    if (p.zero_buf != nullptr) {        // (segnb_bn_bwd_finalize_clear: this layer's forward statistics, consumed by a fused forward)
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            p.zero_buf[(rp * 2) * Cp + c] = 0.0;
            p.zero_buf[(rp * 2 + 1) * Cp + c] = 0.0;
        }
    }
}

// The two per-element expressions of the apply pass, shared by both forms of the kernel (one expression tree = one contraction
// decision of the compiler: the lean form stays bit-identical to this one, and to conv_wgrad_c8roll_kernel's recomputed dy)
__device__ __forceinline__ float bn_dz_elem(float d, float yv, float mu, float sc, float sh, int act, float slope) {
    return d * 1.f * act_grad((yv - mu) * sc + sh + 0.f, act, slope);
}
__device__ __forceinline__ float bn_apply_elem(float yv, float d, float mu, float is, float a, float c1, float c2) {
    const float yh = (yv - mu) * is;
    return a * (d - c1 - yh * c2);
}

template <typename T, bool ACC = false>       // ACC: dy += result (gradient of a multi-consumer tensor: no separate segnb_add pass)
__global__ __launch_bounds__(NTHR) void bn_bwd_apply_kernel(const T* __restrict__ y, int ld_y, EwShape s,
                                                            const float* __restrict__ coef,
                                                            const float* __restrict__ bcoef,
                                                            const T* __restrict__ dz, int ld_dz, T* __restrict__ dy,
                                                            int ld_dy, float* __restrict__ dbias, int C,
                                                            const BnBwdParams bp, const T* __restrict__ g, int ld_g,
                                                            int act, float slope) {
    __shared__ float sred[32 * 8];
    __shared__ float sb3[3][32 * 8];
    for (int i = threadIdx.x; i < 32 * 8; i += NTHR) sred[i] = 0.f;
    const int tx = threadIdx.x % s.CT, ty = threadIdx.x / s.CT;
    const int cc = blockIdx.y * s.CT + tx;
    const bool active = cc < s.CPP;
    const int c0 = active ? cc * 8 : 0;
    // four pixels per trip, all eight loads issued before the first use (one pixel per trip = one dependent round trip
    // per wave in flight: beside the weight-gradient stream the pass ran at a third of its stand-alone rate); the first
    // trip's loads go out BEFORE the coefficient prologue (sums -> fp64 arithmetic -> LDS -> barrier)
    const long long npix = (long long)s.N * s.H * s.W;
    const long long stride = (long long)gridDim.x * s.PY;
    long long pix0 = (long long)blockIdx.x * s.PY + ty;
    Raw8<T> ry[4], rd[4], ro[4];
    auto issue = [&](long long p0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long pix = p0 + u * stride;
            const long long pc = pix < npix ? pix : p0;
            load_raw(y + pc * ld_y + c0, ry[u]);
            if (g != nullptr)
                load_raw(g + pc * ld_g + c0, rd[u]);
            else
                load_raw(dz + pc * ld_dz + c0, rd[u]);
            if constexpr (ACC) load_raw(dy + pc * ld_dy + c0, ro[u]);
        }
    };
    if (active && pix0 < npix) issue(pix0);
    if (bp.bcoef != nullptr) {
        // fused finalize (mirror of the forward kernel): every block derives (a, c1, c2) of its channels from the
        // sums; the blocks of column 0 publish them, write dgamma / dbeta and clear this layer's FORWARD statistics
        const int nch = s.CT * 8;
        if ((int)threadIdx.x < nch) {
            const int c = blockIdx.y * nch + threadIdx.x;
            float a = 0.f, c1 = 0.f, c2 = 0.f;
            if (c < s.Cp) {
                const bool owner = blockIdx.x == 0;
                bn_bwd_coef(bp, coef, s.Cp, c, owner, a, c1, c2);
                if (owner) {
                    bp.bcoef[c] = a;
                    bp.bcoef[s.Cp + c] = c1;
                    bp.bcoef[2 * s.Cp + c] = c2;
                    if (bp.zero_buf != nullptr) {
#pragma unroll
                        for (int rp = 0; rp < REPL; ++rp) {
                            bp.zero_buf[(rp * 2) * s.Cp + c] = 0.0;
                            bp.zero_buf[(rp * 2 + 1) * s.Cp + c] = 0.0;
                        }
                    }
                }
            }
            sb3[0][threadIdx.x] = a;
            sb3[1][threadIdx.x] = c1;
            sb3[2][threadIdx.x] = c2;
        }
    }
    __syncthreads();
    float mu[8], is[8], a[8], c1[8], c2[8], sb[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = coef[c0 + e];
        sh[e] = coef[s.Cp + c0 + e];
        mu[e] = coef[2 * s.Cp + c0 + e];
        is[e] = coef[3 * s.Cp + c0 + e];
        a[e] = bp.bcoef != nullptr ? sb3[0][tx * 8 + e] : bcoef[c0 + e];
        c1[e] = bp.bcoef != nullptr ? sb3[1][tx * 8 + e] : bcoef[s.Cp + c0 + e];
        c2[e] = bp.bcoef != nullptr ? sb3[2][tx * 8 + e] : bcoef[2 * s.Cp + c0 + e];
        sb[e] = 0.f;
    }
    if (active) {
        while (pix0 < npix) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long pix = pix0 + u * stride;
                if (pix >= npix) break;
                float yv[8], d[8];
                unpack_raw(ry[u], yv);
                unpack_raw(rd[u], d);
                if (g != nullptr) {
                    // dz was never written: recompute it from the incoming gradient exactly as the reduce pass did
                    // (same expression, same rounding to the storage type)
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] = bn_dz_elem(d[e], yv[e], mu[e], sc[e], sh[e], act, slope);
                    round_store8((T*)nullptr, d);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = bn_apply_elem(yv[e], d[e], mu[e], is[e], a[e], c1[e], c2[e]);
                if constexpr (ACC) {          // (the rounded result added to the stored gradient, as segnb_add would)
                    round_store8((T*)nullptr, d);
                    float old[8];
                    unpack_raw(ro[u], old);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sb[e] += d[e];
                        d[e] = __fadd_rn(old[e], d[e]);      // (no contraction with the product above)
                    }
                    store8(dy + pix * ld_dy + c0, d);
                } else {
                    round_store8(dy + pix * ld_dy + c0, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sb[e] += d[e];
                }
            }
            pix0 += 4 * stride;
            if (pix0 < npix) issue(pix0);
        }
    }
    if (dbias != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&sred[tx * 8 + e], sb[e]);
        __syncthreads();
        for (int i = threadIdx.x; i < s.CT * 8; i += NTHR) {
            const int ch = blockIdx.y * s.CT * 8 + i;
            if (ch < C) atomicAdd(&dbias[ch], sred[i]);
        }
    }
}

// (A LEAN form of this pass -- <= 96 registers, constants in LDS, so that its blocks become resident BESIDE a weight-gradient block,
// which holds 416 of a SIMD lane's 512 registers -- was built and measured in round 5: beside a stream of wide weight gradients the
// pass itself went from 38.4 to 31.8 us (tools/coresidency_probe.py: x 1.43 -> x 1.25 of its stand-alone time), the training step
// from 4.835 to 4.869 ms (+0.7 %; LinkNet34 +1.5 %): the weight gradients it now shares CUs with finish later and the data gradients
// behind the pass get fewer free CUs.  Removed; profiles/r05_ab.txt.)

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long long n, float lr) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        pv.x -= lr * gv.x; pv.y -= lr * gv.y; pv.z -= lr * gv.z; pv.w -= lr * gv.w;
        reinterpret_cast<float4*>(p)[i] = pv;
    }
    for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride)
        p[i] -= lr * g[i];
}

// ------------------------------------------------------------------------------------------------
// small generic NHWC ops for the other model families (LinkNet34 / FCDenseNet / UNet16)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHR) void add_kernel(const T* __restrict__ a, int ld_a, const T* __restrict__ b, int ld_b,
                                                   T* __restrict__ out, int ld_out, long long npix, int CPP) {
    const long long total = npix * CPP;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / CPP;
        const int c0 = (int)(i - pix * CPP) * 8;
        float x[8], yv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = yv[e] = 0.f;
        if (a != nullptr) load8(a + pix * ld_a + c0, x);          // (a missing operand counts as zeros: copy / clear)
        if (b != nullptr) load8(b + pix * ld_b + c0, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += yv[e];
        store8(out + pix * ld_out + c0, x);
    }
}

// per-channel sum / sum of squares of an NHWC tensor (BatchNorm statistics of an arbitrary input:
// pre-activation BatchNorm of tiramisu.py:12,50)
template <typename T>
__global__ __launch_bounds__(NTHR) void bn_stats_kernel(const T* __restrict__ x, int ld, EwShape s,
                                                        double* __restrict__ stats, int stats_ld) {
    __shared__ float sred[SRED_FLOATS];
    const int tx = threadIdx.x % s.CT, ty = threadIdx.x / s.CT;
    const int cc = blockIdx.y * s.CT + tx;
    const bool active = cc < s.CPP;
    const int c0 = active ? cc * 8 : 0;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    const long long npix = (long long)s.N * s.H * s.W;
    if (active)
        for (long long pix = (long long)blockIdx.x * s.PY + ty; pix < npix; pix += (long long)gridDim.x * s.PY) {
            float v[8];
            load8(x + pix * ld + c0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += v[e];
                s2[e] += v[e] * v[e];
            }
        }
    double* rep = stats + (long long)(blockIdx.x % REPL) * 2 * stats_ld;
    block_channel_sum2(s1, s2, s, tx, blockIdx.y * s.CT, rep, rep + stats_ld, sred);
}

// MaxPool2d(k, stride, pad) forward (floor mode) and its gather-form backward: every input pixel checks the
// (up to ceil(k/stride)^2) windows that contain it and takes the window's gradient if it is that window's
// FIRST maximum in scan order (torch's tie rule).
template <typename T>
__global__ __launch_bounds__(NTHR) void maxpool_fwd_kernel(const T* __restrict__ x, int ld_x, int N, int H, int W, int CPP,
                                                           int k, int st, int pd, int Ho, int Wo, T* __restrict__ out,
                                                           int ld_out, unsigned char* __restrict__ idx) {
    const long long total = (long long)N * Ho * Wo * CPP;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % CPP) * 8;
        long long q = i / CPP;
        const int wo = (int)(q % Wo);
        q /= Wo;
        const int ho = (int)(q % Ho);
        const int n = (int)(q / Ho);
        float m[8];
        unsigned am[8];                    // window position a * k + b of the FIRST maximum in scan order (torch's tie rule)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            m[e] = -INFINITY;
            am[e] = 255u;
        }
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                const int hi = ho * st - pd + a, wi = wo * st - pd + b;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
                    float v[8];
                    load8(x + (((long long)n * H + hi) * W + wi) * ld_x + c0, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (v[e] > m[e] || am[e] == 255u) {
                            m[e] = v[e];
                            am[e] = (unsigned)(a * k + b);
                        }
                }
            }
        store8(out + (((long long)n * Ho + ho) * Wo + wo) * ld_out + c0, m);
        if (idx != nullptr) {
            uint2 pk;
            pk.x = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
            pk.y = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
            *reinterpret_cast<uint2*>(idx + (i / CPP) * (long long)(CPP * 8) + c0) = pk;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NTHR) void maxpool_bwd_kernel(const T* __restrict__ x, int ld_x, const T* __restrict__ go,
                                                           int ld_go, int N, int H, int W, int CPP, int k, int st,
                                                           int pd, int Ho, int Wo, T* __restrict__ dx, int ld_dx) {
    const long long total = (long long)N * H * W * CPP;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % CPP) * 8;
        long long q = i / CPP;
        const int w = (int)(q % W);
        q /= W;
        const int h = (int)(q % H);
        const int n = (int)(q / H);
        float me[8], g[8];
        load8(x + (((long long)n * H + h) * W + w) * ld_x + c0, me);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        // windows (ho, wo) with ho*st - pd <= h <= ho*st - pd + k - 1
        int ho_lo = (h + pd - k + st) / st;
        if (h + pd - k + 1 < 0) ho_lo = 0;
        int wo_lo = (w + pd - k + st) / st;
        if (w + pd - k + 1 < 0) wo_lo = 0;
        const int ho_hi = min((h + pd) / st, Ho - 1), wo_hi = min((w + pd) / st, Wo - 1);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                // is (h, w) the first maximum of window (ho, wo)?
                bool win[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) win[e] = true;
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b) {
                        const int hi = ho * st - pd + a, wi = wo * st - pd + b;
                        if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
                        if (hi == h && wi == w) continue;
                        const bool before = hi < h || (hi == h && wi < w);
                        float v[8];
                        load8(x + (((long long)n * H + hi) * W + wi) * ld_x + c0, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (before ? v[e] >= me[e] : v[e] > me[e]) win[e] = false;
                    }
                float gv[8];
                load8(go + (((long long)n * Ho + ho) * Wo + wo) * ld_go + c0, gv);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (win[e]) g[e] += gv[e];
            }
        store8(dx + (((long long)n * H + h) * W + w) * ld_dx + c0, g);
    }
}

// The same from the argmax positions the forward pass recorded (segnb_maxpool_fwd idx): a pixel reads one index vector and
// one gradient vector per covering window (<= ceil(k / stride)^2 of them) instead of re-scanning every window --
// 3x3 stride 2 at 256x256x64 (the ResNet stem of LinkNet34 at 512x512): 2.4 ms -> the price of a streaming pass.
// I32: N * H * W * CPP < 2^31 -- the index decomposition on 32-bit integers with reciprocal divisions (three 64-bit divisions per
// item were half of the kernel's issue slots: 89 -> 6x us on LinkNet34's stem, 16 x 256 x 256 x 64)
template <typename T, bool I32>
__global__ __launch_bounds__(NTHR) void maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx, const T* __restrict__ go,
                                                               int ld_go, int N, int H, int W, int CPP, int k, int st,
                                                               int pd, int Ho, int Wo, T* __restrict__ dx, int ld_dx,
                                                               const T* __restrict__ go2, int ld_go2) {
    // go2 (segnb_maxpool_bwd_add): a second gradient of the pooled tensor (two consumers: linknet.py:41-62's first BasicBlock and
    // its identity branch); the routed value is round(go + go2), what segnb_add would have stored
    const long long total = (long long)N * H * W * CPP;
    const FastDiv d_cpp(CPP), d_w(W), d_h(H);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c0, w, h, n;
        if constexpr (I32) {
            const int ii = (int)i;
            const int q0 = d_cpp.div(ii);
            c0 = (ii - q0 * CPP) * 8;
            const int q1 = d_w.div(q0);
            w = q0 - q1 * W;
            n = d_h.div(q1);
            h = q1 - n * H;
        } else {
            c0 = (int)(i % CPP) * 8;
            long long q = i / CPP;
            w = (int)(q % W);
            q /= W;
            h = (int)(q % H);
            n = (int)(q / H);
        }
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        int ho_lo = (h + pd - k + st) / st;
        if (h + pd - k + 1 < 0) ho_lo = 0;
        int wo_lo = (w + pd - k + st) / st;
        if (w + pd - k + 1 < 0) wo_lo = 0;
        const int ho_hi = min((h + pd) / st, Ho - 1), wo_hi = min((w + pd) / st, Wo - 1);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const unsigned mine = (unsigned)((h - (ho * st - pd)) * k + (w - (wo * st - pd)));
                const long long wpix = ((long long)n * Ho + ho) * Wo + wo;
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + wpix * (long long)(CPP * 8) + c0);
                float gv[8];
                load8(go + wpix * ld_go + c0, gv);
                if (go2 != nullptr) {
                    float g2[8];
                    load8(go2 + wpix * ld_go2 + c0, g2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[e] = round_as(gv[e] + g2[e], (const T*)nullptr);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                    if (a == mine) g[e] += gv[e];
                }
            }
        store8(dx + (((long long)n * H + h) * W + w) * ld_dx + c0, g);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_f32_kernel(const T* __restrict__ a, int ld, int N, int H, int W, int C,
                                        float* __restrict__ out) {
    const long long total = (long long)N * C * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long hw = i % ((long long)H * W);
        const long long q = i / ((long long)H * W);
        const int c = (int)(q % C);
        const long long n = q / C;
        out[i] = Elem<T>::to_f32(a[(n * H * W + hw) * ld + c]);
    }
}

int check_ew(int N, int H, int W, int Cp) {
    SEGNB_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0, "bad NHWC shape (Cp % 8 != 0?)");
    SEGNB_CHECK_ARG((long long)N * H * W * 4 < (1ll << 31), "pixel count exceeds int32");
    return 0;
}

}  // namespace

extern "C" int segnb_bn_finalize(double* stats, int C, int Cp, double count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, long long* nbt, int training, float* coef,
                                 segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_finalize, stats, C, Cp, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, training, coef, stream);
    SEGNB_CHECK_ARG(coef != nullptr && C > 0 && Cp >= C && Cp % 8 == 0, "bad channel counts");
    SEGNB_CHECK_ARG(training ? stats != nullptr : (running_mean != nullptr && running_var != nullptr),
                    "missing statistics source");
    BnFwdParams fp = {stats, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, C, training, coef, nullptr};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(Cp, 64)), dim3(64), 0, (hipStream_t)stream, fp, Cp);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// The finalize of a layer whose activation pass does not exist (its consumer applies the activation while it loads:
// segnb_conv_fprop_tf): coefficients + running statistics as segnb_bn_finalize(training = 1), but under the FUSED protocol of
// segnb_bn_fwd_fused -- the forward statistics stay for this layer's backward (segnb_bn_bwd_apply_fused*) to clear, and the
// backward accumulators clear_sums are cleared here.
extern "C" int segnb_bn_finalize_keep(const double* stats, int C, int Cp, double count, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, long long* nbt,
                                      float* coef, double* clear_sums, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_finalize_keep, stats, C, Cp, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, coef, clear_sums, stream);
    SEGNB_CHECK_ARG(stats != nullptr && coef != nullptr && C > 0 && Cp >= C && Cp % 8 == 0, "bad arguments");
    BnFwdParams fp = {stats, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, C, 1, coef, clear_sums, 1};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(Cp, 64)), dim3(64), 0, (hipStream_t)stream, fp, Cp);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

static int launch_bn_act_fwd(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef,
                             int act, float slope, const float* dropmul, void* out, int ld_out, void* pool_out,
                             int ld_pool, void* up_out, int ld_up, const void* res, int ld_res, const BnFwdParams& fp,
                             const char* who, segnb_stream_t stream, const HeadFwd* hd = nullptr, const OutStats* so = nullptr);

extern "C" int segnb_bn_act_fwd(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                                const float* coef, int act, float slope, const float* dropmul, void* out,
                                int ld_out, void* pool_out, int ld_pool, void* up_out, int ld_up, const void* res,
                                int ld_res, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_act_fwd, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, pool_out, ld_pool, up_out, ld_up, res, ld_res, stream);
    BnFwdParams fp = {};
    return launch_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, pool_out, ld_pool,
                             up_out, ld_up, res, ld_res, fp, "segnb_bn_act_fwd", stream);
}

extern "C" int segnb_bn_fwd_fused(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                  const double* stats, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, long long* nbt,
                                  float* coef, double* bwd_sums_to_clear, int act, float slope, const float* dropmul,
                                  void* out, int ld_out, void* pool_out, int ld_pool, void* up_out, int ld_up,
                                  const void* res, int ld_res, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_fwd_fused, dtype, y, ld_y, N, H, W, C, Cp, stats, gamma, beta, eps, momentum, running_mean, running_var, nbt, coef, bwd_sums_to_clear, act, slope, dropmul, out, ld_out, pool_out, ld_pool, up_out, ld_up, res, ld_res, stream);
    SEGNB_CHECK_ARG(stats != nullptr && coef != nullptr && C > 0 && Cp >= C, "missing statistics / coefficient buffer");
    BnFwdParams fp = {stats, (double)N * H * W, gamma, beta, eps, momentum, running_mean, running_var, nbt, C, 1, coef,
                      bwd_sums_to_clear};
    return launch_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, nullptr, act, slope, dropmul, out, ld_out, pool_out, ld_pool,
                             up_out, ld_up, res, ld_res, fp, "segnb_bn_fwd_fused", stream);
}

// segnb_bn_act_fwd that also ACCUMULATES the per-channel statistics (sum, sum of squares, fp64, replicated) of the tensor it
// writes into channel range [0, Cp) of a table with row stride out_stats_ld; and segnb_bn_fwd_fused whose statistics are such a
// range.  FCDenseNet's dense blocks (tiramisu.py:9-44): the statistics of a concat prefix are those of its slices, which do not
// change from layer to layer -- each slice is summed once, by the pass that writes it.
extern "C" int segnb_bn_act_fwd_stats(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef,
                                      int act, float slope, const float* dropmul, void* out, int ld_out, double* out_stats,
                                      int out_stats_ld, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_act_fwd_stats, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, out_stats, out_stats_ld, stream);
    SEGNB_CHECK_ARG(out != nullptr && out_stats != nullptr, "NULL tensor");
    BnFwdParams fp = {};
    const OutStats so = {out_stats, out_stats_ld};
    return launch_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, out, ld_out, nullptr, 0, nullptr, 0,
                             nullptr, 0, fp, "segnb_bn_act_fwd_stats", stream, nullptr, &so);
}

extern "C" int segnb_bn_fwd_fused_ld(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                     const double* stats, int stats_ld, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, long long* nbt, float* coef,
                                     double* bwd_sums_to_clear, int act, float slope, const float* dropmul, void* out,
                                     int ld_out, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_fwd_fused_ld, dtype, y, ld_y, N, H, W, C, Cp, stats, stats_ld, gamma, beta, eps, momentum, running_mean, running_var, nbt, coef, bwd_sums_to_clear, act, slope, dropmul, out, ld_out, stream);
    SEGNB_CHECK_ARG(stats != nullptr && coef != nullptr && out != nullptr && C > 0 && Cp >= C && stats_ld >= Cp,
                    "missing statistics / coefficient buffer, or a statistics stride below Cp");
    BnFwdParams fp = {stats, (double)N * H * W, gamma, beta, eps, momentum, running_mean, running_var, nbt, C, 1, coef,
                      bwd_sums_to_clear, 0, stats_ld};
    return launch_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, nullptr, act, slope, dropmul, out, ld_out, nullptr, 0, nullptr, 0,
                             nullptr, 0, fp, "segnb_bn_fwd_fused_ld", stream);
}

extern "C" int segnb_head_fused_ok(int K, int Cp) {
    if (K < 1 || K > 4 || Cp < 8 || Cp % 8 != 0 || Cp > 256) return 0;
    const int cpp = Cp / 8;
    return (cpp & (cpp - 1)) == 0;
}

extern "C" int segnb_bn_fwd_fused_head(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                       const double* stats, const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, long long* nbt, float* coef,
                                       double* bwd_sums_to_clear, int act, float slope, const float* dropmul, void* out,
                                       int ld_out, const float* head_w, const float* head_b, int K, float* logits,
                                       segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_fwd_fused_head, dtype, y, ld_y, N, H, W, C, Cp, stats, gamma, beta, eps, momentum, running_mean, running_var, nbt, coef, bwd_sums_to_clear, act, slope, dropmul, out, ld_out, head_w, head_b, K, logits, stream);
    SEGNB_CHECK_ARG(stats != nullptr && coef != nullptr && C > 0 && Cp >= C, "missing statistics / coefficient buffer");
    SEGNB_CHECK_ARG(segnb_head_fused_ok(K, Cp), "classifier variant: 1..4 classes, Cp / 8 a power of two <= 32 (segnb_head_fused_ok)");
    BnFwdParams fp = {stats, (double)N * H * W, gamma, beta, eps, momentum, running_mean, running_var, nbt, C, 1, coef,
                      bwd_sums_to_clear};
    const HeadFwd hd = {head_w, head_b, logits, K, C};
    return launch_bn_act_fwd(dtype, y, ld_y, N, H, W, Cp, nullptr, act, slope, dropmul, out, ld_out, nullptr, 0, nullptr, 0,
                             nullptr, 0, fp, "segnb_bn_fwd_fused_head", stream, &hd);
}

static int launch_bn_act_fwd(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef,
                             int act, float slope, const float* dropmul, void* out, int ld_out, void* pool_out,
                             int ld_pool, void* up_out, int ld_up, const void* res, int ld_res, const BnFwdParams& fp,
                             const char* who, segnb_stream_t stream, const HeadFwd* hd, const OutStats* so) {
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y != nullptr && (out || pool_out || up_out || hd), "NULL tensor");
    const EwShape s = make_shape(N, H, W, Cp);
    if (so != nullptr) {
        // the variant that also accumulates the statistics of what it writes
        SEGNB_CHECK_ARG(hd == nullptr && pool_out == nullptr && res == nullptr && so->p != nullptr && so->ld >= Cp,
                        "output statistics: no pooling / residual / classifier, statistics stride >= Cp");
        const dim3 grid = make_grid(s, (long long)N * H * W, 2048);
#define SEGNB_FWDS(TT)                                                                                              \
    hipLaunchKernelGGL((bn_act_fwd_kernel<TT, false, false, 0, true>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const TT*)y, \
                       ld_y, s, coef, act, slope, dropmul, (TT*)out, ld_out, (TT*)nullptr, 0, (TT*)up_out, ld_up,    \
                       (const TT*)nullptr, 0, fp, HeadFwd{}, *so)
        if (dtype == SEGNB_BF16) {
            SEGNB_FWDS(bf16_t);
        } else if (dtype == SEGNB_F32) {
            SEGNB_FWDS(float);
        } else {
            segnb_set_error("%s: unknown dtype %d", who, dtype);
            return SEGNB_E_BADARG;
        }
#undef SEGNB_FWDS
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    if (hd != nullptr) {
        // the classifier variant: a pixel's channel chunks are CT consecutive lanes of one wave (one block column)
        SEGNB_CHECK_ARG(pool_out == nullptr && res == nullptr && s.CT == s.CPP && hd->K >= 1 && hd->K <= 4 && hd->w && hd->logits,
                        "classifier variant: no pooling / residual, Cp / 8 a power of two <= 32, 1..4 classes (segnb_head_fused_ok)");
        const dim3 grid = make_grid(s, (long long)N * H * W, 4096);
#define SEGNB_FWDH(TT, HKK)                                                                                         \
    hipLaunchKernelGGL((bn_act_fwd_kernel<TT, false, false, HKK>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const TT*)y, \
                       ld_y, s, coef, act, slope, dropmul, (TT*)out, ld_out, (TT*)nullptr, 0, (TT*)up_out, ld_up,    \
                       (const TT*)nullptr, 0, fp, *hd)
        if (dtype == SEGNB_BF16) {
            if (hd->K == 1) SEGNB_FWDH(bf16_t, 1); else SEGNB_FWDH(bf16_t, 4);
        } else if (dtype == SEGNB_F32) {
            if (hd->K == 1) SEGNB_FWDH(float, 1); else SEGNB_FWDH(float, 4);
        } else {
            segnb_set_error("%s: unknown dtype %d", who, dtype);
            return SEGNB_E_BADARG;
        }
#undef SEGNB_FWDH
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    // pooling: one item per pixel column of a row pair, over an even padded width
    const long long items = pool_out != nullptr ? (long long)N * ((H + 1) / 2) * (2 * ((W + 1) / 2)) : (long long)N * H * W;
    const dim3 grid = make_grid(s, items, pool_out != nullptr ? 2048 : 4096);
#define SEGNB_FWD(TT, P, R)                                                                                         \
    hipLaunchKernelGGL((bn_act_fwd_kernel<TT, P, R>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const TT*)y, ld_y, s, \
                       coef, act, slope, dropmul, (TT*)out, ld_out, (TT*)pool_out, ld_pool, (TT*)up_out, ld_up,     \
                       (const TT*)res, ld_res, fp)
#define SEGNB_FWD_ALL(TT)                                            \
    if (pool_out != nullptr) {                                       \
        if (res != nullptr) SEGNB_FWD(TT, true, true); else SEGNB_FWD(TT, true, false);      \
    } else {                                                         \
        if (res != nullptr) SEGNB_FWD(TT, false, true); else SEGNB_FWD(TT, false, false);    \
    }
    if (dtype == SEGNB_BF16) {
        SEGNB_FWD_ALL(bf16_t)
    } else if (dtype == SEGNB_F32) {
        SEGNB_FWD_ALL(float)
    } else {
        segnb_set_error("%s: unknown dtype %d", who, dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_FWD_ALL
#undef SEGNB_FWD
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_bn_act_bwd_reduce(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                                       const float* coef, int act, float slope, const float* dropmul,
                                       const void* g_direct, int ld_gd, const void* g_pool, int ld_gp,
                                       const void* g_up, int ld_gu, void* dz, int ld_dz, double* sums,
                                       const void* res, int ld_res, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_act_bwd_reduce, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, g_direct, ld_gd, g_pool, ld_gp, g_up, ld_gu, dz, ld_dz, sums, res, ld_res, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y != nullptr, "NULL tensor");
    SEGNB_CHECK_ARG(g_direct || g_pool || g_up, "no gradient source");
    SEGNB_CHECK_ARG(dz != nullptr || sums != nullptr, "a pass without dz needs the sums (the apply pass recomputes dz: "
                    "segnb_bn_bwd_apply_direct / segnb_bn_bwd_apply_fused_src)");
    SEGNB_CHECK_ARG(!(res && g_pool), "residual input and pooled gradient cannot be combined");
    const EwShape s = make_shape(N, H, W, Cp);
    const bool hd = g_direct != nullptr, hp = g_pool != nullptr, hu = g_up != nullptr;
    const long long items = hp ? (long long)N * ((H + 1) / 2) * (2 * ((W + 1) / 2)) : (long long)N * H * W;
    const dim3 grid = make_grid(s, items);
    const int variant = (hd ? 1 : 0) | (hp ? 2 : 0) | (hu ? 4 : 0);
#define SEGNB_RED(TT, D, P, U, R)                                                                                   \
    SEGNB_LAUNCH_FORKABLE((bn_act_bwd_reduce_kernel<TT, D, P, U, false, R>), grid, dim3(NTHR), 0, (hipStream_t)stream, \
                       (const TT*)y, ld_y, s, coef, act, slope, dropmul, (const TT*)g_direct, ld_gd,              \
                       (const TT*)g_pool, ld_gp, (const TT*)g_up, ld_gu, (TT*)dz, ld_dz, sums, (const TT*)res, ld_res, BnBwdParams{})
#define SEGNB_RED_R(TT, D, U)                                                                                       \
    if (res != nullptr) SEGNB_RED(TT, D, false, U, true); else SEGNB_RED(TT, D, false, U, false)
#define SEGNB_RED_ALL(TT)                                                       \
    switch (variant) {                                                          \
        case 1: SEGNB_RED_R(TT, true, false); break;                            \
        case 2: SEGNB_RED(TT, false, true, false, false); break;                \
        case 3: SEGNB_RED(TT, true, true, false, false); break;                 \
        case 4: SEGNB_RED_R(TT, false, true); break;                            \
        case 5: SEGNB_RED_R(TT, true, true); break;                             \
        case 6: SEGNB_RED(TT, false, true, true, false); break;                 \
        default: SEGNB_RED(TT, true, true, true, false); break;                 \
    }
    if (dtype == SEGNB_BF16) {
        SEGNB_RED_ALL(bf16_t)
    } else if (dtype == SEGNB_F32) {
        SEGNB_RED_ALL(float)
    } else {
        segnb_set_error("segnb_bn_act_bwd_reduce: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_RED_ALL
#undef SEGNB_RED_R
#undef SEGNB_RED
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_bn_act_bwd_reduce_add(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef, int act,
                                           float slope, const float* dropmul, const void* g_direct, int ld_gd, const void* g_add,
                                           int ld_ga, void* dz, int ld_dz, double* sums, const void* res, int ld_res,
                                           segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_act_bwd_reduce_add, dtype, y, ld_y, N, H, W, Cp, coef, act, slope, dropmul, g_direct, ld_gd, g_add, ld_ga, dz, ld_dz, sums, res, ld_res, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y != nullptr && g_direct != nullptr && g_add != nullptr, "NULL tensor");
    SEGNB_CHECK_ARG(dz != nullptr || sums != nullptr, "a pass without dz needs the sums");
    const EwShape s = make_shape(N, H, W, Cp);
    const dim3 grid = make_grid(s, (long long)N * H * W);
#define SEGNB_RED_A(TT, R)                                                                                                     \
    SEGNB_LAUNCH_FORKABLE((bn_act_bwd_reduce_kernel<TT, true, false, false, false, R, true>), grid, dim3(NTHR), 0, (hipStream_t)stream, \
                          (const TT*)y, ld_y, s, coef, act, slope, dropmul, (const TT*)g_direct, ld_gd, (const TT*)nullptr, 0,    \
                          (const TT*)g_add, ld_ga, (TT*)dz, ld_dz, sums, (const TT*)res, ld_res, BnBwdParams{})
    if (dtype == SEGNB_BF16) {
        if (res != nullptr) SEGNB_RED_A(bf16_t, true); else SEGNB_RED_A(bf16_t, false);
    } else if (dtype == SEGNB_F32) {
        if (res != nullptr) SEGNB_RED_A(float, true); else SEGNB_RED_A(float, false);
    } else {
        segnb_set_error("segnb_bn_act_bwd_reduce_add: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_RED_A
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_head_bn_bwd(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const float* coef,
                                 int act, float slope, const float* dropmul, const float* head_w, int K, const float* dlogits,
                                 void* dz, int ld_dz, double* sums, float* dw, float* db, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_head_bn_bwd, dtype, y, ld_y, N, H, W, C, Cp, coef, act, slope, dropmul, head_w, K, dlogits, dz, ld_dz, sums, dw, db, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y && coef && head_w && dlogits && sums && C > 0 && Cp >= C, "NULL tensor");      // (dz NULL: sums only)
    SEGNB_CHECK_ARG(segnb_head_fused_ok(K, Cp), "1..4 classes, Cp / 8 a power of two <= 32 (segnb_head_fused_ok)");
    const EwShape s = make_shape(N, H, W, Cp);
    const dim3 grid = make_grid(s, (long long)N * H * W);
    float* part = segnb_head_scratch((size_t)grid.x * grid.y * K * (s.CT * 8 + 1) * sizeof(float), (hipStream_t)stream);
    if (part == nullptr) return SEGNB_E_BADARG;
#define SEGNB_HBB(TT, KMM)                                                                                          \
    hipLaunchKernelGGL((head_bn_bwd_kernel<TT, KMM>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const TT*)y, ld_y, s, coef, \
                       act, slope, dropmul, head_w, C, K, dlogits, (TT*)dz, ld_dz, sums, part)
    if (dtype == SEGNB_BF16) {
        if (K == 1) SEGNB_HBB(bf16_t, 1); else SEGNB_HBB(bf16_t, 4);
    } else if (dtype == SEGNB_F32) {
        if (K == 1) SEGNB_HBB(float, 1); else SEGNB_HBB(float, 4);
    } else {
        segnb_set_error("segnb_head_bn_bwd: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_HBB
    SEGNB_LAUNCH_CHECK();
    if (dw != nullptr || db != nullptr) {
        segnb_head_bwd_finish(part, (int)grid.x, (int)grid.y, K, C, s.CT, dw, db, (hipStream_t)stream);
        SEGNB_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int segnb_head_bn_bwd_apply(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const float* coef,
                                       const double* sums, const float* gamma, float* bcoef, float* dgamma, float* dbeta,
                                       int accumulate, double* fwd_stats_to_clear, int act, float slope, const float* dropmul,
                                       const float* head_w, int K, const float* dlogits, void* dy, int ld_dy,
                                       segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_head_bn_bwd_apply, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate,
                      fwd_stats_to_clear, act, slope, dropmul, head_w, K, dlogits, dy, ld_dy, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y && coef && sums && bcoef && head_w && dlogits && dy && C > 0 && Cp >= C, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_head_fused_ok(K, Cp), "1..4 classes, Cp / 8 a power of two <= 32 (segnb_head_fused_ok)");
    SEGNB_CHECK_ARG(ld_y % 8 == 0 && ld_dy % 8 == 0 && ld_y >= Cp && ld_dy >= Cp, "bad strides");
    const EwShape s = make_shape(N, H, W, Cp);
    const dim3 grid = make_grid(s, (long long)N * H * W);
    const BnBwdParams bp = {sums, (double)N * H * W, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
#define SEGNB_HBA(TT, KMM)                                                                                                    \
    SEGNB_LAUNCH_FORKABLE((head_bn_apply_kernel<TT, KMM>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const TT*)y, ld_y, s, coef, \
                          act, slope, dropmul, head_w, C, K, dlogits, (TT*)dy, ld_dy, bp)
    if (dtype == SEGNB_BF16) {
        if (K == 1) SEGNB_HBA(bf16_t, 1); else SEGNB_HBA(bf16_t, 4);
    } else if (dtype == SEGNB_F32) {
        if (K == 1) SEGNB_HBA(float, 1); else SEGNB_HBA(float, 4);
    } else {
        segnb_set_error("segnb_head_bn_bwd_apply: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_HBA
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// The bias gradients of MANY convolutions without BatchNorm in one launch: job j adds the replicated sums of dz (what
// segnb_bn_act_bwd_reduce accumulated) into the bias gradient and clears the sums for the next step -- what one
// segnb_bn_bwd_finalize(gamma = NULL) launch per layer did (FCDenseNet103: 238 launches of 4 us per step on the dependent chain;
// nothing reads a bias gradient before the end of backward).
struct BiasGradJob {
    double* sums;      // [REPL][2][Cp]
    float* gb;         // [C] or NULL (no bias: the sums are only cleared)
    int C, Cp;
};
__global__ __launch_bounds__(256) void bias_grad_multi_kernel(const BiasGradJob* __restrict__ jobs) {
    const BiasGradJob j = jobs[blockIdx.x];
    for (int c = threadIdx.x; c < j.Cp; c += blockDim.x) {
        double v[REPL];
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) v[rp] = j.sums[(rp * 2) * j.Cp + c];
        double t = 0.0;
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) t += v[rp];
        if (c < j.C && j.gb != nullptr) j.gb[c] += (float)t;
#pragma unroll
        for (int rp = 0; rp < REPL; ++rp) {
            j.sums[(rp * 2) * j.Cp + c] = 0.0;
            j.sums[(rp * 2 + 1) * j.Cp + c] = 0.0;
        }
    }
}

extern "C" int segnb_bias_grad_job_bytes(void) { return (int)sizeof(BiasGradJob); }

extern "C" int segnb_bias_grad_multi(const void* jobs, int njobs, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bias_grad_multi, jobs, njobs, stream);
    SEGNB_CHECK_ARG(jobs != nullptr && njobs > 0, "bad job table");
    hipLaunchKernelGGL(bias_grad_multi_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, (const BiasGradJob*)jobs);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_bn_bwd_finalize(double* sums, int C, int Cp, double count, const float* gamma,
                                     const float* coef, float* bcoef, float* dgamma, float* dbeta,
                                     int accumulate, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_finalize, sums, C, Cp, count, gamma, coef, bcoef, dgamma, dbeta, accumulate, stream);
    SEGNB_CHECK_ARG(sums && coef && bcoef && C > 0 && Cp >= C && Cp % 8 == 0, "bad arguments");
    BnBwdParams bp = {sums, count, gamma, dgamma, dbeta, C, accumulate, bcoef, nullptr};
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(Cp, 64)), dim3(64), 0, (hipStream_t)stream, bp, coef, Cp);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// the same, and the layer's FORWARD statistics buffer cleared (what the fused apply launches do on the way): for a layer
// whose apply pass does not exist because its only consumer recomputes dy (segnb_conv_wgrad_bnapply)
extern "C" int segnb_bn_bwd_finalize_clear(double* sums, int C, int Cp, double count, const float* gamma,
                                           const float* coef, float* bcoef, float* dgamma, float* dbeta,
                                           int accumulate, double* fwd_stats_to_clear, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_finalize_clear, sums, C, Cp, count, gamma, coef, bcoef, dgamma, dbeta, accumulate, fwd_stats_to_clear, stream);
    SEGNB_CHECK_ARG(sums && coef && bcoef && C > 0 && Cp >= C && Cp % 8 == 0, "bad arguments");
    BnBwdParams bp = {sums, count, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(Cp, 64)), dim3(64), 0, (hipStream_t)stream, bp, coef, Cp);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

static int launch_bn_bwd_apply(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef,
                               const float* bcoef, const void* dz, int ld_dz, void* dy, int ld_dy, float* dbias, int C,
                               const BnBwdParams& bp, const char* who, segnb_stream_t stream, const void* g = nullptr,
                               int ld_g = 0, int act = 0, float slope = 0.f, bool acc = false) {
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y && coef && (bcoef || bp.bcoef) && (dz || g) && dy, "NULL tensor");
    const EwShape s = make_shape(N, H, W, Cp);
    const dim3 grid = make_grid(s, (long long)N * H * W, 1536);
    if (acc && dtype == SEGNB_BF16)
        SEGNB_LAUNCH_FORKABLE((bn_bwd_apply_kernel<bf16_t, true>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)y,
                           ld_y, s, coef, bcoef, (const bf16_t*)dz, ld_dz, (bf16_t*)dy, ld_dy, dbias, C, bp,
                           (const bf16_t*)g, ld_g, act, slope);
    else if (acc && dtype == SEGNB_F32)
        SEGNB_LAUNCH_FORKABLE((bn_bwd_apply_kernel<float, true>), grid, dim3(NTHR), 0, (hipStream_t)stream, (const float*)y,
                           ld_y, s, coef, bcoef, (const float*)dz, ld_dz, (float*)dy, ld_dy, dbias, C, bp,
                           (const float*)g, ld_g, act, slope);
    else if (dtype == SEGNB_BF16)
        SEGNB_LAUNCH_FORKABLE(bn_bwd_apply_kernel<bf16_t>, grid, dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)y,
                           ld_y, s, coef, bcoef, (const bf16_t*)dz, ld_dz, (bf16_t*)dy, ld_dy, dbias, C, bp,
                           (const bf16_t*)g, ld_g, act, slope);
    else if (dtype == SEGNB_F32)
        SEGNB_LAUNCH_FORKABLE(bn_bwd_apply_kernel<float>, grid, dim3(NTHR), 0, (hipStream_t)stream, (const float*)y,
                           ld_y, s, coef, bcoef, (const float*)dz, ld_dz, (float*)dy, ld_dy, dbias, C, bp,
                           (const float*)g, ld_g, act, slope);
    else {
        segnb_set_error("%s: unknown dtype %d", who, dtype);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_bn_bwd_apply(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                                  const float* coef, const float* bcoef, const void* dz, int ld_dz, void* dy,
                                  int ld_dy, float* dbias, int C, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply, dtype, y, ld_y, N, H, W, Cp, coef, bcoef, dz, ld_dz, dy, ld_dy, dbias, C, stream);
    BnBwdParams bp = {};
    return launch_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, bcoef, dz, ld_dz, dy, ld_dy, dbias, C, bp,
                               "segnb_bn_bwd_apply", stream);
}

extern "C" int segnb_bn_bwd_apply_direct(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                                         const float* coef, const float* bcoef, int act, float slope, const void* g,
                                         int ld_g, void* dy, int ld_dy, float* dbias, int C, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply_direct, dtype, y, ld_y, N, H, W, Cp, coef, bcoef, act, slope, g, ld_g, dy, ld_dy, dbias, C, stream);
    SEGNB_CHECK_ARG(g != nullptr, "NULL gradient");
    BnBwdParams bp = {};
    return launch_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, bcoef, nullptr, 0, dy, ld_dy, dbias, C, bp,
                               "segnb_bn_bwd_apply_direct", stream, g, ld_g, act, slope);
}

extern "C" int segnb_bn_bwd_apply_fused(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                        const float* coef, const double* sums, const float* gamma, float* bcoef,
                                        float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear,
                                        const void* dz, int ld_dz, void* dy, int ld_dy, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply_fused, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate, fwd_stats_to_clear, dz, ld_dz, dy, ld_dy, stream);
    SEGNB_CHECK_ARG(sums != nullptr && bcoef != nullptr && C > 0 && Cp >= C, "missing sums / coefficient buffer");
    BnBwdParams bp = {sums, (double)N * H * W, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
    return launch_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, nullptr, dz, ld_dz, dy, ld_dy, nullptr, C, bp,
                               "segnb_bn_bwd_apply_fused", stream);
}

extern "C" int segnb_bn_bwd_apply_fused_acc(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                            const float* coef, const double* sums, const float* gamma, float* bcoef,
                                            float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear,
                                            const void* dz, int ld_dz, void* dy, int ld_dy, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply_fused_acc, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate, fwd_stats_to_clear, dz, ld_dz, dy, ld_dy, stream);
    SEGNB_CHECK_ARG(sums != nullptr && bcoef != nullptr && C > 0 && Cp >= C, "missing sums / coefficient buffer");
    SEGNB_CHECK_ARG(dz != nullptr && dz != dy, "the accumulating form needs dz in a buffer of its own");
    BnBwdParams bp = {sums, (double)N * H * W, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
    return launch_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, nullptr, dz, ld_dz, dy, ld_dy, nullptr, C, bp,
                               "segnb_bn_bwd_apply_fused_acc", stream, nullptr, 0, 0, 0.f, true);
}

// (A one-launch BatchNorm backward for small tensors -- a block owning an 8-channel group, every pixel in registers -- was built in
// round 4 and measured slower than the two launches it replaced, 40 vs 18 us at 14 x 14 x 512: DESIGN 11.13; removed in round 5.)

extern "C" int segnb_bn_bwd_apply_fused_direct(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                               const float* coef, const double* sums, const float* gamma,
                                               float* bcoef, float* dgamma, float* dbeta, int accumulate,
                                               double* fwd_stats_to_clear, int act, float slope, const void* g,
                                               int ld_g, void* dy, int ld_dy, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply_fused_direct, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate, fwd_stats_to_clear, act, slope, g, ld_g, dy, ld_dy, stream);
    SEGNB_CHECK_ARG(sums != nullptr && bcoef != nullptr && C > 0 && Cp >= C, "missing sums / coefficient buffer");
    SEGNB_CHECK_ARG(g != nullptr, "NULL gradient");
    BnBwdParams bp = {sums, (double)N * H * W, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
    return launch_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, nullptr, nullptr, 0, dy, ld_dy, nullptr, C, bp,
                               "segnb_bn_bwd_apply_fused_direct", stream, g, ld_g, act, slope);
}

extern "C" int segnb_bn_bwd_apply_fused_direct_acc(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                                   const float* coef, const double* sums, const float* gamma,
                                                   float* bcoef, float* dgamma, float* dbeta, int accumulate,
                                                   double* fwd_stats_to_clear, int act, float slope, const void* g,
                                                   int ld_g, void* dy, int ld_dy, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply_fused_direct_acc, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate, fwd_stats_to_clear, act, slope, g, ld_g, dy, ld_dy, stream);
    SEGNB_CHECK_ARG(sums != nullptr && bcoef != nullptr && C > 0 && Cp >= C, "missing sums / coefficient buffer");
    SEGNB_CHECK_ARG(g != nullptr && g != dy, "the accumulating form needs the incoming gradient in a buffer of its own");
    BnBwdParams bp = {sums, (double)N * H * W, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
    return launch_bn_bwd_apply(dtype, y, ld_y, N, H, W, Cp, coef, nullptr, nullptr, 0, dy, ld_dy, nullptr, C, bp,
                               "segnb_bn_bwd_apply_fused_direct_acc", stream, g, ld_g, act, slope, true);
}

// Second pass of a layer whose dz was NEVER stored: the gradient sources are read again and dz recomputed exactly as
// segnb_bn_act_bwd_reduce (dz = NULL) computed it for the sums -- MaxPool2d routing, Upsample sum, Dropout2d multiplier --
// and dy = round(a * (dz - c1 - yhat * c2)) leaves; (a, c1, c2) come from the sums inside the launch as in
// segnb_bn_bwd_apply_fused.  One tensor write and one tensor read less than reduce(-> dz) + apply(dz -> dy).
extern "C" int segnb_bn_bwd_apply_fused_src(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                            const float* coef, const double* sums, const float* gamma, float* bcoef,
                                            float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear, int act,
                                            float slope, const float* dropmul, const void* g_direct, int ld_gd,
                                            const void* g_pool, int ld_gp, const void* g_up, int ld_gu, void* dy, int ld_dy,
                                            segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_bwd_apply_fused_src, dtype, y, ld_y, N, H, W, C, Cp, coef, sums, gamma, bcoef, dgamma, dbeta, accumulate, fwd_stats_to_clear, act, slope, dropmul, g_direct, ld_gd, g_pool, ld_gp, g_up, ld_gu, dy, ld_dy, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(y && coef && sums && bcoef && dy && C > 0 && Cp >= C, "NULL tensor / missing sums or coefficient buffer");
    SEGNB_CHECK_ARG(g_direct || g_pool || g_up, "no gradient source");
    SEGNB_CHECK_ARG(dy != g_direct && dy != g_pool && dy != g_up, "dy must not alias a gradient source (windows are re-read)");
    const EwShape s = make_shape(N, H, W, Cp);
    const bool hd = g_direct != nullptr, hp = g_pool != nullptr, hu = g_up != nullptr;
    const long long items = hp ? (long long)N * ((H + 1) / 2) * (2 * ((W + 1) / 2)) : (long long)N * H * W;
    const dim3 grid = make_grid(s, items);
    const int variant = (hd ? 1 : 0) | (hp ? 2 : 0) | (hu ? 4 : 0);
    const BnBwdParams bp = {sums, (double)N * H * W, gamma, dgamma, dbeta, C, accumulate, bcoef, fwd_stats_to_clear};
#define SEGNB_APS(TT, D, P, U)                                                                                      \
    SEGNB_LAUNCH_FORKABLE((bn_act_bwd_reduce_kernel<TT, D, P, U, true>), grid, dim3(NTHR), 0, (hipStream_t)stream, \
                       (const TT*)y, ld_y, s, coef, act, slope, dropmul, (const TT*)g_direct, ld_gd,              \
                       (const TT*)g_pool, ld_gp, (const TT*)g_up, ld_gu, (TT*)dy, ld_dy, nullptr, (const TT*)nullptr, 0, bp)
#define SEGNB_APS_ALL(TT)                                                       \
    switch (variant) {                                                          \
        case 1: SEGNB_APS(TT, true, false, false); break;                       \
        case 2: SEGNB_APS(TT, false, true, false); break;                       \
        case 3: SEGNB_APS(TT, true, true, false); break;                        \
        case 4: SEGNB_APS(TT, false, false, true); break;                       \
        case 5: SEGNB_APS(TT, true, false, true); break;                        \
        case 6: SEGNB_APS(TT, false, true, true); break;                        \
        default: SEGNB_APS(TT, true, true, true); break;                        \
    }
    if (dtype == SEGNB_BF16) {
        SEGNB_APS_ALL(bf16_t)
    } else if (dtype == SEGNB_F32) {
        SEGNB_APS_ALL(float)
    } else {
        segnb_set_error("segnb_bn_bwd_apply_fused_src: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_APS_ALL
#undef SEGNB_APS
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// InPlaceABN's backend affine (lib/modules/abn/functions.py:94,112,118 bind the first inplace_abn release): the scale the
// kernels apply is |weight| + eps, and the weight gradient is sign(weight) * sum(dz * yhat).  Two C-element launches around the
// BatchNorm entry points (which take the effective scale as their gamma): recordable in a launch list, no torch operator.
__global__ void abn_scale_kernel(const float* __restrict__ w, float eps, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fabsf(w[i]) + eps;
}
__global__ void abn_dscale_kernel(const float* __restrict__ w, float* __restrict__ dscale, float* __restrict__ dw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float d = dscale[i];
        dw[i] += w[i] > 0.f ? d : -d;       // d(|w| + eps)/dw as the backend signs it: +1 for w > 0, -1 otherwise
        dscale[i] = 0.f;                    // consumed: the buffer accumulates the next backward from zero
    }
}

extern "C" int segnb_abn_scale(const float* w, float eps, float* out, int n, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_abn_scale, w, eps, out, n, stream);
    SEGNB_CHECK_ARG(w && out && n > 0, "NULL tensor");
    hipLaunchKernelGGL(abn_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, eps, out, n);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_abn_dscale(const float* w, float* dscale, float* dw, int n, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_abn_dscale, w, dscale, dw, n, stream);
    SEGNB_CHECK_ARG(w && dscale && dw && n > 0, "NULL tensor");
    hipLaunchKernelGGL(abn_dscale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, dscale, dw, n);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

// torch.optim.RMSprop (alpha, eps; no momentum / centering / weight decay) and torch.optim.Adam (betas, eps; no
// amsgrad / weight decay) over the flat buffers: get_optimizer('rms' | 'adam') of torch_train.py:73-77
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, long long n,
                               float lr, float alpha, float eps) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float v = alpha * sq[i] + (1.f - alpha) * gi * gi;
        sq[i] = v;
        p[i] -= lr * gi / (sqrtf(v) + eps);
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float beta1, float beta2,
                            float inv_sqrt_bc2, float eps) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;         // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

extern "C" int segnb_rmsprop_step(float* p, const float* g, float* square_avg, long long n, float lr, float alpha,
                                  float eps, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_rmsprop_step, p, g, square_avg, n, lr, alpha, eps, stream);
    SEGNB_CHECK_ARG(p && g && square_avg && n > 0, "bad arguments");
    int grid = ceil_div(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, square_avg, n, lr, alpha, eps);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                               float beta1, float beta2, float eps, int step, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_adam_step, p, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, stream);
    SEGNB_CHECK_ARG(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    int grid = ceil_div(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n,
                       (float)((double)lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_sgd_step(float* p, const float* g, long long n, float lr, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_sgd_step, p, g, n, lr, stream);
    SEGNB_CHECK_ARG(p && g && n > 0, "bad arguments");
    SEGNB_CHECK_ARG((((uintptr_t)p | (uintptr_t)g) & 15) == 0, "buffers must be 16-byte aligned");
    int grid = ceil_div(n / 4 + 1, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, n, lr);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

#define SEGNB_DISPATCH_T(CALL_BF16, CALL_F32, NAME)                 \
    if (dtype == SEGNB_BF16) {                                      \
        CALL_BF16;                                                  \
    } else if (dtype == SEGNB_F32) {                                \
        CALL_F32;                                                   \
    } else {                                                        \
        segnb_set_error(NAME ": unknown dtype %d", dtype);          \
        return SEGNB_E_BADARG;                                      \
    }

extern "C" int segnb_add(int dtype, const void* a, int ld_a, const void* b, int ld_b, void* out, int ld_out, int N,
                         int H, int W, int Cp, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_add, dtype, a, ld_a, b, ld_b, out, ld_out, N, H, W, Cp, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(out, "NULL tensor");
    const long long npix = (long long)N * H * W;
    int grid = ceil_div(npix * (Cp / 8), NTHR);
    if (grid > 4096) grid = 4096;
    SEGNB_DISPATCH_T(
        hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)a, ld_a,
                           (const bf16_t*)b, ld_b, (bf16_t*)out, ld_out, npix, Cp / 8),
        hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const float*)a, ld_a,
                           (const float*)b, ld_b, (float*)out, ld_out, npix, Cp / 8),
        "segnb_add")
    SEGNB_LAUNCH_CHECK();
    return 0;
}

static int launch_bn_stats(int dtype, const void* x, int ld, int N, int H, int W, int Cp, double* stats, int stats_ld,
                           const char* who, segnb_stream_t stream) {
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(x && stats && stats_ld >= Cp, "NULL tensor / statistics stride below the channel count");
    const EwShape s = make_shape(N, H, W, Cp);
    const dim3 grid = make_grid(s, (long long)N * H * W);
    if (dtype == SEGNB_BF16)
        hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)x, ld, s, stats, stats_ld);
    else if (dtype == SEGNB_F32)
        hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(NTHR), 0, (hipStream_t)stream, (const float*)x, ld, s, stats, stats_ld);
    else {
        segnb_set_error("%s: unknown dtype %d", who, dtype);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_bn_stats(int dtype, const void* x, int ld, int N, int H, int W, int Cp, double* stats,
                              segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_stats, dtype, x, ld, N, H, W, Cp, stats, stream);
    return launch_bn_stats(dtype, x, ld, N, H, W, Cp, stats, Cp, "segnb_bn_stats", stream);
}

extern "C" int segnb_bn_stats_ld(int dtype, const void* x, int ld, int N, int H, int W, int Cp, double* stats, int stats_ld,
                                 segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_bn_stats_ld, dtype, x, ld, N, H, W, Cp, stats, stats_ld, stream);
    return launch_bn_stats(dtype, x, ld, N, H, W, Cp, stats, stats_ld, "segnb_bn_stats_ld", stream);
}

// ------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode='bilinear') of the DecoderBlock's non-deconvolution branch (lib/models/unet16.py:42-46;
// align_corners=False, torch's default): output row 2i = 0.25 in[i-1] + 0.75 in[i], row 2i+1 = 0.75 in[i] + 0.25 in[i+1] with the
// neighbour index clamped to the image (columns alike).  fp32 arithmetic in torch's order -- rows of column-interpolated values
// -- rounded once.  The backward is the exact transpose in gather form: an input pixel collects from the <= 4 x 4 outputs that
// read it, so no atomics and a fixed summation order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilin_src(int o, int n, int& i0, int& i1, float& w0, float& w1) {
    const int i = o >> 1;
    if (o & 1) { i0 = i; i1 = i + 1 < n ? i + 1 : n - 1; w0 = 0.75f; w1 = 0.25f; }
    else       { i0 = i > 0 ? i - 1 : 0; i1 = i; w0 = 0.25f; w1 = 0.75f; }
}
template <typename T>
__global__ __launch_bounds__(NTHR) void upsample_bilinear2x_fwd_kernel(const T* __restrict__ x, int ld_x, int N, int H, int W, int CPP,
                                                                       T* __restrict__ out, int ld_out) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * Ho * Wo * CPP;
    for (long long i = blockIdx.x * (long long)NTHR + threadIdx.x; i < total; i += (long long)gridDim.x * NTHR) {
        const int cc = (int)(i % CPP);
        long long q = i / CPP;
        const int ox = (int)(q % Wo);
        q /= Wo;
        const int oy = (int)(q % Ho);
        const int n = (int)(q / Ho);
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        bilin_src(oy, H, y0, y1, wy0, wy1);
        bilin_src(ox, W, x0, x1, wx0, wx1);
        float a[8], b[8], c[8], d[8], v[8];
        const T* base = x + (long long)n * H * W * ld_x + cc * 8;
        load8(base + ((long long)y0 * W + x0) * ld_x, a);
        load8(base + ((long long)y0 * W + x1) * ld_x, b);
        load8(base + ((long long)y1 * W + x0) * ld_x, c);
        load8(base + ((long long)y1 * W + x1) * ld_x, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = wy0 * (wx0 * a[e] + wx1 * b[e]) + wy1 * (wx0 * c[e] + wx1 * d[e]);
        store8(out + (((long long)n * Ho + oy) * Wo + ox) * ld_out + cc * 8, v);
    }
}
template <typename T>
__global__ __launch_bounds__(NTHR) void upsample_bilinear2x_bwd_kernel(const T* __restrict__ go, int ld_go, int N, int H, int W, int CPP,
                                                                       T* __restrict__ dx, int ld_dx) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * H * W * CPP;
    for (long long i = blockIdx.x * (long long)NTHR + threadIdx.x; i < total; i += (long long)gridDim.x * NTHR) {
        const int cc = (int)(i % CPP);
        long long q = i / CPP;
        const int ix = (int)(q % W);
        q /= W;
        const int iy = (int)(q % H);
        const int n = (int)(q / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int oy = 2 * iy - 2; oy <= 2 * iy + 3; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            int y0, y1;
            float wy0, wy1;
            bilin_src(oy, H, y0, y1, wy0, wy1);
            const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = 2 * ix - 2; ox <= 2 * ix + 3; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                int x0, x1;
                float wx0, wx1;
                bilin_src(ox, W, x0, x1, wx0, wx1);
                const float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
                if (wx == 0.f) continue;
                float g[8];
                load8(go + (((long long)n * Ho + oy) * Wo + ox) * ld_go + cc * 8, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (wy * wx) * g[e];
            }
        }
        store8(dx + (((long long)n * H + iy) * W + ix) * ld_dx + cc * 8, acc);
    }
}

extern "C" int segnb_upsample_bilinear2x_fwd(int dtype, const void* x, int ld_x, int N, int H, int W, int Cp, void* out, int ld_out,
                                             segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_upsample_bilinear2x_fwd, dtype, x, ld_x, N, H, W, Cp, out, ld_out, stream);
    if (int rc = check_ew(N, 2 * H, 2 * W, Cp)) return rc;
    SEGNB_CHECK_ARG(x && out && ld_x >= Cp && ld_out >= Cp, "bad arguments");
    int grid = ceil_div((long long)N * 4 * H * W * (Cp / 8), NTHR);
    if (grid > 16384) grid = 16384;
    SEGNB_DISPATCH_T(
        hipLaunchKernelGGL(upsample_bilinear2x_fwd_kernel<bf16_t>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)x,
                           ld_x, N, H, W, Cp / 8, (bf16_t*)out, ld_out),
        hipLaunchKernelGGL(upsample_bilinear2x_fwd_kernel<float>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const float*)x,
                           ld_x, N, H, W, Cp / 8, (float*)out, ld_out),
        "segnb_upsample_bilinear2x_fwd")
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_upsample_bilinear2x_bwd(int dtype, const void* g_out, int ld_go, int N, int H, int W, int Cp, void* dx,
                                             int ld_dx, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_upsample_bilinear2x_bwd, dtype, g_out, ld_go, N, H, W, Cp, dx, ld_dx, stream);
    if (int rc = check_ew(N, 2 * H, 2 * W, Cp)) return rc;
    SEGNB_CHECK_ARG(g_out && dx && ld_go >= Cp && ld_dx >= Cp, "bad arguments");
    int grid = ceil_div((long long)N * H * W * (Cp / 8), NTHR);
    if (grid > 16384) grid = 16384;
    SEGNB_DISPATCH_T(
        hipLaunchKernelGGL(upsample_bilinear2x_bwd_kernel<bf16_t>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)g_out,
                           ld_go, N, H, W, Cp / 8, (bf16_t*)dx, ld_dx),
        hipLaunchKernelGGL(upsample_bilinear2x_bwd_kernel<float>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const float*)g_out,
                           ld_go, N, H, W, Cp / 8, (float*)dx, ld_dx),
        "segnb_upsample_bilinear2x_bwd")
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_maxpool_fwd(int dtype, const void* x, int ld_x, int N, int H, int W, int Cp, int k, int stride,
                                 int pad, void* out, int ld_out, unsigned char* idx, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_maxpool_fwd, dtype, x, ld_x, N, H, W, Cp, k, stride, pad, out, ld_out, idx, stream);
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(x && out && k >= 1 && stride >= 1 && pad >= 0 && pad < k, "bad pooling arguments");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    SEGNB_CHECK_ARG(Ho > 0 && Wo > 0, "empty pooled output");
    int grid = ceil_div((long long)N * Ho * Wo * (Cp / 8), NTHR);
    if (grid > 8192) grid = 8192;
    SEGNB_DISPATCH_T(
        hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)x,
                           ld_x, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (bf16_t*)out, ld_out, idx),
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const float*)x,
                           ld_x, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (float*)out, ld_out, idx),
        "segnb_maxpool_fwd")
    SEGNB_LAUNCH_CHECK();
    return 0;
}

static int maxpool_bwd_impl(int dtype, const void* x, int ld_x, const void* g_out, int ld_go, int N, int H, int W, int Cp, int k,
                            int stride, int pad, void* dx, int ld_dx, const unsigned char* idx, const void* g_out2, int ld_go2,
                            segnb_stream_t stream);

extern "C" int segnb_maxpool_bwd(int dtype, const void* x, int ld_x, const void* g_out, int ld_go, int N, int H, int W,
                                 int Cp, int k, int stride, int pad, void* dx, int ld_dx, const unsigned char* idx,
                                 segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_maxpool_bwd, dtype, x, ld_x, g_out, ld_go, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, stream);
    return maxpool_bwd_impl(dtype, x, ld_x, g_out, ld_go, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, nullptr, 0, stream);
}

extern "C" int segnb_maxpool_bwd_add(int dtype, const void* x, int ld_x, const void* g_out, int ld_go, const void* g_out2,
                                     int ld_go2, int N, int H, int W, int Cp, int k, int stride, int pad, void* dx, int ld_dx,
                                     const unsigned char* idx, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_maxpool_bwd_add, dtype, x, ld_x, g_out, ld_go, g_out2, ld_go2, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, stream);
    SEGNB_CHECK_ARG(g_out2 != nullptr && idx != nullptr, "the two-source form takes the recorded argmax positions");
    return maxpool_bwd_impl(dtype, x, ld_x, g_out, ld_go, N, H, W, Cp, k, stride, pad, dx, ld_dx, idx, g_out2, ld_go2, stream);
}

static int maxpool_bwd_impl(int dtype, const void* x, int ld_x, const void* g_out, int ld_go, int N, int H, int W, int Cp, int k,
                            int stride, int pad, void* dx, int ld_dx, const unsigned char* idx, const void* g_out2, int ld_go2,
                            segnb_stream_t stream) {
    if (int rc = check_ew(N, H, W, Cp)) return rc;
    SEGNB_CHECK_ARG(x && g_out && dx && k >= 1 && stride >= 1 && pad >= 0 && pad < k && k * k < 255, "bad pooling arguments");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    int grid = ceil_div((long long)N * H * W * (Cp / 8), NTHR);
    if (grid > 8192) grid = 8192;
    if (idx != nullptr) {
        const bool i32 = (long long)N * H * W * (Cp / 8) < (1ll << 31);
        if (i32) {
            SEGNB_DISPATCH_T(
                hipLaunchKernelGGL((maxpool_bwd_idx_kernel<bf16_t, true>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, idx,
                                   (const bf16_t*)g_out, ld_go, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (bf16_t*)dx, ld_dx,
                                   (const bf16_t*)g_out2, ld_go2),
                hipLaunchKernelGGL((maxpool_bwd_idx_kernel<float, true>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, idx,
                                   (const float*)g_out, ld_go, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (float*)dx, ld_dx,
                                   (const float*)g_out2, ld_go2),
                "segnb_maxpool_bwd")
            SEGNB_LAUNCH_CHECK();
            return 0;
        }
        SEGNB_DISPATCH_T(
            hipLaunchKernelGGL((maxpool_bwd_idx_kernel<bf16_t, false>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, idx,
                               (const bf16_t*)g_out, ld_go, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (bf16_t*)dx, ld_dx,
                               (const bf16_t*)g_out2, ld_go2),
            hipLaunchKernelGGL((maxpool_bwd_idx_kernel<float, false>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, idx,
                               (const float*)g_out, ld_go, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (float*)dx, ld_dx,
                               (const float*)g_out2, ld_go2),
            "segnb_maxpool_bwd")
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    SEGNB_DISPATCH_T(
        hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const bf16_t*)x,
                           ld_x, (const bf16_t*)g_out, ld_go, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (bf16_t*)dx, ld_dx),
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, (const float*)x,
                           ld_x, (const float*)g_out, ld_go, N, H, W, Cp / 8, k, stride, pad, Ho, Wo, (float*)dx, ld_dx),
        "segnb_maxpool_bwd")
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_nhwc_to_nchw_f32(int dtype, const void* a, int ld, int N, int H, int W, int C, float* out,
                                      segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_nhwc_to_nchw_f32, dtype, a, ld, N, H, W, C, out, stream);
    SEGNB_CHECK_ARG(a && out && N > 0 && H > 0 && W > 0 && C > 0 && ld >= C, "bad arguments");
    int grid = ceil_div((long long)N * C * H * W, 256);
    if (grid > 8192) grid = 8192;
    SEGNB_DISPATCH_T(
        hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)a, ld, N, H, W, C, out),
        hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)a, ld, N, H, W, C, out),
        "segnb_nhwc_to_nchw_f32")
    SEGNB_LAUNCH_CHECK();
    return 0;
}
