// 1x1 classifier head (few classes) and the per-pixel binary losses / metrics, gfx950.
//
// head : nn.Conv2d(filters, num_classes, 1) of lib/models/zf_unet.py:58,93 (tiramisu.py:162,
//        unet16.py:111): NHWC activations in, fp32 NCHW logits out (the reference's output layout).
// loss : lib/losses.py:7-101 and lib/metrics.py:9-43 -- one streaming pass produces every global
//        sum any of the losses/metrics needs; a second pass writes d(loss)/d(logits).
#include "common.h"

#include <mutex>
#include <vector>

namespace {

constexpr int MAXK = 8;  // classes handled by the direct head kernels

// ------------------------------------------------------------------------------------------------
// head forward: one thread per pixel, weights in LDS
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a, int ld_a, long long npix,
                                                       long long hw, int C, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int K,
                                                       float* __restrict__ logits) {
    extern __shared__ float sw[];  // [K][C8]
    const int C8 = (C + 7) & ~7;
    for (int i = threadIdx.x; i < K * C8; i += blockDim.x) {
        const int k = i / C8, c = i - k * C8;
        sw[i] = c < C ? w[k * C + c] : 0.f;
    }
    __syncthreads();
    for (long long pix = blockIdx.x * (long long)blockDim.x + threadIdx.x; pix < npix;
         pix += (long long)gridDim.x * blockDim.x) {
        float acc[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) acc[k] = 0.f;
        for (int c0 = 0; c0 < C8; c0 += 8) {
            float v[8];
            load8(a + pix * ld_a + c0, v);
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[k] = fmaf(v[e], sw[k * C8 + c0 + e], acc[k]);
                }
        }
        const long long n = pix / hw, r = pix - n * hw;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) logits[(n * K + k) * hw + r] = acc[k] + (bias != nullptr ? bias[k] : 0.f);
    }
}

// head forward for WIDE activations (C > 64: the FCDenseNet head reads 256 channels, tiramisu.py:162-164): with one thread per pixel a
// load instruction of a wave touched 64 different 512-byte pixels (1.1 TB/s); here CT consecutive lanes hold the 8-channel chunks
// of ONE pixel (a wave reads whole pixels, contiguous), each lane walks the chunks tx, tx + CT, ... and the lanes' partial dot
// products meet in a shuffle tree.  KM = compile-time bound of the class count.
template <typename T, int KM>
__global__ __launch_bounds__(256) void head_fwd_wide_kernel(const T* __restrict__ a, int ld_a, long long npix, long long hw, int C,
                                                            const float* __restrict__ w, const float* __restrict__ bias, int K,
                                                            float* __restrict__ logits, int CT) {
    extern __shared__ float sww[];  // [K][C8] (zero-padded)
    const int PY = 256 / CT;
    const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const int CPP = (C + 7) >> 3, C8 = CPP * 8;
    for (int i = threadIdx.x; i < K * C8; i += blockDim.x) {
        const int k = i / C8, c = i - k * C8;
        sww[i] = c < C ? w[k * C + c] : 0.f;
    }
    __syncthreads();
    for (long long pix = (long long)blockIdx.x * PY + ty; pix < npix; pix += (long long)gridDim.x * PY) {
        float acc[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) acc[k] = 0.f;
        for (int cc = tx; cc < CPP; cc += CT) {
            float v[8];
            load8(a + pix * ld_a + cc * 8, v);
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    const float4 w0 = *reinterpret_cast<const float4*>(sww + k * C8 + cc * 8);
                    const float4 w1 = *reinterpret_cast<const float4*>(sww + k * C8 + cc * 8 + 4);
                    acc[k] = fmaf(v[0], w0.x, acc[k]); acc[k] = fmaf(v[1], w0.y, acc[k]);
                    acc[k] = fmaf(v[2], w0.z, acc[k]); acc[k] = fmaf(v[3], w0.w, acc[k]);
                    acc[k] = fmaf(v[4], w1.x, acc[k]); acc[k] = fmaf(v[5], w1.y, acc[k]);
                    acc[k] = fmaf(v[6], w1.z, acc[k]); acc[k] = fmaf(v[7], w1.w, acc[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < KM; ++k)
            for (int off = 1; off < CT; off <<= 1) acc[k] += __shfl_xor(acc[k], off);
        if (tx == 0) {
            const long long n = pix / hw, r = pix - n * hw;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) logits[(n * K + k) * hw + r] = acc[k] + (bias != nullptr ? bias[k] : 0.f);
        }
    }
}

// head backward: da[pix][c] = sum_k dl[pix][k] w[k][c]; dw[k][c] += sum_pix dl[pix][k] a[pix][c]; db[k] += sum dl
// thread (tx = 8-channel chunk, ty = pixel lane), same mapping as the norm/act kernels.
// KM = compile-time bound of the class count: 1 for binary heads (the per-class register arrays sized for MAXK = 8
// classes left two waves per SIMD: 2.2 TB/s on a streaming pass), MAXK otherwise
template <typename T, int KM>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ a, int ld_a, long long npix,
                                                       long long hw, int C, int Cp, const float* __restrict__ w,
                                                       int K, const float* __restrict__ dl, T* __restrict__ da,
                                                       int ld_da, float* __restrict__ part, int CT) {
    __shared__ float sred[256 * 8];
    const int PY = 256 / CT;
    const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const int cc = blockIdx.y * CT + tx;
    const bool active = cc * 8 < Cp;
    const int c0 = active ? cc * 8 : 0;
    float wv[KM][8];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[k][e] = (k < K && c0 + e < C) ? w[k * C + c0 + e] : 0.f;
    float gw[KM][8];
    float gb[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        gb[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) gw[k][e] = 0.f;
    }
    if (active) {
        // four pixels per trip: all their loads go out before the first use (one pixel per trip left one dependent
        // load chain per wave in flight: 2 TB/s on a pure streaming pass); K == 1 needs no pixel -> (n, r) division
        const int stride = gridDim.x * PY;
        for (int pix0 = blockIdx.x * PY + ty; pix0 < (int)npix; pix0 += 4 * stride) {     // npix < 2^31 (checked)
            float av[4][8], g[4][KM];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pix = pix0 + u * stride;
                ok[u] = pix < (int)npix;
                const int pc = ok[u] ? pix : pix0;
                load8(a + (long long)pc * ld_a + c0, av[u]);
                if (KM == 1) {
                    g[u][0] = dl[pc];
                } else {
                    const int n = pc / (int)hw, r = pc - n * (int)hw;
#pragma unroll
                    for (int k = 0; k < KM; ++k) g[u][k] = k < K ? dl[((long long)n * K + k) * hw + r] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = 0.f;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        const float gk = g[u][k];
                        gb[k] += gk;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            d[e] = fmaf(gk, wv[k][e], d[e]);
                            gw[k][e] = fmaf(gk, av[u][e], gw[k][e]);
                        }
                    }
                if (da != nullptr) store8(da + (long long)(pix0 + u * stride) * ld_da + c0, d);
            }
        }
    }
    // ---- reproducible reduction (LDS and global float atomics made dw / db differ run to run) ----
    // within the block: the PY pixel lanes of a channel chunk are summed in lane order through LDS, class by class;
    // across blocks: every block writes its partial sums to its own row of `part`, head_bwd_finish_kernel adds them in
    // block order.
    float* sg = sred;                                   // [PY][CT][8] floats = 8 KB
    float* prow = part + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * (K * (CT * 8 + 1));
#pragma unroll
    for (int k = 0; k < KM; ++k)
        if (k < K) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sg[(ty * CT + tx) * 8 + e] = gw[k][e];
            __syncthreads();
            if (threadIdx.x < CT * 8) {
                float sum = 0.f;
                for (int j = 0; j < PY; ++j) sum += sg[j * CT * 8 + threadIdx.x];
                prow[k * (CT * 8 + 1) + threadIdx.x] = sum;
            }
            __syncthreads();
            // bias: the pixel lanes of chunk column 0 hold the same pixels as every other column
            if (tx == 0) sg[ty] = gb[k];
            __syncthreads();
            if (threadIdx.x == 0) {
                float sum = 0.f;
                for (int j = 0; j < PY; ++j) sum += sg[j];
                prow[k * (CT * 8 + 1) + CT * 8] = sum;
            }
        }
}


// ------------------------------------------------------------------------------------------------
// classifier head that is a SMALL CONVOLUTION over an activated tensor (linknet.py:62: Conv2d(32, classes, 2, padding=1); the
// 1 x 1 heads are the one-position case).  Memory-bound: the general implicit-GEMM kernels spend a 32-wide channel tile on one
// class (6 TFLOP/s, 2-3 x the time of one pass over the activations).
//   forward : one thread per OUTPUT pixel, weights [class][position][C8] in LDS; the window positions re-read their neighbours'
//             pixels from cache
//   backward: head_bwd_kernel's mapping over the INPUT pixels with K * T virtual classes (class, window position): the gradient
//             of virtual class (k, t) at input pixel (h, w) is dlogits[k] at output pixel (h + pad - t / kw, w + pad - t % kw).
//             act >= 0: the result is dz = round(round(da) * act'(a)) of the convolution + activation that PRODUCED a (no
//             BatchNorm in between: act' has the sign of the activated value) and the per-channel sums of dz go to `sums`
//             (that layer's bias gradient) -- its segnb_bn_act_bwd_reduce pass folded into this one
// ------------------------------------------------------------------------------------------------
template <typename T, int NTM>
__global__ __launch_bounds__(256) void head_conv_fwd_kernel(const T* __restrict__ a, int ld_a, int N, int Hi, int Wi, int C,
                                                            const float* __restrict__ w, int kh, int kw, int pad,
                                                            const float* __restrict__ bias, int K, int Ho, int Wo,
                                                            float* __restrict__ logits, int CT) {
    // CT consecutive lanes hold the 8-channel chunks of ONE pixel (a wave reads whole pixels, contiguous: with one thread per
    // pixel every load instruction touched 64 pixels' lines, and the window multiplies the L1 traffic by its size); all window
    // positions' loads go out before the first use; the lanes' partial dot products meet in a shuffle tree.
    // NTM = compile-time bound of the window size.
    extern __shared__ float swc[];  // [K][T][C8]
    const int CPP = (C + 7) >> 3, C8 = CPP * 8, NT = kh * kw;
    for (int i = threadIdx.x; i < K * NT * C8; i += blockDim.x) {
        const int k = i / (NT * C8), r = i - k * (NT * C8), t = r / C8, c = r - t * C8;
        swc[i] = c < C ? w[((long long)k * C + c) * NT + t] : 0.f;
    }
    __syncthreads();
    const int PY = 256 / CT;
    const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const bool lane_ok = tx < CPP;
    const int hw = Ho * Wo, npix = N * hw;
    for (int pix = blockIdx.x * PY + ty; pix < npix; pix += gridDim.x * PY) {
        const int n = pix / hw, r = pix - n * hw, ho = r / Wo, wo = r - ho * Wo;
        float v[NTM][8];
        bool ok[NTM];
#pragma unroll
        for (int t = 0; t < NTM; ++t) {
            const int hi = ho - pad + t / kw, wi = wo - pad + t % kw;
            ok[t] = t < NT && lane_ok && (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi;
            const long long off = ok[t] ? ((long long)(n * Hi + hi) * Wi + wi) * ld_a + tx * 8 : 0;
            load8(a + off, v[t]);
        }
        float acc[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) acc[k] = 0.f;
#pragma unroll
        for (int t = 0; t < NTM; ++t) {
            if (!ok[t]) continue;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < K) {
                    const float4 w0 = *reinterpret_cast<const float4*>(swc + (k * NT + t) * C8 + tx * 8);
                    const float4 w1 = *reinterpret_cast<const float4*>(swc + (k * NT + t) * C8 + tx * 8 + 4);
                    acc[k] = fmaf(v[t][0], w0.x, acc[k]); acc[k] = fmaf(v[t][1], w0.y, acc[k]);
                    acc[k] = fmaf(v[t][2], w0.z, acc[k]); acc[k] = fmaf(v[t][3], w0.w, acc[k]);
                    acc[k] = fmaf(v[t][4], w1.x, acc[k]); acc[k] = fmaf(v[t][5], w1.y, acc[k]);
                    acc[k] = fmaf(v[t][6], w1.z, acc[k]); acc[k] = fmaf(v[t][7], w1.w, acc[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K)
                for (int off = 1; off < CT; off <<= 1) acc[k] += __shfl_xor(acc[k], off);
        if (tx == 0) {
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < K) logits[((long long)n * K + k) * hw + r] = acc[k] + (bias != nullptr ? bias[k] : 0.f);
        }
    }
}

template <typename T, int KM>
__global__ __launch_bounds__(256) void head_conv_bwd_kernel(const T* __restrict__ a, int ld_a, int N, int Hi, int Wi, int C, int Cp,
                                                            const float* __restrict__ w, int kh, int kw, int pad, int K, int Ho,
                                                            int Wo, const float* __restrict__ dl, int act, float slope,
                                                            T* __restrict__ da, int ld_da, float* __restrict__ part,
                                                            double* __restrict__ sums, int CT) {
    __shared__ float sred[256 * 8];
    const int PY = 256 / CT;
    const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const int cc = blockIdx.y * CT + tx;
    const bool active = cc * 8 < Cp;
    const int c0 = active ? cc * 8 : 0;
    const int NT = kh * kw, KT = K * NT;
    // the pixel walk covers the union of the input and the output grid: every input pixel for da / dw, every output pixel for db
    const int Hu = Hi > Ho ? Hi : Ho, Wu = Wi > Wo ? Wi : Wo;
    const int npix = N * Hu * Wu;
    // weights of the virtual classes: registers for one class (the 1 x 1 binary head), LDS beyond (32 registers per 4 classes
    // cost the streaming pass its third wave per SIMD)
    __shared__ float swv[KM > 1 ? KM * 64 : 1];
    float wv1[8];
    if constexpr (KM > 1) {
        for (int i = threadIdx.x; i < KM * 64; i += 256) {
            const int j = i >> 6, c = blockIdx.y * CT * 8 + (i & 63);
            swv[i] = (j < KT && (i & 63) < CT * 8 && c < C) ? w[((long long)(j / NT) * C + c) * NT + j % NT] : 0.f;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) wv1[e] = (c0 + e < C) ? w[(long long)(c0 + e) * NT] : 0.f;
    }
    float gw[KM][8], gb[KM], sz[8];
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        gb[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) gw[j][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sz[e] = 0.f;
    if (active) {
        // PPT pixels per trip, all their loads out before the first use (as head_bwd_kernel: one dependent load chain per wave in
        // flight streams at 2 TB/s)
        constexpr int PPT = KM <= 4 ? 4 : 2;
        const int stride = gridDim.x * PY;
        for (int pix0 = blockIdx.x * PY + ty; pix0 < npix; pix0 += PPT * stride) {
            float av[PPT][8], g[PPT][KM], gc[PPT][KM];
            bool in_ok[PPT];
            long long ioff[PPT];
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
                const int pix = pix0 + u * stride;
                const bool ok = pix < npix;
                const int pc = ok ? pix : pix0;
                const int n = pc / (Hu * Wu), r = pc - n * (Hu * Wu), h = r / Wu, x = r - h * Wu;
                in_ok[u] = ok && h < Hi && x < Wi;
                ioff[u] = ((long long)(n * Hi + (h < Hi ? h : 0)) * Wi + (x < Wi ? x : 0));
                load8(a + ioff[u] * ld_a + c0, av[u]);
#pragma unroll
                for (int j = 0; j < KM; ++j) {
                    float gv = 0.f;
                    if (j < KT && in_ok[u]) {
                        const int k = j / NT, t = j - k * NT;
                        const int ho = h + pad - t / kw, wo = x + pad - t % kw;
                        if ((unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo)
                            gv = dl[((long long)(n * K + k) * Ho + ho) * Wo + wo];
                    }
                    g[u][j] = gv;
                    // the class's own gradient at output pixel (h, x): the bias gradient (class k rides in row k of the registers)
                    gc[u][j] = (j < K && ok && h < Ho && x < Wo) ? dl[((long long)(n * K + j) * Ho + h) * Wo + x] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
#pragma unroll
                for (int j = 0; j < KM; ++j) gb[j] += gc[u][j];
                if (!in_ok[u]) continue;
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = 0.f;
#pragma unroll
                for (int j = 0; j < KM; ++j)
                    if (j < KT) {
                        const float gj = g[u][j];
                        float wj[8];
                        if constexpr (KM > 1) {
                            const float4 w0 = *reinterpret_cast<const float4*>(swv + j * 64 + tx * 8);
                            const float4 w1 = *reinterpret_cast<const float4*>(swv + j * 64 + tx * 8 + 4);
                            wj[0] = w0.x; wj[1] = w0.y; wj[2] = w0.z; wj[3] = w0.w;
                            wj[4] = w1.x; wj[5] = w1.y; wj[6] = w1.z; wj[7] = w1.w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) wj[e] = wv1[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            d[e] = fmaf(gj, wj[e], d[e]);
                            gw[j][e] = fmaf(gj, av[u][e], gw[j][e]);
                        }
                    }
                if (act >= 0) {
                    const float neg = act == SEGNB_ACT_RELU ? 0.f : (act == SEGNB_ACT_LEAKY ? slope : 1.f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gr = round_as(d[e], da);
                        d[e] = round_as(av[u][e] > 0.f ? gr : gr * neg, da);
                        sz[e] += d[e];
                    }
                }
                if (da != nullptr) store8(da + ioff[u] * ld_da + c0, d);
            }
        }
    }
    // ---- reproducible reduction of dw / db: as head_bwd_kernel (per-block partial sums, summed in block order by the finish kernel)
    float* sg = sred;                                   // [PY][CT][8] floats = 8 KB
    float* prow = part + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * (KT * (CT * 8 + 1));
#pragma unroll
    for (int j = 0; j < KM; ++j)
        if (j < KT) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sg[(ty * CT + tx) * 8 + e] = gw[j][e];
            __syncthreads();
            if (threadIdx.x < CT * 8) {
                float sum = 0.f;
                for (int q = 0; q < PY; ++q) sum += sg[q * CT * 8 + threadIdx.x];
                prow[j * (CT * 8 + 1) + threadIdx.x] = sum;
            }
            __syncthreads();
            // bias of class j (j < K): accumulated in register row j, published in the class's position-0 row j * NT
            if (tx == 0) sg[ty] = gb[j];
            __syncthreads();
            if (threadIdx.x == 0 && j < K) {
                float sum = 0.f;
                for (int q = 0; q < PY; ++q) sum += sg[q];
                prow[(j * NT) * (CT * 8 + 1) + CT * 8] = sum;
            }
        }
    if (act >= 0 && sums != nullptr) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) sg[(ty * CT + tx) * 8 + e] = sz[e];
        __syncthreads();
        if (threadIdx.x < CT * 8) {
            double sum = 0.0;
            for (int q = 0; q < PY; ++q) sum += (double)sg[q * CT * 8 + threadIdx.x];
            const int c = blockIdx.y * CT * 8 + threadIdx.x;
            if (c < Cp) atomicAdd(&sums[(long long)((blockIdx.x % SEGNB_STAT_REPLICAS) * 2) * Cp + c], sum);
        }
    }
}

// dw[k][c] += sum over the pixel blocks (fixed order) of their partial sums; db[k] likewise (channel-chunk row 0).
// One 256-thread block per output element: lanes stride the blocks (up to 2048 rows: 8 dependent loads per lane instead of 32),
// a fixed shuffle tree per wave, the four waves' sums added in wave order.
// T > 1 (head_conv_bwd_kernel): the K rows of a block are K / T classes x T window positions, row j = class j / T, position
// j % T, and dw is the parameter's own layout [class][C][T]; db is taken from the rows of position 0
__global__ __launch_bounds__(256) void head_bwd_finish_kernel(const float* __restrict__ part, int gx, int gy, int K, int C,
                                                              int CT, float* __restrict__ dw, float* __restrict__ db, int T) {
    __shared__ float sw4[4];
    const int o = blockIdx.x;                           // 0 .. K*C-1: weights, K*C .. K*C+K-1: biases
    const int t = threadIdx.x;
    const int row = K * (CT * 8 + 1);
    int k, by, idx;
    int c = 0;
    if (o < K * C) {
        k = o / C;
        c = o - k * C;
        by = c / (CT * 8);
        idx = c - by * (CT * 8);
    } else {
        k = (o - K * C) * T;       // (class o - K C: its position-0 row)
        by = 0;
        idx = CT * 8;
    }
    float sum = 0.f;
    for (int bx = t; bx < gx; bx += 256) sum += part[(long long)(by * gx + bx) * row + k * (CT * 8 + 1) + idx];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((t & 63) == 0) sw4[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
        const float tot = ((sw4[0] + sw4[1]) + sw4[2]) + sw4[3];
        if (o < K * C) {
            if (dw != nullptr) dw[((long long)(k / T) * C + c) * T + k % T] += tot;
        } else if (db != nullptr) {
            db[k / T] += tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------
struct PixTerms {
    float p;    // sigmoid(x)
    float e;    // double-sigmoid BCE element:  -t*logsigmoid(x) + log(1 + sigmoid(x))      (losses.py:51-53)
    float de;   // d e / d x = (1-p) * (p/(1+p) - t)
    float f;    // focal element (1-pt)^2 * e, pt = exp(-e)                                   (losses.py:90-95)
    float df;   // d f / d x
};

__device__ __forceinline__ PixTerms pix_terms(float x, float t, float gamma = 2.f) {
    PixTerms o;
    // logsigmoid(x) = min(x,0) - log1p(exp(-|x|)), as ATen computes it
    const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
    const float p = expf(ls);
    o.p = p;
    o.e = -t * ls + log1pf(p);
    o.de = (1.f - p) * (p / (1.f + p) - t);
    const float pt = expf(-o.e);
    const float om = 1.f - pt;
    if (gamma == 2.f) {                        // the reference default (losses.py:84): exact products, no powf
        o.f = om * om * o.e;
        o.df = (2.f * om * pt * o.e + om * om) * o.de;
    } else if (om > 0.f) {                     // (1-pt)^gamma * e;  d/dx = (gamma (1-pt)^(gamma-1) pt e + (1-pt)^gamma) de
        const float pg1 = powf(om, gamma - 1.f), pg = pg1 * om;
        o.f = pg * o.e;
        o.df = (gamma * pg1 * pt * o.e + pg) * o.de;
    } else {
        o.f = gamma == 0.f ? o.e : 0.f;
        o.df = gamma == 0.f ? o.de : 0.f;
    }
    return o;
}

__device__ __forceinline__ void loss_finalize_math(const double* sums, const segnb_loss_spec& sp, float* __restrict__ out) {
    const double n = sums[6];
    const double I = sums[2], U = sums[3] + sums[4];
    const double bce = sp.bce_sum ? sums[0] : sums[0] / n;
    const double focal = sp.focal_mean ? sums[1] / n : sums[1];
    const double eps = (double)sp.eps, sm = (double)sp.smooth;
    const double Dj = U - I + eps, Ds = U - I + sm, Dd = U + eps;
    const double jac = 1.0 - I / Dj;
    const double sjac = 1.0 - (I + sm) / Ds;
    const double dice = 1.0 - 2.0 * I / Dd;
    const double loss = ((double)sp.w_bce * bce + (double)sp.w_focal * focal + (double)sp.w_jaccard * jac +
                         (double)sp.w_sjaccard * sjac + (double)sp.w_dice * dice) / (double)sp.norm;
    // d(loss)/dI and d(loss)/dU of the region terms (before the 1/norm factor)
    const double GI = (double)sp.w_jaccard * (-(U + eps) / (Dj * Dj)) + (double)sp.w_sjaccard * (-(U + 2.0 * sm) / (Ds * Ds)) +
                      (double)sp.w_dice * (-2.0 / Dd);
    const double GU = (double)sp.w_jaccard * (I / (Dj * Dj)) + (double)sp.w_sjaccard * ((I + sm) / (Ds * Ds)) +
                      (double)sp.w_dice * (2.0 * I / (Dd * Dd));
    out[0] = (float)loss;
    out[1] = (float)(I / (U - I + 1e-7));   // JaccardScore, metrics.py:14-20
    out[2] = (float)(sums[5] / n);          // PixelAccuracy, metrics.py:30-40
    out[3] = (float)GI;
    out[4] = (float)GU;
    out[5] = (float)bce;
    out[6] = (float)n;
    out[7] = 0.f;
}

constexpr int LOSS_REPL = 8;       // replicas of the sums in the one-launch form (work buffer: LOSS_REPL * 8 + 1 doubles <= 128)
// FIN: the LAST block to finish (a ticket counter behind the sums) turns the sums into the result vector and leaves the work
// buffer zeroed for the next call -- zero fill, reduction and finalize in one launch (segnb_seg_loss_reduce_finalize)
template <bool FIN>
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ x,
                                                          const long long* __restrict__ tg, long long n,
                                                          double* __restrict__ sums, int vec, float gamma,
                                                          segnb_loss_spec sp, float* __restrict__ fin) {
    double s[6] = {0, 0, 0, 0, 0, 0};
    auto term = [&](float xv, long long tgv) {
        const float t = tgv != 0 ? 1.f : 0.f;
        // the reference multiplies by target.float(): any integer label value; binary masks are 0/1
        const float tf = (float)tgv;
        const PixTerms q = pix_terms(xv, tf, gamma);
        s[0] += q.e;
        s[1] += q.f;
        s[2] += q.p * tf;
        s[3] += q.p;
        s[4] += tf;
        s[5] += ((q.p > 0.5f) == (t != 0.f)) ? 1.0 : 0.0;
    };
    // four consecutive elements per lane and trip, their three 16-byte loads issued together (one element per trip
    // was four dependent round trips per lane: 27 us for 19 MB); vec = both arrays 16-byte aligned
    const long long n4 = vec ? n / 4 : 0;
#pragma unroll 2
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < n4; j += (long long)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[j];
        const longlong2 t01 = reinterpret_cast<const longlong2*>(tg)[2 * j];
        const longlong2 t23 = reinterpret_cast<const longlong2*>(tg)[2 * j + 1];
        term(xv.x, t01.x);
        term(xv.y, t01.y);
        term(xv.z, t23.x);
        term(xv.w, t23.y);
    }
    for (long long i = 4 * n4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        term(x[i], tg[i]);
    __shared__ double sh[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double v = wave_sum(s[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    // FIN: the sums live in LOSS_REPL replicas of 8 doubles (a block adds to replica blockIdx % LOSS_REPL: same-address atomics
    // serialise, and this launch runs up to 2048 blocks), the ticket behind them
    double* const dst = FIN ? sums + (blockIdx.x % LOSS_REPL) * 8 : sums;
    // (FIN: RETURNING atomics -- the wave waits until they are performed at the device's coherence point, so the ticket below is
    // taken after this block's sums without a __threadfence(): a fence writes the XCD's L2 back, and one per block made this
    // launch 52 us at 512 blocks / 125 us at 2048, profiles/r04_ab.txt)
    double ret = 0.0;
    if (threadIdx.x < 6) {
        const double v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        ret = atomicAdd(&dst[threadIdx.x], v);
    }
    if (threadIdx.x == 6 && blockIdx.x == 0) ret = atomicAdd(&dst[6], (double)n);
    if constexpr (FIN) {
        __shared__ int last;
        if (ret == -1.2345e300) fin[7] = 1.f;           // (never: keeps the returned values alive)
        // The hand-over, ONE recipe with the split K of conv_fprop_ws_kernel (fprop_dma.hip: ks_publish; MI355X_MICROARCH.md,
        // inter-workgroup visibility):
        //   release side  -- the block's contribution is performed at the device's coherence point (there: write-through `sc1`
        //                    stores; here: RETURNING agent-scope atomics), every contributing wave drains (s_waitcnt vmcnt(0): the
        //                    returned values have arrived, so the adds are done), a workgroup barrier, then ONE relaxed agent-scope
        //                    add draws the ticket;
        //   acquire side  -- the block whose ticket is the last one issues an agent-scope ACQUIRE fence (buffer_inv sc1 + wait)
        //                    before it reads, and reads through coherent (agent-scope atomic) loads.
        // A release FENCE per block instead (what __threadfence() compiles to) writes the XCD's L2 back: 52 us at 512 blocks / 125 us
        // at 2048 for this launch (profiles/r04_ab.txt); tests/test_hip_ops.py pins the result against the three-launch form at
        // 512 .. 2048 blocks
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* ticket = reinterpret_cast<unsigned*>(sums + LOSS_REPL * 8);
            const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = tk == gridDim.x - 1 ? 1 : 0;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (last) {
            // the LAST block: every replica value through one coherent load per thread (56 loads in flight at once -- read back one
            // by one with returning atomics they were 56 dependent round trips to the coherence point: ~50 us), summed by thread 0
            __shared__ double stot[LOSS_REPL * 8];
            if (threadIdx.x < LOSS_REPL * 8) {
                stot[threadIdx.x] = __hip_atomic_load(&sums[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sums[threadIdx.x] = 0.0;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double tot[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    tot[k] = 0.0;
                    for (int rp = 0; rp < LOSS_REPL; ++rp) tot[k] += stot[rp * 8 + k];
                }
                loss_finalize_math(tot, sp, fin);
                *reinterpret_cast<unsigned*>(sums + LOSS_REPL * 8) = 0u;
            }
        }
    }
}

__global__ void loss_finalize_kernel(const double* __restrict__ sums, segnb_loss_spec sp, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_finalize_math(sums, sp, out);
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ tg,
                                                       long long n, const float* __restrict__ fin, segnb_loss_spec sp,
                                                       const float* __restrict__ grad_out, float* __restrict__ dx,
                                                       int vec) {
    const float go = (grad_out != nullptr ? grad_out[0] : 1.f) / sp.norm;
    const float GI = fin[3], GU = fin[4];
    const float inv_n = 1.f / fin[6];   // GLOBAL pixel count (== n on one GPU; all-reduced sums in a DP job)
    const float wb = sp.bce_sum ? sp.w_bce : sp.w_bce * inv_n;
    const float wf = sp.focal_mean ? sp.w_focal * inv_n : sp.w_focal;
    auto grad = [&](float xv, long long tgv) {
        const float tf = (float)tgv;
        const PixTerms q = pix_terms(xv, tf, sp.focal_gamma);
        const float dp = q.p * (1.f - q.p);
        return go * (wb * q.de + wf * q.df + dp * (tf * GI + GU));
    };
    const long long n4 = vec ? n / 4 : 0;
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < n4; j += (long long)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[j];
        const longlong2 t01 = reinterpret_cast<const longlong2*>(tg)[2 * j];
        const longlong2 t23 = reinterpret_cast<const longlong2*>(tg)[2 * j + 1];
        float4 o;
        o.x = grad(xv.x, t01.x);
        o.y = grad(xv.y, t01.y);
        o.z = grad(xv.z, t23.x);
        o.w = grad(xv.w, t23.y);
        reinterpret_cast<float4*>(dx)[j] = o;
    }
    for (long long i = 4 * n4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        dx[i] = grad(x[i], tg[i]);
}


// reduce=False forms (losses.py:53 with reduce=False): the per-pixel loss map and its backward
__global__ __launch_bounds__(256) void loss_map_kernel(const float* __restrict__ x, const long long* __restrict__ tg,
                                                       long long n, int kind, float gamma, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const PixTerms q = pix_terms(x[i], (float)tg[i], gamma);
        out[i] = kind == 0 ? q.e : q.f;
    }
}

__global__ __launch_bounds__(256) void loss_map_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ tg,
                                                           long long n, int kind, float gamma,
                                                           const float* __restrict__ gout, float* __restrict__ dx) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const PixTerms q = pix_terms(x[i], (float)tg[i], gamma);
        dx[i] = gout[i] * (kind == 0 ? q.de : q.df);
    }
}

// max |x| over a flat fp32 buffer (the gradient-explosion monitor of torch_train.py:199-205: one launch over the
// flat gradient buffer instead of one reduction + one host sync per parameter tensor).  |x| as uint bits is monotone.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    const long long n4 = n / 4;                      // flat buffers are 16-byte aligned (FlatParams)
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < n4; j += (long long)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(x)[j];
        m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
    for (long long i = 4 * n4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m != 0) atomicMax(out, m);
}

// PRCurveMeter.update (lib/train_utils.py:109-125): per pixel, bucket = number of thresholds strictly below
// sigmoid(x); one histogram per class of the target.  hist: [2][nthr + 1] unsigned 64-bit, accumulated.
__global__ __launch_bounds__(256) void pr_hist_kernel(const float* __restrict__ x, const long long* __restrict__ tg,
                                                      long long n, const float* __restrict__ thr, int nthr,
                                                      unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned sm[];                 // [2][nthr + 1] counters, then the thresholds
    unsigned* cnt = sm;
    float* th = reinterpret_cast<float*>(sm + 2 * (nthr + 1));
    for (int i = threadIdx.x; i < 2 * (nthr + 1); i += blockDim.x) cnt[i] = 0;
    for (int i = threadIdx.x; i < nthr; i += blockDim.x) th[i] = thr[i];
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float xv = x[i];
        const float p = expf(fminf(xv, 0.f) - log1pf(expf(-fabsf(xv))));       // the loss kernels' sigmoid
        int lo = 0, hi = nthr;                       // first index with th[idx] >= p  ==  #thresholds < p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (th[mid] < p) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&cnt[(tg[i] != 0 ? nthr + 1 : 0) + lo], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (nthr + 1); i += blockDim.x)
        if (cnt[i]) atomicAdd(&hist[i], (unsigned long long)cnt[i]);
}

}  // namespace

extern "C" int segnb_head_fwd(int dtype, const void* a, int ld_a, int N, int H, int W, int C, const float* w,
                              const float* bias, int K, float* logits, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_head_fwd, dtype, a, ld_a, N, H, W, C, w, bias, K, logits, stream);
    SEGNB_CHECK_ARG(a && w && logits, "NULL tensor");
    SEGNB_CHECK_ARG(K >= 1 && K <= MAXK, "head supports 1..8 classes");
    SEGNB_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ld_a % 8 == 0 && ld_a >= ((C + 7) & ~7), "bad shape");
    const long long npix = (long long)N * H * W;
    if (C > 64) {
        // wide activations: chunk-per-lane walk (whole pixels per wave)
        const int cpp = (C + 7) / 8;
        int ct = 1;
        while (ct < cpp && ct < 32) ct <<= 1;
        const int py = 256 / ct;
        long long gx = (npix + py - 1) / py;
        if (gx > 4096) gx = 4096;
        const size_t wsm = (size_t)K * cpp * 8 * sizeof(float);
        SEGNB_CHECK_ARG(wsm <= 48 * 1024, "head: K * C too large");
        if (dtype == SEGNB_BF16)
            (K == 1 ? head_fwd_wide_kernel<bf16_t, 1> : head_fwd_wide_kernel<bf16_t, MAXK>)<<<dim3((unsigned)gx), dim3(256), wsm, (hipStream_t)stream>>>(
                (const bf16_t*)a, ld_a, npix, (long long)H * W, C, w, bias, K, logits, ct);
        else if (dtype == SEGNB_F32)
            (K == 1 ? head_fwd_wide_kernel<float, 1> : head_fwd_wide_kernel<float, MAXK>)<<<dim3((unsigned)gx), dim3(256), wsm, (hipStream_t)stream>>>(
                (const float*)a, ld_a, npix, (long long)H * W, C, w, bias, K, logits, ct);
        else {
            segnb_set_error("segnb_head_fwd: unknown dtype %d", dtype);
            return SEGNB_E_BADARG;
        }
        SEGNB_LAUNCH_CHECK();
        return 0;
    }
    int grid = ceil_div(npix, 256);
    if (grid > 4096) grid = 4096;
    const int smem = K * ((C + 7) & ~7) * 4;
    SEGNB_CHECK_ARG(smem <= 60 * 1024, "head too wide");
    if (dtype == SEGNB_BF16)
        hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(grid), dim3(256), smem, (hipStream_t)stream,
                           (const bf16_t*)a, ld_a, npix, (long long)H * W, C, w, bias, K, logits);
    else if (dtype == SEGNB_F32)
        hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(grid), dim3(256), smem, (hipStream_t)stream, (const float*)a,
                           ld_a, npix, (long long)H * W, C, w, bias, K, logits);
    else {
        segnb_set_error("segnb_head_fwd: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    SEGNB_LAUNCH_CHECK();
    return 0;
}

namespace {
// device scratch of the head backward: one buffer per (device, stream), grown (never shrunk) on demand, behind a mutex.
// Two head backwards that overlap on DIFFERENT streams (two models, a tape on a side stream) get different buffers; launches on
// one stream are ordered by the stream.  hipMalloc happens on the first call of a geometry, i.e. in the eager / recording step;
// replayed launch lists find the same pointer.
float* head_scratch(size_t bytes, hipStream_t stream) {
    struct Ent {
        int dev;
        hipStream_t stream;
        float* buf;
        size_t cap;
    };
    static std::mutex mu;
    static std::vector<Ent> ents;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        segnb_set_error("segnb_head_bwd: hipGetDevice failed");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mu);
    Ent* e = nullptr;
    for (auto& it : ents)
        if (it.dev == dev && it.stream == stream) e = &it;
    // A stream that is being CAPTURED into a HIP graph cannot allocate (hipMalloc invalidates the capture): the graph takes the
    // buffer the eager warm-up steps of this device used (bench.py --graph on / torch.cuda.graph run eager steps first; the
    // graph's replays and eager steps of the same model never overlap)
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    if ((e == nullptr || bytes > e->cap) && hipStreamIsCapturing(stream, &cap_st) == hipSuccess &&
        cap_st == hipStreamCaptureStatusActive) {
        for (auto& it : ents)
            if (it.dev == dev && it.cap >= bytes) return it.buf;
        segnb_set_error("segnb_head_bwd: no scratch buffer to capture with -- run one eager step before capturing a graph");
        return nullptr;
    }
    if (e == nullptr) {
        ents.push_back(Ent{dev, stream, nullptr, 0});
        e = &ents.back();
    }
    if (bytes > e->cap) {
        // (the old buffer may still be read by launches in flight: it is left allocated -- a few hundred KB, at most a
        // handful of times per process)
        const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
        float* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            segnb_set_error("segnb_head_bwd: scratch allocation of %zu bytes failed", want);
            return nullptr;
        }
        e->buf = p;
        e->cap = want;
    }
    return e->buf;
}
}  // namespace

// (for segnb_head_bn_bwd, norm_act.hip: the same partial-sum protocol)
float* segnb_head_scratch(size_t bytes, hipStream_t stream) { return head_scratch(bytes, stream); }
void segnb_head_bwd_finish(const float* part, int gx, int gy, int K, int C, int CT, float* dw, float* db, hipStream_t stream) {
    head_bwd_finish_kernel<<<dim3(K * C + K), dim3(256), 0, stream>>>(part, gx, gy, K, C, CT, dw, db, 1);
}

extern "C" int segnb_head_bwd(int dtype, const void* a, int ld_a, int N, int H, int W, int C, int Cp,
                              const float* w, int K, const float* dlogits, void* da, int ld_da, float* dw,
                              float* db, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_head_bwd, dtype, a, ld_a, N, H, W, C, Cp, w, K, dlogits, da, ld_da, dw, db, stream);
    SEGNB_CHECK_ARG(a && w && dlogits, "NULL tensor");
    SEGNB_CHECK_ARG(K >= 1 && K <= MAXK, "head supports 1..8 classes");
    SEGNB_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && Cp % 8 == 0 && Cp >= C, "bad shape");
    const long long npix = (long long)N * H * W;
    SEGNB_CHECK_ARG(npix < (1ll << 30), "pixel count exceeds the 32-bit index range");
    const int CPP = Cp / 8;
    int ct = 1;
    while (ct < CPP && ct < 32) ct <<= 1;
    const int gy = ceil_div(CPP, ct);
    const int py = 256 / ct;
    long long gx = (npix + py - 1) / py;
    if (gx > 2048 / gy) gx = 2048 / gy;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)gy);
    if (dtype != SEGNB_BF16 && dtype != SEGNB_F32) {
        segnb_set_error("segnb_head_bwd: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    // per-block partial sums of dw / db (summed in block order by head_bwd_finish_kernel): a library-owned scratch buffer
    // per (device, stream), grown on demand
    float* part = head_scratch((size_t)gx * gy * K * (ct * 8 + 1) * sizeof(float), (hipStream_t)stream);
    if (part == nullptr) return SEGNB_E_BADARG;
    if (dtype == SEGNB_BF16)
        (K == 1 ? head_bwd_kernel<bf16_t, 1> : head_bwd_kernel<bf16_t, MAXK>)<<<grid, dim3(256), 0, (hipStream_t)stream>>>(
            (const bf16_t*)a, ld_a, npix, (long long)H * W, C, Cp, w, K, dlogits, (bf16_t*)da, ld_da, part, ct);
    else
        (K == 1 ? head_bwd_kernel<float, 1> : head_bwd_kernel<float, MAXK>)<<<grid, dim3(256), 0, (hipStream_t)stream>>>(
            (const float*)a, ld_a, npix, (long long)H * W, C, Cp, w, K, dlogits, (float*)da, ld_da, part, ct);
    SEGNB_LAUNCH_CHECK();
    if (dw != nullptr || db != nullptr) {
        head_bwd_finish_kernel<<<dim3(K * C + K), dim3(256), 0, (hipStream_t)stream>>>(part, (int)gx, gy, K, C, ct, dw, db, 1);
        SEGNB_LAUNCH_CHECK();
    }
    return 0;
}


extern "C" int segnb_head_conv_ok(int C, int K, int kh, int kw) {
    return (K >= 1 && kh >= 1 && kw >= 1 && K * kh * kw <= MAXK && C >= 1 && C <= 64) ? 1 : 0;
}

extern "C" int segnb_head_conv_fwd(int dtype, const void* a, int ld_a, int N, int Hi, int Wi, int C, const float* w, int kh, int kw,
                                   int pad, const float* bias, int K, float* logits, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_head_conv_fwd, dtype, a, ld_a, N, Hi, Wi, C, w, kh, kw, pad, bias, K, logits, stream);
    SEGNB_CHECK_ARG(a && w && logits, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_head_conv_ok(C, K, kh, kw), "classes x window positions <= 8 and C <= 64 (segnb_head_conv_ok)");
    SEGNB_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && pad >= 0 && ld_a % 8 == 0 && ld_a >= ((C + 7) & ~7), "bad shape");
    const int Ho = Hi + 2 * pad - kh + 1, Wo = Wi + 2 * pad - kw + 1;
    SEGNB_CHECK_ARG(Ho > 0 && Wo > 0, "empty output");
    const long long npix = (long long)N * Ho * Wo;
    SEGNB_CHECK_ARG(npix < (1ll << 30) && (long long)N * Hi * Wi < (1ll << 30), "pixel count exceeds the 32-bit index range");
    const int cpp = (C + 7) / 8;
    int ct = 1;
    while (ct < cpp) ct <<= 1;
    const int py = 256 / ct;
    long long gx = (npix + py - 1) / py;
    if (gx > 16384) gx = 16384;
    const int grid = (int)gx;
    const int smem = K * kh * kw * cpp * 8 * 4;
    const int nt = kh * kw;
#define SEGNB_HEAD_CONV_FWD(TT, NTM_)                                                                                         \
    hipLaunchKernelGGL((head_conv_fwd_kernel<TT, NTM_>), dim3(grid), dim3(256), smem, (hipStream_t)stream, (const TT*)a, ld_a, N, Hi, \
                       Wi, C, w, kh, kw, pad, bias, K, Ho, Wo, logits, ct)
    if (dtype == SEGNB_BF16) {
        if (nt == 1) SEGNB_HEAD_CONV_FWD(bf16_t, 1);
        else if (nt <= 4) SEGNB_HEAD_CONV_FWD(bf16_t, 4);
        else SEGNB_HEAD_CONV_FWD(bf16_t, MAXK);
    } else if (dtype == SEGNB_F32) {
        if (nt == 1) SEGNB_HEAD_CONV_FWD(float, 1);
        else if (nt <= 4) SEGNB_HEAD_CONV_FWD(float, 4);
        else SEGNB_HEAD_CONV_FWD(float, MAXK);
    } else {
        segnb_set_error("segnb_head_conv_fwd: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
#undef SEGNB_HEAD_CONV_FWD
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_head_conv_bwd(int dtype, const void* a, int ld_a, int N, int Hi, int Wi, int C, int Cp, const float* w, int kh,
                                   int kw, int pad, int K, const float* dlogits, int act, float slope, void* da, int ld_da,
                                   float* dw, float* db, double* sums, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_head_conv_bwd, dtype, a, ld_a, N, Hi, Wi, C, Cp, w, kh, kw, pad, K, dlogits, act, slope, da, ld_da, dw, db,
                      sums, stream);
    SEGNB_CHECK_ARG(a && w && dlogits, "NULL tensor");
    SEGNB_CHECK_ARG(segnb_head_conv_ok(C, K, kh, kw), "classes x window positions <= 8 and C <= 64 (segnb_head_conv_ok)");
    SEGNB_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && pad >= 0 && Cp % 8 == 0 && Cp >= C && ld_a % 8 == 0 && ld_a >= Cp, "bad shape");
    // (the kernel's LDS weight rows are 64 floats wide and indexed by padded channel chunk: a wider padded view would read past them)
    SEGNB_CHECK_ARG(Cp <= 64, "padded channel count above 64 (the weight rows in LDS hold 64 channels)");
    SEGNB_CHECK_ARG(da == nullptr || (ld_da % 8 == 0 && ld_da >= Cp), "bad gradient stride");
    SEGNB_CHECK_ARG(act < 0 || ((act == SEGNB_ACT_NONE || act == SEGNB_ACT_RELU || act == SEGNB_ACT_LEAKY) && da != nullptr),
                    "act: -1 (plain gradient) or the producing layer's activation, with da");
    if (dtype != SEGNB_BF16 && dtype != SEGNB_F32) {
        segnb_set_error("segnb_head_conv_bwd: unknown dtype %d", dtype);
        return SEGNB_E_BADARG;
    }
    const int Ho = Hi + 2 * pad - kh + 1, Wo = Wi + 2 * pad - kw + 1;
    SEGNB_CHECK_ARG(Ho > 0 && Wo > 0, "empty output");
    const int Hu = Hi > Ho ? Hi : Ho, Wu = Wi > Wo ? Wi : Wo;
    const long long npix = (long long)N * Hu * Wu;
    SEGNB_CHECK_ARG(npix < (1ll << 30), "pixel count exceeds the 32-bit index range");
    const int KT = K * kh * kw;
    const int CPP = Cp / 8;
    int ct = 1;
    while (ct < CPP && ct < 32) ct <<= 1;
    const int gy = ceil_div(CPP, ct);
    const int py = 256 / ct;
    long long gx = (npix + py - 1) / py;
    if (gx > 2048 / gy) gx = 2048 / gy;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)gy);
    float* part = head_scratch((size_t)gx * gy * KT * (ct * 8 + 1) * sizeof(float), (hipStream_t)stream);
    if (part == nullptr) return SEGNB_E_BADARG;
#define SEGNB_HEAD_CONV_BWD(TT, KM_)                                                                                              \
    head_conv_bwd_kernel<TT, KM_><<<grid, dim3(256), 0, (hipStream_t)stream>>>((const TT*)a, ld_a, N, Hi, Wi, C, Cp, w, kh, kw, pad, K, \
                                                                              Ho, Wo, dlogits, act, slope, (TT*)da, ld_da, part, sums, ct)
    if (dtype == SEGNB_BF16) {
        if (KT == 1) SEGNB_HEAD_CONV_BWD(bf16_t, 1);
        else if (KT <= 4) SEGNB_HEAD_CONV_BWD(bf16_t, 4);
        else SEGNB_HEAD_CONV_BWD(bf16_t, MAXK);
    } else {
        if (KT == 1) SEGNB_HEAD_CONV_BWD(float, 1);
        else if (KT <= 4) SEGNB_HEAD_CONV_BWD(float, 4);
        else SEGNB_HEAD_CONV_BWD(float, MAXK);
    }
#undef SEGNB_HEAD_CONV_BWD
    SEGNB_LAUNCH_CHECK();
    if (dw != nullptr || db != nullptr) {
        head_bwd_finish_kernel<<<dim3(KT * C + K), dim3(256), 0, (hipStream_t)stream>>>(part, (int)gx, gy, KT, C, ct, dw, db, kh * kw);
        SEGNB_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int segnb_seg_loss_reduce(const float* logits, const long long* target, long long n, float focal_gamma,
                                     double* sums, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_seg_loss_reduce, logits, target, n, focal_gamma, sums, stream);
    SEGNB_CHECK_ARG(logits && target && sums && n > 0, "bad arguments");
    // few blocks: every block ends in six double atomics on the SAME six addresses (1568 blocks = 9.4 k serialised
    // atomics were most of the 27 us)
    int grid = ceil_div(n, 256 * 4);
    if (grid > 512) grid = 512;
    const int vec = (((uintptr_t)logits | (uintptr_t)target) & 15) == 0;
    hipLaunchKernelGGL(loss_reduce_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, n, sums, vec,
                       focal_gamma, segnb_loss_spec{}, (float*)nullptr);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_seg_loss_reduce_finalize(const float* logits, const long long* target, long long n,
                                              const segnb_loss_spec* spec, double* work, float* out, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_seg_loss_reduce_finalize, logits, target, n, spec, work, out, stream);
    SEGNB_CHECK_ARG(logits && target && spec && work && out && n > 0, "bad arguments");
    SEGNB_CHECK_ARG(spec->norm != 0.f, "loss norm must be non-zero");
    int grid = ceil_div(n, 256 * 4);
    if (grid > 512) grid = 512;            // (measured: 512 blocks 23.7 us, 1024: 26.7, 2048: 31.5, 4096: 32.1 -- the per-block atomics)
    const int vec = (((uintptr_t)logits | (uintptr_t)target) & 15) == 0;
    hipLaunchKernelGGL(loss_reduce_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, n, work, vec,
                       spec->focal_gamma, *spec, out);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_seg_loss_finalize(const double* sums, const segnb_loss_spec* spec, float* out,
                                       segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_seg_loss_finalize, sums, spec, out, stream);
    SEGNB_CHECK_ARG(sums && spec && out, "bad arguments");
    SEGNB_CHECK_ARG(spec->norm != 0.f, "loss norm must be non-zero");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, *spec, out);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_seg_loss_bwd(const float* logits, const long long* target, long long n,
                                  const double* sums, const float* fin, const segnb_loss_spec* spec,
                                  const float* grad_out, float* dlogits, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_seg_loss_bwd, logits, target, n, sums, fin, spec, grad_out, dlogits, stream);
    (void)sums;
    SEGNB_CHECK_ARG(logits && target && fin && spec && dlogits && n > 0, "bad arguments");
    int grid = ceil_div(n, 256 * 4);
    if (grid > 2048) grid = 2048;
    const int vec = (((uintptr_t)logits | (uintptr_t)target | (uintptr_t)dlogits) & 15) == 0;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, n, fin, *spec,
                       grad_out, dlogits, vec);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_seg_loss_map(const float* logits, const long long* target, long long n, int kind, float gamma,
                                  float* out, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_seg_loss_map, logits, target, n, kind, gamma, out, stream);
    SEGNB_CHECK_ARG(logits && target && out && n > 0 && (kind == 0 || kind == 1), "bad arguments");
    int grid = ceil_div(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(loss_map_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, n, kind, gamma, out);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_seg_loss_map_bwd(const float* logits, const long long* target, long long n, int kind, float gamma,
                                      const float* grad_out, float* dlogits, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_seg_loss_map_bwd, logits, target, n, kind, gamma, grad_out, dlogits, stream);
    SEGNB_CHECK_ARG(logits && target && grad_out && dlogits && n > 0 && (kind == 0 || kind == 1), "bad arguments");
    int grid = ceil_div(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(loss_map_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, n, kind,
                       gamma, grad_out, dlogits);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_absmax_f32(const float* x, long long n, float* out, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_absmax_f32, x, n, out, stream);
    SEGNB_CHECK_ARG(x && out && n > 0 && ((uintptr_t)x & 15) == 0, "bad arguments");
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) {
        segnb_set_error("segnb_absmax_f32: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    int grid = ceil_div(n, 256 * 16);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, reinterpret_cast<unsigned*>(out));
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_pr_histogram(const float* logits, const long long* target, long long n, const float* thresholds,
                                  int nthr, unsigned long long* hist, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_pr_histogram, logits, target, n, thresholds, nthr, hist, stream);
    SEGNB_CHECK_ARG(logits && target && thresholds && hist && n > 0 && nthr > 0 && nthr <= 4096, "bad arguments");
    int grid = ceil_div(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    const int smem = (2 * (nthr + 1) + nthr) * 4;
    hipLaunchKernelGGL(pr_hist_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, logits, target, n, thresholds,
                       nthr, hist);
    SEGNB_LAUNCH_CHECK();
    return 0;
}
