// Tiled inference on the device: lib/tiles.py ImageSlicer.split / merge, the D4 test-time augmentation of
// lib/augmentations.py:476-511 and the sigmoid of inria_submit.py:249, as two gather-form kernels
//   segnb_tiles_gather : padded (reflect-101) image -> batch of D4-transformed tiles, NCHW fp32 (the model input)
//   segnb_tiles_merge  : logits of every (tile, transform) -> weighted, de-augmented, normalised probability map
// Gather form on both sides: no atomics, every output element is produced by one thread in a fixed order (the
// reference's float64 accumulation order over tiles is kept).
#include "common.h"

namespace {

// D4 element k of tta_d4_aug, as an index map: output (r, c) of the S x S transformed tile reads source (sr, sc).
// k: 0 id, 1..3 rot90 x k (counter-clockwise), 4 fliplr, 5..7 fliplr(rot90 x (k-4)).
__device__ __forceinline__ void d4_src(int k, int S, int r, int c, int& sr, int& sc) {
    if (k >= 4) c = S - 1 - c;                    // fliplr is applied last: undo it first
    switch (k & 3) {                              // rot90(m, q)[r][c]
        case 0: sr = r; sc = c; break;
        case 1: sr = c; sc = S - 1 - r; break;
        case 2: sr = S - 1 - r; sc = S - 1 - c; break;
        default: sr = S - 1 - c; sc = r; break;
    }
}

// reflect-101 of a coordinate in [-n+1, 2n-2] into [0, n)
__device__ __forceinline__ int reflect101(int v, int n) {
    if (v < 0) v = -v;
    if (v >= n) v = 2 * n - 2 - v;
    return v;
}

__global__ void tiles_gather_kernel(const float* __restrict__ img, int H, int W, int C, int mt, int ml,
                                    const int* __restrict__ crops, int first, int count, int S,
                                    float* __restrict__ out) {
    // out[b][ch][r][c], b = local index of item (tile, k) = first + b
    const long long total = (long long)count * C * S * S;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % S);
        long long q = i / S;
        const int r = (int)(q % S);
        q /= S;
        const int ch = (int)(q % C);
        const int b = (int)(q / C);
        const int item = first + b, tile = item >> 3, k = item & 7;
        int sr, sc;
        d4_src(k, S, r, c, sr, sc);
        const int y = reflect101(crops[2 * tile + 1] + sr - mt, H);
        const int x = reflect101(crops[2 * tile] + sc - ml, W);
        out[i] = img[((long long)y * W + x) * C + ch];
    }
}

// the same gather from the uint8 image itself, NormalizeImage (lib/augmentations.py:452-460: (x * scale - mean) / std) in
// registers: a 5000 x 5000 x 3 Inria image is uploaded as 75 MB instead of 300 MB and never exists as floats
struct TileNorm {
    float scale, mean[8], inv_std[8];
};
__global__ void tiles_gather_u8_kernel(const unsigned char* __restrict__ img, int H, int W, int C, int mt, int ml,
                                       const int* __restrict__ crops, int first, int count, int S, TileNorm nm,
                                       float* __restrict__ out) {
    const long long total = (long long)count * C * S * S;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % S);
        long long q = i / S;
        const int r = (int)(q % S);
        q /= S;
        const int ch = (int)(q % C);
        const int b = (int)(q / C);
        const int item = first + b, tile = item >> 3, k = item & 7;
        int sr, sc;
        d4_src(k, S, r, c, sr, sc);
        const int y = reflect101(crops[2 * tile + 1] + sr - mt, H);
        const int x = reflect101(crops[2 * tile] + sc - ml, W);
        out[i] = ((float)img[((long long)y * W + x) * C + ch] * nm.scale - nm.mean[ch]) * nm.inv_std[ch];
    }
}

__global__ void tiles_merge_kernel(const float* __restrict__ logits, int K, int S, const int* __restrict__ crops,
                                   int ntiles, int step, int nx, int ny, const double* __restrict__ weight,
                                   int H, int W, int mt, int ml, float* __restrict__ out) {
    // one thread per (pixel, class): the tiles covering a padded pixel form a small grid range (crops are a regular
    // grid of pitch `step`): ty in [ylo, yhi], tx in [xlo, xhi]
    const long long total = (long long)H * W * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kc = (int)(i % K);
        const long long p = i / K;
        const int x = (int)(p % W) + ml, y = (int)(p / W) + mt;           // padded coordinates
        int tyh = y / step, txh = x / step;
        if (tyh > ny - 1) tyh = ny - 1;
        if (txh > nx - 1) txh = nx - 1;
        int tyl = (y - S + step) / step, txl = (x - S + step) / step;      // smallest t with t*step + S > y
        if (y - S + 1 <= 0) tyl = 0;
        if (x - S + 1 <= 0) txl = 0;
        double acc = 0.0, norm = 0.0;
        for (int ty = tyl; ty <= tyh; ++ty)
            for (int tx = txl; tx <= txh; ++tx) {
                const int tile = ty * nx + tx;
                const int r = y - crops[2 * tile + 1], c = x - crops[2 * tile];
                if ((unsigned)r >= (unsigned)S || (unsigned)c >= (unsigned)S) continue;
                // tta_d4_deaug: average over the 8 transforms of the value that lands on (r, c) after undoing it =
                // element (rr, cc) of transformed prediction k with d4_src(k, rr, cc) == (r, c)
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // inverse of d4_src: where does source (r, c) appear in the transformed tile?
                    int rr, cc;
                    switch (k & 3) {
                        case 0: rr = r; cc = c; break;
                        case 1: rr = S - 1 - c; cc = r; break;
                        case 2: rr = S - 1 - r; cc = S - 1 - c; break;
                        default: rr = c; cc = S - 1 - r; break;
                    }
                    if (k >= 4) cc = S - 1 - cc;
                    const float lg = logits[(((long long)(tile * 8 + k) * K + kc) * S + rr) * S + cc];
                    s += 1.f / (1.f + expf(-lg));
                }
                const float v = s * 0.125f;                        // float32 average, as the reference's arrays
                const double w = weight[r * S + c];
                acc += (double)v * w;
                norm += w;
            }
        if (norm < 2.220446049250313e-16) norm = 2.220446049250313e-16;
        out[i] = (float)(acc / norm);
    }
}

}  // namespace

extern "C" int segnb_tiles_gather(const float* image, int H, int W, int C, int margin_top, int margin_left,
                                  const int* crops_xy, int first_item, int count, int S, float* out,
                                  segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_tiles_gather, image, H, W, C, margin_top, margin_left, crops_xy, first_item, count, S, out, stream);
    SEGNB_CHECK_ARG(image && crops_xy && out && H > 1 && W > 1 && C > 0 && S > 0 && count > 0 && first_item >= 0,
                    "bad arguments");
    const long long total = (long long)count * C * S * S;
    int grid = ceil_div(total, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(tiles_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, image, H, W, C, margin_top,
                       margin_left, crops_xy, first_item, count, S, out);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_tiles_gather_u8(const unsigned char* image, int H, int W, int C, int margin_top, int margin_left,
                                     const int* crops_xy, int first_item, int count, int S, float scale, const float* mean,
                                     const float* stdv, float* out, segnb_stream_t stream) {
    SEGNB_PLAN_REFUSE("segnb_tiles_gather_u8 takes host mean / std arrays");
    SEGNB_CHECK_ARG(image && crops_xy && out && mean && stdv && H > 1 && W > 1 && C > 0 && C <= 8 && S > 0 && count > 0 &&
                        first_item >= 0,
                    "bad arguments");
    TileNorm nm;
    nm.scale = scale;
    for (int e = 0; e < 8; ++e) {
        SEGNB_CHECK_ARG(e >= C || stdv[e] != 0.f, "std must be non-zero");
        nm.mean[e] = e < C ? mean[e] : 0.f;
        nm.inv_std[e] = e < C ? 1.0f / stdv[e] : 0.f;
    }
    const long long total = (long long)count * C * S * S;
    int grid = ceil_div(total, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(tiles_gather_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, image, H, W, C, margin_top,
                       margin_left, crops_xy, first_item, count, S, nm, out);
    SEGNB_LAUNCH_CHECK();
    return 0;
}

extern "C" int segnb_tiles_merge(const float* logits, int K, int S, const int* crops_xy, int ntiles, int step, int nx,
                                 int ny, const double* weight, int H, int W, int margin_top, int margin_left,
                                 float* out, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_tiles_merge, logits, K, S, crops_xy, ntiles, step, nx, ny, weight, H, W, margin_top, margin_left, out, stream);
    SEGNB_CHECK_ARG(logits && crops_xy && weight && out && K > 0 && S > 0 && step > 0 && step <= S && nx * ny == ntiles,
                    "bad arguments");
    const long long total = (long long)H * W * K;
    int grid = ceil_div(total, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(tiles_merge_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, K, S, crops_xy,
                       ntiles, step, nx, ny, weight, H, W, margin_top, margin_left, out);
    SEGNB_LAUNCH_CHECK();
    return 0;
}
