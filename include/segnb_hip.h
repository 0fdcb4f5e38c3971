/* segnb_hip.h -- C ABI of libsegnb_hip.so: the MI355X (gfx950) kernels behind the reference's
 * training/inference hot path (torch_train.py:176-190 driving lib/models/<model>.py, lib/losses.py,
 * lib/metrics.py).
 *
 * The reference has no FFI of its own for this path: every primitive below replaces a torch.nn /
 * ATen(cuDNN) call the reference makes (cited per entry point).  The only native-op precedent in
 * the reference is the `inplace_abn` backend contract (lib/modules/abn/functions.py:12-15, :81-118:
 * functions return a success flag, the Python wrapper raises RuntimeError) -- the status-code
 * convention here follows it.
 *
 * Conventions
 *   - every function returns 0 on success, a hipError_t (>0) or SEGNB_E_* (<0) otherwise; nothing
 *     throws across the ABI; segnb_last_error() gives a thread-local message
 *   - every function is asynchronous on the explicit `stream` (a hipStream_t), allocates nothing,
 *     never synchronises: graph-capturable
 *   - pointers are DEVICE pointers borrowed for the duration of the call unless marked host
 *   - activations are NHWC ("pixel-major"), element type `dtype` (f32 or bf16), with an explicit
 *     pixel stride `ld` (elements) so a tensor may be a channel slice of a wider concat buffer
 *     (zero-copy torch.cat, lib/models/zf_unet.py:78-90).  Channel counts are padded to a multiple
 *     of 8 (16-byte vectors); pad channels hold zeros.
 *   - parameters / gradients / logits / targets cross the boundary in the reference's own layouts
 *     (OIHW fp32 weights, NCHW fp32 logits, int64 targets).
 */
#ifndef SEGNB_HIP_H
#define SEGNB_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* segnb_stream_t; /* hipStream_t */

enum { SEGNB_F32 = 0, SEGNB_BF16 = 1 };
enum { SEGNB_ACT_NONE = 0, SEGNB_ACT_RELU = 1, SEGNB_ACT_LEAKY = 2 };
enum { SEGNB_E_BADARG = -1, SEGNB_E_UNSUPPORTED = -2 };

#define SEGNB_MAX_TAPS 64
/* Per-channel accumulators (`stats` of segnb_conv_fprop, `sums` of segnb_bn_act_bwd_reduce) are kept in
 * SEGNB_STAT_REPLICAS interleaved copies, [replica][2][Cp] fp64, so that the hundreds of workgroups that end
 * with an atomic add per channel do not serialise on one address; the finalize calls sum and re-zero them. */
#define SEGNB_STAT_REPLICAS 16

const char* segnb_last_error(void);
int segnb_version(void);
/* number of CUs of the current device (grid sizing); <0 on error */
int segnb_device_cus(void);
/* Kernel-selection knobs for A/B measurements and tests (results never depend on them beyond rounding order):
 *   "fprop_dma"      0 = never use the direct-to-LDS 3x3 pipeline (fprop_dma.hip), 1 = use it where it applies
 *   "fprop_dma_cfg"  -1 = automatic tile configuration, n >= 0 = force configuration n
 *   "fprop_rw"       0 = never use the resident-weights pipeline of the thin layers (fprop_rw.hip)
 *   "fprop_dma_dbg"  timing builds (parts of the pipeline removed; results are WRONG when non-zero)
 *   "conv_cu_pct"    percentage (10..100) of the CUs the persistent convolution kernels size their grids for
 * Defaults come from the environment variables SEGNB_FPROP_DMA / SEGNB_FPROP_DMA_CFG.  Not thread-safe.
 * (The weight-gradient launches read SEGNB_WG_CU_FRACTION / SEGNB_WG_CU_FRACTION_THIN once: share of the CUs their
 * pixel split is sized for, default 0.5 / 1.0; segnb_conv_wgrad_slabs reports the resulting slab count.) */
int segnb_tune(const char* key, int value);
/* segnb_tune("wg_cu_pct", pct) as a call that may sit inside a recorded launch list: the share (%) of the CUs the wide
 * weight-gradient launches split their pixels for -- what segnb_conv_wgrad_slabs reports and what segnb_conv_wgrad* then expect as
 * nslab -- from now on; 0 = the default (SEGNB_WG_CU_FRACTION, 0.5).  A convolution must be launched under the share its workspace
 * was sized under.  (UNet16, unet16.py:50-108: 100 % -- its weight gradients are the longer stream; LinkNet34 loses 7 % with it.) */
int segnb_wg_cu_share(int pct);
/* Timing builds: in-kernel shader-clock stamps of block 0 of the last direct-to-LDS convolution launched with
 * "fprop_dma_dbg" = 32.  host_dst: HOST buffer of 3 x 256 x 4 unsigned 64-bit values ([wave role][tap][event]);
 * synchronises the device (tools/stamps.py). */
int segnb_debug_stamps(unsigned long long* host_dst);
/* Replay guard (segnb.engine.ReplayGuard; SEGNB_REPLAY_GUARD=1): while segnb_tune("call_census", 1) is on, every top-level entry
 * point executed on the calling thread -- called directly or from segnb_plan_run -- is counted by name.  Writes the census as
 * "name count\n" lines into the HOST buffer buf (cap bytes, NUL-terminated, truncated if it does not fit) and clears it.
 * A step replayed from recorded lists must execute exactly the launches the step that recorded them executed: a launch made by
 * host code next to a recorded list and missing from the replay path shows as a difference (round 4's batched bias gradients
 * were lost that way from the third step of a geometry on). */
int segnb_debug_census(char* buf, int cap);

/* ---------------------------------------------------------------------------------------------
 * Generalised gather-convolution geometry.  One launch computes, for every image n and every
 * (qh, qw) in [0,QH)x[0,QW):
 *     out[n, qh*out_step + oh0, qw*out_step + ow0, co] =
 *         bias[co] + sum_t sum_ci in[n, qh*in_step + dh[t], qw*in_step + dw[t], ci] * W[co][t][ci]
 * (out-of-range input pixels read as zero).  With the right tap table this one form covers
 *   nn.Conv2d fwd (any k/stride/pad; zf_unet.py:8, linknet.py:41, tiramisu.py:14),
 *   its data gradient (stride 1: flipped taps; stride 2: one launch per output parity),
 *   nn.ConvTranspose2d fwd (per output parity; linknet.py:16, tiramisu.py:65, unet16.py:38) and its
 *   data gradient -- i.e. aten::convolution / convolution_backward(input) of the reference.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int N, Hi, Wi, Ci;   /* input  [N,Hi,Wi,Ci]  Ci % 8 == 0 */
    int Ho, Wo, Co;      /* output [N,Ho,Wo,Co]  Co % 8 == 0 */
    int ld_in, ld_out;   /* pixel strides in elements */
    int QH, QW;          /* positions iterated per image */
    int in_step, out_step;
    int oh0, ow0;
    int ntaps;
    int dh[SEGNB_MAX_TAPS];
    int dw[SEGNB_MAX_TAPS];
} segnb_conv_geom;

/* Implicit-GEMM forward on MFMA.  wpacked: [Co][ntaps*Ci] of `dtype` (segnb_pack_weight).
 * bias: fp32 [bias_n] (the real, unpadded parameter; channels >= bias_n get 0) or NULL.  stats: fp64 [SEGNB_STAT_REPLICAS][2][Co] (sum, sum of squares of the STORED outputs over all
 * written pixels, accumulated atomically; caller zeroes) or NULL -- the BatchNorm batch statistics
 * of nn.BatchNorm2d in training mode (zf_unet.py:9,15) fused into the conv epilogue. */
int segnb_conv_fprop(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                     const float* bias, int bias_n, void* out, double* stats, segnb_stream_t stream);

/* Weight gradient (aten::convolution_backward weight part), same geometry as the fprop launch:
 *   sum_s dwp[s][co][t][ci] (+)= sum_{n,q} dout[n, oh(q), ow(q), co] * in[n, ih(q,t), iw(q,t), ci]
 * dwp: fp32 workspace of nslab = segnb_conv_wgrad_slabs(g, dtype) slabs [Co][ntaps*Ci]; the result is slab 0.
 *   nslab == 1 : accumulated with fp32 atomics (split over pixel ranges) into a workspace the caller zeroed
 *                (segnb_unpack_wgrad re-zeroes it);
 *   nslab  > 1 : every slab is OVERWRITTEN with the partial sum of one pixel range by plain stores (global float
 *                atomics run at ~1.3 TB/s chip-wide on MI355X -- measured as a 58 us floor per layer), then
 *                slab 0 += slabs 1.. in a fixed order (bitwise reproducible); slabs 1.. are scratch.
 * The slab count depends on the channel counts, the tap set and the output width -- not on the batch or the
 * height -- so a workspace can be sized once per convolution and input width. */
int segnb_conv_wgrad_slabs(const segnb_conv_geom* g, int dtype);
int segnb_conv_wgrad(const segnb_conv_geom* g, int dtype, const void* in, const void* dout,
                     float* dwp, int nslab, segnb_stream_t stream);
/* Where the NEXT weight-gradient launch of the calling thread (segnb_conv_wgrad / _upcat / _bnapply / _tf) delivers its result:
 * straight into the parameter's own gradient -- nn.Conv2d.weight.grad of torch_train.py:188's backward(), fp32
 * [Co_total][Ci_total][KH][KW] -- instead of slab 0 of the packed workspace:
 *     gw[co * s_out + (ci_off + ci) * s_in + kpos[t]]  (+)=  sum_s dwp[s][co][t][ci]      for co < Co, ci < Ci
 * The sum over the pixel-split slabs (fixed order: bitwise reproducible), the [Co][tap][Ci] -> [Co][Ci][tap] transposition and the
 * accumulation into the flat gradient buffer are ONE pass (launches with a single slab write the gradient from their
 * accumulator registers); the per-layer slab reduction, the batched segnb_unpack_wgrad_multi pass over every workspace (read,
 * re-zero, read-modify-write of the gradient: 0.5 GB of a ZF_UNET step) and its job tables are not needed for such a layer.
 * accumulate != 0: added to what gw holds (torch's .grad accumulation); 0: gw is overwritten.  The workspace keeps its
 * contract (nslab slabs; scratch afterwards).  Recordable; consumed by the next weight-gradient entry point, whatever kernel
 * serves it. */
typedef struct {
    float* gw;          /* first element of the parameter's gradient */
    long long s_out;    /* floats between consecutive output channels of the parameter (Ci_total * KH * KW) */
    int s_in;           /* floats between consecutive input channels (KH * KW) */
    int ci_off;         /* input channel 0 of the launch is input channel ci_off of the parameter (a concat segment on its own) */
    int Ci, Co;         /* real channel counts: the launch's channels beyond them are padding and are dropped */
    int accumulate;
    int ntaps;
    int kpos[SEGNB_MAX_TAPS];     /* kernel position kh * KW + kw of packed tap t */
} segnb_wgrad_target;
int segnb_wgrad_target_arm(const segnb_wgrad_target* t);
/* Weight gradient whose dout operand is not in memory: it is the BatchNorm-backward apply of the layer,
 *     dout = round(a * (round(g * act'(z)) - c1 - yhat * c2)),   z = (y - mean) * scale + shift,  yhat = (y - mean) * invstd
 * (lib/modules/abn/functions.py:118's dx, the arithmetic and roundings of segnb_bn_bwd_apply_direct), recomputed from the
 * incoming gradient g and the pre-BatchNorm output y while the tile is staged.  For the FIRST convolution of a network
 * (lib/models/zf_unet.py:44 conv_224.l1: no data gradient, so nothing else reads that tensor): the apply pass and its
 * 103 MB tensor (bs=32, 224x224) leave the serial tail of backward.  coef: [4][Cp] of segnb_bn_finalize, bcoef: [3][Cp]
 * of segnb_bn_bwd_finalize.  bf16, stride-1 3x3, <= 32 output channels (segnb_conv_wgrad_bnapply_ok). */
int segnb_conv_wgrad_bnapply_ok(const segnb_conv_geom* g, int dtype);
int segnb_conv_wgrad_bnapply(const segnb_conv_geom* g, int dtype, const void* in, const void* gsrc, int ld_g,
                             const void* y, int ld_y, const float* coef, const float* bcoef, int Cp, int act,
                             float slope, float* dwp, int nslab, segnb_stream_t stream);

/* Parameter-layout <-> packed-GEMM-layout.  Packed matrix is [Mp][ntaps][Cp]; element (mp,t,cp)
 * maps to w[mmap[mp]*s_m + cmap[cp]*s_c + tap_off[t]] (maps are device int32 arrays, -1 = padding).
 * tap_off is a HOST array of ntaps element offsets (kh*s_kh + kw*s_kw).
 * segnb_unpack_wgrad CONSUMES dwp (slab 0): it is zeroed as it is read, ready for the next step's atomics. */
int segnb_pack_weight(const float* w, void* wpacked, int dtype, int Mp, int Cp, int ntaps,
                      long long s_m, long long s_c, const int* tap_off_host, const int* mmap,
                      const int* cmap, segnb_stream_t stream);
int segnb_unpack_wgrad(float* dwp, float* gw, int Mp, int Cp, int ntaps, long long s_m,
                       long long s_c, const int* tap_off_host, const int* mmap, const int* cmap,
                       int accumulate, segnb_stream_t stream);

/* Batched forms: one launch for every weight matrix of a model (one for every gradient).  `jobs` is a DEVICE
 * array of njobs records, each segnb_pack_job_bytes() long:
 *   { const float* param_or_grad; void* packed; const int* mmap; const int* cmap; int64 s_m, s_c;
 *     int32 Mp, Cp, ntaps, dtype, block_start, nslab; int32 tap_off[SEGNB_MAX_TAPS]; }
 *     (nslab: unpack jobs only -- partial slabs to sum in slab order; 0 or 1 = a single slab)
 * sorted by block_start; job k owns blocks [block_start_k, block_start_k + segnb_pack_job_blocks(...)) (LDS-tiled
 * transposes: both the parameter side and the packed side are accessed in contiguous runs).
 * segnb_pack_job_blocks returns -1 for kernels wider than 3x3 (use the single-job calls for those).
 * unpack ADDS into the gradient and re-zeroes the workspace (slab 0). */
int segnb_pack_job_bytes(void);
int segnb_pack_job_blocks(int Mp, int Cp, int ntaps, long long s_m, long long s_c);
int segnb_pack_weight_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream);
int segnb_unpack_wgrad_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream);
/* the same job table for the jobs segnb_pack_job_blocks refuses (parameter tensors with more than 3 x 3 positions: the 7 x 7 stem
 * of linknet.py:13-20 / dilated_resnet.py, the 4 x 4 ConvTranspose2d of unet16.py:40-47): element-wise kernels,
 * segnb_pack_elem_job_blocks(Mp, Cp, ntaps) blocks per job; unpack ADDS to the gradient and clears the workspace */
int segnb_pack_elem_job_blocks(int Mp, int Cp, int ntaps);
/* BOTH matrices of a plain 3x3 convolution from one read of its parameter (bf16 only): what the reference's
 * nn.Conv2d(ci, co, 3) weight (lib/models/zf_unet.py:5-32) becomes at every step's start -- the forward matrix
 * [Cop][9][Cip] and the data-gradient matrix [Cip][9][Cop].  `jobs`: DEVICE array of records, each
 * segnb_pack_pair_job_bytes() long:
 *   { const float* w; void* packed_fwd; void* packed_dgrad; int32 Ci, Co, Cip, Cop, block_start, pad;
 *     int32 tap_fwd[9], tap_dgrad[9]; }      (tap_*[t] = kernel position kh * 3 + kw of packed tap t)
 * sorted by block_start; a job owns segnb_pack_pair_job_blocks(Co, Ci, Cop, Cip) blocks (-1: bad shape).  packed_dgrad may be
 * NULL (a layer without a data gradient: forward matrix only).  Real channel c is packed channel c; the padding channels are
 * written as zeros. */
int segnb_pack_pair_job_bytes(void);
int segnb_pack_pair_job_blocks(int Co, int Ci, int Cop, int Cip);
int segnb_pack_weight_pair_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream);
/* torch.optim.SGD.step() -- plain SGD, w -= lr * g, torch_train.py:71,190 -- on the parameters of a pair-job table AND their weight pack
 * in ONE pass: every tile of a parameter is read once, updated with segnb_sgd_step's expression (bit-identical parameters), written back,
 * rounded and stored as both packed matrices.  flat_p / flat_g: the parallel flat fp32 parameter / gradient buffers the jobs' `w` point
 * into.  Saves the second read of every parameter the next forward's segnb_pack_weight_pair_multi would make.  Not recordable
 * (optimizer.step() is host code between the recorded lists).  The parameters the table does not cover: segnb_sgd_ranges -- the update
 * over `nranges` element ranges, ranges = DEVICE int64 [nranges][3] = (start, length, index of the range's first element in the
 * concatenation of all ranges), total = the sum of the lengths. */
int segnb_sgd_pack_pair_multi(const void* jobs, int njobs, int total_blocks, float* flat_p, const float* flat_g, float lr,
                              segnb_stream_t stream);
int segnb_sgd_ranges(float* p, const float* g, const long long* ranges, int nranges, long long total, float lr, segnb_stream_t stream);
int segnb_pack_weight_elem_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream);
int segnb_unpack_wgrad_elem_multi(const void* jobs, int njobs, int total_blocks, segnb_stream_t stream);

/* NCHW fp32 network input -> NHWC `dtype`, channels zero-padded to Cp (torch_train.py:177 hands the
 * model a float32 [N,3,H,W] batch, lib/common.py:70). */
int segnb_pack_input_nchw(const float* x, int N, int C, int H, int W, void* out, int dtype, int Cp,
                          int ld_out, segnb_stream_t stream);

/* The network input as the dataset holds it: uint8 HWC [N][H][W][C] (cv2.imread, lib/common.py:44), 1 <= C <= 8.
 * NormalizeImage (lib/augmentations.py:452-460: x * scale - mean) / std), the HWC -> CHW move (lib/common.py:70) and the
 * float conversion happen in registers: v = (u8 * scale - mean[c]) * (1 / std[c]) in fp32, stored as `dtype`.
 * mean / std: HOST arrays of C floats (copied into the launch).   */
int segnb_pack_input_u8(const unsigned char* img, int N, int H, int W, int C, float scale, const float* mean,
                        const float* stdv, void* out, int dtype, int Cp, int ld_out, segnb_stream_t stream);
/* The FIRST convolution (3x3, stride 1, <= 32 output channels, bf16) reading the uint8 image directly: no packed copy
 * is read at all.  x_packed (optional, [N][H][W][ld_packed] bf16, 8 channels written): the normalised pixels, written
 * once by the tile that owns them -- the x operand of this layer's weight gradient; NULL for inference.  The result
 * is bit-identical to segnb_pack_input_u8 + segnb_conv_fprop.  segnb_conv_fprop_u8_ok: 1 if the geometry is served. */
int segnb_conv_fprop_u8_ok(const segnb_conv_geom* g, int dtype);
int segnb_conv_fprop_u8(const segnb_conv_geom* g, const unsigned char* img, int C, float scale, const float* mean,
                        const float* stdv, const void* wpacked, const float* bias, int bias_n, void* out,
                        double* stats, void* x_packed, int ld_packed, segnb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tiled inference: ImageSlicer.split / merge of lib/tiles.py:99-161, the D4 test-time
 * augmentation of lib/augmentations.py:476-511 and the sigmoid of inria_submit.py:249.
 * crops_xy: device int32 [ntiles][2] = (x, y) of each tile in the PADDED image (ImageSlicer.crops), a regular grid of
 * pitch `step`, nx tiles per row.  Items are numbered item = tile*8 + k, k = index into tta_d4_aug's 8 transforms.
 * ------------------------------------------------------------------------------------------- */

/* out[b][c][r][s] (NCHW fp32, b = item first_item + b) = transform k of tile `tile` of the reflect-101 padded
 * image (fp32 HWC, already normalised): the batch inria_submit.predict_tiled feeds the model with. */
int segnb_tiles_gather(const float* image, int H, int W, int C, int margin_top, int margin_left,
                       const int* crops_xy, int first_item, int count, int S, float* out,
                       segnb_stream_t stream);
/* The same batch gathered from the UNNORMALISED uint8 HWC image (what cv2.imread hands inria_submit.py:298), NormalizeImage
 * (lib/augmentations.py:452-460: (x * scale - mean) / std, inria_submit.py:238) in registers; mean / stdv: HOST arrays of C
 * floats (C <= 8).  A 5000 x 5000 x 3 Inria image crosses PCIe as 75 MB instead of 300 MB.  Not recordable. */
int segnb_tiles_gather_u8(const unsigned char* image, int H, int W, int C, int margin_top, int margin_left,
                          const int* crops_xy, int first_item, int count, int S, float scale, const float* mean,
                          const float* stdv, float* out, segnb_stream_t stream);

/* logits: fp32 [ntiles*8][K][S][S] (model outputs of every item).  out: fp32 [H][W][K] = merge(deaug(sigmoid)):
 * per tile the mean over the 8 un-transformed predictions (float32, the reference's order), then the weighted
 * mean over the tiles covering a pixel (float64, tile order) with weight[S][S] (fp64: pyramid or ones), cropped. */
int segnb_tiles_merge(const float* logits, int K, int S, const int* crops_xy, int ntiles, int step, int nx,
                      int ny, const double* weight, int H, int W, int margin_top, int margin_left, float* out,
                      segnb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm (+activation, +Dropout2d, +MaxPool2d(2), +nearest x2 upsample) -- the same 4-phase split
 * as inplace_abn's mean_var / forward / edz_eydz / backward (lib/modules/abn/functions.py:81,94,112,118).
 * ------------------------------------------------------------------------------------------- */

/* stats [SEGNB_STAT_REPLICAS][2][Cp] (from segnb_conv_fprop) -> per-channel affine + running-stat update.
 * coef: fp32 [4][Cp] = scale(=gamma*invstd), beta, mean, invstd; z = (y - mean)*scale + beta.  Channels
 * >= C get all-zero rows.  training != 0: batch stats, running_mean/var updated with `momentum`
 * (unbiased var, nn.BatchNorm2d semantics), *nbt += 1, and `stats` is CONSUMED (re-zeroed for the next
 * step).  training == 0: running stats (eval). */
int segnb_bn_finalize(double* stats, int C, int Cp, double count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, long long* nbt, int training, float* coef,
                      segnb_stream_t stream);

/* a = dropmul[n][c] * act((y - mean)*scale + beta + res).  res: optional residual input added BEFORE the
 * activation (the identity branch of a ResNet BasicBlock, linknet.py:45-48 via torchvision resnet34), NULL = none.
 *   coef may be NULL (identity affine).  dropmul: fp32
 * [N][Cp] multiplier table (Dropout2d replay format; NULL = none).  Optional extra outputs:
 * pool_out = MaxPool2d(2) of a (floor mode), up_out = nearest x2 upsample of a (each with its own ld).
 * Replaces BatchNorm2d+ReLU (zf_unet.py:15-16), Dropout2d (:31), MaxPool2d (:41), Upsample (:42). */
int segnb_bn_act_fwd(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                     const float* coef, int act, float slope, const float* dropmul, void* out,
                     int ld_out, void* pool_out, int ld_pool, void* up_out, int ld_up, const void* res,
                     int ld_res, segnb_stream_t stream);

/* dz = act'(z) * dropmul * (g_direct + maxpool_bwd(g_pool) + upsample_bwd(g_up)); any source may be
 * NULL.  Writes dz; accumulates sums[r][0][c] += sum dz, sums[r][1][c] += sum dz*yhat
 * (fp64 [SEGNB_STAT_REPLICAS][2][Cp]). */
int segnb_bn_act_bwd_reduce(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                            const float* coef, int act, float slope, const float* dropmul,
                            const void* g_direct, int ld_gd, const void* g_pool, int ld_gp,
                            const void* g_up, int ld_gu, void* dz, int ld_dz, double* sums,
                            const void* res, int ld_res, segnb_stream_t stream);
/* The same pass over TWO same-size gradient sources (a tensor with two consumers: the identity branches of linknet.py:41-62's
 * BasicBlocks): g = round(g_direct + g_add) -- exactly what segnb_add would have stored -- without that pass over the three tensors. */
int segnb_bn_act_bwd_reduce_add(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef, int act,
                                float slope, const float* dropmul, const void* g_direct, int ld_gd, const void* g_add, int ld_ga,
                                void* dz, int ld_dz, double* sums, const void* res, int ld_res, segnb_stream_t stream);
/* (with a residual input, dz is also the gradient of the residual branch)
 * dz may be NULL when the only source is g_direct and there is no dropout table and no residual: a sums-only pass
 * (one tensor write less); the layer's dy then comes from segnb_bn_bwd_apply_direct. */

/* segnb_bn_finalize + segnb_bn_act_fwd in ONE launch (training mode): every block derives the coefficients of its
 * channels from `stats`; the first block column also writes `coef` (for the backward pass), updates the running
 * statistics / *nbt and zeroes `bwd_sums_to_clear` (this layer's backward accumulators, NULL to skip).  `stats` is
 * NOT consumed here -- other blocks are still reading it: segnb_bn_bwd_apply_fused of the same layer clears it (a
 * caller that runs no backward must clear it itself before the next forward).  count = N*H*W. */
int segnb_bn_fwd_fused(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                       const double* stats, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, long long* nbt, float* coef,
                       double* bwd_sums_to_clear, int act, float slope, const float* dropmul, void* out,
                       int ld_out, void* pool_out, int ld_pool, void* up_out, int ld_up, const void* res,
                       int ld_res, segnb_stream_t stream);

/* segnb_bn_bwd_finalize + segnb_bn_bwd_apply in ONE launch: bcoef from `sums` per block, the first block column
 * writes bcoef / dgamma / dbeta and zeroes `fwd_stats_to_clear` (the forward statistics of this layer, NULL to
 * skip).  `sums` is NOT consumed here: the next segnb_bn_fwd_fused of the layer clears it. */
int segnb_bn_bwd_apply_fused(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                             const float* coef, const double* sums, const float* gamma, float* bcoef,
                             float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear,
                             const void* dz, int ld_dz, void* dy, int ld_dy, segnb_stream_t stream);

/* The same with dy += result instead of dy = result: the input of a PRE-activation BatchNorm (tiramisu.py:12-13) has
 * other consumers, whose gradients are already in dy -- the separate segnb_add pass of the accumulation folded in (the
 * result, rounded to the storage type, is added to the stored value as segnb_add would).  dz must not alias dy. */
int segnb_bn_bwd_apply_fused_acc(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                 const float* coef, const double* sums, const float* gamma, float* bcoef,
                                 float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear,
                                 const void* dz, int ld_dz, void* dy, int ld_dy, segnb_stream_t stream);

/* ... and with dz recomputed from the incoming gradient g as in segnb_bn_bwd_apply_fused_direct: a pre-activation
 * BatchNorm + ReLU whose activation has ONE consumer (tiramisu.py:12-14) needs no dz tensor at all -- sums-only reduce,
 * then this launch: dy += BatchNorm-backward(act'(z) * g). */
int segnb_bn_bwd_apply_fused_direct_acc(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                        const float* coef, const double* sums, const float* gamma, float* bcoef,
                                        float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear, int act,
                                        float slope, const void* g, int ld_g, void* dy, int ld_dy, segnb_stream_t stream);
/* The apply pass of a layer whose dz was never stored (segnb_bn_act_bwd_reduce with dz = NULL, now allowed for every source
 * combination): the gradient sources are read again, dz is recomputed as the reduction pass computed it (MaxPool2d(2) routing to
 * the first maximum, 2 x 2 sum of an upsampled gradient, Dropout2d multiplier, act') and dy = round(a * (dz - c1 - yhat * c2)) is
 * written; (a, c1, c2), dgamma, dbeta and the clearing of the forward statistics as segnb_bn_bwd_apply_fused.  Reads
 * sources + y, writes dy: one tensor write and one read less than reduce(-> dz) + apply(dz -> dy) (the backward phase of
 * lib/modules/abn/functions.py:118 without the stored dz).  dy must not alias a source. */
int segnb_bn_bwd_apply_fused_src(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const float* coef,
                                 const double* sums, const float* gamma, float* bcoef, float* dgamma, float* dbeta,
                                 int accumulate, double* fwd_stats_to_clear, int act, float slope, const float* dropmul,
                                 const void* g_direct, int ld_gd, const void* g_pool, int ld_gp, const void* g_up, int ld_gu,
                                 void* dy, int ld_dy, segnb_stream_t stream);
/* InPlaceABN with the backend's affine form inside a fused plan (lib/models/linknet.py:12-21 on lib/modules/abn/functions.py:
 * 94,112,118): segnb_abn_scale writes the effective scale out = |w| + eps that the BatchNorm entry points then take as their
 * gamma; their dgamma output goes to a scratch vector, and segnb_abn_dscale adds sign(w) * dscale into the parameter's
 * gradient (+1 for w > 0, -1 otherwise, as the backend) and clears the scratch. */
/* segnb_bn_finalize(training = 1) under the fused protocol of segnb_bn_fwd_fused, for a layer whose activation pass does not
 * exist (its consumer applies BatchNorm + activation while it loads: segnb_conv_fprop_tf): coef and the running statistics are
 * written, the forward statistics are LEFT for the layer's backward (segnb_bn_bwd_apply_fused*) to clear, the backward
 * accumulators clear_sums [16][2][Cp] are cleared. */
int segnb_bn_finalize_keep(const double* stats, int C, int Cp, double count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, long long* nbt, float* coef,
                           double* clear_sums, segnb_stream_t stream);
int segnb_abn_scale(const float* w, float eps, float* out, int n, segnb_stream_t stream);
int segnb_abn_dscale(const float* w, float* dscale, float* dw, int n, segnb_stream_t stream);

/* sums -> bcoef fp32 [3][Cp] = (gamma*invstd, mean(dz), mean(dz*yhat)); dgamma/dbeta (C entries)
 * assigned or accumulated.  `sums` is CONSUMED (re-zeroed). */
int segnb_bn_bwd_finalize(double* sums, int C, int Cp, double count, const float* gamma,
                          const float* coef, float* bcoef, float* dgamma, float* dbeta,
                          int accumulate, segnb_stream_t stream);
/* the same, clearing the layer's forward statistics buffer [SEGNB_STAT_REPLICAS][2][Cp] too (NULL: no) -- what the fused
 * apply launches do on the way; for layers without an apply pass (segnb_conv_wgrad_bnapply) */
int segnb_bn_bwd_finalize_clear(double* sums, int C, int Cp, double count, const float* gamma, const float* coef,
                                float* bcoef, float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear,
                                segnb_stream_t stream);

/* dy = bcoef0 * (dz - bcoef1 - yhat*bcoef2) written to dy (may alias dz);
 * dbias[c] += sum dy (fp32 atomics; NULL to skip). */
int segnb_bn_bwd_apply(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                       const float* coef, const float* bcoef, const void* dz, int ld_dz, void* dy,
                       int ld_dy, float* dbias, int C, segnb_stream_t stream);

/* The same with dz recomputed on the fly from the layer's incoming gradient g (dz = act'(z) * g, rounded to the storage
 * type exactly as segnb_bn_act_bwd_reduce would have stored it): for layers whose gradient has a single direct source
 * (the first convolution of every ZF_UNET block, zf_unet.py:22-23) dz never goes to memory.  dy may alias g. */
int segnb_bn_bwd_apply_direct(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp,
                              const float* coef, const float* bcoef, int act, float slope, const void* g, int ld_g,
                              void* dy, int ld_dy, float* dbias, int C, segnb_stream_t stream);

/* segnb_bn_bwd_apply_fused with dz recomputed from g as in segnb_bn_bwd_apply_direct (finalize + apply in one launch, dz
 * never in memory). */
int segnb_bn_bwd_apply_fused_direct(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp,
                                    const float* coef, const double* sums, const float* gamma, float* bcoef,
                                    float* dgamma, float* dbeta, int accumulate, double* fwd_stats_to_clear, int act,
                                    float slope, const void* g, int ld_g, void* dy, int ld_dy, segnb_stream_t stream);

/* out = a + b (skip ADD of linknet.py:77-79; gradient accumulation of multi-consumer tensors); out may alias a.
 * A NULL operand counts as zeros: (NULL, b) copies b, (NULL, NULL) clears out -- the strided copies / clears of the
 * executor as recordable launches (segnb_plan_*) */
int segnb_add(int dtype, const void* a, int ld_a, const void* b, int ld_b, void* out, int ld_out, int N,
              int H, int W, int Cp, segnb_stream_t stream);
/* stats[r][0][c] += sum x, stats[r][1][c] += sum x^2 of an arbitrary NHWC tensor: batch statistics for a
 * PRE-activation BatchNorm (tiramisu.py:12,50), same replicated layout as the conv epilogue's */
int segnb_bn_stats(int dtype, const void* x, int ld, int N, int H, int W, int Cp, double* stats,
                   segnb_stream_t stream);
/* nn.MaxPool2d(k, stride, pad), floor mode (resnet stem maxpool 3x3 s2 p1, linknet.py:44) and its backward
 * (gradient to the first maximum of each window) */
/* nn.Upsample(scale_factor=2, mode='bilinear') (align_corners=False) of DecoderBlock's non-deconvolution branch
 * (lib/models/unet16.py:42-46) and its backward (the exact transpose, gather form): x [N][H][W][Cp] -> out [N][2H][2W][Cp];
 * g_out [N][2H][2W][Cp] -> dx [N][H][W][Cp].  H, W: the LOW-resolution size. */
int segnb_upsample_bilinear2x_fwd(int dtype, const void* x, int ld_x, int N, int H, int W, int Cp, void* out, int ld_out,
                                  segnb_stream_t stream);
int segnb_upsample_bilinear2x_bwd(int dtype, const void* g_out, int ld_go, int N, int H, int W, int Cp, void* dx, int ld_dx,
                                  segnb_stream_t stream);

/* idx (optional, uint8 [N][Ho][Wo][Cp]): the forward records the window position a*k+b of each maximum; handed to the
 * backward it replaces the re-scan of every covering window (NULL on either side: the re-scanning backward). */
int segnb_maxpool_fwd(int dtype, const void* x, int ld_x, int N, int H, int W, int Cp, int k, int stride,
                      int pad, void* out, int ld_out, unsigned char* idx, segnb_stream_t stream);
int segnb_maxpool_bwd(int dtype, const void* x, int ld_x, const void* g_out, int ld_go, int N, int H, int W,
                      int Cp, int k, int stride, int pad, void* dx, int ld_dx, const unsigned char* idx,
                      segnb_stream_t stream);
/* The same with TWO gradients of the pooled tensor (it has two consumers: the first BasicBlock's convolution and its identity
 * branch, linknet.py:41-62 via resnet34): the routed value is round(g_out + g_out2), what segnb_add would have stored -- without
 * that pass.  idx (the argmax positions segnb_maxpool_fwd recorded) is required. */
int segnb_maxpool_bwd_add(int dtype, const void* x, int ld_x, const void* g_out, int ld_go, const void* g_out2, int ld_go2, int N,
                          int H, int W, int Cp, int k, int stride, int pad, void* dx, int ld_dx, const unsigned char* idx,
                          segnb_stream_t stream);
/* NHWC `dtype` -> fp32 NCHW (logits of a head that is not a 1x1 conv: linknet.py:62) */
int segnb_nhwc_to_nchw_f32(int dtype, const void* a, int ld, int N, int H, int W, int C, float* out,
                           segnb_stream_t stream);

/* Convolution with an affine + activation epilogue: out = act(v),
 *   coef == NULL : v = conv + bias                          (nn.Conv2d -> nn.ReLU with no BatchNorm between them: unet16.py:12-21,
 *                                                            35-40; linknet.py:57-62) -- no separate activation pass
 *   coef != NULL : v = (conv + bias - mean) * scale + shift  (eval-mode BatchNorm folded in; coef = [4][Co] of segnb_bn_finalize
 *                                                            with training = 0: validate(), torch_train.py:248-265, and inference)
 * act: SEGNB_ACT_NONE / RELU / LEAKY(slope).  No statistics (a training-mode BatchNorm needs the raw output first). */
typedef struct {
    const float* coef;
    int act;
    float slope;
} segnb_act_epilogue;
int segnb_conv_fprop_act(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, const float* bias,
                         int bias_n, void* out, const segnb_act_epilogue* ep, segnb_stream_t stream);

/* A DATA-GRADIENT launch whose epilogue also does the BatchNorm-backward reduction of the layer that produced its
 * output's forward counterpart: out = g (the gradient of that layer's activation, exactly what segnb_conv_fprop
 * writes), and sums[r][0][c] += sum dz, sums[r][1][c] += sum dz * yhat with dz = round(g * act'((y - mean) * scale +
 * shift)) -- segnb_bn_act_bwd_reduce(g_direct = out, dz = NULL, no dropout) without its own pass over y and g (the
 * edz_eydz phase of lib/modules/abn/functions.py:112 folded into the producer of dz).  y: [N][Ho][Wo][ld_y] `dtype`,
 * coef: [4][Co] of segnb_bn_finalize, sums: [16][2][Co] fp64.  _ok: 1 when a fused kernel serves the geometry (thin
 * stride-1 3x3 layers on conv_fprop_rw_kernel); otherwise call the two entry points separately. */
typedef struct {
    const void* y;
    int ld_y;
    const float* coef;
    double* sums;
    int act;
    float slope;
} segnb_bn_reduce_epilogue;
int segnb_conv_fprop_bnreduce_ok(const segnb_conv_geom* g, int dtype);
/* A data gradient that is NEVER STORED (tiramisu.py:9-20: norm -> relu -> conv(C -> 16) of a dense layer -- its data gradient has
 * K = 16 x 9 and C up to ~1100 output channels: the launch is the bytes of its output, which the BatchNorm backward then reads once
 * for the reduction and once for the apply).  Two launches that each recompute the tile instead:
 *   segnb_conv_fprop_bnsums   the reduction of segnb_conv_fprop_bnreduce, nothing written (reads in, y)
 *   segnb_conv_fprop_bnapply  once the sums are complete: dz = round(round(g) * act'(z)) again, (a, c1, c2) from the sums (dgamma +=,
 *                             dbeta +=, bcoef written: the fused finalize of segnb_bn_bwd_apply_fused), and
 *                             dx = round(a * (dz - c1 - yhat * c2))  [accumulate: dx = round(dx + that)]
 * = segnb_conv_fprop_bnreduce + segnb_bn_bwd_apply_fused_direct(_acc) bit for bit in dx, with 4 tensor transits instead of 6.
 * gamma: [C] or NULL; C real BatchNorm channels (<= g->Co, the others give dx = 0); count = N * Ho * Wo; the sums are only READ
 * (segnb_bn_fwd_fused clears them in the next forward).  _ok: served (bf16, stride-1 3x3, Ci <= 24, Co > 32: the general kernel). */
typedef struct {
    const void* y;
    int ld_y;
    const float* coef;
    const double* sums;
    const float* gamma;
    int C;
    double count;
    float* bcoef;
    float* dgamma;
    float* dbeta;
    int act;
    float slope;
    void* dx;
    int ld_dx;
    int accumulate;
} segnb_bn_apply_epilogue;
int segnb_conv_fprop_bnapply_ok(const segnb_conv_geom* g, int dtype);
int segnb_conv_fprop_bnsums(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                            const segnb_bn_reduce_epilogue* ep, segnb_stream_t stream);
int segnb_conv_fprop_bnapply(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked,
                             const segnb_bn_apply_epilogue* ep, segnb_stream_t stream);

/* coef == NULL: the producing layer is a convolution + activation WITHOUT BatchNorm (linknet.py:58-61 finaldeconv1 -> finalrelu1 ->
 * finalconv2, unet16.py:12-21) and y is its ACTIVATED output: the launch then stores out = dz = round(round(g) * act'(y)) (act' from
 * the sign of the activated value) and sums[r][0][c] += sum dz -- that layer's segnb_bn_act_bwd_reduce pass (y, coef NULL, dz) folded
 * into the data gradient that produces g.  _actmask_ok: 1 when a fused kernel serves the geometry (stride-1 3 x 3 window; 32 -> <= 32
 * channels with any padding, width >= 32: conv_roll_kernel;  Ci % 64 == 0 -> > 32 channels, width > 8: the MASK instantiation of
 * conv_fprop_ws_kernel -- the VGG-style encoder / decoder convolutions of unet16.py:73-108). */
int segnb_conv_fprop_actmask_ok(const segnb_conv_geom* g, int dtype);

/* conv -> Dropout2d -> [statistics of the result] in ONE launch (a dense layer of tiramisu.py:9-20: norm -> relu -> conv(C -> 16) ->
 * Dropout2d(0.2), its 16 channels written into their slice of the block's concat buffer, whose later BatchNorms need the slice's
 * batch statistics):
 *     out[n, h, w, co] = round(round(acc + bias[co]) * dropmul[n * ld_drop + co])      -- the bits segnb_conv_fprop followed by
 *                                                                                          segnb_bn_act_fwd(coef NULL, ACT_NONE, dropmul) stores
 *     stats[r][0][co] += sum out,  stats[r][1][co] += sum out^2   over the pixels of the launch's blocks, rows stats_ld doubles apart
 *                                                                  (a [REPLICAS][2][stats_ld] table: segnb_bn_stats_ld); stats NULL: none
 * instead of the convolution + a 5 us pass over sixteen channels, ninety times per FCDenseNet103 forward.  dropmul: [N][ld_drop] fp32,
 * required.  _ok: bf16, stride-1 3 x 3 window, <= 32 output channels behind > 96 input channels (conv_fprop_deepk_kernel /
 * conv_fprop_s1x9_kernel). */
int segnb_conv_fprop_drop_ok(const segnb_conv_geom* g, int dtype);
int segnb_conv_fprop_drop(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, const float* bias, int bias_n,
                          void* out, const float* dropmul, int ld_drop, double* stats, int stats_ld, segnb_stream_t stream);

/* CONSUMER-SIDE BatchNorm: a convolution (or weight-gradient) operand that is NOT in memory -- it is recomputed from what the
 * producing layer left there while the kernel stages its input rows (conv_roll_kernel, fprop_roll.hip):
 *   SEGNB_TF_ACT    operand = round(drop * act((src - mean) * scale + shift)): what segnb_bn_act_fwd would have written from the
 *                   pre-BatchNorm tensor src (`in` of the launch) -- the _Conv3BN activation of lib/models/zf_unet.py:12-17 applied by
 *                   the NEXT convolution's loads; the activated tensor never exists;
 *   SEGNB_TF_BNBWD  operand = dy = a * (dz - c1 - yhat * c2), dz = src * drop * act'(z): what segnb_bn_bwd_apply(_direct) would
 *                   have written from the incoming gradient src (`in`; with act = SEGNB_ACT_NONE: dz itself) and this layer's
 *                   pre-BatchNorm output y -- the data gradient of a layer without its BatchNorm-backward apply pass
 *                   (lib/modules/abn/functions.py:118 folded into the consumer).  Evaluated in fp32 in a folded form:
 *                   equal to the two-launch form up to fp32 rounding before the one bf16 rounding (fprop_roll.hip).
 * coef = [4][Cp] of segnb_bn_finalize, bcoef = [3][Cp] of segnb_bn_bwd_finalize, drop = [N][Cp] Dropout2d multipliers or NULL.
 * Pixels outside the image are zero AFTER the transform (the zero padding of the convolution).  segnb_conv_fprop_tf takes the
 * statistics / BatchNorm-reduce epilogues of segnb_conv_fprop / segnb_conv_fprop_bnreduce (stats, bn: either may be NULL). */
#define SEGNB_TF_ACT 1
#define SEGNB_TF_BNBWD 2
typedef struct {
    int kind;
    const void* y;
    int ld_y;
    const float* coef;
    const float* bcoef;
    const float* drop;
    int Cp;
    int act;
    float slope;
} segnb_operand_tf;
int segnb_conv_fprop_tf_ok(const segnb_conv_geom* g, int dtype, int kind);
int segnb_conv_fprop_tf(const segnb_conv_geom* g, int dtype, const void* in, const segnb_operand_tf* tf, const void* wpacked,
                        const float* bias, int bias_n, void* out, double* stats, const segnb_bn_reduce_epilogue* bn,
                        segnb_stream_t stream);
/* The weight gradient of such a layer (segnb_conv_wgrad's protocol, result in slab 0 of dwp): tf_in (SEGNB_TF_ACT or NULL) describes
 * how `in` becomes the convolution's input x, tf_dout (SEGNB_TF_BNBWD or NULL) how `dout` (g or dz) and tf_dout->y become dy.  With
 * both, the data gradient (segnb_conv_fprop_tf) and the weight gradient of a layer read the same three tensors -- g, y and the
 * producing layer's y -- and neither the activated input nor dy is ever written (conv_wgrad_roll_kernel, wgrad_roll.hip). */
int segnb_conv_wgrad_tf_ok(const segnb_conv_geom* g, int dtype);
int segnb_conv_wgrad_tf(const segnb_conv_geom* g, int dtype, const void* in, const segnb_operand_tf* tf_in, const void* dout,
                        const segnb_operand_tf* tf_dout, float* dwp, int nslab, segnb_stream_t stream);

/* 1 if segnb_conv_fprop serves this 4x4 / stride-2 gather (ntaps 16, in_step 2: the data gradient of an
 * Upsample(scale_factor=2) -> conv3x3 segment on the low-resolution grid, lib/models/zf_unet.py:42,78-90; also the data gradient
 * of ConvTranspose2d(4, 2, 1), unet16.py:38 / linknet.py:16) on the plane-gather form of the direct-to-LDS pipeline; 0: it
 * would run on the general gather kernel.  A plan uses it to decide between the segmented and the plain data gradient of a
 * decoder block (segnb.engine.UpCatConvOp). */
int segnb_conv_fprop_upd_ok(const segnb_conv_geom* g, int dtype);

/* VIRTUAL CONCAT.  The first convolution of a ZF_UNET decoder block reads torch.cat([Upsample(scale_factor=2)(u), skip], 1)
 * (lib/models/zf_unet.py:42,78-90).  These entry points take the two tensors as they are: the first src->Cu input channels of
 * the geometry are the NEAREST x2 upsample of src->u [N][Hi/2][Wi/2][ld_u] -- input pixel (h, w) of those channels is u pixel
 * (h >> 1, w >> 1), resolved in the kernels' tile fetch -- and the remaining g->Ci - Cu channels come from `in`
 * ([N][Hi][Wi][ld_in], the skip tensor, channel 0 = logical channel Cu).  The upsampled copy (4x the size of u) is never
 * written or read.  Same weights / workspace layouts as segnb_conv_fprop / segnb_conv_wgrad over the concatenated input.
 * segnb_conv_upcat_ok: 1 if BOTH are served for the geometry (bf16, stride-1 3x3, even Hi / Wi; Cu a multiple of the
 * kernels' channel chunk); otherwise the caller materialises the upsampled copy and uses the plain entry points. */
typedef struct {
    const void* u;
    int Cu;
    int ld_u;
} segnb_upcat_src;
int segnb_conv_upcat_ok(const segnb_conv_geom* g, int dtype, int Cu);
/* The DATA GRADIENT of such a layer (geometry g: dy -> gradient of the concatenated input) with the Upsample(x2) backward
 * fused into its store pass: the first dst->Cu output channels are not stored at this resolution -- every 2 x 2 window is
 * summed (fp32 sum of the four bf16 values, rounded once: exactly what summing the stored slice afterwards computes) and
 * written to dst->u [N][Ho/2][Wo/2][ld_u], the gradient of u; the remaining channels go to `out` as usual (channel
 * offsets unchanged: out's first Cu channels are left untouched).  For the thin, HBM-bound level (<= 96 output channels,
 * fprop_rw.hip): the multiply-adds are not what bounds it, the 4x-sized gradient slice is.  *_ok: 1 if served. */
int segnb_conv_fprop_upsum_ok(const segnb_conv_geom* g, int dtype, int Cu);
int segnb_conv_fprop_upsum(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, void* out,
                           const segnb_upcat_src* dst, segnb_stream_t stream);
int segnb_conv_fprop_upcat(const segnb_conv_geom* g, int dtype, const void* in, const segnb_upcat_src* src,
                           const void* wpacked, const float* bias, int bias_n, void* out, double* stats,
                           segnb_stream_t stream);
int segnb_conv_wgrad_upcat(const segnb_conv_geom* g, int dtype, const void* in, const segnb_upcat_src* src,
                           const void* dout, float* dwp, int nslab, segnb_stream_t stream);

/* The FORWARD of such a segment on the low-resolution tensor u [N][H][W][Ci], ADDED to out [N][2H][2W][Co]:
 *     out[n, 2Y + py, 2X + px, :] += sum_{a, b < 2} u[n, Y + py - 1 + a, X + px - 1 + b, :] . W[py, px][:, (a, b), :]
 * = conv3x3(pad 1)(Upsample(scale_factor=2)(u)) (lib/models/zf_unet.py:42,78-90) through the ConvTranspose2d(4, 2, 1)
 * identity, the four output phases in ONE launch; `out` holds the skip segment's convolution (+ bias) on entry.
 * wpacked: [4 phases][CoW][4 taps][Ci] bf16 -- phase p = 2 py + px, the tap lists of the phase launches of a 4x4 / stride-2 /
 * pad-1 transposed convolution (segnb.convplan.convt_fwd), summed from the 3x3 parameter by masked pack jobs.
 * stats: BatchNorm statistics of the FINAL stored values (fp64 [SEGNB_STAT_REPLICAS][2][Co]) or NULL.  bf16, Ci % 64 == 0,
 * Ci >= 128, Co > 32, W >= 12 (segnb_upconv_fprop_acc_ok); other shapes keep the 9-tap launch over the concat buffer. */
int segnb_upconv_fprop_acc_ok(int N, int H, int W, int Ci, int Co, int ld_out, int dtype);
/* The forward of a ConvTranspose2d(k=4, stride=2, pad=1) itself (unet16.py:30, DecoderBlock) on that kernel: the same four
 * phases in one launch, but nothing is accumulated -- out = bias + the phase sums -- and bias (fp32, bias_n <= Co entries, or
 * NULL / 0) is added in the accumulator staging.  wpacked = [4][CoW][4][Ci] as above (the packed matrices of
 * segnb.convplan.convt_fwd(4, 2, 1)'s launches, contiguous).  *_ok: 1 if served (else the four phase launches of
 * segnb_conv_fprop). */
int segnb_upconv_fprop_ok(int N, int H, int W, int Ci, int Co, int ld_out, int dtype);
int segnb_upconv_fprop(int dtype, int N, int H, int W, int Ci, int ld_in, const void* in, const void* wpacked, int Co, int CoW,
                       const float* bias, int bias_n, void* out, int ld_out, double* stats, segnb_stream_t stream);
/* ... with the activation that follows it (unet16.py:38-40: ConvTranspose2d -> ReLU) applied in the accumulator staging: out =
 * act(bias + the phase sums); ep->coef must be NULL (no folded BatchNorm), ep->slope in [0, 1].  No statistics. */
int segnb_upconv_fprop_act(int dtype, int N, int H, int W, int Ci, int ld_in, const void* in, const void* wpacked, int Co, int CoW,
                           const float* bias, int bias_n, void* out, int ld_out, const segnb_act_epilogue* ep, segnb_stream_t stream);
int segnb_upconv_fprop_acc(int dtype, int N, int H, int W, int Ci, int ld_in, const void* in, const void* wpacked, int Co,
                           int CoW, void* out, int ld_out, double* stats, segnb_stream_t stream);
int segnb_conv_fprop_bnreduce(const segnb_conv_geom* g, int dtype, const void* in, const void* wpacked, void* out,
                              const segnb_bn_reduce_epilogue* ep, segnb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 classifier head with a handful of classes (zf_unet.py:58,93; tiramisu.py:162; unet16.py:111):
 * fp32 NCHW logits out, fp32 NCHW dlogits in.
 * ------------------------------------------------------------------------------------------- */
int segnb_head_fwd(int dtype, const void* a, int ld_a, int N, int H, int W, int C, const float* w,
                   const float* bias, int K, float* logits, segnb_stream_t stream);
/* da = dlogits . w ; dw (+)= dlogits^T . a ; db (+)= sum dlogits   (dw/db fp32 atomics; caller zeroes
 * unless accumulating) */
int segnb_head_bwd(int dtype, const void* a, int ld_a, int N, int H, int W, int C, int Cp,
                   const float* w, int K, const float* dlogits, void* da, int ld_da, float* dw,
                   float* db, segnb_stream_t stream);

/* Classifier head that is a SMALL CONVOLUTION over an activated tensor (linknet.py:62 `finalconv3 = nn.Conv2d(32, num_classes, 2,
 * padding=1)`; a 1 x 1 head -- unet16.py:111 -- is kh = kw = 1, pad = 0), stride 1, zero padding `pad`:
 *   logits[n][k][ho][wo] = bias[k] + sum_{i < kh, j < kw, c < C} a[n][ho - pad + i][wo - pad + j][c] * w[k][c][i][j]
 * a: [N][Hi][Wi][ld_a] `dtype`; w: the PARAMETER itself, fp32 [K][C][kh][kw] (no packed copy); logits / dlogits: fp32 NCHW
 * [N][K][Ho][Wo], Ho = Hi + 2 pad - kh + 1 (the reference's output layout).  Served when K * kh * kw <= 8 and C <= 64
 * (segnb_head_conv_ok); otherwise the convolution runs as segnb_conv_fprop.
 * _bwd: dw[k][c][i][j] += sum dlogits * a, db[k] += sum dlogits (fp32, the parameters' gradient layout; block-ordered partial
 * sums: reproducible), da[n][h][w][c] = sum dlogits[n][k][h + pad - i][w + pad - j] * w[k][c][i][j] (channels C..Cp-1: zero).
 * act < 0: da is that plain gradient.  act = SEGNB_ACT_NONE / RELU / LEAKY: `a` is the output of a convolution + activation WITHOUT
 * BatchNorm (unet16.py:12-21 ConvRelu, linknet.py:58-61) whose only consumer is this head -- da is then dz = round(round(g) *
 * act'(a)) (act' read from the sign of the activated value) and sums[r][0][c] += sum dz (fp64 [SEGNB_STAT_REPLICAS][2][Cp], that
 * layer's bias-gradient sums): its segnb_bn_act_bwd_reduce pass over (g, a) is folded into this launch. */
int segnb_head_conv_ok(int C, int K, int kh, int kw);
int segnb_head_conv_fwd(int dtype, const void* a, int ld_a, int N, int Hi, int Wi, int C, const float* w, int kh, int kw, int pad,
                        const float* bias, int K, float* logits, segnb_stream_t stream);
int segnb_head_conv_bwd(int dtype, const void* a, int ld_a, int N, int Hi, int Wi, int C, int Cp, const float* w, int kh, int kw,
                        int pad, int K, const float* dlogits, int act, float slope, void* da, int ld_da, float* dw, float* db,
                        double* sums, segnb_stream_t stream);

/* Statistics of a concat prefix WITHOUT a pass over the prefix (FCDenseNet's dense blocks, tiramisu.py:9-44: layer l normalises
 * [input | slice 1 .. slice l-1] with its own BatchNorm; the batch statistics of those channels are the same for every layer).
 * One table [REPLICAS][2][ld] per concat buffer, channel c of the buffer at column c:
 *   segnb_bn_stats_ld        segnb_bn_stats into channel range [stats, stats + Cp) of a table with row stride stats_ld
 *   segnb_bn_act_fwd_stats   segnb_bn_act_fwd (no pooling / upsampling / residual) that also accumulates the statistics of the
 *                            tensor it WRITES into such a range -- the pass that writes a slice sums it
 *   segnb_bn_fwd_fused_ld    segnb_bn_fwd_fused whose statistics are such a range (row stride stats_ld)
 * The table is cleared by the caller once per step (not by the backward of any one layer: several layers read it). */
int segnb_bn_stats_ld(int dtype, const void* x, int ld, int N, int H, int W, int Cp, double* stats, int stats_ld,
                      segnb_stream_t stream);
int segnb_bn_act_fwd_stats(int dtype, const void* y, int ld_y, int N, int H, int W, int Cp, const float* coef, int act,
                           float slope, const float* dropmul, void* out, int ld_out, double* out_stats, int out_stats_ld,
                           segnb_stream_t stream);
int segnb_bn_fwd_fused_ld(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const double* stats,
                          int stats_ld, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, long long* nbt, float* coef, double* bwd_sums_to_clear, int act, float slope,
                          const float* dropmul, void* out, int ld_out, segnb_stream_t stream);

/* Bias gradients of many convolutions WITHOUT BatchNorm in one launch (the executor models' backward: unet16.py:12-21,
 * tiramisu.py:14,52 -- Conv2d(bias=True) followed by no normalisation): job = {double* sums [REPLICAS][2][Cp] as
 * segnb_bn_act_bwd_reduce accumulated them; float* gb [C] or NULL; int C, Cp} (segnb_bias_grad_job_bytes() bytes each, on the
 * device): gb += sum over the replicas of row 0, then the sums are cleared -- segnb_bn_bwd_finalize(gamma = NULL) per layer. */
int segnb_bias_grad_job_bytes(void);
int segnb_bias_grad_multi(const void* jobs, int njobs, segnb_stream_t stream);

/* The network's LAST BatchNorm + activation (+ Dropout2d) layer and the classifier behind it as ONE pass over the layer's
 * pre-BatchNorm output y, in both directions (zf_unet.py:56-58,91-93: double_conv_layer -> Conv2d(filters, num_classes, 1);
 * autograd's BatchNorm / ReLU / Dropout2d / Conv2d backward nodes of the same lines):
 *   segnb_bn_fwd_fused_head = segnb_bn_fwd_fused (same finalize protocol, same rounding of the activated values to the storage
 *     type) + segnb_head_fwd on those values; out may be NULL -- the activated tensor then never exists in memory;
 *   segnb_head_bn_bwd = segnb_head_bwd (da rounded to the storage type as it would have been stored; dw / db accumulate, the
 *     activated values recomputed from y) + segnb_bn_act_bwd_reduce (dz stored, sums accumulated).  The layer's apply pass
 *     (segnb_bn_bwd_apply_fused on dz) follows as usual.
 * segnb_head_fused_ok: 1..4 classes, Cp / 8 a power of two <= 32. */
int segnb_head_fused_ok(int K, int Cp);
int segnb_bn_fwd_fused_head(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const double* stats,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, long long* nbt, float* coef, double* bwd_sums_to_clear, int act, float slope,
                            const float* dropmul, void* out, int ld_out, const float* head_w, const float* head_b, int K,
                            float* logits, segnb_stream_t stream);
int segnb_head_bn_bwd(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const float* coef, int act,
                      float slope, const float* dropmul, const float* head_w, int K, const float* dlogits, void* dz, int ld_dz,
                      double* sums, float* dw, float* db, segnb_stream_t stream);
/* The second pass of that layer when segnb_head_bn_bwd was given dz = NULL (sums only): dz is a function of the d(logits) map and y
 * alone, so it is recomputed here -- the expressions and roundings of segnb_head_bn_bwd, bit for bit -- and
 *     dy = round(a * (dz - c1 - yhat * c2))
 * leaves, with (a, c1, c2) taken from the sums inside the launch (the fused finalize of segnb_bn_bwd_apply_fused: bcoef, dgamma /
 * dbeta, the forward statistics cleared).  The Cp-channel dz tensor between the two passes (lib/models/zf_unet.py:56-58,91-93:
 * 103 MB at 224 x 224, bs 32) is neither written nor read. */
int segnb_head_bn_bwd_apply(int dtype, const void* y, int ld_y, int N, int H, int W, int C, int Cp, const float* coef,
                            const double* sums, const float* gamma, float* bcoef, float* dgamma, float* dbeta, int accumulate,
                            double* fwd_stats_to_clear, int act, float slope, const float* dropmul, const float* head_w, int K,
                            const float* dlogits, void* dy, int ld_dy, segnb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-pixel binary losses and metrics (lib/losses.py:7-101, lib/metrics.py:9-43).
 *   loss = (w_bce*bce2 + w_focal*focal + w_jaccard*jaccard + w_sjaccard*smooth_jaccard + w_dice*dice)/norm
 * bce2 is the reference's double-sigmoid BCE (losses.py:51-53).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float w_bce, w_focal, w_jaccard, w_sjaccard, w_dice;
    float smooth;      /* SmoothJaccardLoss smooth (100) */
    float eps;         /* 1e-7 of JaccardLoss / DiceLoss */
    float norm;        /* divisor of the weighted sum */
    int focal_mean;    /* FocalLossBinary size_average */
    int bce_sum;       /* BCEWithSigmoidLoss(size_average=False): sum instead of mean (losses.py:47,53) */
    float focal_gamma; /* FocalLossBinary gamma (losses.py:84; 2 = the reference default) */
} segnb_loss_spec;

/* sums fp64 [8]: sum bce2, sum focal, sum p*t, sum p, sum t, #correct@0.5, #pixels, (unused);
 * accumulated atomically, caller zeroes.  A data-parallel job all-reduces these 8 doubles. */
int segnb_seg_loss_reduce(const float* logits, const long long* target, long long n, float focal_gamma,
                          double* sums, segnb_stream_t stream);
/* out fp32 [8]: loss, soft IoU (metrics.py:14-20), pixel accuracy, GI, GU, bce mean, -, - */
int segnb_seg_loss_finalize(const double* sums, const segnb_loss_spec* spec, float* out,
                            segnb_stream_t stream);
/* segnb_seg_loss_reduce + segnb_seg_loss_finalize (and the zero fill of the sums before them) as ONE launch on one device:
 * work = 128 doubles, ZERO on entry (allocate once, zeroed) and left zero on return -- the last block to finish turns the sums into
 * `out` and clears them.  A data-parallel job keeps the two-launch form (the sums are all-reduced in between). */
int segnb_seg_loss_reduce_finalize(const float* logits, const long long* target, long long n,
                                   const segnb_loss_spec* spec, double* work, float* out, segnb_stream_t stream);
/* dlogits = (*grad_out) * dloss/dlogits  (grad_out: device fp32 scalar, the (B*loss).backward() seed
 * of torch_train.py:187-188) */
int segnb_seg_loss_bwd(const float* logits, const long long* target, long long n,
                       const double* sums, const float* fin, const segnb_loss_spec* spec,
                       const float* grad_out, float* dlogits, segnb_stream_t stream);

/* reduce=False forms (BCEWithSigmoidLoss(reduce=False), losses.py:47-53): per-pixel loss map, kind 0 = double-sigmoid
 * BCE element, 1 = focal element, and its backward dlogits = grad_out[i] * d(map[i])/d(logits[i]) */
int segnb_seg_loss_map(const float* logits, const long long* target, long long n, int kind, float gamma,
                       float* out, segnb_stream_t stream);
int segnb_seg_loss_map_bwd(const float* logits, const long long* target, long long n, int kind, float gamma,
                           const float* grad_out, float* dlogits, segnb_stream_t stream);

/* *out = max |x[i]| over a flat, 16-byte aligned fp32 buffer: the gradient-explosion monitor of
 * torch_train.py:199-205 (there: one reduction and one host sync per parameter tensor) as ONE launch over the flat
 * gradient buffer. */
int segnb_absmax_f32(const float* x, long long n, float* out, segnb_stream_t stream);

/* PRCurveMeter.update (lib/train_utils.py:109-125): hist[c][b] += #pixels of target class c (0 / non-zero) whose
 * sigmoid(logit) exceeds exactly b of the `nthr` ascending thresholds; hist: unsigned 64-bit [2][nthr + 1],
 * accumulated (caller zeroes).  tp/fp at threshold i = suffix sums over b > i. */
int segnb_pr_histogram(const float* logits, const long long* target, long long n, const float* thresholds,
                       int nthr, unsigned long long* hist, segnb_stream_t stream);

/* plain SGD p -= lr * g over a flat buffer (torch.optim.SGD of torch_train.py:71) */
int segnb_sgd_step(float* p, const float* g, long long n, float lr, segnb_stream_t stream);

/* torch.optim.RMSprop(lr, alpha, eps) -- v = alpha*v + (1-alpha)*g^2; p -= lr*g/(sqrt(v)+eps) -- and
 * torch.optim.Adam(lr, betas, eps) at step number `step` (1-based; bias corrections computed on the host in fp64)
 * over flat buffers: get_optimizer('rms' | 'adam'), torch_train.py:73-77.  State buffers are caller-owned. */
int segnb_rmsprop_step(float* p, const float* g, float* square_avg, long long n, float lr, float alpha,
                       float eps, segnb_stream_t stream);
int segnb_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                    float beta1, float beta2, float eps, int step, segnb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Launch plans: the launch list of a model's forward (or backward) replayed from C.  The reference drives its step from
 * Python one torch operator at a time (torch_train.py:183-190); this path's ~300 launches per step cost the Python
 * launcher 10-14 us each, which bounds small-batch steps.  Between segnb_plan_begin and segnb_plan_end every top-level
 * entry point called on this thread records itself (pointer arguments by value, host structs copied) and executes
 * normally; segnb_plan_run replays the list with the same pointers -- the caller keeps the buffers alive and unchanged in
 * place.  Entry points that read HOST arrays (segnb_pack_weight, segnb_unpack_wgrad, the uint8 input functions) and
 * segnb_tune make the plan unusable: segnb_plan_end then returns *plan_out = NULL.
 * segnb_stream_fork / _join: `side` waits for what has been issued on `main` / `main` waits for `side` (events).
 * ------------------------------------------------------------------------------------------- */
int segnb_plan_begin(void);
int segnb_plan_end(void** plan_out, int* nops);
int segnb_plan_run(void* plan);
int segnb_plan_destroy(void* plan);
int segnb_stream_fork(segnb_stream_t main_stream, segnb_stream_t side_stream);
/* The same dependency without a marker packet between two dependent kernels of `main`: segnb_stream_fork_arm(main) BEFORE the last
 * launch the side stream has to wait for, segnb_stream_fork_commit(main, side) after it.  When that launch is a BatchNorm-backward
 * apply pass (segnb_bn_bwd_apply*, the launches lib/modules/abn/functions.py:118's dx precedes every weight gradient with) its
 * completion event rides on the kernel's own dispatch; otherwise commit behaves as segnb_stream_fork.  (MI355X: +1.7 us instead of
 * +5.5 us on the main queue per fork, tools/fork_cost.hip.)  Nothing else may be launched on `main` between that launch and commit. */
int segnb_stream_fork_arm(segnb_stream_t main_stream);
int segnb_stream_fork_commit(segnb_stream_t main_stream, segnb_stream_t side_stream);
int segnb_stream_join(segnb_stream_t main_stream, segnb_stream_t side_stream);
/* hipEventRecord(event, stream) as a recordable call: timing events inside a replayed launch list (bench.py) */
int segnb_event_record(void* event, segnb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
