// Shared device/host helpers for libsegnb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/segnb_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct bf16_t {
    unsigned short bits;
};

// ------------------------------------------------------------------------------------------------
// error plumbing (never throw across the ABI)
// ------------------------------------------------------------------------------------------------
void segnb_set_error(const char* fmt, ...);

#define SEGNB_CHECK_ARG(cond, msg)                                   \
    do {                                                             \
        if (!(cond)) {                                               \
            segnb_set_error("%s: bad argument: %s", __func__, msg);  \
            return SEGNB_E_BADARG;                                   \
        }                                                            \
    } while (0)

// A launch that can carry the completion event of an armed cross-stream fork (segnb_stream_fork_arm / _commit, runtime.hip): the
// event rides on the kernel's own dispatch packet (hipExtLaunchKernelGGL stopEvent) instead of a marker packet of its own between
// two dependent kernels of the queue -- tools/fork_cost.hip: +1.7 us instead of +5.5 us per fork on MI355X.
hipEvent_t segnb_take_armed_event(hipStream_t stream);
#define SEGNB_LAUNCH_FORKABLE(kernel, grid, block, shmem, stream, ...)                                          \
    do {                                                                                                        \
        hipEvent_t seg_ev_ = segnb_take_armed_event(stream);                                                    \
        if (seg_ev_ != nullptr)                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, seg_ev_, 0, __VA_ARGS__);        \
        else                                                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                \
    } while (0)

#define SEGNB_LAUNCH_CHECK()                                                                  \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess) {                                                              \
            segnb_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));       \
            return (int)e__;                                                                  \
        }                                                                                     \
    } while (0)

// ------------------------------------------------------------------------------------------------
// launch plans (runtime.hip: segnb_plan_begin / _end / _run): while a plan is being recorded on this thread every
// top-level C-ABI call appends itself -- function pointer + its arguments, host structs copied into the plan -- and then
// executes as usual; segnb_plan_run replays the list from C (the Python launcher needs 10-14 us per launch).
// ------------------------------------------------------------------------------------------------
#include <functional>
#include <type_traits>
#include <tuple>
bool segnb_plan_recording();
void segnb_plan_push(std::function<int()> op, const char* name);
const void* segnb_plan_dup(const void* p, size_t bytes);
struct SegnbPlanScope {          // nested entry points (conv_wgrad_partial -> conv_wgrad) record once, at the top
    bool top;
    SegnbPlanScope();
    ~SegnbPlanScope();
};
template <class T>
inline T segnb_plan_keep(T v) {
    // (a HOST struct passed by pointer is copied into the plan by an overload below: without one the recorded call would replay
    // with a dangling pointer -- the compiler refuses the entry point instead)
    static_assert(!(std::is_pointer<T>::value && std::is_class<typename std::remove_pointer<T>::type>::value),
                  "segnb_plan_keep: add an overload that copies this struct into the plan");
    return v;
}
inline const segnb_conv_geom* segnb_plan_keep(const segnb_conv_geom* g) {
    return g ? (const segnb_conv_geom*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_loss_spec* segnb_plan_keep(const segnb_loss_spec* g) {
    return g ? (const segnb_loss_spec*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_bn_reduce_epilogue* segnb_plan_keep(const segnb_bn_reduce_epilogue* g) {
    return g ? (const segnb_bn_reduce_epilogue*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_bn_apply_epilogue* segnb_plan_keep(const segnb_bn_apply_epilogue* g) {
    return g ? (const segnb_bn_apply_epilogue*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_upcat_src* segnb_plan_keep(const segnb_upcat_src* g) {
    return g ? (const segnb_upcat_src*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_operand_tf* segnb_plan_keep(const segnb_operand_tf* g) {
    return g ? (const segnb_operand_tf*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_wgrad_target* segnb_plan_keep(const segnb_wgrad_target* g) {
    return g ? (const segnb_wgrad_target*)segnb_plan_dup(g, sizeof(*g)) : g;
}
inline const segnb_act_epilogue* segnb_plan_keep(const segnb_act_epilogue* g) {
    return g ? (const segnb_act_epilogue*)segnb_plan_dup(g, sizeof(*g)) : g;
}
template <class... P, class... A>
inline void segnb_plan_record_call(const char* name, int (*fn)(P...), A... args) {
    static_assert(sizeof...(P) == sizeof...(A), "argument count");
    std::tuple<P...> kept(segnb_plan_keep((P)args)...);
    segnb_plan_push([fn, kept]() { return std::apply(fn, kept); }, name);
}
// first statement of a recordable extern "C" entry point (entry points that take HOST arrays -- tap offsets, mean / std --
// are not recordable: SEGNB_PLAN_REFUSE marks the plan being recorded as unusable)
extern int g_segnb_census_on;          // segnb_tune("call_census"): count the top-level entry points by name (segnb_debug_census)
void segnb_census(const char* name);
#define SEGNB_PLAN_RECORD(fn, ...)                                                           \
    SegnbPlanScope plan_scope__;                                                             \
    if (plan_scope__.top && g_segnb_census_on) segnb_census(#fn);                            \
    if (plan_scope__.top && segnb_plan_recording()) segnb_plan_record_call(#fn, fn, __VA_ARGS__)
void segnb_plan_refuse(const char* why);
#define SEGNB_PLAN_REFUSE(why)                                         \
    SegnbPlanScope plan_scope__;                                       \
    if (plan_scope__.top && g_segnb_census_on) segnb_census(__func__); \
    if (plan_scope__.top && segnb_plan_recording()) segnb_plan_refuse(why)

int segnb_num_cus();
int segnb_knob_fprop_dma();       // runtime.hip: segnb_tune() knobs
int segnb_knob_fprop_dma_cfg();
int segnb_knob_fprop_dma_dbg();
int segnb_knob_fprop_nostats();   // 1: launches without statistics run the statistics-free instantiation of conv_fprop_ws_kernel
int segnb_knob_rw_store_waves();  // 2 or 4 store waves in conv_fprop_rw_kernel
int segnb_knob_bnreduce_fused();  // 1: segnb_conv_fprop_bnreduce_ok may say yes
int segnb_knob_fprop_deepk();     // 1: conv_fprop_deepk_kernel serves the shapes it applies to
int segnb_knob_fprop_thin();      // 1: conv_thin_kernel (fprop_thin.hip) serves <= 16 -> >= 48 channel stride-1 3x3 launches
// 1 = launched, 0 = not served (fprop_thin.hip); bn: the BatchNorm-backward reduction epilogue of segnb_conv_fprop_bnreduce or NULL
int segnb_fprop_thin_try(const segnb_conv_geom* g, const void* in, const void* wpacked, void* out, hipStream_t stream,
                         const segnb_bn_reduce_epilogue* bn);
int segnb_fprop_thin_ok(const segnb_conv_geom* g);
int segnb_knob_fprop_upd();       // 1: 4x4 / stride-2 gathers (ntaps 16, in_step 2) on the plane-gather form of conv_fprop_ws_kernel
int segnb_knob_fprop_drop();      // 1: segnb_conv_fprop_drop_ok may say yes
int segnb_knob_fprop_mask();      // 1: conv_fprop_ws_kernel serves activation-mask data gradients (segnb_conv_fprop_bnreduce, coef NULL)
int segnb_fprop_dma_actmask_ok(const segnb_conv_geom* g);
int segnb_fprop_roll_actmask_ok(const segnb_conv_geom* g);     // fprop_roll.hip: conv_roll_kernel (EPI = 3) serves g      // fprop_dma.hip: the MASK instantiation serves g
int segnb_knob_fprop_mf16();      // 1: conv_fprop_ws_kernel issues v_mfma_f32_16x16x32_bf16, 0: 32x32x16
int segnb_knob_fprop_rw();
int segnb_knob_fprop_ksplit();   // conv_fprop_ws_kernel split K: 0 off, 1 automatic (default), 2 / 4 forced where it applies
// head backward's per-(device, stream) partial-sum scratch and its fixed-order finish launch (head_loss.hip)
float* segnb_head_scratch(size_t bytes, hipStream_t stream);
void segnb_head_bwd_finish(const float* part, int gx, int gy, int K, int C, int CT, float* dw, float* db, hipStream_t stream);
int segnb_knob_pack_blocks();    // persistent blocks of segnb_pack_weight_multi (0: one block per tile)
int segnb_knob_fprop_roll();     // 0: off, 1: conv_roll_kernel with 16-column strips, 2: 32-column strips (segnb_tune "fprop_roll")
int segnb_knob_wg_cu_pct();      // segnb_tune "wg_cu_pct": 0 = default share of the CUs for the 64x64-tile weight gradients
int segnb_knob_conv_cus();        // CUs the persistent fprop / dgrad kernels size their grids for (segnb_tune "conv_cu_pct")
int segnb_fprop_dma_read_stamps(unsigned long long* host_dst);
// fast path of segnb_conv_wgrad (wgrad_s1.hip): 1 = handled, 0 = not applicable, else error
// (drop / ld_drop / stats_ld: the Dropout2d multipliers and the statistics row stride of segnb_conv_fprop_drop; NULL / 0 / 0: off)
int segnb_fprop_s1_try(const segnb_conv_geom* g, const void* in, const void* wpacked, const float* bias, int bias_n,
                       void* out, double* stats, hipStream_t stream, const float* drop = nullptr, int ld_drop = 0,
                       int stats_ld = 0);
// direct-to-LDS pipeline for Ci % 64 == 0 (fprop_dma.hip): 1 = handled, 0 = not applicable, else error
int segnb_fprop_dma_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked,
                        unsigned w_bytes, const float* bias, int bias_n, void* out, double* stats,
                        hipStream_t stream, const segnb_act_epilogue* ep = nullptr, const segnb_upcat_src* uc = nullptr,
                        const segnb_bn_reduce_epilogue* bn = nullptr);
// resident-weights pipeline for the thin layers, Ci <= 96 and Co <= 96 (fprop_rw.hip)
int segnb_fprop_rw_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked,
                       unsigned w_bytes, const float* bias, int bias_n, void* out, double* stats,
                       hipStream_t stream, const segnb_bn_reduce_epilogue* bn = nullptr,
                       const segnb_act_epilogue* ep = nullptr, const segnb_upcat_src* uc = nullptr,
                       const segnb_upcat_src* upsum = nullptr);
// rolling-window kernel for the thin layers, weights in registers, no block-level synchronisation (fprop_roll.hip)
int segnb_fprop_roll_try(const segnb_conv_geom* g, const void* in, unsigned in_bytes, const void* wpacked, unsigned w_bytes,
                         const float* bias, int bias_n, void* out, double* stats, hipStream_t stream,
                         const segnb_bn_reduce_epilogue* bn = nullptr, const segnb_operand_tf* tf = nullptr,
                         const segnb_upcat_src* uc = nullptr);
// first layer: 8-channel (3 padded) input, <= 32 output channels (fprop_c8.hip)
int segnb_fprop_c8_try(const segnb_conv_geom* g, const void* in, const void* wpacked, const float* bias, int bias_n,
                       void* out, double* stats, hipStream_t stream, const segnb_act_epilogue* ep = nullptr);
// bna (optional): the dy operand is not in memory -- it is the BatchNorm-backward apply of the layer, recomputed while
// the tile is staged: dy = round(a (round(g act'(z)) - c1 - yhat c2)) from the incoming gradient g and the pre-BatchNorm
// output y (segnb_conv_wgrad_bnapply; the arithmetic and roundings of bn_bwd_apply_kernel's direct form)
struct segnb_wgrad_bnapply {
    const void* g;
    int ld_g;
    const void* y;
    int ld_y;
    const float* coef;      // [4][Cp]: scale, shift, mean, invstd (segnb_bn_finalize)
    const float* bcoef;     // [3][Cp]: a, c1, c2 (segnb_bn_bwd_finalize)
    int Cp;
    int act;
    float slope;
};
// tgt: the armed segnb_wgrad_target or NULL.  With a target a single-slab launch writes the parameter's gradient from its
// accumulators (returns 2 instead of 1: nothing left to do); launches with several slabs leave them unreduced for
// segnb_wgrad_to_param
int segnb_wgrad_s1_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab,
                       hipStream_t stream, bool partial, const segnb_wgrad_bnapply* bna = nullptr,
                       const segnb_upcat_src* uc = nullptr, const segnb_wgrad_target* tgt = nullptr);      // partial: leave the nslab slabs unreduced
// runtime.hip: the target armed by segnb_wgrad_target_arm on this thread (disarmed by the call), or NULL
const segnb_wgrad_target* segnb_take_wgrad_target();
// wgrad_s1.hip: sum the nslab slabs [Cop][ntaps][Cip] of dwp (fixed order) into the parameter-layout gradient of tgt; rezero: slab 0
// is cleared afterwards (the atomics of the general kernel need a zeroed workspace)
void segnb_wgrad_to_param(float* dwp, int Cop, int ntaps, int Cip, int nslab, const segnb_wgrad_target* tgt, bool rezero,
                          hipStream_t stream);
int segnb_wgrad_s1_slabs(const segnb_conv_geom* g);
void segnb_slab_reduce(float* dwp, long long total, int nslab, hipStream_t stream);
// rolling-window weight gradient of the thin layers (wgrad_roll.hip); tfx / tfd: operands recomputed on load (or NULL)
bool segnb_wgrad_roll_applies(const segnb_conv_geom* g);
int segnb_wgrad_roll_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab, hipStream_t stream,
                         bool partial, const segnb_operand_tf* tfx = nullptr, const segnb_operand_tf* tfd = nullptr);
int segnb_knob_wgrad_roll();
// the first layer (8 padded input channels): conv_wgrad_c8roll_kernel; bna: dy recomputed from (g, y), dout ignored
bool segnb_wgrad_c8roll_applies(const segnb_conv_geom* g);
int segnb_wgrad_c8roll_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab, hipStream_t stream,
                           bool partial, const segnb_wgrad_bnapply* bna = nullptr);
int segnb_knob_wgrad_c8roll();
// strided / wide-window tile kernel (wgrad_s1.hip: conv_wgrad_sx_kernel): same protocol
int segnb_wgrad_sx_try(const segnb_conv_geom* g, const void* in, const void* dout, float* dwp, int nslab, hipStream_t stream,
                       bool partial);
int segnb_wgrad_sx_slabs(const segnb_conv_geom* g);
// forward / data gradient of strided, transposed-phase, 2x2, 1x1 and 16-channel-multiple convolutions on halo tiles (fprop_sx.hip)
int segnb_fprop_sx_try(const segnb_conv_geom* g, const void* in, const void* wpacked, const float* bias, int bias_n, void* out,
                       double* stats, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// element helpers: 8 channels per thread ("chunk8"), fp32 math
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ static __forceinline__ float from_f32(float v) { return v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float to_f32(bf16_t v) { return bf16_bits_to_f32(v.bits); }
    __device__ static __forceinline__ bf16_t from_f32(float v) {
        bf16_t r;
        r.bits = f32_to_bf16_bits(v);
        return r;
    }
    __device__ static __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

// load / store 8 consecutive channels as fp32
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// two fp32 -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32 through the vector conversion (two scalar
// conversions + shift + or compile to four instructions; same RNE result)
typedef float segnb_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 segnb_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const segnb_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, segnb_bf16x2_t));
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    uint4 u;
    u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
    u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
// value as it will read back from a T-typed store
__device__ __forceinline__ float round_as(float v, const float*) { return v; }
__device__ __forceinline__ float round_as(float v, const bf16_t*) {
    return bf16_bits_to_f32(f32_to_bf16_bits(v));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
