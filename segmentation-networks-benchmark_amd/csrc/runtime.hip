// Error plumbing and device queries for libsegnb_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

static thread_local char g_err[512] = "";

void segnb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int segnb_num_cus() {
    static int cus = 0;
    if (cus > 0) return cus;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cus;
}

extern "C" const char* segnb_last_error(void) { return g_err; }
extern "C" int segnb_version(void) { return 1; }
extern "C" int segnb_device_cus(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        segnb_set_error("segnb_device_cus: no HIP device");
        return -1;
    }
    return segnb_num_cus();
}

// ---- launch plans ---------------------------------------------------------------------------------------------------
#include <memory>
#include <vector>
namespace {
struct Plan {
    std::vector<std::function<int()>> ops;
    std::vector<const char*> names;
    std::vector<std::unique_ptr<unsigned char[]>> arena;
    bool refused = false;
    char why[128] = "";
};
thread_local Plan* g_rec = nullptr;
thread_local int g_depth = 0;
}  // namespace

bool segnb_plan_recording() { return g_rec != nullptr; }
void segnb_plan_push(std::function<int()> op, const char* name) {
    g_rec->ops.push_back(std::move(op));
    g_rec->names.push_back(name);
}
const void* segnb_plan_dup(const void* p, size_t bytes) {
    g_rec->arena.emplace_back(new unsigned char[bytes]);
    memcpy(g_rec->arena.back().get(), p, bytes);
    return g_rec->arena.back().get();
}
void segnb_plan_refuse(const char* why) {
    g_rec->refused = true;
    snprintf(g_rec->why, sizeof(g_rec->why), "%s", why);
}
SegnbPlanScope::SegnbPlanScope() : top(g_depth == 0) { ++g_depth; }
SegnbPlanScope::~SegnbPlanScope() { --g_depth; }

// segnb_tune("plan_profile", 1): segnb_plan_run times every replayed call on the host (per entry point); ("plan_profile", 2)
// prints the table to stderr and clears it; 0 = off
#include <chrono>
#include <map>
#include <string>
namespace {
int g_plan_profile = 0;
std::map<std::string, std::pair<long, double>> g_plan_prof;
int plan_run_profiled(Plan* p) {
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = p->ops[i]();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        auto& e = g_plan_prof[p->names[i]];
        e.first += 1;
        e.second += us;
        if (rc != 0) return rc;
    }
    return 0;
}
void plan_profile_dump() {
    double tot = 0;
    long n = 0;
    for (auto& kv : g_plan_prof) {
        fprintf(stderr, "plan_profile %-34s %7ld calls %9.1f us  %6.2f us/call\n", kv.first.c_str(), kv.second.first,
                kv.second.second, kv.second.second / kv.second.first);
        tot += kv.second.second;
        n += kv.second.first;
    }
    fprintf(stderr, "plan_profile %-34s %7ld calls %9.1f us  %6.2f us/call\n", "total", n, tot, n ? tot / n : 0.0);
    g_plan_prof.clear();
}
}  // namespace

// ---- call census (replay guard) ---------------------------------------------------------------------------------------
int g_segnb_census_on = 0;
namespace {
thread_local std::map<std::string, long> g_census;
}
void segnb_census(const char* name) { ++g_census[name]; }
extern "C" int segnb_debug_census(char* buf, int cap) {
    SEGNB_CHECK_ARG(buf != nullptr && cap > 0, "NULL buffer");
    std::string out;
    for (auto& kv : g_census) out += kv.first + " " + std::to_string(kv.second) + "\n";
    g_census.clear();
    snprintf(buf, (size_t)cap, "%s", out.c_str());
    return 0;
}

extern "C" int segnb_plan_begin(void) {
    delete g_rec;            // (a recording abandoned by an exception on the host side)
    g_rec = new Plan();
    return 0;
}

// *plan_out = the recorded plan, or NULL when a call that cannot be replayed was made while recording (segnb_last_error says
// which); *nops = number of recorded launches
extern "C" int segnb_plan_end(void** plan_out, int* nops) {
    SEGNB_CHECK_ARG(g_rec != nullptr && plan_out != nullptr, "no plan is being recorded");
    Plan* p = g_rec;
    g_rec = nullptr;
    if (nops != nullptr) *nops = (int)p->ops.size();
    if (p->refused) {
        segnb_set_error("segnb_plan_end: not replayable: %s", p->why);
        delete p;
        *plan_out = nullptr;
        return 0;
    }
    *plan_out = p;
    return 0;
}

extern "C" int segnb_plan_run(void* plan) {
    SEGNB_CHECK_ARG(plan != nullptr && g_rec == nullptr, "NULL plan, or a plan is being recorded");
    Plan* p = (Plan*)plan;
    if (g_plan_profile) return plan_run_profiled(p);
    for (auto& op : p->ops) {
        const int rc = op();
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int segnb_plan_destroy(void* plan) {
    delete (Plan*)plan;
    return 0;
}

// `side` waits for everything issued so far on `main` (fork) / `main` waits for `side` (join): what the two-stream backward
// needs, as recordable entry points (events from a small ring: an event may be re-recorded once its wait has been issued)
namespace {
hipEvent_t next_event() {
    static thread_local hipEvent_t ring[64];
    static thread_local int n = 0, made = 0;
    if (made < 64) {
        if (hipEventCreateWithFlags(&ring[made], hipEventDisableTiming) != hipSuccess) return nullptr;
        ++made;
        return ring[made - 1];
    }
    n = (n + 1) & 63;
    return ring[n];
}
int wait_on(hipStream_t waiter, hipStream_t signaller, const char* who) {
    hipEvent_t ev = next_event();
    hipError_t e = ev == nullptr ? hipErrorOutOfMemory : hipEventRecord(ev, signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
    if (e != hipSuccess) {
        segnb_set_error("%s: %s", who, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
}  // namespace
// Armed fork: segnb_stream_fork_arm(main) before the LAST launch the side stream has to wait for, segnb_stream_fork_commit(main,
// side) after it.  If that launch is one that can carry an event (SEGNB_LAUNCH_FORKABLE: the BatchNorm-backward apply passes) the
// event is part of its dispatch packet and the main queue holds no marker between that kernel and the next one; otherwise -- no
// such launch in between, or the stream is being captured into a graph -- commit is an ordinary segnb_stream_fork.
namespace {
thread_local hipEvent_t g_armed_ev = nullptr;
thread_local hipStream_t g_armed_stream = nullptr;
thread_local bool g_armed_taken = false;
}  // namespace
// Every SEGNB_LAUNCH_FORKABLE launch on the armed stream between arm and commit carries an event; commit waits on the one of the
// LAST such launch (the stream is in order: it covers the earlier ones).  A second forkable launch used to find the event taken
// and go out bare, so the side stream's dependency on it was silently dropped (today every caller issues exactly one).
hipEvent_t segnb_take_armed_event(hipStream_t stream) {
    if (g_armed_ev == nullptr || stream != g_armed_stream) return nullptr;
    if (g_armed_taken) {
        hipEvent_t ev = next_event();        // (an event is recorded once per dispatch: a later launch takes a fresh one)
        if (ev == nullptr) return nullptr;   // commit then waits on the earlier launch only -- and says so below
        g_armed_ev = ev;
    }
    g_armed_taken = true;
    return g_armed_ev;
}
extern "C" int segnb_stream_fork_arm(segnb_stream_t main_stream) {
    SEGNB_PLAN_RECORD(segnb_stream_fork_arm, main_stream);
    g_armed_ev = nullptr;
    g_armed_taken = false;
    static const bool off = getenv("SEGNB_FORK_ON_DISPATCH") != nullptr && getenv("SEGNB_FORK_ON_DISPATCH")[0] == '0';
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (off || hipStreamIsCapturing((hipStream_t)main_stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 0;
    g_armed_ev = next_event();
    g_armed_stream = (hipStream_t)main_stream;
    return 0;
}
extern "C" int segnb_stream_fork_commit(segnb_stream_t main_stream, segnb_stream_t side_stream) {
    SEGNB_PLAN_RECORD(segnb_stream_fork_commit, main_stream, side_stream);
    const bool carried = g_armed_ev != nullptr && g_armed_taken && g_armed_stream == (hipStream_t)main_stream;
    hipEvent_t ev = g_armed_ev;
    g_armed_ev = nullptr;
    g_armed_taken = false;
    if (!carried) return wait_on((hipStream_t)side_stream, (hipStream_t)main_stream, "segnb_stream_fork_commit");
    const hipError_t e = hipStreamWaitEvent((hipStream_t)side_stream, ev, 0);
    if (e != hipSuccess) {
        segnb_set_error("segnb_stream_fork_commit: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
extern "C" int segnb_stream_fork(segnb_stream_t main_stream, segnb_stream_t side_stream) {
    SEGNB_PLAN_RECORD(segnb_stream_fork, main_stream, side_stream);
    return wait_on((hipStream_t)side_stream, (hipStream_t)main_stream, "segnb_stream_fork");
}
// hipEventRecord as a recordable entry point: bench.py's per-launch timing events are part of the replayed lists, so the
// kernels are timed in the configuration that is benchmarked (the same launcher, the same overlap of the two streams)
extern "C" int segnb_event_record(void* event, segnb_stream_t stream) {
    SEGNB_PLAN_RECORD(segnb_event_record, event, stream);
    SEGNB_CHECK_ARG(event != nullptr, "NULL event");
    const hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    if (e != hipSuccess) {
        segnb_set_error("segnb_event_record: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
extern "C" int segnb_stream_join(segnb_stream_t main_stream, segnb_stream_t side_stream) {
    SEGNB_PLAN_RECORD(segnb_stream_join, main_stream, side_stream);
    return wait_on((hipStream_t)main_stream, (hipStream_t)side_stream, "segnb_stream_join");
}

// ---- tuning knobs (A/B measurements and tests; defaults come from the environment once) ------------------------
static int g_fprop_dma = -2;       // -2 = not initialised, 0 = off, 1 = on
static int g_fprop_dma_cfg = -2;   // -1 = automatic, >= 0 forced configuration

int segnb_knob_fprop_dma() {
    if (g_fprop_dma == -2) {
        const char* e = getenv("SEGNB_FPROP_DMA");
        g_fprop_dma = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_fprop_dma;
}
int segnb_knob_fprop_dma_cfg() {
    if (g_fprop_dma_cfg == -2) {
        const char* e = getenv("SEGNB_FPROP_DMA_CFG");
        g_fprop_dma_cfg = e ? atoi(e) : -1;
    }
    return g_fprop_dma_cfg;
}
static int g_fprop_rw = 1;
int segnb_knob_fprop_rw() { return g_fprop_rw && segnb_knob_fprop_dma_cfg() < 0; }
static int g_fprop_roll = -2;      // conv_roll_kernel (fprop_roll.hip): 0 off, 1 = 16-column strips, 2 = 32-column strips
int segnb_knob_fprop_roll() {
    if (g_fprop_roll == -2) {
        const char* e = getenv("SEGNB_FPROP_ROLL");
        g_fprop_roll = e != nullptr ? atoi(e) : 2;
    }
    return segnb_knob_fprop_dma() ? g_fprop_roll : 0;
}
// conv_wgrad_roll_kernel (wgrad_roll.hip) for the plain segnb_conv_wgrad of the thin layers: 0 off (default), 1 on.  Measured
// on MI355X: 59.4 us against 60.8 us of conv_wgrad_s1x9_kernel alone, 5.41 / 5.39 against 5.37 / 5.39 ms per step in situ
// (profiles/r04_ab.txt) -- no gain, so the plain path stays on the tile kernel; segnb_conv_wgrad_tf always runs on it.
// blocks of the batched weight pack (segnb_pack_weight_multi): 0 = one per tile; > 0: that many persistent blocks (a pack running
// beside the forward's first levels, SEGNB_PACK_OVERLAP=1, throttled to a share of the HBM bandwidth)
static int g_pack_blocks = 0;
int segnb_knob_pack_blocks() { return g_pack_blocks; }

static int g_wgrad_roll = 0;          // segnb_tune "wgrad_roll"
int segnb_knob_wgrad_roll() { return g_wgrad_roll; }
// conv_wgrad_c8roll_kernel (wgrad_roll.hip) for the first layer's weight gradient (8 padded input channels): 1 on, 0 off
static int g_wgrad_c8roll = 1;        // segnb_tune "wgrad_c8roll"
int segnb_knob_wgrad_c8roll() { return g_wgrad_c8roll; }
static int g_conv_cu_pct = 100;
int segnb_knob_conv_cus() {
    const int n = segnb_num_cus() * g_conv_cu_pct / 100;
    return n < 1 ? 1 : n;
}
static int g_wg_cu_pct = 0;        // 0 = SEGNB_WG_CU_FRACTION / built-in default; else % of the CUs for the wide weight gradients
int segnb_knob_wg_cu_pct() { return g_wg_cu_pct; }
static int g_fprop_deepk = -2;     // few pixels x few channels x deep K forwards on conv_fprop_deepk_kernel (1, default)
int segnb_knob_fprop_deepk() {
    if (g_fprop_deepk == -2) {
        const char* e = getenv("SEGNB_FPROP_DEEPK");
        g_fprop_deepk = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_fprop_deepk;
}
static int g_fprop_thin = 1;       // thin input (<= 16 channels), wide output: conv_thin_kernel (tune key only: tests, A/B)
int segnb_knob_fprop_thin() { return g_fprop_thin; }
static int g_fprop_mf16 = -2;      // conv_fprop_ws_kernel on v_mfma_f32_16x16x32_bf16 (1, default) or 32x32x16 (0)
int segnb_knob_fprop_mf16() {
    if (g_fprop_mf16 == -2) {
        const char* e = getenv("SEGNB_FPROP_MF16");
        g_fprop_mf16 = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_fprop_mf16;
}
static int g_fprop_upd = -2;       // plane-gather form of conv_fprop_ws_kernel for 4x4 / stride-2 gathers (A/B: SEGNB_FPROP_UPD=0)
int segnb_knob_fprop_upd() {
    if (g_fprop_upd == -2) {
        const char* e = getenv("SEGNB_FPROP_UPD");
        g_fprop_upd = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_fprop_upd;
}
static int g_fprop_mask = -2;      // conv_fprop_ws_kernel's activation-mask data gradient (MASK instantiation; A/B: SEGNB_FPROP_MASK=0)
int segnb_knob_fprop_mask() {
    if (g_fprop_mask == -2) {
        const char* e = getenv("SEGNB_FPROP_MASK");
        g_fprop_mask = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_fprop_mask;
}
static int g_fprop_drop = -2;      // segnb_conv_fprop_drop_ok may say yes (A/B: SEGNB_FPROP_DROP=0)
int segnb_knob_fprop_drop() {
    if (g_fprop_drop == -2) {
        const char* e = getenv("SEGNB_FPROP_DROP");
        g_fprop_drop = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_fprop_drop;
}
static int g_fprop_nostats = 1;    // conv_fprop_ws_kernel: statistics-free instantiation for launches without statistics (A/B)
int segnb_knob_fprop_nostats() { return g_fprop_nostats; }
static int g_rw_store_waves = 4;   // store waves of conv_fprop_rw_kernel: 4 (default) or 2 (round 1)
int segnb_knob_rw_store_waves() { return g_rw_store_waves; }
static int g_bnreduce_fused = -2;  // data-gradient launches that also do the next BatchNorm-backward reduction (A/B: SEGNB_BNREDUCE_FUSED=0)
int segnb_knob_bnreduce_fused() {
    if (g_bnreduce_fused == -2) {
        const char* e = getenv("SEGNB_BNREDUCE_FUSED");
        g_bnreduce_fused = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_bnreduce_fused;
}
// split K of conv_fprop_ws_kernel (fprop_dma.hip: ksplit_factor): 0 off, 1 automatic (default), 2 / 4 forced (A/B: SEGNB_FPROP_KSPLIT)
static int g_fprop_ksplit = -2;
int segnb_knob_fprop_ksplit() {
    if (g_fprop_ksplit == -2) {
        const char* e = getenv("SEGNB_FPROP_KSPLIT");
        g_fprop_ksplit = e != nullptr ? atoi(e) : 1;
    }
    return g_fprop_ksplit;
}
static int g_fprop_dma_dbg = 0;
int segnb_knob_fprop_dma_dbg() { return g_fprop_dma_dbg; }
// The share of the CUs the WIDE weight-gradient launches size their pixel split for (segnb_conv_wgrad_slabs and the launches
// that follow), as a RECORDABLE call: a model whose side stream is the longer one (UNet16: 12 ms of weight gradients against a
// 17 ms dependent chain that ends 1.7 ms earlier) takes all CUs for them, another in the same process keeps the default half;
// a recorded launch list sets and restores the value around its weight gradients.  0 = the built-in / environment default.
extern "C" int segnb_wg_cu_share(int pct) {
    SEGNB_PLAN_RECORD(segnb_wg_cu_share, pct);
    g_wg_cu_pct = pct < 0 ? 0 : (pct > 100 ? 100 : pct);
    return 0;
}

// The armed weight-gradient target (include/segnb_hip.h): per thread, like the recording state; consumed by the next
// segnb_conv_wgrad* entry point
static thread_local segnb_wgrad_target g_wg_target;
static thread_local bool g_wg_target_armed = false;
extern "C" int segnb_wgrad_target_arm(const segnb_wgrad_target* t) {
    SEGNB_PLAN_RECORD(segnb_wgrad_target_arm, t);
    SEGNB_CHECK_ARG(t != nullptr && t->gw != nullptr, "NULL target");
    SEGNB_CHECK_ARG(t->ntaps >= 1 && t->ntaps <= SEGNB_MAX_TAPS && t->Ci > 0 && t->Co > 0 && t->s_in >= t->ntaps && t->ci_off >= 0 &&
                        t->s_out >= (long long)(t->ci_off + t->Ci) * t->s_in, "bad target layout");
    for (int i = 0; i < t->ntaps; ++i) SEGNB_CHECK_ARG(t->kpos[i] >= 0 && t->kpos[i] < t->s_in, "bad kernel position");
    g_wg_target = *t;
    g_wg_target_armed = true;
    return 0;
}
const segnb_wgrad_target* segnb_take_wgrad_target() {
    if (!g_wg_target_armed) return nullptr;
    g_wg_target_armed = false;
    return &g_wg_target;
}

extern "C" int segnb_tune(const char* key, int value) {
    SEGNB_PLAN_REFUSE("segnb_tune inside a recorded plan");
    SEGNB_CHECK_ARG(key != nullptr, "NULL key");
    if (strcmp(key, "fprop_dma") == 0) {
        g_fprop_dma = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_dma_cfg") == 0) {
        g_fprop_dma_cfg = value < 0 ? -1 : value;
        return 0;
    }
    if (strcmp(key, "fprop_roll") == 0) {
        g_fprop_roll = value < 0 ? 0 : value;
        return 0;
    }
    if (strcmp(key, "pack_blocks") == 0) {
        g_pack_blocks = value < 0 ? 0 : value;
        return 0;
    }
    if (strcmp(key, "wgrad_roll") == 0) {
        g_wgrad_roll = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "wgrad_c8roll") == 0) {
        g_wgrad_c8roll = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_rw") == 0) {
        g_fprop_rw = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "conv_cu_pct") == 0) {        // CUs the persistent convolution kernels size their grids for (%)
        g_conv_cu_pct = value < 10 ? 10 : (value > 100 ? 100 : value);
        return 0;
    }
    if (strcmp(key, "plan_profile") == 0) {
        if (value == 2) plan_profile_dump();
        else g_plan_profile = value;
        return 0;
    }
    if (strcmp(key, "wg_cu_pct") == 0) {          // takes effect for plans made afterwards (segnb_conv_wgrad_slabs)
        g_wg_cu_pct = value < 0 ? 0 : (value > 100 ? 100 : value);
        return 0;
    }
    if (strcmp(key, "fprop_drop") == 0) {
        g_fprop_drop = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_mask") == 0) {
        g_fprop_mask = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_nostats") == 0) {
        g_fprop_nostats = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "rw_store_waves") == 0) {
        g_rw_store_waves = value == 2 ? 2 : (value == 8 ? 8 : 4);
        return 0;
    }
    if (strcmp(key, "bnreduce_fused") == 0) {
        g_bnreduce_fused = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_deepk") == 0) {
        g_fprop_deepk = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_thin") == 0) {
        g_fprop_thin = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_mf16") == 0) {
        g_fprop_mf16 = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "call_census") == 0) {
        g_segnb_census_on = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "fprop_ksplit") == 0) {
        g_fprop_ksplit = value < 0 ? 0 : value;
        return 0;
    }
    if (strcmp(key, "fprop_dma_dbg") == 0) {      // timing builds only: results are WRONG when non-zero
        g_fprop_dma_dbg = value;
        return 0;
    }
    segnb_set_error("segnb_tune: unknown key '%s'", key);
    return SEGNB_E_BADARG;
}

// timing builds: in-kernel time stamps of block 0 of the last conv_fprop_ws_kernel launch made with
// segnb_tune("fprop_dma_dbg", 32).  host_dst: 3 x 256 x 4 unsigned 64-bit shader clocks (matrix / weight / halo wave).
extern "C" int segnb_debug_stamps(unsigned long long* host_dst) {
    SEGNB_CHECK_ARG(host_dst != nullptr, "NULL destination");
    return segnb_fprop_dma_read_stamps(host_dst);
}
