// Host cost of one kernel launch on this box (hipLaunchKernelGGL of an empty kernel, small and 600-byte arguments,
// one stream, GPU kept busy), and of an event record + cross-stream wait.   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/launch_cost tools/launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[150]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = b.v[3]; }
__global__ void k_spin(int n, int* p) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(100); if (p && n < 0) *p = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s, s2;
    hipStreamCreate(&s); hipStreamCreate(&s2);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    Big b{};
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 2000, nullptr);
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, nullptr);
        double t1 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, b, nullptr);
        double t2 = now();
        for (int i = 0; i < N; ++i) { hipEventRecord(ev, s); hipStreamWaitEvent(s2, ev, 0); }
        double t3 = now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, (i & 1) ? s : s2, nullptr); }
        double t4 = now();
        hipDeviceSynchronize();
        double t5 = now();
        printf("rep %d: small-arg launch %.2f us, 600-byte-arg launch %.2f us, record+wait %.2f us, alternating streams %.2f us/launch; drain %.0f us\n",
               rep, (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N, (t4 - t3) / N, t5 - t4);
    }
    return 0;
}
