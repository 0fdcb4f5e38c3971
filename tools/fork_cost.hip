// GPU-side cost of a cross-stream "fork" between two dependent kernels of one queue:
//   (a) kernels back to back, (b) hipEventRecord between every pair (+ hipStreamWaitEvent on a second stream),
//   (c) the event carried by the kernel's own dispatch (hipExtLaunchKernelGGL stopEvent) + the same wait.
// hipcc --offload-arch=gfx950 -O2 -o tools/_bin/fork_cost tools/fork_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_work(float* p, int n) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}
int main() {
    hipStream_t s, s2;
    hipStreamCreate(&s); hipStreamCreate(&s2);
    hipEvent_t evs[64];
    for (auto& e : evs) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    float* p; hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    const int N = 400, WORK = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(t0, s);
            for (int i = 0; i < N; ++i) {
                hipEvent_t ev = evs[i & 63];
                if (mode == 2 || mode == 3) {
                    hipExtLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, nullptr, ev, 0, p, WORK);
                    hipStreamWaitEvent(s2, ev, 0);
                    if (mode == 3) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s2, p + 512, WORK / 2);
                } else {
                    hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, p, WORK);
                    if (mode == 1) { hipEventRecord(ev, s); hipStreamWaitEvent(s2, ev, 0); }
                }
            }
            hipEventRecord(t1, s);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, t0, t1);
            if (rep) printf("mode %d (%s): %.2f us per kernel\n", mode,
                            mode == 0 ? "back to back" : mode == 1 ? "hipEventRecord + wait between" : mode == 2 ? "stopEvent on the dispatch + wait" : "stopEvent + wait + kernel on the other stream",
                            ms * 1e3 / N);
        }
    }
    return 0;
}
