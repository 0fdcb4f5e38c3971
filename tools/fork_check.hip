// correctness of a cross-stream wait on an event carried by a kernel's own dispatch (hipExtLaunchKernelGGL stopEvent):
// the waiting stream's kernel must see everything the signalling kernel wrote.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_write(int* p, int n, int v, int spin) {
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(50);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_check(const int* p, int n, int v, int* bad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (p[i] != v) atomicAdd(bad, 1);
}
int main() {
    hipStream_t s, s2;
    CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t evs[64], back[64];
    for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : back) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int n = 1 << 22;
    int *p, *bad;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(p, 0, n * 4));
    for (int it = 1; it <= 300; ++it) {
        hipEvent_t ev = evs[it & 63], bk = back[it & 63];
        CK(hipStreamWaitEvent(s, back[(it - 1) & 63], 0));            // the writer must not overtake the previous check
        hipExtLaunchKernelGGL(k_write, dim3(512), dim3(256), 0, s, nullptr, ev, 0, p, n, it, 20);
        CK(hipGetLastError());
        CK(hipStreamWaitEvent(s2, ev, 0));
        hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, s2, p, n, it, bad);
        CK(hipEventRecord(bk, s2));
    }
    CK(hipDeviceSynchronize());
    int h = -1;
    CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    printf("mismatches: %d\n", h);
    return h != 0;
}
