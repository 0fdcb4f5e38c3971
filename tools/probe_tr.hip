// Hardware probe: lane semantics of ds_read_b64_tr_b16 (gfx950).  Build+run: hipcc --offload-arch=gfx950 probe_tr.hip -o /tmp/probe_tr && /tmp/probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8 * 32];
  for (int i = threadIdx.x; i < 8 * 32; i += 64) lds[i] = (short)((i / 32) * 100 + (i % 32));   // value = row*100 + col
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3, h = l >> 5;
  const int row = 4 * h + q, col = 16 * (g & 1) + 4 * p;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 32 + col));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = a[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short hbuf[256]; hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int h = l >> 5, g = l >> 4, i = l & 15;
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      int expect = (4 * h + e) * 100 + 16 * (g & 1) + i;   // row 4h+e, column cbase+i
      printf(" %4d%s", hbuf[l * 4 + e], hbuf[l * 4 + e] == expect ? "" : "!");
      bad += hbuf[l * 4 + e] != expect;
    }
    printf("\n");
  }
  printf("mismatches vs expected mapping: %d\n", bad);
  return 0;
}
