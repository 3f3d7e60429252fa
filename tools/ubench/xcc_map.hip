// Which XCD does workgroup i of a 1-D grid run on?  (s_getreg HW_REG_XCC_ID per workgroup)
// build: hipcc --offload-arch=gfx950 -O2 xcc_map.hip -o xcc_map ; run on the GPU box: ./xcc_map [lds_bytes]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void who(int* out) {
  extern __shared__ char smem[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)(xcc & 15); out[2 * blockIdx.x + 1] = (int)hwid; }
  if (smem[threadIdx.x] == 77) out[0] = -1;
}
int main(int argc, char** argv) {
  const int lds = argc > 1 ? atoi(argv[1]) : 0, n = 256;
  int* d; hipMalloc(&d, 2 * n * sizeof(int));
  if (lds > 65536) hipFuncSetAttribute((const void*)who, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(who, dim3(n), dim3(256), lds, 0, d);
    int h[2 * 256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("xcc of workgroup i (lds %d): ", lds);
    for (int i = 0; i < 64; ++i) printf("%d", h[2 * i]);
    int ok = 1; for (int i = 0; i < n; ++i) ok &= h[2 * i] == (i % 8);
    printf(" ... i %% 8 for all %d: %s\n", n, ok ? "yes" : "NO");
  }
  return 0;
}
