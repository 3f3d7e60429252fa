#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// C[16][16] = A[16][32] * B[32][16], assumed layouts:
// A: lane l -> row l%16, k = 8*(l/16)+t ; B: lane l -> col l%16, k = 8*(l/16)+t ; C: lane l -> col l%16, row 4*(l/16)+r
__global__ void k(const float* A, const float* B, float* C) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (__bf16)A[(l % 16) * 32 + 8 * (l / 16) + t];
    b[t] = (__bf16)B[(8 * (l / 16) + t) * 16 + (l % 16)];
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * (l / 16) + r) * 16 + (l % 16)] = acc[r];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hC[256], ref[256];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int t = 0; t < 32; ++t) s += hA[i * 32 + t] * hB[t * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hC[i] - ref[i]));
  printf("mfma 16x16x32 bf16 layout max err = %g\n", err);
  return err > 1e-3;
}
