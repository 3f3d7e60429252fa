// Micro-benchmark: how plain fp32 VALU work overlaps with bf16 MFMAs on one SIMD of gfx950.
//   ./mix   -> cycles per loop body for {MFMA only, VALU only, interleaved} at 1 and 2 waves per SIMD
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mix.hip -o tools/ubench/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int KIND>   // per body: NM MFMAs, NV VALU ops after each MFMA; KIND 0 = v_fma, 1 = mixed ops, 2 = packed
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* cyc) {
  f32x4 acc[8];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  float v[8];
  f32x2 vp[4];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 4; ++i) vp[i] = (f32x2){threadIdx.x * 0.002f + i, threadIdx.x * 0.003f - i};
  const float c0 = out[0], c1 = out[1];
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < NM) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int i = (m * NV + q) & 7;
        if (KIND == 0) v[i] = __builtin_fmaf(v[i], c0, c1);
        else if (KIND == 1) {
          if ((q & 3) == 0) v[i] = __builtin_amdgcn_fmed3f(v[i], -c0, c0);
          else if ((q & 3) == 1) v[i] = v[i] - c1;
          else if ((q & 3) == 2) v[i] = __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
          else v[i] = c1 + __builtin_fabsf(v[i]);
        } else {
          vp[i & 3] = __builtin_elementwise_fma(vp[i & 3], (f32x2){c0, c0}, (f32x2){c1, c1});   // register pairs: no moves
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + vp[i & 3][i >> 2 & 1] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int KIND>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, (2 + 256 * 512) * sizeof(float)); hipMalloc(&cyc, 8);
  float h[2] = {1.0001f, 0.5f};
  hipMemcpy(out, h, 8, hipMemcpyHostToDevice);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NM, NV, KIND><<<256, threads>>>(out, 10, cyc);
  hipEventRecord(e0);
  k<NM, NV, KIND><<<256, threads>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_body_ns = ms * 1e6 / iters;
  printf("%-34s waves/SIMD %d: %8.1f ns per body (8 groups)  = %6.1f cycles @2.4GHz per group; clock64 ticks/body %.1f\n", name, threads / 256,
         per_body_ns, per_body_ns * 2.4 / 8, (double)c / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<8, 0, 0>("8 MFMA", threads);
    run<0, 5, 0>("40 v_fma (5 per group)", threads);
    run<0, 5, 1>("40 mixed VALU", threads);
    run<0, 5, 2>("40 v_pk_fma", threads);
    run<8, 2, 0>("8 x (MFMA + 2 v_fma)", threads);
    run<8, 4, 0>("8 x (MFMA + 4 v_fma)", threads);
    run<8, 5, 0>("8 x (MFMA + 5 v_fma)", threads);
    run<8, 5, 1>("8 x (MFMA + 5 mixed)", threads);
    run<8, 8, 0>("8 x (MFMA + 8 v_fma)", threads);
    run<8, 3, 2>("8 x (MFMA + 3 v_pk_fma)", threads);
  }
  return 0;
}
