// Device-side building blocks of the bf16 kernels (bt_bf16.hip: multi-launch line search and
// fixed-step gradient; bt16_persist.hip: the persistent single-launch solve): 64-row tiles,
// 16-byte-chunk swizzled LDS tiles, the bf16-MFMA GEMM-1 loop with W fragments streamed
// straight from L2 into registers.
#pragma once
#include "tile_device.hpp"

namespace lasso {
namespace bf16dev {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) __bf16 lds_bf16;

constexpr int kRows = 64;            // rows per tile
constexpr int kThreads = 512;
constexpr int kWaves = 8;

// byte offset of 16-byte chunk `chunk` (8 bf16) of row `row` in a tile with ROWB bytes per row
template <int ROWB>
__device__ __forceinline__ int tile16_off(int row, int chunk) {
  return row * ROWB + ((chunk ^ (row & 15)) << 4);
}

__device__ __forceinline__ bf16x8 to_bf16x8(const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
  return o;
}

// 8 consecutive floats of row `row` starting at column c0 (zero outside [0,n) x [0,cols))
__device__ __forceinline__ void load8(const float* __restrict__ src, int64_t ld, int row, int n, int c0, int cols,
                                      bool vec, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  if (row >= n || c0 >= cols) return;
  const float* p = src + (int64_t)row * ld + c0;
  if (vec && c0 + 8 <= cols) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < cols) v[e] = p[e];
  }
}

// acc[rb][cb] += A-tile(rows 16rb.., K) * Wq1 fragments of this wave (r columns 32w + 16cb ..)
// W fragments stream through a 4-deep register ring, two steps ahead of their use; the loop
// is rolled (4 steps per trip) so that the compiler cannot hoist the whole stream.
template <int K>
__device__ __forceinline__ void gemm1_bf16(const lds_char* at, const bf16x8* __restrict__ wq, int lane,
                                           f32x4 (&acc)[4][2]) {
  constexpr int S1 = K / 32;
  static_assert(S1 % 4 == 0, "K must be a multiple of 128");
  const int i = lane & 15, kg = lane >> 4;
  bf16x8 b[4][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    b[0][cb] = wq[(0 * 2 + cb) * 64 + lane];
    b[1][cb] = wq[(1 * 2 + cb) * 64 + lane];
  }
#pragma unroll 1
  for (int j = 0; j < S1 / 4; ++j) {
    static_for<4>([&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      const int s = 4 * j + u;
      const int sp = min(s + 2, S1 - 1);          // the last two prefetches re-read the final step
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) b[(u + 2) % 4][cb] = wq[(sp * 2 + cb) * 64 + lane];
      bf16x8 a[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[rb] = *(const lds_bf16x8*)(at + tile16_off<K * 2>(16 * rb + i, 4 * s + kg));
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b[u][cb], acc[rb][cb], 0, 0, 0);
    });
  }
}

// r = acc - x (bf16 x), returns sum r^2 of this lane; optionally writes r (bf16) into the r tile
template <bool STORE>
__device__ __forceinline__ float residual_epilogue(f32x4 (&acc)[4][2], const __bf16* __restrict__ X, int64_t ldx,
                                                   int row0, int n, int d, int wid, int lane, lds_char* rt) {
  const int cl = lane & 15, q = lane >> 4;
  float rss = 0.0f;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 16 * rb + 4 * q + rg, cc = 32 * wid + 16 * cb + cl;
        float xv = 0.0f;
        if (row0 + r < n && cc < d) xv = (float)X[(int64_t)(row0 + r) * ldx + cc];
        const float res = acc[rb][cb][rg] - xv;
        acc[rb][cb][rg] = res;
        rss = fmaf(res, res, rss);
        if constexpr (STORE)
          *(lds_bf16*)(rt + tile16_off<kFistaD * 2>(r, cc >> 3) + 2 * (cc & 7)) = (__bf16)res;
      }
  return rss;
}

}  // namespace bf16dev
}  // namespace lasso
