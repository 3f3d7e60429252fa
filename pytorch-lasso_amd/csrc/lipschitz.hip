// Deterministic device-side Lipschitz constant  L = lambda_max(W^T W)
// (replaces lasso/linear/solvers/ista.py:8-14, which forms the fp32 Gram on the
// device, copies it to the host and runs ARPACK -- not run-to-run reproducible).
//
// Method (all fp64, all on the GPU, no atomics => bitwise reproducible):
//   G  = W W^T            (m x m, m = min(d,k); the non-zero spectrum of W^T W)
//   P0 = G / tr(G);   P_{i+1} = P_i^2 / tr(P_i)^2 ... i.e. repeated squaring with
//   trace normalisation, P_p ~ G^(2^p) / c
//   L  = <G, P_p>_F / tr(P_p) = sum_i lambda_i^(N+1) / sum_i lambda_i^N,  N = 2^p
// which converges to lambda_max from below with relative error <= ~1/(e*N) per
// eigenvalue clustered at the top (p = 20: ~3.5e-7 worst case, exact to fp64
// rounding for any realistic gap).  Cost: p+1 small fp64 GEMMs (2 m^3 flop each,
// 33.5 MFLOP at m = 256) -- launch-latency bound, not on the FISTA hot loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lasso_kernels.h"

namespace lasso {
namespace {

constexpr int kLipTile = 32;   // output tile per workgroup (256 threads, 2x2 per thread)

// sum of the diagonal of an m x m matrix, fixed order, computed redundantly per block
__device__ double block_trace(const double* __restrict__ A, int m, int ld, double* sh) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < m; i += blockDim.x) acc += A[(size_t)i * ld + i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const double t = sh[0];
  __syncthreads();
  return t;
}

// G[i][j] = sum_t w(i,t) w(j,t),  w(i,t) = W[i*si + t*st]  (fp32 in, fp64 accumulate).
// Output is zero-padded to mp x mp (mp multiple of 32).
__global__ __launch_bounds__(256) void gram_f64_kernel(const float* __restrict__ W, int64_t si,
                                                       int64_t st, int m, int len, int mp,
                                                       double* __restrict__ G) {
  __shared__ double sa[kLipTile][kLipTile + 1], sb[kLipTile][kLipTile + 1];
  const int i0 = blockIdx.y * kLipTile, j0 = blockIdx.x * kLipTile;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, 2x2 outputs each
  double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
  for (int t0 = 0; t0 < len; t0 += kLipTile) {
    for (int e = threadIdx.x; e < kLipTile * kLipTile; e += 256) {
      const int r = e / kLipTile, c = e % kLipTile;   // r: row in tile, c: t index
      const int t = t0 + c;
      sa[r][c] = (i0 + r < m && t < len) ? (double)W[(int64_t)(i0 + r) * si + (int64_t)t * st] : 0.0;
      sb[r][c] = (j0 + r < m && t < len) ? (double)W[(int64_t)(j0 + r) * si + (int64_t)t * st] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < kLipTile; ++c) {
      const double a0 = sa[ty][c], a1 = sa[ty + 16][c], b0 = sb[tx][c], b1 = sb[tx + 16][c];
      c00 = fma(a0, b0, c00); c01 = fma(a0, b1, c01);
      c10 = fma(a1, b0, c10); c11 = fma(a1, b1, c11);
    }
    __syncthreads();
  }
  G[(size_t)(i0 + ty) * mp + j0 + tx] = c00;
  G[(size_t)(i0 + ty) * mp + j0 + tx + 16] = c01;
  G[(size_t)(i0 + ty + 16) * mp + j0 + tx] = c10;
  G[(size_t)(i0 + ty + 16) * mp + j0 + tx + 16] = c11;
}

// C = (A/s)(A/s)^T with s = tr(A)   (A symmetric => both operands row-contiguous)
__global__ __launch_bounds__(256) void square_f64_kernel(const double* __restrict__ A, int mp,
                                                         double* __restrict__ C) {
  __shared__ double sa[kLipTile][kLipTile + 1], sb[kLipTile][kLipTile + 1];
  __shared__ double sh[256];
  const double inv = 1.0 / block_trace(A, mp, mp, sh);
  const int i0 = blockIdx.y * kLipTile, j0 = blockIdx.x * kLipTile;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
  for (int t0 = 0; t0 < mp; t0 += kLipTile) {
    for (int e = threadIdx.x; e < kLipTile * kLipTile; e += 256) {
      const int r = e / kLipTile, c = e % kLipTile;
      sa[r][c] = A[(size_t)(i0 + r) * mp + t0 + c] * inv;
      sb[r][c] = A[(size_t)(j0 + r) * mp + t0 + c] * inv;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < kLipTile; ++c) {
      const double a0 = sa[ty][c], a1 = sa[ty + 16][c], b0 = sb[tx][c], b1 = sb[tx + 16][c];
      c00 = fma(a0, b0, c00); c01 = fma(a0, b1, c01);
      c10 = fma(a1, b0, c10); c11 = fma(a1, b1, c11);
    }
    __syncthreads();
  }
  C[(size_t)(i0 + ty) * mp + j0 + tx] = c00;
  C[(size_t)(i0 + ty) * mp + j0 + tx + 16] = c01;
  C[(size_t)(i0 + ty + 16) * mp + j0 + tx] = c10;
  C[(size_t)(i0 + ty + 16) * mp + j0 + tx + 16] = c11;
}

// out[0] = <G, P>_F / tr(P)   (single block, fixed summation order)
__global__ __launch_bounds__(1024) void rayleigh_trace_kernel(const double* __restrict__ G,
                                                              const double* __restrict__ P, int mp,
                                                              double* __restrict__ out) {
  __shared__ double sh[1024];
  const double tr = block_trace(P, mp, mp, sh);
  double acc = 0.0;
  const size_t total = (size_t)mp * mp;
  for (size_t e = threadIdx.x; e < total; e += 1024) acc = fma(G[e], P[e], acc);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0] / tr;
}

}  // namespace

size_t lipschitz_workspace_bytes(int64_t d, int64_t k) {
  const int64_t m = d < k ? d : k;
  const int64_t mp = (m + kLipTile - 1) / kLipTile * kLipTile;
  return (size_t)(3 * mp * mp + 32) * sizeof(double);
}

// Enqueue the whole computation; the result lands in ((double*)workspace)[0].
hipError_t launch_lipschitz(const float* W, int64_t ldw, int64_t d, int64_t k, void* workspace,
                            int squarings, hipStream_t stream) {
  const bool rows = d <= k;                 // G = W W^T (rows) or W^T W (columns)
  const int m = (int)(rows ? d : k), len = (int)(rows ? k : d);
  const int mp = (m + kLipTile - 1) / kLipTile * kLipTile;
  double* out = static_cast<double*>(workspace);
  double* G = out + 32;
  double* P[2] = {G + (size_t)mp * mp, G + 2 * (size_t)mp * mp};
  const dim3 grid(mp / kLipTile, mp / kLipTile);
  hipLaunchKernelGGL(gram_f64_kernel, grid, dim3(256), 0, stream, W, rows ? ldw : (int64_t)1,
                     rows ? (int64_t)1 : ldw, m, len, mp, G);
  const double* src = G;
  for (int p = 0; p < squarings; ++p) {
    hipLaunchKernelGGL(square_f64_kernel, grid, dim3(256), 0, stream, src, mp, P[p & 1]);
    src = P[p & 1];
  }
  hipLaunchKernelGGL(rayleigh_trace_kernel, dim3(1), dim3(1024), 0, stream, G, src, mp, out);
  return hipGetLastError();
}

}  // namespace lasso
