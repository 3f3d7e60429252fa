// lasso_loss (reference lasso/linear/dict_learning.py:10-13):
//     loss = (0.5*||X - Z W^T||^2 + alpha*||Z||_1) / n
// One workgroup per 16-row tile: Z tile -> LDS (A operand), the same MFMA GEMM-1
// stream as the FISTA kernel computes r = Z W^T - x, then sum r^2 and sum |z| are
// reduced in a fixed order to two floats per tile; a second one-block kernel folds the
// per-tile pairs (fp64 accumulation, fixed order) into the scalar loss.
// Roofline: MFMA-bound, 2*n*d*k flop (half a FISTA iteration), state read once.
#include "tile_device.hpp"

namespace lasso {

template <int K>
__global__ __launch_bounds__(kFistaThreads, 2) void objective_tile_kernel(const ObjectiveParams p) {
  constexpr int NW = kFistaWaves;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const zt = rings + NW * kRingBytesPerWave;
  lds_f32* const red = (lds_f32*)(zt + kTileM * K * 4);

  TileCtx<K> c;
  c.init(p.Wp, p.Wp /*unused*/, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;
  dma_step(c.w1, c.voff1, c.ring);
  dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    float l1 = 0.0f;
    visit_tile4<K, kFistaThreads>(p.Z, p.ldz, row0, p.n, p.k, [&](int r, int cc, const f32x4& v) {
      l1 += (__builtin_fabsf(v[0]) + __builtin_fabsf(v[1])) + (__builtin_fabsf(v[2]) + __builtin_fabsf(v[3]));
      *(lds_f32x4*)(zt + tile_chunk_off<K>(r, cc)) = v;
    });
    f32x4 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = 32 * wid + 16 * cb + n;
        float v = 0.0f;
        if ((row0 + r) < p.n && cc < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + cc];
        acc[cb][rg] = -v;
      }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    // tail refill = steps 0/1 of the next tile (W is tile independent)
    gemm1_stream_sp<K>(c, zt, acc, c.w1, c.w1 + 32, c.voff1);
    float rss = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) rss = fmaf(acc[cb][rg], acc[cb][rg], rss);
    rss = wave_sum(rss);
    l1 = wave_sum(l1);
    if (lane == 0) { red[2 * wid] = rss; red[2 * wid + 1] = l1; }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (tid == 0) {
      float a = 0.0f, b = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
      p.partials[2 * tile] = a;
      p.partials[2 * tile + 1] = b;
    }
    __builtin_amdgcn_s_barrier();   // red[] / zt reuse
  }
  LASSO_WAIT_VMCNT(0);
}

// sums[0] = sum r^2, sums[1] = sum |z| (fp64, this shard);  if loss_out != nullptr also
// loss_out[0] = (0.5*sums[0] + alpha*sums[1]) / n_total   (single-GPU convenience)
__global__ __launch_bounds__(256) void objective_finalize_kernel(const float* __restrict__ partials,
                                                                 int ntiles, double alpha,
                                                                 double n_total,
                                                                 double* __restrict__ sums,
                                                                 float* __restrict__ loss_out) {
  __shared__ double sa[256], sb[256];
  double a = 0.0, b = 0.0;
  for (int t = threadIdx.x; t < ntiles; t += 256) { a += partials[2 * t]; b += partials[2 * t + 1]; }
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[0] = sa[0]; sums[1] = sb[0];
    if (loss_out) loss_out[0] = (float)((0.5 * sa[0] + alpha * sb[0]) / n_total);
  }
}

template <int K>
static hipError_t launch_obj_k(const ObjectiveParams& p, int grid, hipStream_t stream) {
  const size_t lds = (size_t)kFistaWaves * kRingBytesPerWave + (size_t)kTileM * K * 4 + 128;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&objective_tile_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(objective_tile_kernel<K>, dim3(grid), dim3(kFistaThreads), lds, stream, p);
  return hipGetLastError();
}

hipError_t launch_objective(const ObjectiveParams& p, int kpad, int grid, double alpha,
                            double n_total, double* sums, float* loss_out, hipStream_t stream) {
  hipError_t e;
  switch (kpad) {
    case 256: e = launch_obj_k<256>(p, grid, stream); break;
    case 512: e = launch_obj_k<512>(p, grid, stream); break;
    case 1024: e = launch_obj_k<1024>(p, grid, stream); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(objective_finalize_kernel, dim3(1), dim3(256), 0, stream, p.partials, p.ntiles,
                     alpha, n_total, sums, loss_out);
  return hipGetLastError();
}

// Large shapes (d > 256 or k > 1024): R = X - Z W^T by the general GEMM (gemm.hip), then
// one pass that folds sum R^2 and sum |Z| into per-workgroup pairs (fixed grid => fixed
// summation order) for objective_finalize_kernel.
__global__ __launch_bounds__(256) void objective_reduce_kernel(const float* __restrict__ R, int64_t nd,
                                                               const float* __restrict__ Z, int64_t ldz,
                                                               int n, int k, float* __restrict__ partials) {
  __shared__ float sa[256], sb[256];
  float a = 0.0f, b = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nd; i += stride) a = fmaf(R[i], R[i], a);
  const int64_t nk = (int64_t)n * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nk; i += stride)
    b += __builtin_fabsf(Z[(i / k) * ldz + i % k]);
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = sa[0]; partials[2 * blockIdx.x + 1] = sb[0]; }
}

// sums / loss from a residual that is already in memory (convolutional objective)
hipError_t launch_objective_reduce(const float* R, int64_t nd, const float* Z, int64_t ldz, int n, int k,
                                   float* partials, int grid, double alpha, double n_total, double* sums,
                                   float* loss_out, hipStream_t stream) {
  hipLaunchKernelGGL(objective_reduce_kernel, dim3(grid), dim3(256), 0, stream, R, nd, Z, ldz, n, k, partials);
  hipLaunchKernelGGL(objective_finalize_kernel, dim3(1), dim3(256), 0, stream, partials, grid, alpha, n_total,
                     sums, loss_out);
  return hipGetLastError();
}

hipError_t launch_objective_generic(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* Z,
                                    int64_t ldz, int n, int d, int k, float* R, float* partials, int grid,
                                    double alpha, double n_total, double* sums, float* loss_out,
                                    hipStream_t stream) {
  hipError_t e = launch_gemm_nt_sub(Z, ldz, W, ldw, X, ldx, R, d, n, d, k, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(objective_reduce_kernel, dim3(grid), dim3(256), 0, stream, R, (int64_t)n * d, Z, ldz, n, k,
                     partials);
  hipLaunchKernelGGL(objective_finalize_kernel, dim3(1), dim3(256), 0, stream, partials, grid, alpha, n_total,
                     sums, loss_out);
  return hipGetLastError();
}

}  // namespace lasso
