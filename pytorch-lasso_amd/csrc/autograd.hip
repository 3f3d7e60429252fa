// Reverse-mode derivative of the unrolled fixed-step ISTA/FISTA solve (SURVEY.md 8f row f4:
// the reference's ista() is plain autograd-traceable torch code, ista.py:57-104, and its
// README advertises back-propagation through the solver).
//
// Forward, per iteration i (c_i = momentum coefficient, p_i = y_i, y_0 = z_0):
//     r_i = y_i W^T - x ;  g_i = r_i W ;  u_i = y_i - lr g_i ;  z_{i+1} = S(u_i)
//     y_{i+1} = (1 + c_i) z_{i+1} - c_i z_i
// Backward, i = T-1 .. 0, with adjoints zb_j (of z_j) and yb_j (of y_j):
//     zb_{i+1} += (1 + c_i) yb_{i+1} ;  zb_i = -c_i yb_{i+1}
//     ub = [z_{i+1} != 0] zb_{i+1}          (softshrink passes the gradient where |u| > lambda)
//     gb = -lr ub ;  rb = gb W^T ;  yb_i = ub + rb W
//     Wb += r_i^T gb + rb^T y_i ;  xb -= rb
// and finally z0b = zb_0 + yb_0.  The five products per iteration are the library's MFMA
// GEMMs (gemm.hip: three [n x .] products; mstep.hip gram_tn: the two reductions over the
// batch); this file holds the elementwise glue.  The forward pass keeps the iterates z_0..z_T
// (the `trace`); y_i and r_i are recomputed from them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"

namespace lasso {
namespace {

inline int grid_for(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, 4096); }

// zb_next += (1+c) yb ; zb_cur = -c yb ; ub = mask(z_next) zb_next ; gb = -lr ub
__global__ __launch_bounds__(256) void bw_prox_kernel(float* __restrict__ zb_next, float* __restrict__ zb_cur,
                                                      const float* __restrict__ yb, const float* __restrict__ z_next,
                                                      float* __restrict__ ub, float* __restrict__ gb, int64_t total,
                                                      float c, float lr) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float y = yb[i];
    const float zn = zb_next[i] + (1.0f + c) * y;
    zb_next[i] = zn;
    zb_cur[i] = -c * y;
    const float u = z_next[i] != 0.0f ? zn : 0.0f;
    ub[i] = u;
    gb[i] = -lr * u;
  }
}

// y = z + c (z - z_prev)      (c == 0 or z_prev == nullptr: y = z)
__global__ __launch_bounds__(256) void bw_point_kernel(const float* __restrict__ z, const float* __restrict__ z_prev,
                                                       float* __restrict__ y, int64_t total, float c) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float v = z[i];
    y[i] = z_prev ? v + c * (v - z_prev[i]) : v;
  }
}

// a += s1 * b (+ s2 * c)
__global__ __launch_bounds__(256) void bw_axpy_kernel(float* __restrict__ a, const float* __restrict__ b, float s1,
                                                      const float* __restrict__ c, float s2, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    a[i] += s1 * b[i] + (c ? s2 * c[i] : 0.0f);
}

}  // namespace

hipError_t launch_bw_prox(float* zb_next, float* zb_cur, const float* yb, const float* z_next, float* ub, float* gb,
                          int64_t total, float c, float lr, hipStream_t stream) {
  hipLaunchKernelGGL(bw_prox_kernel, dim3(grid_for(total)), dim3(256), 0, stream, zb_next, zb_cur, yb, z_next, ub, gb,
                     total, c, lr);
  return hipGetLastError();
}
hipError_t launch_bw_point(const float* z, const float* z_prev, float* y, int64_t total, float c, hipStream_t stream) {
  hipLaunchKernelGGL(bw_point_kernel, dim3(grid_for(total)), dim3(256), 0, stream, z, z_prev, y, total, c);
  return hipGetLastError();
}
hipError_t launch_bw_axpy(float* a, const float* b, float s1, const float* c, float s2, int64_t total,
                          hipStream_t stream) {
  hipLaunchKernelGGL(bw_axpy_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a, b, s1, c, s2, total);
  return hipGetLastError();
}

}  // namespace lasso
