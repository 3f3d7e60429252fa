// Convolutional ISTA/FISTA (reference lasso/conv2d/ista.py:7-49) and
// the Toeplitz Lipschitz bound (lasso/conv2d/lip_const.py:96-135).
//
//   x_hat = conv_transpose2d(z, W)      (synthesis, ista.py:19)
//   g     = conv2d(x_hat - x, W)        (its adjoint,  ista.py:20)
//   z+    = S_{alpha lr}(y - lr g), momentum and global stop rule as in the linear solver.
//
// Layout: the code z [N][K][Hz][Wz] is kept as a MATRIX Zm [M = N*Hz*Wz][K] ("one row per
// code pixel") for the whole solve, so both convolutions become the dense MFMA GEMMs the
// rest of the library already has, against ONE weight matrix Wt [C*kh*kw][K]:
//   COLSt [CKK][M] = Wt Ym^T                    (gemm_nt_kernel: every code pixel's patch)
//   R     [N][C][H][W] = overlap-add(COLSt) - x (conv_residual_kernel, gather form: no atomics)
//   RC    [M][CKK]     = patches of R            (conv_patches_kernel, im2col, pixel-major)
//   G     [M][K]       = RC W^T                  (gemm_nt_kernel, contraction over CKK)
//   prox / momentum / sum|z - z+|                (generic_prox_kernel on the matrices)
// COLSt is stored tap-major so that the overlap-add gather of neighbouring pixels touches
// neighbouring addresses; RC is pixel-major (rows = contraction-contiguous GEMM operand).  Stride and padding live
// only in the index arithmetic of those two kernels.
// Rooflines: the GEMMs are MFMA-bound (2*2*M*CKK*K flop per iteration); the two
// data-movement kernels are HBM-bound (each reads or writes the M*CKK patch matrix once).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Zm[(n,u,v)][k] <- z[n][k][u][v]   (to_rows != 0)   or the inverse: per image a K x P
// matrix transpose through a 32 x 33 LDS tile (both sides coalesced); blockIdx.z = image.
__global__ __launch_bounds__(256) void conv_relayout_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int K, int P, int to_rows) {
  __shared__ float t[32][33];
  const int64_t img = (int64_t)blockIdx.z * K * P;
  // source matrix of this image: rows x cols, destination: cols x rows
  const int rows = to_rows ? K : P, cols = to_rows ? P : K;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < rows && c < cols) ? src[img + (int64_t)r * cols + c] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;        // destination row = c, column = r
    if (c < cols && r < rows) dst[img + (int64_t)c * rows + r] = t[tx][i];
  }
}

// R[n][c][i][j] = sum_{a,b} COLSt[(c,a,b)][(n,u,v)] - x[n][c][i][j],
// u = (i + ph - a)/sh, v = (j + pw - b)/sw where they are integers inside the code grid
__global__ __launch_bounds__(256) void conv_residual_kernel(const float* __restrict__ colst, const float* __restrict__ x,
                                                            float* __restrict__ r, const ConvGeom g) {
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % g.W);
    const int i = (int)((e / g.W) % g.H);
    const int c = (int)((e / ((int64_t)g.W * g.H)) % g.C);
    const int n = (int)(e / ((int64_t)g.W * g.H * g.C));
    float acc = 0.0f;
    if (g.sh == 1 && g.sw == 1) {               // the usual case: no divisions in the tap loops
      const int a_lo = max(0, i + g.ph - (g.Hz - 1)), a_hi = min(g.kh - 1, i + g.ph);
      const int b_lo = max(0, j + g.pw - (g.Wz - 1)), b_hi = min(g.kw - 1, j + g.pw);
      for (int a = a_lo; a <= a_hi; ++a) {
        const int u = i + g.ph - a;
        const float* row = colst + ((int64_t)n * g.Hz + u) * g.Wz + (j + g.pw);
        const int64_t tb = ((int64_t)c * g.kh + a) * g.kw;
        for (int b = b_lo; b <= b_hi; ++b) acc += row[(tb + b) * M - b];
      }
    } else {
      for (int a = 0; a < g.kh; ++a) {
        const int ii = i + g.ph - a;
        if (ii < 0 || ii % g.sh != 0) continue;
        const int u = ii / g.sh;
        if (u >= g.Hz) continue;
        for (int b = 0; b < g.kw; ++b) {
          const int jj = j + g.pw - b;
          if (jj < 0 || jj % g.sw != 0) continue;
          const int v = jj / g.sw;
          if (v >= g.Wz) continue;
          const int64_t t = ((int64_t)c * g.kh + a) * g.kw + b;
          acc += colst[t * M + ((int64_t)n * g.Hz + u) * g.Wz + v];
        }
      }
    }
    r[e] = acc - (x ? x[e] : 0.0f);
  }
}

// RC[(n,u,v)][(c,a,b)] = R[n][c][u*sh - ph + a][v*sw - pw + b]   (0 outside the image), row
// stride ldr (the tap count rounded up to a multiple of 4 so that the GEMM can stage it with
// 16-byte loads; the padding columns are zero)
__global__ __launch_bounds__(256) void conv_patches_kernel(const float* __restrict__ r, float* __restrict__ rc,
                                                           int ldr, const ConvGeom g) {
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  const int ckk = g.C * g.kh * g.kw;
  const int q4 = ldr / 4;                        // float4 groups per patch row
  const int64_t total = M * q4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / q4;
    int t = (int)(e - m * q4) * 4;
    const int v = (int)(m % g.Wz), u = (int)((m / g.Wz) % g.Hz);
    const int n = (int)(m / ((int64_t)g.Wz * g.Hz));
    int b = t % g.kw, a = (t / g.kw) % g.kh, c = t / (g.kw * g.kh);     // decoded once, then stepped
    const int i0 = u * g.sh - g.ph, j0 = v * g.sw - g.pw;
    f32x4 out;
#pragma unroll
    for (int s = 0; s < 4; ++s, ++t) {
      float val = 0.0f;
      const int i = i0 + a, j = j0 + b;
      if (t < ckk && i >= 0 && i < g.H && j >= 0 && j < g.W) val = r[(((int64_t)n * g.C + c) * g.H + i) * g.W + j];
      out[s] = val;
      if (++b == g.kw) { b = 0; if (++a == g.kh) { a = 0; ++c; } }
    }
    *(f32x4*)(rc + m * ldr + (e - m * q4) * 4) = out;
  }
}

// Wp[k][t] = w[k][t] with the row stride padded like RC (zeros in the padding)
__global__ __launch_bounds__(256) void conv_pad_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                         int ckk, int ldr) {
  const int total = K * ldr;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int k = e / ldr, t = e % ldr;
    wp[e] = t < ckk ? w[(int64_t)k * ckk + t] : 0.0f;
  }
}

// Wt[(c,a,b)][k] = w[k][c][a][b]
__global__ __launch_bounds__(256) void conv_pack_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int K,
                                                          int ckk) {
  const int total = K * ckk;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int k = e % K, t = e / K;
    wt[e] = w[(int64_t)k * ckk + t];
  }
}

// lip_const.py:96-135.  taps [O][I][T] (T = ksize^2; O <= I after the reference's swap).
// One workgroup per o: power(o, f) = sum_i (sum_t tap*cos(phase[t][f]))^2 + (... sin ...)^2,
// out[o] = max_f power.  phase[t][f] = w0[f]*h0[t] + w1[f]*h1[t] (fp32, like the reference).
__global__ __launch_bounds__(256) void conv_lip_kernel(const float* __restrict__ taps, int64_t so, int64_t si, int I,
                                                       int ks, int padding, const float* __restrict__ freq,
                                                       int sample, float* __restrict__ out_max) {
  extern __shared__ float sh_taps[];          // [I][T]
  __shared__ float sred[256];
  const int T = ks * ks, o = blockIdx.x;
  for (int e = threadIdx.x; e < I * T; e += 256) sh_taps[e] = taps[(int64_t)o * so + (int64_t)(e / T) * si + e % T];
  __syncthreads();
  float best = 0.0f;
  for (int f = threadIdx.x; f < sample * sample; f += 256) {
    const float w0 = freq[f / sample], w1 = freq[f % sample];
    float power = 0.0f;
    for (int i = 0; i < I; ++i) {
      float re = 0.0f, im = 0.0f;
      for (int t = 0; t < T; ++t) {
        const float h0 = 1.0f + (float)(padding - ks + t / ks), h1 = 1.0f + (float)(padding - ks + t % ks);
        const float a0 = w0 * h0, a1 = w1 * h1;
        const float ph = a0 + a1;
        float sn, cs;
        sincosf(ph, &sn, &cs);
        re = fmaf(sh_taps[i * T + t], cs, re);
        im = fmaf(sh_taps[i * T + t], sn, im);
      }
      power += re * re + im * im;
    }
    best = fmaxf(best, power);
  }
  sred[threadIdx.x] = best;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sred[threadIdx.x] = fmaxf(sred[threadIdx.x], sred[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out_max[o] = sred[0];
}

__global__ void conv_lip_sum_kernel(const float* __restrict__ maxes, int O, int take_sqrt, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.0f;
    for (int o = 0; o < O; ++o) s += maxes[o];
    out[0] = take_sqrt ? (double)sqrtf(s) : (double)s;
  }
}


// ---- patch front end (image -> patches -> centre -> dict_learning ->
// reconstruct; the reference's Omniglot notebook that did this is not in the checkout, so
// the layout follows torch.nn.functional.unfold: row = (n, u, v), column = (c, a, b)) -------
// One wave per patch row: copy the patch, optionally subtract its mean (kept in means[]).
__global__ __launch_bounds__(256) void patches_extract_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                              int64_t ld, float* __restrict__ means,
                                                              const ConvGeom g, int center) {
  const int lane = threadIdx.x & 63;
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  const int d = g.C * g.kh * g.kw;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    const int v = (int)(m % g.Wz), u = (int)((m / g.Wz) % g.Hz);
    const int n = (int)(m / ((int64_t)g.Wz * g.Hz));
    float sum = 0.0f;
    for (int t = lane; t < d; t += 64) {
      const int b = t % g.kw, a = (t / g.kw) % g.kh, c = t / (g.kw * g.kh);
      const float val = img[(((int64_t)n * g.C + c) * g.H + u * g.sh + a) * g.W + v * g.sw + b];
      out[m * ld + t] = val;
      sum += val;
    }
    if (center) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float mean = sum / (float)d;
      for (int t = lane; t < d; t += 64) out[m * ld + t] -= mean;
      if (means && lane == 0) means[m] = mean;
    }
  }
}

// img[n][c][i][j] = average over the patches covering the pixel of (patch value + its mean);
// pixels no patch covers are 0
__global__ __launch_bounds__(256) void patches_reconstruct_kernel(const float* __restrict__ pat, int64_t ld,
                                                                  const float* __restrict__ means,
                                                                  float* __restrict__ img, const ConvGeom g) {
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % g.W), i = (int)((e / g.W) % g.H);
    const int c = (int)((e / ((int64_t)g.W * g.H)) % g.C);
    const int n = (int)(e / ((int64_t)g.W * g.H * g.C));
    float acc = 0.0f;
    int cnt = 0;
    for (int a = 0; a < g.kh; ++a) {
      const int ii = i - a;
      if (ii < 0 || ii % g.sh != 0 || ii / g.sh >= g.Hz) continue;
      for (int b = 0; b < g.kw; ++b) {
        const int jj = j - b;
        if (jj < 0 || jj % g.sw != 0 || jj / g.sw >= g.Wz) continue;
        const int64_t m = ((int64_t)n * g.Hz + ii / g.sh) * g.Wz + jj / g.sw;
        acc += pat[m * ld + ((int64_t)c * g.kh + a) * g.kw + b] + (means ? means[m] : 0.0f);
        ++cnt;
      }
    }
    img[e] = cnt ? acc / (float)cnt : 0.0f;
  }
}

inline int grid_for(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, 8192); }

}  // namespace

hipError_t launch_conv_relayout(const float* src, float* dst, int N, int K, int P, int to_rows, hipStream_t stream) {
  if ((int64_t)N * K * P == 0) return hipSuccess;
  const int rows = to_rows ? K : P, cols = to_rows ? P : K;
  for (int n0 = 0; n0 < N; n0 += 65535) {        // gridDim.z limit
    const int nb = std::min(N - n0, 65535);
    hipLaunchKernelGGL(conv_relayout_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, nb), dim3(256), 0, stream,
                       src + (int64_t)n0 * K * P, dst + (int64_t)n0 * K * P, K, P, to_rows);
  }
  return hipGetLastError();
}

hipError_t launch_conv_pack_w(const float* w, float* wt, float* wp, int K, int ckk, int ldr, hipStream_t stream) {
  hipLaunchKernelGGL(conv_pack_w_kernel, dim3(grid_for((int64_t)K * ckk)), dim3(256), 0, stream, w, wt, K, ckk);
  hipLaunchKernelGGL(conv_pad_w_kernel, dim3(grid_for((int64_t)K * ldr)), dim3(256), 0, stream, w, wp, K, ckk, ldr);
  return hipGetLastError();
}

// R = conv_transpose2d(rows Ym) - x :  GEMM into COLSt, then the gather
hipError_t launch_conv_residual(const float* Ym, const float* Wt, const float* x, float* colst, float* r,
                                const ConvGeom& g, hipStream_t stream) {
  const int ckk = g.C * g.kh * g.kw;
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  hipError_t e = launch_gemm_nt_sub(Wt, g.K, Ym, g.K, nullptr, 0, colst, M, ckk, (int)M, g.K, stream, /*add=*/1);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  hipLaunchKernelGGL(conv_residual_kernel, dim3(grid_for(total)), dim3(256), 0, stream, colst, x, r, g);
  return hipGetLastError();
}

// G [M][K] = conv2d(R, W) in row layout: pixel-major patches, then G = RC Wp^T on the general
// MFMA GEMM (contraction over the taps)
hipError_t launch_conv_gradient(const float* r, const float* Wp, float* rc, int ldr, float* G, const ConvGeom& g,
                                hipStream_t stream) {
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  hipLaunchKernelGGL(conv_patches_kernel, dim3(grid_for(M * (ldr / 4))), dim3(256), 0, stream, r, rc, ldr, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_gemm_nt_sub(rc, ldr, Wp, ldr, nullptr, 0, G, g.K, (int)M, g.K, ldr, stream, /*add=*/1);
}

hipError_t launch_patches_extract(const float* img, float* out, int64_t ld, float* means, const ConvGeom& g,
                                  int center, hipStream_t stream) {
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  if (M == 0) return hipSuccess;
  hipLaunchKernelGGL(patches_extract_kernel, dim3((unsigned)std::min<int64_t>((M + 3) / 4, 8192)), dim3(256), 0,
                     stream, img, out, ld, means, g, center);
  return hipGetLastError();
}

hipError_t launch_patches_reconstruct(const float* pat, int64_t ld, const float* means, float* img,
                                      const ConvGeom& g, hipStream_t stream) {
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(patches_reconstruct_kernel, dim3(grid_for(total)), dim3(256), 0, stream, pat, ld, means, img, g);
  return hipGetLastError();
}

hipError_t launch_conv_lip(const float* taps, int O, int I, int64_t so, int64_t si, int ks, int padding,
                           const float* freq, int sample, int take_sqrt, float* maxes, double* out,
                           hipStream_t stream) {
  const size_t lds = (size_t)I * ks * ks * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv_lip_kernel, dim3(O), dim3(256), lds, stream, taps, so, si, I, ks, padding, freq, sample,
                     maxes);
  hipLaunchKernelGGL(conv_lip_sum_kernel, dim3(1), dim3(64), 0, stream, maxes, O, take_sqrt, out);
  return hipGetLastError();
}

}  // namespace lasso
