// M-step kernels: the constrained dictionary update of
// lasso/linear/dict_learning.py:56-103 in GRAM FORM.
//
// With A = Z^T Z [k,k] and B = Z^T X [k,d] (row-shard partial sums, all-reduced by
// the host across GPUs), the reference's Gauss-Seidel atom sweep
//     R += z_j d_j^T ; u = z_j^T R ; d_j <- u/||u|| ; R -= z_j d_j^T     (:85-101)
// is u_j = B_j - sum_i A_ji d_i + A_jj d_j with the CURRENT atoms d_i.  We keep
// U = B - A D^T (one GEMM) and sweep blocks of 32 atoms:
//   sweep_block_kernel   one wave, sequential inside the block, U rows in registers;
//   trailing_update      U[j' > block] -= A[j', block] * dD[block]   (many workgroups).
// A degenerate atom (||u|| < eps, :92-98) is replaced by a caller-supplied unit vector
// and removed from the model (its effective new atom is 0, as zeroing Z[:,j] does).
//
//   gram_tn_kernel   C = P^T Q     fp32 MFMA, reduce over the n rows of the shard
//   (C = C0 - A B^T lives in gemm.hip)
// Rooflines: the two GEMM kernels are MFMA-bound (2nk^2 + 2nkd and 2k^2 d flop); the
// sweep is a latency-bound dependency chain of k steps (time reported, no roofline).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"
#include "static_for.hpp"

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// C[pc x qc] = P^T Q,  P [n x pc] (ldp), Q [n x qc] (ldq).  64x64 block per workgroup,
// 4 waves x (32x32), samples in chunks of 32 through LDS (row stride 80 floats:
// the per-MFMA operand read "16 consecutive floats of 4 consecutive rows" is
// conflict-free).  sym != 0: P == Q, only blocks bj >= bi are computed and mirrored.
// ---------------------------------------------------------------------------
constexpr int kGB = 64, kGS = 64, kGLd = 80;

// VEC: every row segment is 16-byte aligned and the column counts are multiples of 4,
// so the staging loads are float4 (the usual case: k, d, ld multiples of 4).
template <bool VEC>
__global__ __launch_bounds__(256) void gram_tn_kernel(const float* __restrict__ P, int64_t ldp, int pc,
                                                      const float* __restrict__ Q, int64_t ldq, int qc,
                                                      int n, float* __restrict__ C, int64_t ldc, int sym,
                                                      int rows_per_split, int64_t split_stride) {
  if (sym && blockIdx.x < blockIdx.y) return;
  // split-n: blockIdx.z handles samples [z*rows_per_split, (z+1)*rows_per_split) and
  // writes its partial product to C + z*split_stride (summed in fixed order afterwards)
  const int n_lo = blockIdx.z * rows_per_split;
  P += (int64_t)n_lo * ldp;
  Q += (int64_t)n_lo * ldq;
  n = min(rows_per_split, n - n_lo);
  C += (int64_t)blockIdx.z * split_stride;
  __shared__ __attribute__((aligned(16))) float sp[kGS][kGLd], sq[kGS][kGLd];
  const int i0 = blockIdx.y * kGB, j0 = blockIdx.x * kGB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, q = lane >> 4;
  f32x4 acc[2][2] = {};
  // staging map: thread -> rows srow + 16h (h = 0..3), 4 consecutive columns
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  f32x4 stg[2][4];   // [P/Q][h]

  auto load_chunk = [&](int s0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = s0 + srow + 16 * h;
      if constexpr (VEC) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        stg[0][h] = (r < n && i0 + scol < pc) ? *(const f32x4*)(P + (int64_t)r * ldp + i0 + scol) : z;
        stg[1][h] = (r < n && j0 + scol < qc) ? *(const f32x4*)(Q + (int64_t)r * ldq + j0 + scol) : z;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cp = i0 + scol + e, cq = j0 + scol + e;
          stg[0][h][e] = (r < n && cp < pc) ? P[(int64_t)r * ldp + cp] : 0.0f;
          stg[1][h][e] = (r < n && cq < qc) ? Q[(int64_t)r * ldq + cq] : 0.0f;
        }
      }
    }
  };

  load_chunk(0);
  for (int s0 = 0; s0 < n; s0 += kGS) {
    __syncthreads();                       // previous chunk's fragment reads are done
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      *(f32x4*)(&sp[srow + 16 * h][scol]) = stg[0][h];
      *(f32x4*)(&sq[srow + 16 * h][scol]) = stg[1][h];
    }
    __syncthreads();
    if (s0 + kGS < n) load_chunk(s0 + kGS);   // global loads fly under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < kGS / 4; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = sp[4 * ks + q][32 * wr + 16 * m + l15];
#pragma unroll
      for (int m = 0; m < 2; ++m) b[m] = sq[4 * ks + q][32 * wc + 16 * m + l15];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = i0 + 32 * wr + 16 * mi + 4 * q + rg, cc = j0 + 32 * wc + 16 * nj + l15;
        if (r < pc && cc < qc) {
          C[(int64_t)r * ldc + cc] = acc[mi][nj][rg];
          if (sym && blockIdx.x != blockIdx.y) C[(int64_t)cc * ldc + r] = acc[mi][nj][rg];
        }
      }
}

// C[r][c] = sum_s part[s][r][c]  (fixed order)
__global__ __launch_bounds__(256) void sum_splits_kernel(const float* __restrict__ part, int splits,
                                                         int64_t split_stride, int rows, int cols,
                                                         float* __restrict__ C, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  float acc = 0.0f;
  for (int s = 0; s < splits; ++s) acc += part[(int64_t)s * split_stride + idx];
  C[(idx / cols) * ldc + idx % cols] = acc;
}

// Wave-wide sum on the ALU path (no LDS crossbar): DPP row_shr 1,2,4,8 leaves each
// 16-lane row's total in its last lane; four readlanes + scalar-operand adds give the
// wave total in a fixed order.  Result is wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float x) {
  int v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
  v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8
  v = __float_as_int(x);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(v, 15));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(v, 31));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(v, 47));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(v, 63));
  return (r0 + r1) + (r2 + r3);
}

// counter-based standard normal (splitmix-style hash + Box-Muller); used only when the
// caller supplies no replacement pool for degenerate atoms
__device__ __forceinline__ float counter_normal(unsigned long long seed, unsigned a, unsigned b) {
  unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * (((unsigned long long)a << 32) | b) + 0xD1B54A32D192ED03ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  const float u1 = ((unsigned)(x >> 40) + 1.0f) * (1.0f / 16777217.0f);
  const float u2 = (unsigned)((x >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// ---------------------------------------------------------------------------
// Sequential sweep over the kSweepBlock atoms [j0, j0+JB) -- NW waves, wave w owns the
// feature panel [256w, 256w+256) and lane l its features 256w + 4l .. +3 (d <= 256*NW).
// NW == 1 (d <= 256, the tuned case) has no barrier at all inside the sweep; for wider
// rows the squared norm of an atom is the fixed-order sum of the waves' partial sums,
// exchanged through LDS with one barrier per atom.
// ---------------------------------------------------------------------------
// FULL: all kSweepBlock atoms of the block exist (no per-atom branch at all, so hipcc can
// overlap the deferred row updates of atom a with the reduction chain of atom a+1).
// A degenerate atom (||u|| < eps, :92) leaves the model here (new atom = 0, dD = -old);
// its replacement direction is written afterwards by degenerate_fixup_kernel.
template <bool FULL, int NW>
__global__ __launch_bounds__(64 * NW) void sweep_block_kernel(const SweepParams p, int j0) {
  constexpr int JB = kSweepBlock;
  constexpr int DP = 256 * NW;                                     // == p.dp
  __shared__ __attribute__((aligned(16))) float sA[JB][JB];   // A[j0+a][j0+b] (symmetric)
  extern __shared__ __attribute__((aligned(16))) float sD_[];      // [JB][DP] old atoms of the block (rows of Dt)
  __shared__ float red[JB][NW];
  const int tid = threadIdx.x;
  const int fo = 4 * tid;                                           // first feature of this lane
  const int nb = FULL ? JB : min(JB, p.k - j0);
  for (int e = tid; e < JB * JB; e += 64 * NW) {
    const int a = e / JB, b = e % JB;
    sA[a][b] = (a < nb && b < nb) ? p.A[(int64_t)(j0 + a) * p.lda + j0 + b] : 0.0f;
  }
  for (int a = 0; a < JB; ++a) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a < nb) v = *(const f32x4*)(p.Dt + (int64_t)(j0 + a) * DP + fo);
    *(f32x4*)(&sD_[a * DP + fo]) = v;
  }
  float u[JB][4];
#pragma unroll
  for (int a = 0; a < JB; ++a) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a < nb) v = *(const f32x4*)(p.U + (int64_t)(j0 + a) * p.ldu + fo);   // padded cols are 0
#pragma unroll
    for (int f = 0; f < 4; ++f) u[a][f] = v[f];
  }
  __syncthreads();
  const float lo = p.positive ? 0.0f : -INFINITY;                 // dict_learning.py:87-88
  const float eps2 = p.eps * p.eps;
  unsigned degmask = 0;
  static_for<JB>([&](auto a_c) {
    constexpr int a = decltype(a_c)::value;
    // the block's coefficients of atom a, A[j0+b][j0+a] = sA[a][b] by symmetry: one batch of
    // broadcast ds_read_b128 instead of a dependent read per later atom
    float cf[JB];
#pragma unroll
    for (int b4 = 0; b4 < JB / 4; ++b4) {
      const f32x4 t4 = *(const f32x4*)(&sA[a][4 * b4]);
      cf[4 * b4] = t4[0]; cf[4 * b4 + 1] = t4[1]; cf[4 * b4 + 2] = t4[2]; cf[4 * b4 + 3] = t4[3];
    }
    const f32x4 dc4 = *(const f32x4*)(&sD_[a * DP + fo]);
    float v[4], dcur[4], ss = 0.0f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      dcur[f] = dc4[f];
      v[f] = fmaxf(fmaf(cf[a], dcur[f], u[a][f]), lo);            // u_j = U_j + A_jj d_j   (:85-88)
      ss = fmaf(v[f], v[f], ss);
    }
    ss = wave_sum_dpp(ss);
    if constexpr (NW > 1) {
      if ((tid & 63) == 0) red[a][tid >> 6] = ss;
      __syncthreads();
      ss = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) ss += red[a][w];
    }
    // ||u|| < eps  <=>  ||u||^2 < eps^2 (:91-92); 1/||u|| by v_rsq_f32 (1 ulp) -- the
    // sqrt + divide pair of :91,:100 would put ~25 dependent instructions on the chain
    const bool deg = ss < eps2;                                     // uniform over the workgroup
    const float inv = deg ? 0.0f : __builtin_amdgcn_rsqf(ss);
    f32x4 dnew, delta;
#pragma unroll
    for (int f = 0; f < 4; ++f) { dnew[f] = v[f] * inv; delta[f] = dnew[f] - dcur[f]; }
    if (FULL || a < nb) {
      *(f32x4*)(p.Dt + (int64_t)(j0 + a) * DP + fo) = dnew;
      *(f32x4*)(p.dD + (int64_t)a * DP + fo) = delta;
      degmask |= (deg ? 1u : 0u) << a;
    } else {
      *(f32x4*)(p.dD + (int64_t)a * DP + fo) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = a + 1; b < JB; ++b) {
#pragma unroll
      for (int f = 0; f < 4; ++f) u[b][f] = fmaf(-cf[b], delta[f], u[b][f]);
    }
  });
  if (tid < nb) p.degenerate[j0 + tid] = (int)((degmask >> tid) & 1u);   // one store, no per-atom branch
}

// Replacement directions for the degenerate atoms, in atom order: the i-th degenerate atom
// takes pool row i (normalised; dict_learning.py:93-96) or a counter-based N(0,1) vector.
__global__ __launch_bounds__(256) void degenerate_fixup_kernel(const SweepParams p) {
  __shared__ int s_idx[kSweepMaxK];
  __shared__ int s_count;
  __shared__ float sh[256];
  // ordered list of the degenerate atoms: every thread scans a contiguous slice of the flags,
  // an exclusive scan over the 256 slice counts gives each slice its place in the list
  // (the usual case -- no degenerate atom at all -- costs one pass instead of k dependent loads)
  __shared__ int s_cnt[256];
  const int per = (p.k + 255) / 256;
  const int lo = min((int)threadIdx.x * per, p.k), hi = min(lo + per, p.k);
  int mine = 0;
  for (int j = lo; j < hi; ++j) mine += p.degenerate[j] != 0;
  s_cnt[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int t = 0; t < 256; ++t) { const int v = s_cnt[t]; s_cnt[t] = c; c += v; }
    s_count = c;
    p.ndeg_in_out[0] = c;
  }
  __syncthreads();
  if (mine) {
    int at = s_cnt[threadIdx.x];
    for (int j = lo; j < hi; ++j)
      if (p.degenerate[j]) s_idx[at++] = j;
  }
  __syncthreads();
  const int cnt = s_count;
  auto direction = [&](int i, int j, int dd) {
    float g = 0.0f;
    if (dd < p.d) {
      if (p.pool && p.pool_rows > 0) g = p.pool[(int64_t)min(i, p.pool_rows - 1) * p.pool_ld + dd];
      else g = counter_normal(p.seed, (unsigned)j, (unsigned)dd);
      if (p.positive) g = fmaxf(g, 0.0f);
    }
    return g;
  };
  for (int i = 0; i < cnt; ++i) {
    const int j = s_idx[i];
    float part = 0.0f;
    for (int dd = threadIdx.x; dd < p.dp; dd += 256) { const float g = direction(i, j, dd); part = fmaf(g, g, part); }
    sh[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    const float inv = 1.0f / sqrtf(sh[0]);
    __syncthreads();
    for (int dd = threadIdx.x; dd < p.dp; dd += 256) p.Dt[(int64_t)j * p.dp + dd] = direction(i, j, dd) * inv;
  }
}

// Deferred form of the replacement (multi-GPU driver): the sweep never reads a replacement
// direction (a degenerate atom leaves the model), so the host can draw exactly as many
// directions as atoms degenerated AFTER the sweep and write them here: the i-th flagged atom
// (atom order) takes pool row i, clamped if `positive`, normalised (dict_learning.py:93-96).
__global__ __launch_bounds__(256) void fill_degenerate_kernel(float* __restrict__ D, int64_t ldd, int d, int k,
                                                              const int* __restrict__ degenerate,
                                                              const float* __restrict__ pool, int pool_rows,
                                                              int64_t pool_ld, int positive) {
  __shared__ float sh[256];
  int i = 0;
  for (int j = 0; j < k; ++j) {
    if (!degenerate[j]) continue;                       // uniform over the block
    const float* row = pool + (int64_t)min(i, pool_rows - 1) * pool_ld;
    float part = 0.0f;
    for (int dd = threadIdx.x; dd < d; dd += 256) {
      float g = row[dd];
      if (positive) g = fmaxf(g, 0.0f);
      part = fmaf(g, g, part);
    }
    sh[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    const float inv = 1.0f / sqrtf(sh[0]);
    __syncthreads();
    for (int dd = threadIdx.x; dd < d; dd += 256) {
      float g = row[dd];
      if (positive) g = fmaxf(g, 0.0f);
      D[(int64_t)dd * ldd + j] = g * inv;
    }
    ++i;
  }
}

// U[j'][:] -= sum_a A[j'][j0+a] * dD[a][:]   for j' >= j0 + JB; one row per workgroup
// iteration, blockIdx.y = panel of 256 features
__global__ __launch_bounds__(256) void trailing_update_kernel(const SweepParams p, int j0) {
  constexpr int JB = kSweepBlock;
  const int dd = 256 * blockIdx.y + threadIdx.x;   // feature
  float dl[JB];
#pragma unroll
  for (int a = 0; a < JB; ++a) dl[a] = p.dD[(int64_t)a * p.dp + dd];
  const int nb = min(JB, p.k - j0);
  for (int r = j0 + JB + blockIdx.x; r < p.k; r += gridDim.x) {
    const float* arow = p.A + (int64_t)r * p.lda + j0;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < JB; ++a) acc = fmaf((a < nb) ? arow[a] : 0.0f, dl[a], acc);
    if (dd < p.d) p.U[(int64_t)r * p.ldu + dd] -= acc;
  }
}

// dst[c][r] = src[r][c] (zero padded to the dst extents)
__global__ void transpose_pad_kernel(const float* __restrict__ src, int64_t lds_, int rows, int cols,
                                     float* __restrict__ dst, int64_t ldd, int drows, int dcols) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    t[i][threadIdx.x] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;     // dst row = c, dst col = r
    if (c < drows && r < dcols) dst[(int64_t)c * ldd + r] = t[threadIdx.x][i];
  }
}

// Elementwise tail of one FISTA iteration for the unfused large-shape path:
//   z_next = softshrink(y - lr*g, lam); dpart = sum|z - z_next|; y = z_next + c (z_next - z); z = z_next
// (ista.py:90,93,98-102).  Fixed grid => deterministic partial sums.
__global__ __launch_bounds__(256) void generic_prox_kernel(float* __restrict__ Z, int64_t ldz,
                                                           float* __restrict__ Y, const float* __restrict__ G,
                                                           int n, int k, float lr, float lam, float coef,
                                                           float* __restrict__ dpart) {
  __shared__ float sh[256];
  float acc = 0.0f;
  const int64_t total = (int64_t)n * k;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / k;
    const int cc = (int)(idx - r * k);
    const float zo = Z[r * ldz + cc];
    const float v = __fsub_rn(Y[idx], __fmul_rn(lr, G[idx]));
    const float zn = __fsub_rn(v, __builtin_amdgcn_fmed3f(v, -lam, lam));
    acc += __builtin_fabsf(__fsub_rn(zo, zn));
    Y[idx] = __fadd_rn(zn, __fmul_rn(coef, __fsub_rn(zn, zo)));
    Z[r * ldz + cc] = zn;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dpart[blockIdx.x] = sh[0];
}

// Z[:, j] = 0 for degenerate atoms (dict_learning.py:98; matters when persist=True)
__global__ void zero_columns_kernel(float* __restrict__ Z, int64_t ldz, int n, int k,
                                    const int* __restrict__ degenerate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * k) return;
  const int c = (int)(idx % k);
  if (degenerate[c]) Z[(idx / k) * ldz + c] = 0.0f;
}

}  // namespace

// Number of sample splits that gives the chip ~3 workgroups per CU (the output has only
// (pc/64)*(qc/64) blocks), each split keeping at least 512 samples.
int gram_splits(int pc, int qc, int n, int sym, int cus) {
  int blocks = ((qc + kGB - 1) / kGB) * ((pc + kGB - 1) / kGB);
  if (sym) blocks = blocks / 2 + (pc + kGB - 1) / kGB / 2 + 1;
  int s = (3 * cus + blocks - 1) / std::max(blocks, 1);
  s = std::min(s, std::max(n / 512, 1));
  return std::max(1, std::min(s, 16));
}

hipError_t launch_gram_tn(const float* P, int64_t ldp, int pc, const float* Q, int64_t ldq, int qc,
                          int n, float* C, int64_t ldc, int sym, float* scratch, int splits,
                          hipStream_t stream) {
  if (!scratch) splits = 1;
  const int rows_per_split = ((n + splits - 1) / splits + kGS - 1) / kGS * kGS;
  splits = std::max(1, (n + rows_per_split - 1) / std::max(rows_per_split, 1));
  const dim3 grid((qc + kGB - 1) / kGB, (pc + kGB - 1) / kGB, splits);
  const bool vec = pc % 4 == 0 && qc % 4 == 0 && ldp % 4 == 0 && ldq % 4 == 0 &&
                   ((uintptr_t)P & 15) == 0 && ((uintptr_t)Q & 15) == 0;
  float* dst = splits > 1 ? scratch : C;
  const int64_t dld = splits > 1 ? qc : ldc;
  const int64_t stride = (int64_t)pc * qc;
  if (vec)
    hipLaunchKernelGGL(gram_tn_kernel<true>, grid, dim3(256), 0, stream, P, ldp, pc, Q, ldq, qc, n, dst,
                       dld, sym, rows_per_split, stride);
  else
    hipLaunchKernelGGL(gram_tn_kernel<false>, grid, dim3(256), 0, stream, P, ldp, pc, Q, ldq, qc, n, dst,
                       dld, sym, rows_per_split, stride);
  if (splits > 1) {
    const int64_t total = (int64_t)pc * qc;
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       scratch, splits, stride, pc, qc, C, ldc);
  }
  return hipGetLastError();
}

template <int NW>
static hipError_t sweep_blocks(const SweepParams& p, hipStream_t stream) {
  const size_t lds = (size_t)kSweepBlock * 256 * NW * 4;
  if (lds > 32 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&sweep_block_kernel<true, NW>), lds);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void*>(&sweep_block_kernel<false, NW>), lds);
    if (e != hipSuccess) return e;
  }
  for (int j0 = 0; j0 < p.k; j0 += kSweepBlock) {
    if (j0 + kSweepBlock <= p.k)
      hipLaunchKernelGGL((sweep_block_kernel<true, NW>), dim3(1), dim3(64 * NW), lds, stream, p, j0);
    else
      hipLaunchKernelGGL((sweep_block_kernel<false, NW>), dim3(1), dim3(64 * NW), lds, stream, p, j0);
    if (j0 + kSweepBlock < p.k) {
      const int rows = p.k - j0 - kSweepBlock;
      hipLaunchKernelGGL(trailing_update_kernel, dim3(std::min(rows, 256), NW), dim3(256), 0, stream, p, j0);
    }
  }
  return hipGetLastError();
}

hipError_t launch_dict_sweep(const SweepParams& p, hipStream_t stream) {
  hipError_t e;
  switch (p.dp) {
    case 256: e = sweep_blocks<1>(p, stream); break;
    case 512: e = sweep_blocks<2>(p, stream); break;
    case 768: e = sweep_blocks<3>(p, stream); break;
    case 1024: e = sweep_blocks<4>(p, stream); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(degenerate_fixup_kernel, dim3(1), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_fill_degenerate(float* D, int64_t ldd, int d, int k, const int* degenerate, const float* pool,
                                  int pool_rows, int64_t pool_ld, int positive, hipStream_t stream) {
  hipLaunchKernelGGL(fill_degenerate_kernel, dim3(1), dim3(256), 0, stream, D, ldd, d, k, degenerate, pool,
                     pool_rows, pool_ld, positive);
  return hipGetLastError();
}

hipError_t launch_transpose_pad(const float* src, int64_t ld_src, int rows, int cols, float* dst,
                                int64_t ld_dst, int drows, int dcols, hipStream_t stream) {
  const int gr = std::max(rows, dcols), gc = std::max(cols, drows);
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((gc + 31) / 32, (gr + 31) / 32), dim3(32, 8), 0, stream,
                     src, ld_src, rows, cols, dst, ld_dst, drows, dcols);
  return hipGetLastError();
}

hipError_t launch_generic_prox(float* Z, int64_t ldz, float* Y, const float* G, int n, int k, float lr,
                               float lam, float coef, float* dpart, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(generic_prox_kernel, dim3(grid), dim3(256), 0, stream, Z, ldz, Y, G, n, k, lr, lam,
                     coef, dpart);
  return hipGetLastError();
}

hipError_t launch_zero_columns(float* Z, int64_t ldz, int n, int k, const int* degenerate,
                               hipStream_t stream) {
  const int64_t total = (int64_t)n * k;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(zero_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     Z, ldz, n, k, degenerate);
  return hipGetLastError();
}

}  // namespace lasso
