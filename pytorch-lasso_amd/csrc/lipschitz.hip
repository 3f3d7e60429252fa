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
// 33.5 MFLOP at m = 256) on v_mfma_f64_16x16x4_f64 -- launch-latency bound, not on the
// FISTA hot loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "lasso_kernels.h"

namespace lasso {
int g_force_standby = 0;
namespace {

constexpr int kLipTile = 32;   // output tile per workgroup (256 threads, 2x2 per thread)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// {lr, alpha lr} of an lr = LASSO_LR_AUTO solve, left by whichever launch finishes lambda_max (as prepare_block would
// from the stored double: the same conversions)
__device__ __forceinline__ void lip_write_lr(const LipLr lr, double lambda_max) {
  if (!lr.slot) return;
  const double step = 1.0 / lambda_max;
  lr.slot[0] = (float)step;
  lr.slot[1] = (float)(lr.alpha * step);
}

constexpr int kLipSpan = 256;      // contraction elements whose loads one workgroup keeps in flight together
constexpr int kLipMaxSplits = 16;  // partial products along the contraction (blockIdx.z)

// C_z[mp x mp] = (s A)(s A)^T restricted to the contraction range of split z, on
// v_mfma_f64_16x16x4_f64; A given as elem(i,t) = A[i*si + t*st] (TIn = float for the Gram of W,
// double for the squarings), rows >= m and t >= len read 0.
// One workgroup = 4 waves = one 32x32 tile (a 16x16 block per wave).  These products are
// tiny and latency-bound, so ALL global loads of a 256-element span of the contraction are
// issued up front (8 chunks of 32 into registers) and then staged through LDS chunk by
// chunk: one exposed memory latency per span instead of one per chunk.
// SCALE: s = 1/tr(A) (A square, ld = si), else s = 1.  Split z writes its partial product to
// C + z*mp*mp; fold_partials_kernel sums the splits in a fixed order.
// MFMA f64 layouts: A/B one double per lane (row l&15, k = l>>4); C/D col = l&15,
// row = (l>>4) + 4*reg.
template <typename TIn, bool SCALE>
__global__ __launch_bounds__(256) void syrk_f64_kernel(const TIn* __restrict__ A, int64_t si, int64_t st,
                                                       int m, int len, int mp, int spans_per_split,
                                                       double* __restrict__ C) {
  constexpr int RS = 33;                            // padded row stride (doubles)
  __shared__ double sa[32][RS], sb[32][RS];
  __shared__ double sh[256];
  auto elem = [&](int64_t off) -> double { return (double)A[off]; };
  double inv = 1.0;
  if constexpr (SCALE) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) acc += elem((int64_t)i * si + i);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    inv = 1.0 / sh[0];
    __syncthreads();
  }
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int iw = 16 * (w >> 1), jw = 16 * (w & 1);
  const int l15 = lane & 15, q = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int span0 = blockIdx.z * spans_per_split;
  for (int sp = span0; sp < span0 + spans_per_split; ++sp) {
    const int tbase = sp * kLipSpan;
    if (tbase >= len) break;
    double ra[8][4], rb[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int e = tid + 256 * h;
        const int r = e >> 5, t = tbase + 32 * c + (e & 31);     // consecutive threads -> consecutive t
        ra[c][h] = (i0 + r < m && t < len) ? elem((int64_t)(i0 + r) * si + (int64_t)t * st) : 0.0;
        rb[c][h] = (j0 + r < m && t < len) ? elem((int64_t)(j0 + r) * si + (int64_t)t * st) : 0.0;
      }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (tbase + 32 * c >= len) break;
      __syncthreads();                               // previous chunk's fragment reads are done
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int e = tid + 256 * h;
        sa[e >> 5][e & 31] = ra[c][h] * inv;
        sb[e >> 5][e & 31] = rb[c][h] * inv;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[iw + l15][4 * ks + q], sb[jw + l15][4 * ks + q], acc, 0, 0, 0);
    }
  }
  double* const Cz = C + (int64_t)blockIdx.z * mp * mp;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
    Cz[(size_t)(i0 + iw + q + 4 * rg) * mp + j0 + jw + l15] = acc[rg];
}

// One squaring  C = (P / tr P)(P / tr P)^T  of the m x m iterate, m = mp <= 256: the same arithmetic as
// syrk_f64_kernel<double, true> (scaled operands, contraction in ascending order, the same reduction tree for
// the trace: bitwise the same C), laid out for latency -- a squaring is 33 MFLOP on 64 workgroups, and the
// general kernel spent 8.6 us on it: the trace reduction in front of the operand loads, then eight staged
// chunks with two barriers each.  Here all operand loads (32 x mp doubles per operand, 16-byte pieces) and the
// diagonal are issued at once, the trace is reduced while they fly (two barriers, then shuffles), both operand
// tiles go to LDS whole (132 KB at mp = 256) and the 64 fp64 MFMA steps run without a barrier in between.
// NB = ceil(mp / 64): 16-byte column groups per thread and row.
// (Round 3 tried one WAVE per 16 x 16 block with the operands loaded straight into the lanes' MFMA slots -- 256
// workgroups, no LDS, no barrier, bitwise the same C: 8.2 us against this kernel's 7.4; the lane layout of the fp64
// MFMA makes those loads 32-byte pieces, 2048 cache-line requests per wave.)
template <int NB>
__global__ __launch_bounds__(256) void square_f64_kernel(const double* __restrict__ A, int mp, double* __restrict__ C,
                                                         int may_skip) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double sq_smem[];
  const int RS = mp + 1;
  double* const sa = sq_smem;                 // [32][RS]
  double* const sb = sq_smem + 32 * RS;       // [32][RS]
  double* const sh = sq_smem + 64 * RS;       // [256]
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tx = tid & 31, ty = tid >> 5;
  f64x2 ra[4][NB], rb[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int r = ty + 8 * a, t = min(2 * (tx + 32 * b), mp - 2);
      ra[a][b] = *reinterpret_cast<const f64x2*>(A + (int64_t)(i0 + r) * mp + t);
      rb[a][b] = *reinterpret_cast<const f64x2*>(A + (int64_t)(j0 + r) * mp + t);
    }
  // tr A: the tree of syrk_f64_kernel (sh[t] += sh[t + s], s = 128 .. 1), its last six levels inside wave 0
  sh[tid] = tid < mp ? A[(int64_t)tid * mp + tid] : 0.0;
  __syncthreads();
  if (tid < 128) sh[tid] += sh[tid + 128];
  __syncthreads();
  if (tid < 64) {
    double v = sh[tid] + sh[tid + 64];
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) v += __shfl_down(v, s2, 64);
    if (tid == 0) sh[0] = v;
  }
  __syncthreads();
  // Round 4: the input of every squaring but the first is normalised -- tr A = sum mu_i^2 / (sum mu_i)^2 of the previous
  // iterate's eigenvalues -- so 1 - tr A < 1e-15 says A already IS the rank-one projector to fp64 precision: this
  // squaring would reproduce it.  The launch then only copies its tile (every workgroup reads the same trace and takes
  // the same branch); dictionaries with a healthy gap get there after 10-13 of the 20 squarings.
#ifdef LASSO_LIP_NOSKIP
  may_skip = 0;
#endif
  if (may_skip && 1.0 - sh[0] < 1e-15) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int e = tid + 256 * h, r = e >> 5, cc = e & 31;
      C[(size_t)(i0 + r) * mp + j0 + cc] = A[(size_t)(i0 + r) * mp + j0 + cc];
    }
    return;
  }
  const double inv = 1.0 / sh[0];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int r = ty + 8 * a, t = 2 * (tx + 32 * b);
      if (t < mp) {
        sa[r * RS + t] = ra[a][b][0] * inv; sa[r * RS + t + 1] = ra[a][b][1] * inv;
        sb[r * RS + t] = rb[a][b][0] * inv; sb[r * RS + t + 1] = rb[a][b][1] * inv;
      }
    }
  __syncthreads();
  const int iw = 16 * (w >> 1), jw = 16 * (w & 1);
  const int l15 = lane & 15, q = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const double* const pa = sa + (iw + l15) * RS + q;
  const double* const pb = sb + (jw + l15) * RS + q;
  for (int c = 0; c < mp / 32; ++c)          // (mp is a multiple of 32)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[32 * c + 4 * ks], pb[32 * c + 4 * ks], acc, 0, 0, 0);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) C[(size_t)(i0 + iw + q + 4 * rg) * mp + j0 + jw + l15] = acc[rg];
}

// The Gram matrix of W over ONE 256-element span of the contraction per workgroup (split z = span z), laid out
// like square_f64_kernel: all loads of both 32 x 256 operand tiles in flight at once (16-byte pieces of fp32),
// converted to fp64 on their way into LDS, 64 MFMA steps without a barrier in between.  Same contraction order
// per split as syrk_f64_kernel<float, false> -- bitwise the same partial products.
// ROWS: elem(i, t) = W[i * ld + t] (G = W W^T), else elem(i, t) = W[t * ld + i] (G = W^T W).
// Needs m % 4 == 0 or ROWS, len % 4 == 0 or !ROWS, ld % 4 == 0, a 16-byte aligned base.
// Round 6: the launch also carries the blocks of a solve's prepare launch (job.gx > 0: z slices >= gs; both read only W) --
// an lr = LASSO_LR_AUTO solve is one launch shorter.
template <bool ROWS>
__global__ __launch_bounds__(256) void gram_span_f64_kernel(const float* __restrict__ W, int64_t ld, int m, int len,
                                                            int mp, double* __restrict__ C, int gs, const PrepareJob job) {
  constexpr int RS = 257;
  extern __shared__ __attribute__((aligned(16))) double gs_smem[];
  if ((int)blockIdx.z >= gs) {
    const int id = (((int)blockIdx.z - gs) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    if (id < job.gx * job.gy) prepare_block(job, id % job.gx, id / job.gx, threadIdx.x, reinterpret_cast<float (*)[33]>(gs_smem));
    return;
  }
  double* const sa = gs_smem;                 // [32][RS]
  double* const sb = gs_smem + 32 * RS;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, tbase = blockIdx.z * 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x4v ra[8], rb[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const int e = tid + 256 * h;
    if constexpr (ROWS) {
      const int r = e >> 6, t = tbase + 4 * (e & 63);
      const int tc = min(t, len - 4);
      ra[h] = *reinterpret_cast<const f32x4v*>(W + (int64_t)min(i0 + r, m - 1) * ld + tc);
      rb[h] = *reinterpret_cast<const f32x4v*>(W + (int64_t)min(j0 + r, m - 1) * ld + tc);
    } else {
      const int tl = e >> 3, i4 = 4 * (e & 7);
      const int64_t tc = min(tbase + tl, len - 1);
      ra[h] = *reinterpret_cast<const f32x4v*>(W + tc * ld + min(i0 + i4, m - 4));
      rb[h] = *reinterpret_cast<const f32x4v*>(W + tc * ld + min(j0 + i4, m - 4));
    }
  }
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const int e = tid + 256 * h;
    if constexpr (ROWS) {
      const int r = e >> 6, tl = 4 * (e & 63);
      const bool okt = tbase + tl < len;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sa[r * RS + tl + u] = (okt && i0 + r < m) ? (double)ra[h][u] : 0.0;
        sb[r * RS + tl + u] = (okt && j0 + r < m) ? (double)rb[h][u] : 0.0;
      }
    } else {
      const int tl = e >> 3, i4 = 4 * (e & 7);
      const bool okt = tbase + tl < len;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sa[(i4 + u) * RS + tl] = (okt && i0 + i4 < m) ? (double)ra[h][u] : 0.0;
        sb[(i4 + u) * RS + tl] = (okt && j0 + i4 < m) ? (double)rb[h][u] : 0.0;
      }
    }
  }
  __syncthreads();
  const int iw = 16 * (w >> 1), jw = 16 * (w & 1);
  const int l15 = lane & 15, q = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const double* const pa = sa + (iw + l15) * RS + q;
  const double* const pb = sb + (jw + l15) * RS + q;
  const int chunks = min(8, (len - tbase + 31) / 32);       // like syrk_f64_kernel: chunks beyond len are skipped
  for (int c = 0; c < chunks; ++c)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[32 * c + 4 * ks], pb[32 * c + 4 * ks], acc, 0, 0, 0);
  double* const Cz = C + (int64_t)blockIdx.z * mp * mp;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Cz[(size_t)(i0 + iw + q + 4 * rg) * mp + j0 + jw + l15] = acc[rg];
}

// wave-wide sum of doubles on the ALU path (DPP row shifts + four readlanes): a fixed order, no LDS round trips
template <int CTRL>
__device__ __forceinline__ double dpp_row_shr_f64(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64_dpp(double x) {
  x += dpp_row_shr_f64<0x111>(x);
  x += dpp_row_shr_f64<0x112>(x);
  x += dpp_row_shr_f64<0x114>(x);
  x += dpp_row_shr_f64<0x118>(x);        // lane 15 of every 16-lane row holds the row's total
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  double r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * i + 15);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * i + 15);
    r[i] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// ALL squarings of a small iterate (m = mp <= 64: dictionaries with d or k <= 64, e.g. 8 x 8 patches) in ONE
// launch of one workgroup: the iterate lives in LDS (two MP x (MP + 2) buffers), one wave per 16 x 16 block of the
// UPPER triangle (the iterate is symmetric: the strictly lower blocks are written as mirror images -- 10 of 16 blocks
// at MP = 64), per squaring {all 2 x MP/4 operand reads of the block issued together, MP/4 fp64 MFMA steps, the block
// scaled by 1 / tr(P)^2, the diagonal waves' share of the NEW trace by DPP sums, ONE barrier}.  Round 3's form read two
// operands per MFMA step in a rolled loop (one LDS round trip per step), scaled both operands, and spent three
// barriers on the trace tree of the per-launch kernels: 3.3 us per squaring; this one: see DESIGN.md 3.3.
// P_{i+1} = P_i P_i^T / tr(P_i)^2 as before; the sums inside a squaring are taken in another (fixed) order.
template <int MP>
__global__ __launch_bounds__(1024) void square_chain_f64_kernel(const double* __restrict__ G, int squarings,
                                                                double* __restrict__ Pout, double* __restrict__ out,
                                                                const LipLr lr) {
  constexpr int RS = MP + 2, NBK = MP / 16, NBLK = NBK * (NBK + 1) / 2;   // (row pitch = 4 banks mod 64: the operand reads and the mirror writes are conflict-free)
  __shared__ double buf[2][MP * RS];
  __shared__ double trp[2][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  // block (bi <= bj) of wave w < NBLK: row-major walk of the upper triangle
  int bi = 0, bj = 0;
  {
    int rem = w;
    for (int r = 0; r < NBK; ++r) {
      const int cnt = NBK - r;
      if (rem < cnt) { bi = r; bj = r + rem; break; }
      rem -= cnt;
    }
  }
  for (int e = tid; e < MP * MP; e += 1024) buf[0][(e / MP) * RS + e % MP] = G[e];
  if (tid < 8) trp[tid >> 2][tid & 3] = 0.0;
  __syncthreads();
  if (w < NBK) {                              // tr(G): wave w adds rows 16 w .. 16 w + 15 of the diagonal
    const double d = lane < 16 ? buf[0][(16 * w + lane) * RS + 16 * w + lane] : 0.0;
    const double t = wave_sum_f64_dpp(d);
    if (lane == 0) trp[0][w] = t;
  }
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int p = 0; p < squarings; ++p) {
    const double* const src = buf[cur];
    double* const dst = buf[cur ^ 1];
    const double tr = ((trp[cur][0] + trp[cur][1]) + trp[cur][2]) + trp[cur][3];
    // tr(P_p) = sum mu_i^2 / (sum mu_i)^2 of the previous iterate's eigenvalues: 1 - tr is twice the weight of
    // everything beside the dominant eigenvector.  Below 1e-15 the iterate IS the rank-one projector to fp64 precision
    // and further squarings reproduce it: stop (every thread reads the same tr; dictionaries with a healthy gap get
    // there after 10-13 of the 20 squarings; a multiple top eigenvalue never does and runs them all).
    if (p > 0 && 1.0 - tr < 1e-15) break;
    const double inv = 1.0 / tr, inv2 = inv * inv;
    if (w < NBLK) {
      double a[MP / 4], b[MP / 4];
      const double* const pa = src + (16 * bi + l15) * RS + q;
      const double* const pb = src + (16 * bj + l15) * RS + q;
#pragma unroll
      for (int ks = 0; ks < MP / 4; ++ks) { a[ks] = pa[4 * ks]; b[ks] = pb[4 * ks]; }
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < MP / 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
      double dg = 0.0;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const double v = acc[rg] * inv2;
        const int r = q + 4 * rg;                                  // row inside the block; column = l15
        dst[(16 * bi + r) * RS + 16 * bj + l15] = v;
        if (bi != bj) dst[(16 * bj + l15) * RS + 16 * bi + r] = v;   // mirror image
        else if (r == l15) dg = v;
      }
      if (bi == bj) {
        const double t = wave_sum_f64_dpp(dg);
        if (lane == 0) trp[cur ^ 1][bi] = t;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  if (!out) {
    for (int e = tid; e < MP * MP; e += 1024) Pout[e] = buf[cur][(e / MP) * RS + e % MP];
    return;
  }
  // out[0] = <G, P>_F / tr(P), the sums of rayleigh_trace_kernel (1024 threads, same chains and trees) with P read
  // from LDS instead of from the copy a separate launch would need: one launch less behind the squarings
  __shared__ double shq[1024];
  const double* const P = buf[cur];
  double tr = 0.0;
  for (int i = tid; i < MP; i += 1024) tr += P[i * RS + i];
  shq[tid] = tr;
  __syncthreads();
  for (int s2 = 512; s2 > 0; s2 >>= 1) {
    if (tid < s2) shq[tid] += shq[tid + s2];
    __syncthreads();
  }
  tr = shq[0];
  __syncthreads();
  double acc = 0.0;
  constexpr int total = MP * MP;
  for (int e0 = tid; e0 < total; e0 += 8 * 1024) {
    double gv[8], pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 1024 * u < total ? e0 + 1024 * u : total - 1;
      gv[u] = G[e];
      pv[u] = P[(e / MP) * RS + e % MP];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 1024 * u < total) acc = fma(gv[u], pv[u], acc);
  }
  shq[tid] = acc;
  __syncthreads();
  for (int s2 = 512; s2 > 0; s2 >>= 1) {
    if (tid < s2) shq[tid] += shq[tid + s2];
    __syncthreads();
  }
  if (tid == 0) { const double lam = shq[0] / tr; out[0] = lam; lip_write_lr(lr, lam); }
}

// C[e] = sum_z part[z][e]  (fixed order)
__global__ __launch_bounds__(256) void fold_partials_kernel(const double* __restrict__ part, int splits, int64_t mm,
                                                            double* __restrict__ C, int* __restrict__ zero4 = nullptr) {
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0;    // flags of the launch behind this one
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= mm) return;
  double v = part[e];
  for (int s = 1; s < splits; ++s) v += part[e + s * mm];
  C[e] = v;
}

// out[0] = <G, P>_F / tr(P)   (single block, fixed summation order)
__global__ __launch_bounds__(1024) void rayleigh_trace_kernel(const double* __restrict__ G,
                                                              const double* __restrict__ P, int mp,
                                                              double* __restrict__ out, const LipLr lr) {
  __shared__ double sh[1024];
  double tr = 0.0;
  for (int i = threadIdx.x; i < mp; i += 1024) tr += P[(size_t)i * mp + i];
  sh[threadIdx.x] = tr;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  tr = sh[0];
  __syncthreads();
  double acc = 0.0;
  const size_t total = (size_t)mp * mp;
  // (same chain of fmas per thread; the loads of eight links are issued together instead of one round trip each)
  for (size_t e0 = threadIdx.x; e0 < total; e0 += 8 * 1024) {
    double gv[8], pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t e = e0 + (size_t)1024 * u < total ? e0 + (size_t)1024 * u : total - 1;
      gv[u] = G[e];
      pv[u] = P[e];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + (size_t)1024 * u < total) acc = fma(gv[u], pv[u], acc);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sh[0] / tr; lip_write_lr(lr, out[0]); }
}

// The same quotient in two launches that use the chip: kRayBlocks workgroups each reduce a contiguous chunk of <G, P>_F
// and of the diagonal of P in a fixed order, one small block adds the partial sums in index order.  (The single block
// above reads 1 MiB through one compute unit: 18 us at m = 256; these two launches take 5.)
constexpr int kRayBlocks = 64;
__global__ __launch_bounds__(256) void rayleigh_partial_kernel(const double* __restrict__ G, const double* __restrict__ P,
                                                               int mp, double* __restrict__ part) {
  __shared__ double sh[2][256];
  const size_t total = (size_t)mp * mp;
  const size_t chunk = (total + kRayBlocks - 1) / kRayBlocks;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
  double acc = 0.0, tr = 0.0;
  for (size_t e0 = lo + threadIdx.x; e0 < hi; e0 += 8 * 256) {
    double gv[8], pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t e = e0 + (size_t)256 * u < hi ? e0 + (size_t)256 * u : hi - 1;
      gv[u] = G[e];
      pv[u] = P[e];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t e = e0 + (size_t)256 * u;
      if (e < hi) {
        acc = fma(gv[u], pv[u], acc);
        if (e / mp == e % mp) tr += pv[u];
      }
    }
  }
  sh[0][threadIdx.x] = acc;
  sh[1][threadIdx.x] = tr;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0][0]; part[2 * blockIdx.x + 1] = sh[1][0]; }
}
__global__ void rayleigh_finish_kernel(const double* __restrict__ part, double* __restrict__ out, const LipLr lr) {
  double num = 0.0, tr = 0.0;
  for (int b = 0; b < kRayBlocks; ++b) { num += part[2 * b]; tr += part[2 * b + 1]; }
  out[0] = num / tr;
  lip_write_lr(lr, out[0]);
}
// `part`: 2 kRayBlocks doubles of scratch (the split-partials region of the workspace is free by then)
static void launch_rayleigh(const double* G, const double* P, int mp, double* out, double* part, hipStream_t stream,
                            const LipLr lr) {
  if (mp < 128) {            // small iterates: one block is the shorter path
    hipLaunchKernelGGL(rayleigh_trace_kernel, dim3(1), dim3(1024), 0, stream, G, P, mp, out, lr);
    return;
  }
  hipLaunchKernelGGL(rayleigh_partial_kernel, dim3(kRayBlocks), dim3(256), 0, stream, G, P, mp, part);
  hipLaunchKernelGGL(rayleigh_finish_kernel, dim3(1), dim3(1), 0, stream, part, out, lr);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Every squaring of a 64 < mp <= 256 iterate AND the Rayleigh quotient in ONE launch (round 4).  The chain of 20
// square_f64_kernel launches + 2 for the quotient cost 7.4 us per squaring (a launch each, whether it squares or --
// once the iterate has converged -- only copies) on a chip that needs 1 us for the arithmetic.  Here the (mp / 32)^2
// workgroups keep their tile, meet at a barrier in global memory after each squaring (tiles written through,
// drained, one relaxed atomic per workgroup; loads past the caches), ALL leave the loop at the first squaring whose
// input already is the rank-one projector (same test, same trace, so the same iterate reaches the quotient as in the
// multi-launch form, where the remaining squarings only copy), and then each reduces its chunk of <G, P> and tr P --
// the chunks and orders of rayleigh_partial_kernel / rayleigh_finish_kernel; the workgroup that arrives last adds
// the partial sums in index order.  Arithmetic per tile is square_f64_kernel's: bitwise the same lambda.
// All workgroups must be resident (64 at mp = 256): every wait is bounded; on a timeout the grid raises flags[1]
// and leaves, and a stand-by launch of the same kernel (one workgroup, every tile itself) redoes the computation
// from G -- it returns at once otherwise.
// ---------------------------------------------------------------------------------------------------------------
struct LipPersist {
  const double* G; double* P0; double* P1; double* part; double* out;
  int* flags;          // [0] barrier arrivals, [1] abort, [2] quotient arrivals
  int mp, squarings, solo;
  const int* run_if;   // nullable: run only if *run_if != 0
  LipLr lr;            // (slot nullable) {lr, alpha lr} next to out[0]
};

__device__ __forceinline__ bool lip_spin_until(const int* flag, int want, int* abort_flag) {
  for (int spins = 0; spins < kStopSpinLimit; ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

template <int NB>
__global__ __launch_bounds__(256) void square_persist_f64_kernel(const LipPersist x) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if (x.run_if && *x.run_if == 0) return;
  extern __shared__ __attribute__((aligned(16))) double sq_smem[];
  const int mp = x.mp, RS = mp + 1;
  double* const sa = sq_smem;                 // [32][RS]
  double* const sb = sq_smem + 32 * RS;       // [32][RS]
  double* const sh = sq_smem + 64 * RS;       // [256]
  __shared__ int sh_ok;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tx = tid & 31, ty = tid >> 5;
  const int side = mp / 32, tiles = side * side, nwg = x.solo ? 1 : (int)gridDim.x;
  const unsigned bytes = (unsigned)((size_t)mp * mp * sizeof(double));
  // device-coherent accesses (sc1): what another workgroup wrote in this launch is read past L1 / L2
  auto ld2 = [](const __amdgpu_buffer_rsrc_t r, size_t idx) {
    return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(idx * 8), 0, 16));
  };
  auto ld1 = [](const __amdgpu_buffer_rsrc_t r, size_t idx) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(idx * 8), 0, 16));
  };
  auto st1 = [](const __amdgpu_buffer_rsrc_t r, size_t idx, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (unsigned)(idx * 8), 0, 16);
  };
  const double* src = x.G;
  for (int p = 0; p < x.squarings; ++p) {
    double* const dst = (p & 1) ? x.P1 : x.P0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes, 0x00020000);
    bool converged = false;
    for (int t = blockIdx.x; t < tiles; t += nwg) {
      const int i0 = (t / side) * 32, j0 = (t % side) * 32;
      f64x2 ra[4][NB], rb[4][NB];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int r = ty + 8 * a, tt = min(2 * (tx + 32 * b), mp - 2);
          ra[a][b] = ld2(rs, (size_t)(i0 + r) * mp + tt);
          rb[a][b] = ld2(rs, (size_t)(j0 + r) * mp + tt);
        }
      // tr A: the tree of square_f64_kernel
      sh[tid] = tid < mp ? ld1(rs, (size_t)tid * mp + tid) : 0.0;
      __syncthreads();
      if (tid < 128) sh[tid] += sh[tid + 128];
      __syncthreads();
      if (tid < 64) {
        double v = sh[tid] + sh[tid + 64];
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) v += __shfl_down(v, s2, 64);
        if (tid == 0) sh[0] = v;
      }
      __syncthreads();
      const double tr = sh[0];
      if (p > 0 && 1.0 - tr < 1e-15) { converged = true; break; }   // (the same for every tile and workgroup)
      const double inv = 1.0 / tr;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int r = ty + 8 * a, tt = 2 * (tx + 32 * b);
          if (tt < mp) {
            sa[r * RS + tt] = ra[a][b][0] * inv; sa[r * RS + tt + 1] = ra[a][b][1] * inv;
            sb[r * RS + tt] = rb[a][b][0] * inv; sb[r * RS + tt + 1] = rb[a][b][1] * inv;
          }
        }
      __syncthreads();
      const int iw = 16 * (w >> 1), jw = 16 * (w & 1);
      const int l15 = lane & 15, q = lane >> 4;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      const double* const pa = sa + (iw + l15) * RS + q;
      const double* const pb = sb + (jw + l15) * RS + q;
      for (int c = 0; c < mp / 32; ++c)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[32 * c + 4 * ks], pb[32 * c + 4 * ks], acc, 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) st1(rd, (size_t)(i0 + iw + q + 4 * rg) * mp + j0 + jw + l15, acc[rg]);
      __syncthreads();                                      // (solo: the next tile reuses the LDS tiles)
    }
    if (converged) break;
    src = dst;
    // every tile of this squaring written before any workgroup reads it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!x.solo) {
      if (tid == 0) {
        __hip_atomic_fetch_add(x.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_ok = lip_spin_until(x.flags, nwg * (p + 1), x.flags + 1);
      }
      __syncthreads();
      if (!sh_ok) return;
    }
  }
  // ---- <G, P> / tr P: chunk c as rayleigh_partial_kernel's workgroup c ----
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(x.part, 0, 2 * kRayBlocks * 8, 0x00020000);
    double* const r0 = sq_smem, * const r1 = sq_smem + 256;
    const size_t total = (size_t)mp * mp;
    const size_t chunk = (total + kRayBlocks - 1) / kRayBlocks;
    for (int c = blockIdx.x; c < kRayBlocks; c += nwg) {
      const size_t lo = (size_t)c * chunk, hi = lo + chunk < total ? lo + chunk : total;
      double acc = 0.0, tr = 0.0;
      for (size_t e0 = lo + tid; e0 < hi; e0 += 8 * 256) {
        double gv[8], pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const size_t e = e0 + (size_t)256 * u < hi ? e0 + (size_t)256 * u : hi - 1;
          gv[u] = x.G[e];
          pv[u] = ld1(rs, e);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const size_t e = e0 + (size_t)256 * u;
          if (e < hi) {
            acc = fma(gv[u], pv[u], acc);
            if (e / mp == e % mp) tr += pv[u];
          }
        }
      }
      __syncthreads();
      r0[tid] = acc;
      r1[tid] = tr;
      __syncthreads();
      for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) { r0[tid] += r0[tid + s2]; r1[tid] += r1[tid + s2]; }
        __syncthreads();
      }
      if (tid == 0) { st1(rp, 2 * c, r0[0]); st1(rp, 2 * c + 1, r1[0]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const bool last = x.solo || __hip_atomic_fetch_add(x.flags + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1;
      if (last) {
        double num = 0.0, tr = 0.0;
        for (int b = 0; b < kRayBlocks; ++b) { num += ld1(rp, 2 * b); tr += ld1(rp, 2 * b + 1); }
        const double lam = num / tr;
        x.out[0] = lam;
        lip_write_lr(x.lr, lam);
      }
    }
  }
}

// splits of a product with contraction length len: one 256-element span per workgroup up to
// kLipMaxSplits, several spans per workgroup beyond that
static void lip_splits(int len, int* splits, int* spans_per_split) {
  const int spans = (len + kLipSpan - 1) / kLipSpan;
  *spans_per_split = (spans + kLipMaxSplits - 1) / kLipMaxSplits;
  *splits = (spans + *spans_per_split - 1) / *spans_per_split;
}

size_t lipschitz_workspace_bytes(int64_t d, int64_t k) {
  const int64_t m = d < k ? d : k;
  const int64_t mp = (m + kLipTile - 1) / kLipTile * kLipTile;
  // out[32] + G + two ping-pong P + the split partials of the product in flight
  return (size_t)((3 + kLipMaxSplits) * mp * mp + 32) * sizeof(double);
}

// Enqueue the whole computation; the result lands in ((double*)workspace)[0].
hipError_t launch_lipschitz(const float* W, int64_t ldw, int64_t d, int64_t k, void* workspace,
                            int squarings, hipStream_t stream, const PrepareJob* job, bool* fused, LipLr lr) {
  if (fused) *fused = false;
  const bool rows = d <= k;                 // G = W W^T (rows) or W^T W (columns)
  const int m = (int)(rows ? d : k), len = (int)(rows ? k : d);
  const int mp = (m + kLipTile - 1) / kLipTile * kLipTile;
  const int64_t mm = (int64_t)mp * mp;
  double* out = static_cast<double*>(workspace);
  double* G = out + 32;
  double* P[2] = {G + mm, G + 2 * mm};
  double* part = G + 3 * mm;
  int gs, gspans, ps, pspans;
  lip_splits(len, &gs, &gspans);
  lip_splits(mp, &ps, &pspans);
  const dim3 fold_grid((unsigned)((mm + 255) / 256));
  const bool span_ok = gspans == 1 && (ldw & 3) == 0 && (((uintptr_t)W) & 15) == 0 && (rows ? (len & 3) == 0 : (m & 3) == 0) &&
                       len >= 4 && m >= 4;
  if (span_ok) {                            // one span per workgroup: the low-latency kernel
    const size_t lds = (size_t)64 * 257 * sizeof(double);
    const void* fn = rows ? (const void*)&gram_span_f64_kernel<true> : (const void*)&gram_span_f64_kernel<false>;
    if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
    static const PrepareJob no_job = [] { PrepareJob j; memset(&j, 0, sizeof(j)); return j; }();
    const int per_slice = (mp / kLipTile) * (mp / kLipTile);
    const int extra = job ? (job->gx * job->gy + per_slice - 1) / per_slice : 0;     // z slices of prepare blocks
    const dim3 grid(mp / kLipTile, mp / kLipTile, gs + extra);
    if (rows) hipLaunchKernelGGL(gram_span_f64_kernel<true>, grid, dim3(256), lds, stream, W, ldw, m, len, mp, gs > 1 ? part : G, gs, job ? *job : no_job);
    else hipLaunchKernelGGL(gram_span_f64_kernel<false>, grid, dim3(256), lds, stream, W, ldw, m, len, mp, gs > 1 ? part : G, gs, job ? *job : no_job);
    if (job && fused) *fused = true;
  } else {
    hipLaunchKernelGGL((syrk_f64_kernel<float, false>), dim3(mp / kLipTile, mp / kLipTile, gs), dim3(256), 0, stream,
                       W, rows ? ldw : (int64_t)1, rows ? (int64_t)1 : ldw, m, len, mp, gspans, gs > 1 ? part : G);
  }
  int* const pflags = reinterpret_cast<int*>(out + 16);
  const bool persist = mp > 64 && mp <= 256 && squarings > 0 && gs > 1;     // (gs > 1: the fold launch clears the flags)
  if (gs > 1) hipLaunchKernelGGL(fold_partials_kernel, fold_grid, dim3(256), 0, stream, part, gs, mm, G,
                                 persist ? pflags : (int*)nullptr);
  const double* src = G;
#ifndef LASSO_LIP_MULTI_LAUNCH
  if (persist) {
    const size_t lds = (size_t)(64 * (mp + 1) + 256) * sizeof(double);
    const int nb = (mp + 63) / 64;
    const void* fn = nb == 2 ? (const void*)&square_persist_f64_kernel<2> : nb == 3 ? (const void*)&square_persist_f64_kernel<3>
                             : (const void*)&square_persist_f64_kernel<4>;
    if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
    LipPersist x;
    x.G = G; x.P0 = P[0]; x.P1 = P[1]; x.part = part; x.out = out; x.flags = pflags;
    x.mp = mp; x.squarings = squarings; x.solo = 0; x.run_if = nullptr; x.lr = lr;
    LipPersist y = x;
    y.solo = 1; y.run_if = pflags + 1;
    int tiles = (mp / kLipTile) * (mp / kLipTile);
    if (g_force_standby) { x.solo = 1; tiles = 1; }        // (test hook: the one-workgroup form as the first launch)
    switch (nb) {
      case 2: hipLaunchKernelGGL(square_persist_f64_kernel<2>, dim3(tiles), dim3(256), lds, stream, x);
              hipLaunchKernelGGL(square_persist_f64_kernel<2>, dim3(1), dim3(256), lds, stream, y); break;
      case 3: hipLaunchKernelGGL(square_persist_f64_kernel<3>, dim3(tiles), dim3(256), lds, stream, x);
              hipLaunchKernelGGL(square_persist_f64_kernel<3>, dim3(1), dim3(256), lds, stream, y); break;
      default: hipLaunchKernelGGL(square_persist_f64_kernel<4>, dim3(tiles), dim3(256), lds, stream, x);
               hipLaunchKernelGGL(square_persist_f64_kernel<4>, dim3(1), dim3(256), lds, stream, y); break;
    }
    return hipGetLastError();
  }
#endif
  if (mp <= 64 && squarings > 0) {         // small iterate: every squaring in one launch of one workgroup
    if (mp == 64) hipLaunchKernelGGL(square_chain_f64_kernel<64>, dim3(1), dim3(1024), 0, stream, G, squarings, P[0], out, lr);
    else hipLaunchKernelGGL(square_chain_f64_kernel<32>, dim3(1), dim3(1024), 0, stream, G, squarings, P[0], out, lr);
    return hipGetLastError();
  }
  if (mp <= 256) {                         // the usual case (d or k <= 256): the low-latency squaring kernel
    const size_t lds = (size_t)(64 * (mp + 1) + 256) * sizeof(double);
    const int nb = (mp + 63) / 64;
    const void* fn = nb == 1 ? (const void*)&square_f64_kernel<1> : nb == 2 ? (const void*)&square_f64_kernel<2>
                   : nb == 3 ? (const void*)&square_f64_kernel<3> : (const void*)&square_f64_kernel<4>;
    if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
    const dim3 grid(mp / kLipTile, mp / kLipTile);
    for (int p = 0; p < squarings; ++p) {
      double* const dst = P[p & 1];
      switch (nb) {
        case 1: hipLaunchKernelGGL(square_f64_kernel<1>, grid, dim3(256), lds, stream, src, mp, dst, p > 0 ? 1 : 0); break;
        case 2: hipLaunchKernelGGL(square_f64_kernel<2>, grid, dim3(256), lds, stream, src, mp, dst, p > 0 ? 1 : 0); break;
        case 3: hipLaunchKernelGGL(square_f64_kernel<3>, grid, dim3(256), lds, stream, src, mp, dst, p > 0 ? 1 : 0); break;
        default: hipLaunchKernelGGL(square_f64_kernel<4>, grid, dim3(256), lds, stream, src, mp, dst, p > 0 ? 1 : 0); break;
      }
      src = dst;
    }
    launch_rayleigh(G, src, mp, out, part, stream, lr);
    return hipGetLastError();
  }
  for (int p = 0; p < squarings; ++p) {
    hipLaunchKernelGGL((syrk_f64_kernel<double, true>), dim3(mp / kLipTile, mp / kLipTile, ps), dim3(256), 0,
                       stream, src, (int64_t)mp, (int64_t)1, mp, mp, mp, pspans, ps > 1 ? part : P[p & 1]);
    if (ps > 1) hipLaunchKernelGGL(fold_partials_kernel, fold_grid, dim3(256), 0, stream, part, ps, mm, P[p & 1]);
    src = P[p & 1];
  }
  launch_rayleigh(G, src, mp, out, part, stream, lr);
  return hipGetLastError();
}

}  // namespace lasso
