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
#include <type_traits>
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

// ---------------------------------------------------------------------------
// The same product on 128 x 128 output blocks (the 16-byte-aligned case with at least one
// full block): 4 waves x (64 x 64) = 16 accumulators per wave, so a k-step of 4 samples is
// 4 + 4 LDS operand reads for 16 MFMAs (the 64 x 64 kernel above: 2 + 2 for 4); samples in
// chunks of 32 through a double-buffered LDS stage (row stride 144 floats: the operand read
// "16 consecutive floats of 4 consecutive rows" is conflict-free per half-wave), the next
// chunk's global loads in flight in registers meanwhile -- one barrier per chunk.
// sym: only the blocks bj >= bi exist (blockIdx.x enumerates them); the mirror image is written
// by sum_splits_sym_kernel, which also folds the sample splits.
// ---------------------------------------------------------------------------
// barrier that orders LDS traffic only: __syncthreads() would also drain the global loads in flight
#define LASSO_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
constexpr int kG2B = 128, kG2S = 32, kG2Ld = 144;

// Gram2 (round 6): a SECOND right operand whose blocks ride in the same launch -- B = P^T Q2 behind the sym product
// A = P^T P (the M-step's two products of a small dictionary: d <= 128 is one column block, half of it padding for
// d = 64 -- still cheaper than a launch of its own on the EM step's chain).  Its partials go to C2 (pitch ldc2).
struct Gram2 { const float* Q2; int64_t ldq2; int qc2; float* C2; int64_t ldc2; int64_t split_stride2; int first;
               int* raise = nullptr; int raise_value = 0; };   // (raise, nullable: *raise = raise_value as the launch starts)
__global__ __launch_bounds__(256, 2) void gram_tn128_kernel(const float* __restrict__ P, int64_t ldp, int pc,
                                                            const float* __restrict__ Q, int64_t ldq, int qc, int n,
                                                            float* __restrict__ C, int64_t ldc, int sym,
                                                            int rows_per_split, int64_t split_stride, const Gram2 g2) {
  if (g2.raise && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    __hip_atomic_store(g2.raise, g2.raise_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int bi, bj;
  if (g2.Q2 && (int)blockIdx.x >= g2.first) {  // a block of the second product: row block bi of P, column block bj of Q2
    const int e = blockIdx.x - g2.first, nbq = (g2.qc2 + kG2B - 1) / kG2B;
    bi = e / nbq;
    bj = e - bi * nbq;
    Q = g2.Q2; ldq = g2.ldq2; qc = g2.qc2; C = g2.C2; ldc = g2.ldc2; split_stride = g2.split_stride2;
  } else if (sym) {                            // blockIdx.x -> (bi, bj), bj >= bi, row by row
    const int nb = (pc + kG2B - 1) / kG2B;
    int rem = blockIdx.x;
    bi = 0;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    bj = bi + rem;
  } else {
    const int nbq = (qc + kG2B - 1) / kG2B;
    bi = blockIdx.x / nbq;
    bj = blockIdx.x - bi * nbq;
  }
  const int n_lo = blockIdx.z * rows_per_split;
  P += (int64_t)n_lo * ldp;
  Q += (int64_t)n_lo * ldq;
  n = min(rows_per_split, n - n_lo);
  C += (int64_t)blockIdx.z * split_stride;
  extern __shared__ __attribute__((aligned(16))) float g2_smem[];
  float* const sp = g2_smem;                               // [2][32][144]
  float* const sq = g2_smem + 2 * kG2S * kG2Ld;            // [2][32][144]
  const int i0 = bi * kG2B, j0 = bj * kG2B;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, l15 = lane & 15, q = lane >> 4;
  f32x4 acc[4][4] = {};
  // staging map: thread -> sample row srow + 8 h (h = 0..3), 4 consecutive columns of each operand
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  f32x4 stg[2][4];
  const bool pin = i0 + scol < pc, qin = j0 + scol < qc;   // columns are multiples of 4: whole groups
  auto load_chunk = [&](int s0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = min(s0 + srow + 8 * h, n - 1);
      const f32x4 vp = *(const f32x4*)(P + (int64_t)r * ldp + (pin ? i0 + scol : 0));
      const f32x4 vq = *(const f32x4*)(Q + (int64_t)r * ldq + (qin ? j0 + scol : 0));
      const float mp = (pin && s0 + srow + 8 * h < n) ? 1.0f : 0.0f, mq = (qin && s0 + srow + 8 * h < n) ? 1.0f : 0.0f;
      stg[0][h] = vp * mp;
      stg[1][h] = vq * mq;
    }
  };
  auto put_chunk = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      *(f32x4*)(sp + (buf * kG2S + srow + 8 * h) * kG2Ld + scol) = stg[0][h];
      *(f32x4*)(sq + (buf * kG2S + srow + 8 * h) * kG2Ld + scol) = stg[1][h];
    }
  };
  if (n > 0) {
    load_chunk(0);
    put_chunk(0);
    if (kG2S < n) load_chunk(kG2S);
    LASSO_LDS_BARRIER();
    int buf = 0;
    for (int s0 = 0; s0 < n; s0 += kG2S, buf ^= 1) {
      const float* const ap = sp + buf * kG2S * kG2Ld + 64 * wr + l15;
      const float* const bp = sq + buf * kG2S * kG2Ld + 64 * wc + l15;
#pragma unroll
      for (int ks = 0; ks < kG2S / 4; ++ks) {
        float a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = ap[(4 * ks + q) * kG2Ld + 16 * m];
#pragma unroll
        for (int m = 0; m < 4; ++m) b[m] = bp[(4 * ks + q) * kG2Ld + 16 * m];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
      }
      if (s0 + kG2S < n) {
        put_chunk(buf ^ 1);                                // the other buffer: its last readers passed the barrier below
        if (s0 + 2 * kG2S < n) load_chunk(s0 + 2 * kG2S);
      }
      LASSO_LDS_BARRIER();
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = i0 + 64 * wr + 16 * mi + 4 * q + rg, cc = j0 + 64 * wc + 16 * nj + l15;
        if (r < pc && cc < qc) C[(int64_t)r * ldc + cc] = acc[mi][nj][rg];
      }
}

// sum_s base[s * stride], s = 0 .. splits-1, added in that order; the loads of kFoldBatch terms are in flight together
// (with many splits -- small dictionaries, shards -- a load per dependent add was a memory round trip per split;
// round 4: 16 instead of 8 -- 64 splits of a shard's Gram product are four round trips, not eight)
#ifndef LASSO_FOLD_BATCH
#define LASSO_FOLD_BATCH 16
#endif
constexpr int kFoldBatch = LASSO_FOLD_BATCH;
__device__ __forceinline__ float ordered_split_sum(const float* __restrict__ base, int64_t stride, int splits) {
  float acc = 0.0f;
  for (int s0 = 0; s0 < splits; s0 += kFoldBatch) {
    float v[kFoldBatch];
#pragma unroll
    for (int u = 0; u < kFoldBatch; ++u) v[u] = base[(int64_t)min(s0 + u, splits - 1) * stride];
#pragma unroll
    for (int u = 0; u < kFoldBatch; ++u)
      if (s0 + u < splits) acc += v[u];
  }
  return acc;
}

// sym product of gram_tn128_kernel: C = sum of the splits' upper 128-blocks, mirrored.  One workgroup
// per 32 x 32 tile (tr, tc), tc >= tr, of the upper triangle: coalesced reads, the transposed copy
// through LDS.  (Tiles below the diagonal inside a diagonal 128-block are computed values too, but
// bitwise equal to their mirror image -- the same products in the same order.)
__device__ __forceinline__ void sum_splits_sym_body(int block, const float* __restrict__ part, int splits,
                                                    int64_t split_stride, int64_t ldpart, int pc,
                                                    float* __restrict__ C, int64_t ldc) {
  __shared__ float t[32][33];
  const int nt = (pc + 31) / 32;
  int rem = block, tr = 0;
  while (rem >= nt - tr) { rem -= nt - tr; ++tr; }
  const int tc = tr + rem;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // this thread's four elements (rows ty + 8 a), each summed over the splits in order; 4 x kFoldBatch loads in flight
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int cc = min(32 * tc + tx, pc - 1);
  for (int s0 = 0; s0 < splits; s0 += kFoldBatch) {
    float v[4][kFoldBatch];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int u = 0; u < kFoldBatch; ++u)
        v[a][u] = part[(int64_t)min(s0 + u, splits - 1) * split_stride + (int64_t)min(32 * tr + ty + 8 * a, pc - 1) * ldpart + cc];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int u = 0; u < kFoldBatch; ++u)
        if (s0 + u < splits) acc[a] += v[a][u];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = ty + 8 * a, r = 32 * tr + i, c = 32 * tc + tx;
    const bool ok = r < pc && c < pc;
    t[i][tx] = ok ? acc[a] : 0.0f;
    if (ok) C[(int64_t)r * ldc + c] = acc[a];
  }
  __syncthreads();
  if (tr != tc)
    for (int i = ty; i < 32; i += 8) {
      const int r = 32 * tc + i, c = 32 * tr + tx;          // the mirror tile
      if (r < pc && c < pc) C[(int64_t)r * ldc + c] = t[tx][i];
    }
}

// The same fold with one element per thread -- four workgroups per 32 x 32 tile (rows ty + 8 a of the tile for
// workgroup a), the mirror image written directly: a k = 256 dictionary has only 36 tiles, and 36 workgroups cannot
// pull the splits' 12 MB in fast enough (round 4; used while the tiles are few).  Same sums, same order.
__global__ __launch_bounds__(256) void sum_splits_sym4_kernel(const float* __restrict__ part, int splits,
                                                              int64_t split_stride, int64_t ldpart, int pc,
                                                              float* __restrict__ C, int64_t ldc) {
  const int nt = (pc + 31) / 32;
  int rem = blockIdx.x >> 2, tr = 0;
  while (rem >= nt - tr) { rem -= nt - tr; ++tr; }
  const int tc = tr + rem, a = blockIdx.x & 3;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r = 32 * tr + ty + 8 * a, c = 32 * tc + tx;
  const float v = ordered_split_sum(part + (int64_t)min(r, pc - 1) * ldpart + min(c, pc - 1), split_stride, splits);
  if (r < pc && c < pc) {
    C[(int64_t)r * ldc + c] = v;
    if (tr != tc) C[(int64_t)c * ldc + r] = v;
  }
}

// sum_splits_sym4_kernel on A and the element-wise fold of a second product B in ONE launch (small dictionaries: the two
// folds of an EM step's Gram products; the same sums in the same order as the separate kernels)
__global__ __launch_bounds__(256) void sum_splits_sym4_b_kernel(const float* __restrict__ part, int splits,
                                                                int64_t split_stride, int64_t ldpart, int pc,
                                                                float* __restrict__ C, int64_t ldc, int nsym4,
                                                                const float* __restrict__ part2, int64_t split_stride2,
                                                                int rows2, int cols2, float* __restrict__ C2, int64_t ldc2) {
  if ((int)blockIdx.x >= nsym4) {
    const int64_t idx = (int64_t)(blockIdx.x - nsym4) * 256 + threadIdx.x;
    if (idx >= (int64_t)rows2 * cols2) return;
    C2[(idx / cols2) * ldc2 + idx % cols2] = ordered_split_sum(part2 + idx, split_stride2, splits);
    return;
  }
  const int nt = (pc + 31) / 32;
  int rem = blockIdx.x >> 2, tr = 0;
  while (rem >= nt - tr) { rem -= nt - tr; ++tr; }
  const int tc = tr + rem, a = blockIdx.x & 3;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r = 32 * tr + ty + 8 * a, c = 32 * tc + tx;
  const float v = ordered_split_sum(part + (int64_t)min(r, pc - 1) * ldpart + min(c, pc - 1), split_stride, splits);
  if (r < pc && c < pc) {
    C[(int64_t)r * ldc + c] = v;
    if (tr != tc) C[(int64_t)c * ldc + r] = v;
  }
}

__global__ __launch_bounds__(256) void sum_splits_sym_kernel(const float* __restrict__ part, int splits,
                                                             int64_t split_stride, int64_t ldpart, int pc,
                                                             float* __restrict__ C, int64_t ldc) {
  sum_splits_sym_body(blockIdx.x, part, splits, split_stride, ldpart, pc, C, ldc);
}

// ---------------------------------------------------------------------------
// Both M-step products in ONE launch on 256 x 256 blocks:  [A | B] = Z^T [Z | X].
// At n = 65536 the 128 x 128 kernel above is HBM-bound, not MFMA-bound: Z (268 MB) does not stay
// in L2 / MALL and every block pair streams its 2 x 128 columns of all n rows -- 3.5 GB per Gram
// step.  A 256 x 256 block reads 512 columns for four times the flops (half the bytes per flop),
// and treating X as further column blocks of the right operand puts A's 10 upper blocks and B's
// k/256 x d/256 blocks into one grid that fills the chip (k = 1024, d = 256: 14 blocks x 18
// sample splits = 252 workgroups).  8 waves x (64 x 128) = 32 accumulators per wave, a k-step of 4
// samples = 4 + 8 LDS operand reads for 32 MFMAs; samples in chunks of 32, double-buffered LDS
// (row stride 272 floats: conflict-free operand reads), next chunk's loads in registers.
// Output: partial [split][k][k + d] (A's block (bi, bj) at columns 256 bj, B behind column k);
// folded by sum_splits_ab_kernel (A mirrored like sum_splits_sym_kernel, B element by element).
// ---------------------------------------------------------------------------
constexpr int kG3B = 256, kG3S = 32, kG3Ld = 272;

// GramRows (the pipelined M-step, round 6): only the blocks of ONE block row `bi` -- A's (bi, bi .. nb-1), then B's --
// with the partials of a split holding that block row alone ([256][k + d] per split), so that a row block can use
// as many sample splits as fill the chip on its own; `clear`: words the first workgroup zeroes (the single-launch
// sweep's flags: the launch sits in front of everything that sets or reads them).
struct GramRows {
  int bi; int bi_hi;                   // block rows bi .. bi_hi - 1 (bi < 0: the whole product, the partials' usual layout)
  int* clear; int nclear;
  int* ticket; int* done; int seq;     // nullable: the workgroup that finishes last writes `seq` to *done (and resets the ticket)
  int* raise = nullptr; int raise_value = 0;   // nullable: *raise = raise_value as the launch starts (lasso_gram_accumulate_signal)
};
__global__ __launch_bounds__(512) void gram_ab256_kernel(const float* __restrict__ Z, int64_t ldz, int k,
                                                         const float* __restrict__ X, int64_t ldx, int d, int n,
                                                         float* __restrict__ part, int rows_per_split, const GramRows gr) {
  const int nb = k / kG3B, nsym = nb * (nb + 1) / 2;
  if (gr.clear && blockIdx.x == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < gr.nclear; i += 512) gr.clear[i] = 0;
  if (gr.raise && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    __hip_atomic_store(gr.raise, gr.raise_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int bi, bj;                                     // bj < nb: A block (bj >= bi);  bj >= nb: B block, X columns 256 (bj - nb)
  if (gr.bi >= 0) {
    int rem = blockIdx.x;                         // block row bi has nb - bi blocks of A and d / 256 of B
    bi = gr.bi;
    while (rem >= nb - bi + d / kG3B) { rem -= nb - bi + d / kG3B; ++bi; }
    bj = bi + rem;                                // (rem >= nb - bi: the B blocks)
  } else if ((int)blockIdx.x < nsym) {
    int rem = blockIdx.x;
    bi = 0;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    bj = bi + rem;
  } else {
    const int e = blockIdx.x - nsym, nbx = d / kG3B;
    bi = e / nbx;
    bj = nb + e % nbx;
  }
  const int n_lo = blockIdx.z * rows_per_split;
  n = min(rows_per_split, n - n_lo);
  const float* const P = Z + (int64_t)n_lo * ldz + kG3B * bi;
  const float* const Q = bj < nb ? Z + (int64_t)n_lo * ldz + kG3B * bj : X + (int64_t)n_lo * ldx + kG3B * (bj - nb);
  const int64_t ldq = bj < nb ? ldz : ldx;
  const int64_t ldc = k + d;
  float* const C = gr.bi >= 0 ? part + ((int64_t)blockIdx.z * (gr.bi_hi - gr.bi) + (bi - gr.bi)) * kG3B * ldc + kG3B * bj
                              : part + (int64_t)blockIdx.z * k * ldc + (int64_t)(kG3B * bi) * ldc + kG3B * bj;
  extern __shared__ __attribute__((aligned(16))) float g3_smem[];
  float* const sp = g3_smem;                               // [2][32][272]
  float* const sq = g3_smem + 2 * kG3S * kG3Ld;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, l15 = lane & 15, q = lane >> 4;
  f32x4 acc[4][8] = {};
  // staging map: thread -> sample rows srow + 8 h (h = 0..3), 4 consecutive columns of each operand
  const int srow = tid >> 6, scol = (tid & 63) * 4;
  f32x4 stg[2][4];
  auto load_chunk = [&](int s0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = min(s0 + srow + 8 * h, n - 1);
      const float m = (s0 + srow + 8 * h < n) ? 1.0f : 0.0f;
      stg[0][h] = *(const f32x4*)(P + (int64_t)r * ldz + scol) * m;
      stg[1][h] = *(const f32x4*)(Q + (int64_t)r * ldq + scol) * m;
    }
  };
  auto put_chunk = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      *(f32x4*)(sp + (buf * kG3S + srow + 8 * h) * kG3Ld + scol) = stg[0][h];
      *(f32x4*)(sq + (buf * kG3S + srow + 8 * h) * kG3Ld + scol) = stg[1][h];
    }
  };
  if (n > 0) {
    load_chunk(0);
    put_chunk(0);
    if (kG3S < n) load_chunk(kG3S);
    LASSO_LDS_BARRIER();
    int buf = 0;
    for (int s0 = 0; s0 < n; s0 += kG3S, buf ^= 1) {
      const float* const ap = sp + buf * kG3S * kG3Ld + 64 * wr + l15;
      const float* const bp = sq + buf * kG3S * kG3Ld + 128 * wc + l15;
#pragma unroll
      for (int ks = 0; ks < kG3S / 4; ++ks) {
        float a[4], b[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = ap[(4 * ks + q) * kG3Ld + 16 * m];
#pragma unroll
        for (int m = 0; m < 8; ++m) b[m] = bp[(4 * ks + q) * kG3Ld + 16 * m];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 8; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
      }
      if (s0 + kG3S < n) {
        put_chunk(buf ^ 1);
        if (s0 + 2 * kG3S < n) load_chunk(s0 + 2 * kG3S);
      }
      LASSO_LDS_BARRIER();
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 8; ++nj)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        C[(int64_t)(64 * wr + 16 * mi + 4 * q + rg) * ldc + 128 * wc + 16 * nj + l15] = acc[mi][nj][rg];
  if (gr.ticket) {
    // "this launch has finished" for a launch on ANOTHER stream that should not start before (pipelined M-step: the
    // later block rows would only take CUs from this one) -- a performance hint, nothing reads data on its strength
    __syncthreads();
    if (tid == 0) {
      const int total = (int)(gridDim.x * gridDim.z);
      if (__hip_atomic_fetch_add(gr.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
        __hip_atomic_store(gr.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gr.done, gr.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// Both folds of gram_ab256_kernel's partials in one launch: workgroups [0, nsym) are sum_splits_sym_kernel's on
// A (k x k), the rest fold B (k x d, behind column k of the partials) element by element -- the same sums in
// the same order, one launch less on the EM step's dependent chain.
__global__ __launch_bounds__(256) void sum_splits_ab_kernel(const float* __restrict__ part, int splits,
                                                            int64_t split_stride, int64_t ldpart, int k, int d, int nsym,
                                                            float* __restrict__ A, float* __restrict__ B) {
  if ((int)blockIdx.x < nsym) {
    sum_splits_sym_body(blockIdx.x, part, splits, split_stride, ldpart, k, A, (int64_t)k);
    return;
  }
  const int64_t idx = (int64_t)(blockIdx.x - nsym) * 256 + threadIdx.x;
  if (idx >= (int64_t)k * d) return;
  const int64_t r = idx / d, c = idx - r * d;
  B[r * d + c] = ordered_split_sum(part + k + r * ldpart + c, split_stride, splits);
}

// Fold of ONE block row of gram_ab256_kernel's row mode into AB [k][k + d] (A | B side by side, row pitch ldab):
// one workgroup per 32 x 32 tile of the block row's columns 256 bi .. k + d; tiles of A below the diagonal (inside
// the diagonal block) are left to their mirror images, tiles of A right of it are also written transposed into the
// block rows below -- after the folds of block rows 0 .. R the rows of block R are complete.  Sums in split order.
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ part, int splits, int64_t split_stride,
                                                        int64_t ldpart, int k, int d, int bi_lo, float* __restrict__ AB,
                                                        int64_t ldab) {
  __shared__ float t[32][33];
  int bi = bi_lo, rem = blockIdx.x;                        // block row bi has 8 tile rows of (k - 256 bi + d) / 32 tiles
  while (rem >= 8 * ((k - kG3B * bi + d) / 32)) { rem -= 8 * ((k - kG3B * bi + d) / 32); ++bi; }
  const int tcols = (k - kG3B * bi + d) / 32;              // tiles per tile row
  const int ti = rem / tcols, tj = rem - ti * tcols;
  const int r0 = 32 * ti, c0 = kG3B * bi + 32 * tj;        // row inside the block row, column of [A | B]
  part += (int64_t)(bi - bi_lo) * kG3B * ldpart;           // this block row's rows inside a split's partial
  const bool isA = c0 < k;
  if (isA && c0 < kG3B * bi + r0) return;                  // below the diagonal: written as a mirror image
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < splits; s0 += kFoldBatch) {
    float v[4][kFoldBatch];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int u = 0; u < kFoldBatch; ++u)
        v[a][u] = part[(int64_t)min(s0 + u, splits - 1) * split_stride + (int64_t)(r0 + ty + 8 * a) * ldpart + c0 + tx];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int u = 0; u < kFoldBatch; ++u)
        if (s0 + u < splits) acc[a] += v[a][u];
  }
  const int gr0 = kG3B * bi + r0;                          // global row of the tile
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = ty + 8 * a;
    t[i][tx] = acc[a];
    AB[(int64_t)(gr0 + i) * ldab + c0 + tx] = acc[a];
  }
  if (!isA) return;
  __syncthreads();
  if (c0 != gr0)      // (a diagonal tile is its own mirror image: the same products in the same order on both sides)
    for (int i = ty; i < 32; i += 8) AB[(int64_t)(c0 + i) * ldab + gr0 + tx] = t[tx][i];   // the mirror tile
}

// one word, written through at agent scope: "the rows of block R of [A | U] are complete" (pipelined M-step)
__global__ void set_flag_kernel(int* flag, int value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave that returns once *word == seq (or after ~0.1 s): stream order behind it then means "after that launch of
// another stream" without an event record on the other stream (~5 us between two of its kernels).
__global__ void wait_word_kernel(const int* word, int seq, int host_memory) {
  if (host_memory) {                     // a word of pinned host memory (e.g. a verdict's "valid" word): one bus read per poll
    for (int spins = 0; spins < (1 << 16); ++spins) {
      if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == seq) return;
      __builtin_amdgcn_s_sleep(64);
    }
    return;
  }
  // (~30 s: what is waited for may sit behind the caller's collectives or a time-sliced GPU; launches behind this one
  // that could not tolerate an early return carry their own check of the word)
  for (int spins = 0; spins < kGateSpinLimit; ++spins) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq) return;      // (the words count up)
    __builtin_amdgcn_s_sleep(8);
  }
}

// U rows of one block row of the pipelined M-step:  U[256][256] = C0 - A Bm^T,  A = rows of [A | B] (K = k atoms),
// Bm = the dictionary [256 features][k], C0 = the B part of the same rows.  ONE WAVE per 16 x 16 output block, the
// operands straight from global memory as 16-byte pieces (lane (l15, q) takes K = 16 c + 4 q .. + 3 of row l15: the
// fragment gemm_nt_kernel reads from its LDS image), ONE accumulator chain over K ascending -- bitwise the product of
// gemm_nt_kernel (launch_gemm_nt_sub), without its staging barriers: the launch is latency, not throughput (256 x 256
// outputs), and it sits on the EM step's dependent chain for block row 0.  The workgroup that finishes last publishes
// the block row's flag (nullable) for the running sweep.
__global__ __launch_bounds__(64) void uprod_rows_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                        int64_t ldb, const float* __restrict__ C0, int64_t ldc0,
                                                        float* __restrict__ U, int64_t ldu, int kk, int* ticket, int* flag,
                                                        int nflags, int flag_value) {
  const int lane = threadIdx.x, l15 = lane & 15, q = lane >> 4;
  const int ti = blockIdx.x >> 4, tj = blockIdx.x & 15;          // 16 x 16 tiles of the 256 x 256 block
  const float* const ap = A + (int64_t)(16 * ti + l15) * lda + 4 * q;
  const float* const bp = Bm + (int64_t)(16 * tj + l15) * ldb + 4 * q;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int kBatch = 8;                                        // 16-float pieces per operand in flight (2 x 8 x 4 VGPRs), double buffered
  f32x4 a[2][kBatch], b[2][kBatch];
  const int npiece = kk / 16;
  auto fetch = [&](int buf, int c0) {
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int c = min(c0 + i, npiece - 1);
      a[buf][i] = *(const f32x4*)(ap + 16 * c);
      b[buf][i] = *(const f32x4*)(bp + 16 * c);
    }
  };
  fetch(0, 0);
  for (int c0 = 0; c0 < npiece; c0 += 2 * kBatch) {
    fetch(1, c0 + kBatch);
#pragma unroll
    for (int i = 0; i < kBatch; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][i][j], b[0][i][j], acc, 0, 0, 0);
    if (c0 + 2 * kBatch < npiece) fetch(0, c0 + 2 * kBatch);
    if (c0 + kBatch < npiece) {
#pragma unroll
      for (int i = 0; i < kBatch; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][i][j], b[1][i][j], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int r = 16 * ti + 4 * q + rg, c = 16 * tj + l15;
    U[(int64_t)r * ldu + c] = C0[(int64_t)r * ldc0 + c] - acc[rg];
  }
  if (flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = 0; i < nflags; ++i) __hip_atomic_store(flag + i, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// C[r][c] = sum_s part[s][r][c]  (fixed order)
__global__ __launch_bounds__(256) void sum_splits_kernel(const float* __restrict__ part, int splits,
                                                         int64_t split_stride, int rows, int cols,
                                                         float* __restrict__ C, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  C[(idx / cols) * ldc + idx % cols] = ordered_split_sum(part + idx, split_stride, splits);
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
// The same sum, same order, with the row totals combined by row_bcast:15 / row_bcast:31 (lane 63 ends up with
// (r2 + r3) + (r0 + r1)): two DPP adds and one readlane in place of four readlanes and three adds.
__device__ __forceinline__ float wave_sum_dpp_bcast(float x) {
  int v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
  v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = __float_as_int(x);
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8
  // (as assembly: the rows the mask leaves out keep x, which the builtin can only express with a zero-filled
  // temporary and a separate add; the s_nop are the VALU-write -> DPP-read wait states hipcc does not see in here)
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
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
template <bool FULL, int NW, int F>
__global__ __launch_bounds__(64 * NW) void sweep_block_kernel(const SweepParams p, int j0) {
  constexpr int JB = kSweepBlock;
  constexpr int DP = 64 * NW * F;                                  // == p.dp
  __shared__ __attribute__((aligned(16))) float sA[JB][JB];   // A[j0+a][j0+b] (symmetric)
  extern __shared__ __attribute__((aligned(16))) float sD_[];      // [JB][DP] old atoms of the block (rows of Dt)
  __shared__ float red[JB][NW];
  const int tid = threadIdx.x;
  const int fo = F * tid;                                           // first feature of this lane
  const int nb = FULL ? JB : min(JB, p.k - j0);
  // F consecutive floats at ptr (16-byte access for F == 4)
  auto ld = [](const float* ptr, float (&v)[F]) {
    if constexpr (F == 4) {
      const f32x4 t = *(const f32x4*)ptr;
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) v[f] = ptr[f];
    }
  };
  auto st = [](float* ptr, const float (&v)[F]) {
    if constexpr (F == 4) {
      *(f32x4*)ptr = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) ptr[f] = v[f];
    }
  };
  for (int e = tid; e < JB * JB; e += 64 * NW) {
    const int a = e / JB, b = e % JB;
    sA[a][b] = (a < nb && b < nb) ? p.A[(int64_t)(j0 + a) * p.lda + j0 + b] : 0.0f;
  }
  for (int a = 0; a < JB; ++a) {
    float v[F] = {};
    if (a < nb) ld(p.Dt + (int64_t)(j0 + a) * DP + fo, v);
    st(&sD_[a * DP + fo], v);
  }
  float u[JB][F];
#pragma unroll
  for (int a = 0; a < JB; ++a) {
#pragma unroll
    for (int f = 0; f < F; ++f) u[a][f] = 0.0f;
    if (a < nb) ld(p.U + (int64_t)(j0 + a) * p.ldu + fo, u[a]);   // padded cols are 0
  }
  __syncthreads();
  const float lo = p.positive ? 0.0f : -INFINITY;                 // dict_learning.py:87-88
  const float eps2 = p.eps * p.eps;
  unsigned degmask = 0;
#ifdef LASSO_ABL_NOSWEEP   // timing ablation only (results invalid): prologue + epilogue
  if (p.k < 0)
#endif
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
    float dcur[F];
    ld(&sD_[a * DP + fo], dcur);
    float v[F], ss = 0.0f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
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
    float dnew[F], delta[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { dnew[f] = v[f] * inv; delta[f] = dnew[f] - dcur[f]; }
    if (FULL || a < nb) {
      st(p.Dt + (int64_t)(j0 + a) * DP + fo, dnew);
      st(p.dD + (int64_t)a * DP + fo, delta);
      degmask |= (deg ? 1u : 0u) << a;
    } else {
      const float zero[F] = {};
      st(p.dD + (int64_t)a * DP + fo, zero);
    }
#pragma unroll
    for (int b = a + 1; b < JB; ++b) {
#pragma unroll
      for (int f = 0; f < F; ++f) u[b][f] = fmaf(-cf[b], delta[f], u[b][f]);
    }
  });
  if (tid < nb) p.degenerate[j0 + tid] = (int)((degmask >> tid) & 1u);   // one store, no per-atom branch
}

// ---------------------------------------------------------------------------
// The whole sweep in ONE launch (dp == 256): the multi-launch form above spends more than half
// of its time in the fixed cost of its 2 k/32 dependent launches (~4.8 us each, against 8 us of
// atom chain per block).  Workgroup 0 ("sweeper") walks the blocks; a "worker" workgroup owns the 32
// U rows of block r >= 2: it keeps them in MFMA accumulators, applies U_r -= A[r, b] dD_b for
// b = 0 .. r-2 as the deltas are published -- in GROUPS of 8 atoms (flags[0] counts groups), so that
// three quarters of the newest block are in when its last group arrives -- and hands the rows over
// (quad-interleaved, the layout of the sweeper's LDS tile).  The sweeper applies the newest block,
// dD_{r-1}, itself from LDS.  Inside the sweeper (round 4):
//   wave 0   the chain of block b: U rows, old atoms and coefficients in registers, no memory wait;
//            in-block rank-one updates on v_mfma_f32_4x4x1; a progress word in LDS after every 8 atoms;
//   wave 1   stages the A blocks of block b + 1, then copies the deltas of block b to global memory
//            behind the chain, a group at a time (written through, drained, then the group count; the
//            last group's drain and flag wait until after the loop-top barrier);
//   waves 2-3 stage the old atoms (from D itself when the caller's dictionary can be read in place)
//            and the U rows of block b + 1, then apply the first 24 deltas of block b to those rows
//            while the chain runs; all four waves add the last 8 at the loop top.
// Hand-offs as in fista_splitk.hip: payload written through (sc1) and drained, then a flag;
// consumers load past their L1/L2 (sc1).  Every spin is bounded: on a timeout (a workgroup is
// not resident) the grid raises flags[1] and leaves; a stand-by launch of the same kernel in
// solo mode (one workgroup doing the workers' updates itself, from the untouched U / D)
// then runs -- it returns at once otherwise.  New atoms go to DtN, never over the old ones.
// Arithmetic of a row block is the same sequence in both modes (updates b = 0 .. r-1 in order,
// each one v_mfma_f32_16x16x4_f32 chain over the 32 atoms of block b, whoever issues its k-steps).
// ---------------------------------------------------------------------------
constexpr int kSpLdB = 272;     // row stride of the [32][256] LDS tiles: MFMA B-operand reads conflict-free
constexpr int kSpLdA = 34;      // row stride of the negated off-diagonal A blocks (MFMA A operand)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The waits of the GATED sweep (pipelined M-step) for rows that another stream is still producing: that stream's work
// includes the caller's collectives (RCCL all-reduces of a stage; the first ones of a process set up connections for
// hundreds of milliseconds), so their bound is ~30 s, not the ~0.2 s that says "a co-operating workgroup is not resident".
__device__ __forceinline__ bool spin_until(const int* flag, int want, int* abort_flag, int limit = kStopSpinLimit) {
  for (int spins = 0; spins < limit; ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// acc[mt][nt] (rows 16 mt + 4 q + rg, columns 64 w + 16 nt + l15) += An[32][32] B[32][256];
// the 32 B values of the lane are fetched up front (one memory latency when they come from HBM)
template <typename BLoad>
__device__ __forceinline__ void sp_mma(f32x4 (&acc)[2][4], const float* __restrict__ sAn, BLoad&& bload, int l15, int q) {
  float b[8][4];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b[s][nt] = bload(4 * s + q, nt);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float a0 = sAn[l15 * kSpLdA + 4 * s + q], a1 = sAn[(16 + l15) * kSpLdA + 4 * s + q];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[s][nt], acc[0][nt], 0, 0, 0);
      acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[s][nt], acc[1][nt], 0, 0, 0);
    }
  }
}

// The sweeper's U rows of the current block in LDS, "quad interleaved": the four rows 4 Q .. 4 Q + 3 at
// one column are one 16-byte word -- the accumulator of a v_mfma_f32_4x4x1 (register = row of the quad,
// lane = column) and of the block-level 16x16x4 products (register = row 4 q + rg) alike.
__device__ __forceinline__ int sp_ub(int row, int col) { return (((row >> 2) * 256 + col) << 2) + (row & 3); }

// One group of a block's product: the 8 delta rows 8 g .. 8 g + 7 (k-steps 2 g, 2 g + 1 of sp_mma, same order)
template <int NT, typename BLoad>
__device__ __forceinline__ void sp_mma_group(f32x4 (&acc)[2][NT], const float* __restrict__ sAn, BLoad&& bload, int g,
                                             int l15, int q) {
  float b[2][NT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[s][nt] = bload(4 * (2 * g + s) + q, nt);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ks = 2 * g + s;
    const float a0 = sAn[l15 * kSpLdA + 4 * ks + q], a1 = sAn[(16 + l15) * kSpLdA + 4 * ks + q];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[s][nt], acc[0][nt], 0, 0, 0);
      acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[s][nt], acc[1][nt], 0, 0, 0);
    }
  }
}

struct SweepPersist {
  float* DtN;          // [k][256] new atoms
  float* dDg;          // [nblk * 32][256] published deltas
  float* Uw;           // [nblk * 32][256] worker results
  int* flags;          // [0] groups of 8 deltas published, [1] abort, [8 + r] rows of block r handed over
  int solo;            // 1: one workgroup, no workers
  int wg_stride;       // worker i is workgroup i * wg_stride (8: every worker on the sweeper's XCD)
  const int* run_if;   // nullable: run only if *run_if != 0 (the stand-by launch)
  int gate0;           // first group of rows that is NOT complete when the launch starts (see gate)
  int gate;            // > 0 (pipelined M-step): the rows of [A | U] arrive in groups of `gate` blocks while the sweep runs;
                       //   flags[kSpRowFlag + R] != 0 once the rows of blocks gate R .. gate R + gate - 1 are complete
                       //   (the groups below gate0 are complete before the launch)
};
constexpr int kSpRowFlag = 192;     // (below the debug stamps at word 256)
constexpr int kSpSelf = 1;      // newest deltas the sweeper applies itself
static_assert(kSpSelf == 1, "the LDS delta buffers are shared with the next block's old atoms");

__global__ __launch_bounds__(256) void sweep_persist_kernel(const SweepParams p, const SweepPersist x) {
  constexpr int JB = kSweepBlock, DP = 256;
  if (x.run_if && *x.run_if == 0) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nblk = (p.k + JB - 1) / JB;
  int* const f_pub = x.flags, * const f_abort = x.flags + 1, * const f_rows = x.flags + 8;
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(x.dDg, 0, nblk * JB * DP * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(x.DtN, 0, nblk * JB * DP * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(x.Uw, 0, nblk * JB * DP * 4, 0x00020000);
  // negated A[rb rows][cb columns] -> sAn (zero beyond k)
  // (every load below is unconditional on a clamped address, the mask applied to the value: a
  // load under a lane mask gets its own s_waitcnt and the batch degenerates into a chain of latencies)
  const int kl = p.k - 1;
  auto load_an = [&](float* sAn, int rb, int cb, int t0, int nt) {
    for (int e = t0; e < JB * JB; e += nt) {
      const int a = e >> 5, c = e & 31, r = JB * rb + a, cc = JB * cb + c;
      const float v = p.A[(int64_t)min(r, kl) * p.lda + min(cc, kl)];
      sAn[a * kSpLdA + c] = (r < p.k && cc < p.k) ? -v : 0.0f;
    }
  };
  const auto dd_global = [&](int b) {
    return [&, b](int kk, int nt) {
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
          drsrc, (unsigned)(((JB * b + kk) * DP + 64 * w + 16 * nt + l15) * 4), 0, 16));
    };
  };

  if (blockIdx.x > 0) {
    // ---------------- worker: rows of block r ------------------------------------------------
    if (blockIdx.x % x.wg_stride) return;
    const int r = blockIdx.x / x.wg_stride + kSpSelf;
    float* const sAn = smem;                                 // [2][32][34]
    __shared__ int sh_dead;
    if (tid == 0) sh_dead = 0;
    if (x.gate > 0 && r / x.gate >= x.gate0) {
      // pipelined M-step: this block's rows of A and U are still being produced (another stream) when the launch
      // starts -- wait for their group's word, then make this CU read them afresh
      __syncthreads();
      if (tid == 0 && !spin_until(x.flags + kSpRowFlag + r / x.gate, 1, f_abort, kGateSpinLimit)) sh_dead = 1;
      __syncthreads();
      if (sh_dead) return;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // (through a buffer descriptor, offset out of range where the element is padding, offsets opaque: behind a select
    // hipcc moved each of these 32 loads under the condition, a branch of its own with an s_waitcnt vmcnt(0) -- 32 memory
    // round trips in a row before a worker could take its first deltas; round 5)
    f32x4 acc[2][4];
    {
      const __amdgpu_buffer_rsrc_t usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, (int)((int64_t)p.k * p.ldu * 4), 0x00020000);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = JB * r + 16 * mt + 4 * q + rg, col = 64 * w + 16 * nt + l15;
            unsigned o = (row < p.k && col < p.d) ? (unsigned)(row * (int)p.ldu + col) * 4u : 0xfffffff0u;   // (the padding is never read)
            asm volatile("" : "+v"(o));
            acc[mt][nt][rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(usrc, o, 0, 0));
          }
    }
    load_an(sAn, r, 0, tid, 256);
    // The deltas arrive in groups of 8 atoms (flags[0] counts the groups): the worker of the block next in
    // line has consumed three quarters of the newest block's deltas when its last group is published.
    // Every wave follows the flag itself; the block of A for b + 1 is fetched while the groups of b arrive.
    bool alive = true;
    for (int b = 0; b <= r - kSpSelf - 1; ++b) {
      __syncthreads();                                       // sAn[b & 1] staged, sAn[(b + 1) & 1] free
      if (sh_dead) return;
      const float* const cur = sAn + (b & 1) * JB * kSpLdA;
      if (b + 1 <= r - kSpSelf - 1) load_an(sAn + ((b + 1) & 1) * JB * kSpLdA, r, b + 1, tid, 256);
      for (int g = 0; g < 4 && alive; ++g) {
        bool ok = true;
        // (gated: the sweeper may itself be waiting for a stage of rows -- its deltas then take as long as that stage)
        if (lane == 0) ok = spin_until(f_pub, 4 * b + g + 1, f_abort, x.gate > 0 ? kGateSpinLimit : kStopSpinLimit);
        alive = __builtin_amdgcn_readfirstlane((int)ok) != 0;
        if (alive) sp_mma_group<4>(acc, cur, dd_global(b), g, l15, q);
      }
      if (!alive) sh_dead = 1;
    }
    __syncthreads();
    if (sh_dead) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)                         // quad-interleaved, as the sweeper's LDS tile (sp_ub)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][nt]), ursrc,
            (unsigned)((JB * r * DP + sp_ub(16 * mt + 4 * q, 64 * w + 16 * nt + l15)) * 4), 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // written through
    __syncthreads();
    if (tid == 0) __hip_atomic_store(f_rows + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ---------------- sweeper ----------------------------------------------------------------
  float* const sA = smem;                                    // [2][32][32]     A[b][b]
  float* const sAp = sA + 2 * JB * JB;                       // [2][2][32][34]  -A[b][b-1], -A[b][b-2]
  // dDl[b & 1] holds the OLD atoms of block b until the chain replaces them, row by row, with the
  // block's deltas (atom a reads its old row, then writes its delta there); dDl[(b & 1) ^ 1] the
  // deltas of block b - 1 until wave 1 has published them, then the old atoms of block b + 1
  float* const dDl = sAp + 4 * JB * kSpLdA;                  // [2][32][272]
  float* const Ub = dDl + 2 * JB * kSpLdB;                   // [32][272]       U rows of the block
  __shared__ volatile int a_staged;                          // blocks whose A blocks wave 1 has staged
  __shared__ volatile int chain_prog;                        // atoms whose deltas the chain has written to dDl
  __shared__ volatile int rows_taken;                        // blocks whose Ub rows wave 0 holds in registers
  __shared__ int sh_abort;
  // (loads first, LDS stores after: one memory latency per staging step, not one per loop trip)
  auto stage_a = [&](int nb, int par, int t0, auto nt_c) {
    constexpr int nt = decltype(nt_c)::value, kPer = (JB * JB + nt - 1) / nt;
    float ra[kPer], rp[kSpSelf][kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = min(t0 + i * nt, JB * JB - 1), a = e >> 5, c = e & 31, r = JB * nb + a;
      const float* const row = p.A + (int64_t)min(r, kl) * p.lda;
      const float va = row[min(JB * nb + c, kl)];
      ra[i] = va * ((r < p.k && JB * nb + c < p.k) ? 1.0f : 0.0f);
#pragma unroll
      for (int h = 0; h < kSpSelf; ++h) {
        const float vp = row[max(JB * (nb - 1 - h), 0) + c];                                     // columns < k
        rp[h][i] = vp * ((r < p.k && nb >= h + 1) ? -1.0f : 0.0f);
      }
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = t0 + i * nt, a = e >> 5, c = e & 31;
      if (e < JB * JB) {
        sA[par * JB * JB + e] = ra[i];
        sAp[(2 * par + 1) * JB * kSpLdA + a * kSpLdA + c] = -ra[i];          // -A[b][b]: A operand of the chain's MFMAs
#pragma unroll
        for (int h = 0; h < kSpSelf; ++h) sAp[(2 * par + h) * JB * kSpLdA + a * kSpLdA + c] = rp[h][i];
      }
    }
  };
  // (nt == 128: thread t0 < 64 stages columns 0 .. 127, the others columns 128 .. 255 -- of every row)
  auto stage_rows = [&](int nb, bool from_worker, int t0, auto nt_c) {
    constexpr int nt = decltype(nt_c)::value, kPer = (JB * DP / 4 + nt - 1) / nt;
    f32x4 rv[kPer];
    auto unit = [&](int i) {                                 // index of the i-th 16-byte unit of this thread
      if constexpr (nt == 128) {
        const int u = (t0 & 63) + 64 * i;                    // 0 .. 1023 inside the half
        return from_worker ? ((u >> 7) * 256 + 128 * (t0 >> 6) + (u & 127)) : ((u >> 5) * 64 + 32 * (t0 >> 6) + (u & 31));
      } else {
        return t0 + i * nt;
      }
    };
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = min(unit(i), JB * DP / 4 - 1), a = e >> 6, c4 = (e & 63) * 4;
      if (from_worker) {                                     // (wave-uniform) already quad-interleaved
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(ursrc, (unsigned)((JB * nb * DP + 4 * e) * 4), 0, 16);
        rv[i] = __builtin_bit_cast(f32x4, t);
      } else {
        rv[i] = *(const f32x4*)(p.U + (int64_t)min(JB * nb + a, kl) * p.ldu + c4);
#pragma unroll
        for (int f = 0; f < 4; ++f)
          if (JB * nb + a >= p.k || c4 + f >= p.d) rv[i][f] = 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = unit(i), a = e >> 6, c4 = (e & 63) * 4;
      if (e < JB * DP / 4) {
        if (from_worker) {
          *(f32x4*)(Ub + 4 * e) = rv[i];
        } else {
#pragma unroll
          for (int f = 0; f < 4; ++f) Ub[sp_ub(a, c4 + f)] = rv[i][f];
        }
      }
    }
  };
  auto stage_old = [&](int nb, int t0, auto nt_c, bool wait_pub) {
    constexpr int nt = decltype(nt_c)::value, kPer = JB * DP / 4 / nt;
    f32x4 rv[kPer];
    if (p.Dsrc) {
      // from D[d][k]: a unit is four atoms of one feature (16 bytes of a row of D).  Neighbouring lanes take
      // neighbouring FEATURES (rows of D: the loads are not coalesced, 16 of 128 bytes per row and instruction, the
      // other seven units of the row come from the cache) so that the four LDS stores of a unit are conflict-free --
      // with neighbouring atoms in neighbouring lanes they are 8-way conflicts, and the chain next door feels them
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int u = t0 + i * nt, c = u & 255, j = JB * nb + 4 * (u >> 8);
        rv[i] = *(const f32x4*)(p.Dsrc + (int64_t)min(c, p.d - 1) * p.ldd + min(j, p.k - 4));
        if (c >= p.d || j >= p.k) rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      float* const dst = dDl + (nb & 1) * JB * kSpLdB;
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int u = t0 + i * nt, c = u & 255, a4 = 4 * (u >> 8);
#pragma unroll
        for (int f = 0; f < 4; ++f) dst[(a4 + f) * kSpLdB + c] = rv[i][f];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = t0 + i * nt, a = e >> 6, c4 = (e & 63) * 4;
      rv[i] = *(const f32x4*)(p.Dt + (int64_t)min(JB * nb + a, kl) * DP + c4);
      if (JB * nb + a >= p.k) rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    (void)wait_pub;   // (the deltas this overwrites were published while their chain ran, before the loop-top barrier)
    float* const dst = dDl + (nb & 1) * JB * kSpLdB;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = t0 + i * nt, a = e >> 6, c4 = (e & 63) * 4;
      *(f32x4*)(dst + a * kSpLdB + c4) = rv[i];
    }
  };
  if (tid == 0) { rows_taken = 0; sh_abort = 0; chain_prog = 0; a_staged = 0; }
  using I64 = std::integral_constant<int, 64>;
  using I128 = std::integral_constant<int, 128>;
  using I256 = std::integral_constant<int, 256>;
  stage_a(0, 0, tid, I256{});
  stage_rows(0, false, tid, I256{});
  stage_old(0, tid, I256{}, false);
  const float lo = p.positive ? 0.0f : -INFINITY;            // dict_learning.py:87-88
  const float eps2 = p.eps * p.eps;
  const bool solo = x.solo != 0;

#ifdef LASSO_SWEEP_TIMING
  long long* const tlog = (long long*)(x.flags + 256);       // [nblk][16] wall_clock64 stamps (debug builds)
#define SP_STAMP(slot) do { if (lane == 0) tlog[b * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define SP_STAMP(slot) do {} while (0)
#endif
  for (int b = 0; b < nblk; ++b) {
    const int par = b & 1;
    // the staging code derives ~100 per-thread addresses from the thread index; opaque per trip, or
    // hipcc hoists them all out of this loop and spills them (each reload then waits for vmcnt(0))
    int tdyn = tid;
    asm volatile("" : "+v"(tdyn));
    __syncthreads();                                         // block b staged; dDl[par ^ 1] = deltas of block b - 1
    if (sh_abort) return;
    if (w == 0) SP_STAMP(0);
    if (b > 0) {
      // U_b -= A[b][b'] dD_b', b' ascending: (solo mode) every b' < b - 2 from the published deltas,
      // then the two newest ones from LDS
      f32x4 acc[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)     // rows 16 mt + 4 q + (0..3): one quad of the interleaved tile
          acc[mt][nt] = *(const f32x4*)(Ub + sp_ub(16 * mt + 4 * q, 64 * w + 16 * nt + l15));
      if (solo) {
        float* const sAn = sAp + (2 * (par ^ 1)) * JB * kSpLdA;   // free until block b + 1 is staged
        for (int bb = 0; bb <= b - kSpSelf - 1; ++bb) {
          __syncthreads();
          load_an(sAn, b, bb, tid, 256);
          __syncthreads();
          sp_mma(acc, sAn, dd_global(bb), l15, q);
        }
        __syncthreads();
      }
      const auto dd_lds = [&](int kk, int nt) { return dDl[((par ^ 1) * JB + kk) * kSpLdB + 64 * w + 16 * nt + l15]; };
      if (solo) sp_mma(acc, sAp + (2 * par) * JB * kSpLdA, dd_lds, l15, q);
      else sp_mma_group<4>(acc, sAp + (2 * par) * JB * kSpLdA, dd_lds, 3, l15, q);   // (waves 2-3 did groups 0 .. 2)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          *(f32x4*)(Ub + sp_ub(16 * mt + 4 * q, 64 * w + 16 * nt + l15)) = acc[mt][nt];
      __syncthreads();
    }
    if (w == 0) {
      SP_STAMP(1);
      // ---- the chain of block b: rows and old atoms in registers.  Lane l holds columns F l .. F l + F - 1 of
      // all 32 rows (F = 4; 2 or 1 when d <= 128 / 64: the columns beyond are zero and stay zero, nobody has to
      // carry them through the chain): u[Q][c] = rows 4 Q .. 4 Q + 3 at column F l + c
      const int j0 = JB * b, nb_at = min(JB, p.k - j0);
      const float* const cA = sA + par * JB * JB;
      float* const dOut = dDl + par * JB * kSpLdB;
      const float* const cAn = sAp + (2 * par + 1) * JB * kSpLdA + (lane & 3);
      unsigned degmask = 0;
      // The rank-one updates of the later rows, u_b -= A[b][a] delta_a, run on the matrix pipe beside the
      // reduction chain: one v_mfma_f32_4x4x1 (16 blocks: rows of a quad x 4 lanes' columns, k = 1 -- a plain
      // fused multiply-add per element) updates a quad of rows at 64 columns; its A operand is the quad's
      // -A[a][4 Q + (lane & 3)], the same for every block.  Rows <= a of the quad that holds atom a are
      // dead by then, whatever lands in them.  The LDS operands of atom a + 1 are fetched before atom a
      // writes its delta (hipcc cannot move a load across that store): no LDS latency on the chain.
      auto chain_wave = [&](auto f_c) {
        constexpr int F = decltype(f_c)::value;
        const int fo = F * lane;
        auto ldF = [](const float* ptr, float (&v)[F]) {
          if constexpr (F == 4) { const f32x4 t = *(const f32x4*)ptr; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
          else if constexpr (F == 2) { const f32x2 t = *(const f32x2*)ptr; v[0] = t[0]; v[1] = t[1]; }
          else v[0] = *ptr;
        };
        auto stF = [](float* ptr, const float (&v)[F]) {
          if constexpr (F == 4) *(f32x4*)ptr = (f32x4){v[0], v[1], v[2], v[3]};
          else if constexpr (F == 2) *(f32x2*)ptr = (f32x2){v[0], v[1]};
          else *ptr = v[0];
        };
        f32x4 u[JB / 4][F];
#pragma unroll
        for (int Q = 0; Q < JB / 4; ++Q)
#pragma unroll
          for (int c = 0; c < F; ++c) u[Q][c] = *(const f32x4*)(Ub + sp_ub(4 * Q, fo + c));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (also keeps hipcc from sinking the loads past the flag)
        rows_taken = b + 1;
        float cqn[JB / 4], caan, dcn[F];
        auto fetch = [&](auto a_c) {
          constexpr int a = decltype(a_c)::value;
#pragma unroll
          for (int Q = (a + 1) / 4; Q < JB / 4; ++Q) cqn[Q] = cAn[a * kSpLdA + 4 * Q];
          caan = cA[a * JB + a];
          ldF(dOut + a * kSpLdB + fo, dcn);                  // the old atom; its delta goes back here
        };
        fetch(std::integral_constant<int, 0>{});
        auto chain = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;     // all 32 atoms exist: no per-atom branch
          static_for<JB>([&](auto a_c) {
            constexpr int a = decltype(a_c)::value, Q0 = a / 4, r0 = a % 4, Qs = (a + 1) / 4;
            float cq[JB / 4];
#pragma unroll
            for (int Q = Qs; Q < JB / 4; ++Q) cq[Q] = cqn[Q];
            const float caa = caan;
            float dc[F];
#pragma unroll
            for (int f = 0; f < F; ++f) dc[f] = dcn[f];
            if constexpr (a + 1 < JB) fetch(std::integral_constant<int, a + 1>{});
            float v[F], ss = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) {
              v[f] = fmaxf(fmaf(caa, dc[f], u[Q0][f][r0]), lo);            // u_j = U_j + A_jj d_j   (:85-88)
              ss = fmaf(v[f], v[f], ss);
            }
#ifndef LASSO_ABL_NORED      // (timing ablations: results invalid)
#ifdef LASSO_SWEEP_RED_READLANE
            ss = wave_sum_dpp(ss);
#else
            ss = wave_sum_dpp_bcast(ss);
#endif
#endif
            const bool deg = ss < eps2;
#ifdef LASSO_ABL_NORSQ
            const float inv = deg ? 0.0f : ss * 0.001f;
#else
            const float inv = deg ? 0.0f : __builtin_amdgcn_rsqf(ss);
#endif
            float dnew[F], delta[F];
#pragma unroll
            for (int f = 0; f < F; ++f) {
              dnew[f] = v[f] * inv;
              delta[f] = (!FULL && a >= nb_at) ? 0.0f : dnew[f] - dc[f];
            }
            if (FULL || a < nb_at) {
#ifndef LASSO_ABL_NOSTORE
              const unsigned so = (unsigned)((j0 + a) * DP * 4);
              if constexpr (F == 4)
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(dnew[0]), __float_as_uint(dnew[1]),
                                                               __float_as_uint(dnew[2]), __float_as_uint(dnew[3])},
                                                       nrsrc, (unsigned)(fo * 4), so, 0);
              else if constexpr (F == 2)
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(dnew[0]), __float_as_uint(dnew[1])},
                                                      nrsrc, (unsigned)(fo * 4), so, 0);
              else
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dnew[0]), nrsrc, (unsigned)(fo * 4), so, 0);
#endif
              degmask |= (deg ? 1u : 0u) << a;
            }
            stF(dOut + a * kSpLdB + fo, delta);
            if constexpr (a % 8 == 7) chain_prog = j0 + a + 1;   // (LDS keeps a wave's writes in order: no wait)
#ifdef LASSO_ABL_NOUPD
            constexpr int Qe = Qs + 1 < JB / 4 ? Qs + 1 : JB / 4;     // only the quad of the next atom
#else
            constexpr int Qe = JB / 4;
#endif
#pragma unroll
            for (int Q = Qs; Q < Qe; ++Q)
#pragma unroll
              for (int c = 0; c < F; ++c)
                u[Q][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(cq[Q], delta[c], u[Q][c], 0, 0, 0);
          });
        };
        if (nb_at == JB) chain(std::true_type{});
        else chain(std::false_type{});
      };
      if (p.d <= 64) chain_wave(std::integral_constant<int, 1>{});
      else if (p.d <= 128) chain_wave(std::integral_constant<int, 2>{});
      else chain_wave(std::integral_constant<int, 4>{});
      if (lane < nb_at) p.degenerate[j0 + lane] = (int)((degmask >> lane) & 1u);
      SP_STAMP(2);
    } else if (w == 1) {
      // ---- wave 1: stage the A blocks of block b + 1, then publish the deltas of THIS block behind the chain,
      // eight atoms at a time (LDS -> global, written through, drained, then the group count)
      const int nb = b + 1;
      SP_STAMP(4);
      if (b > 0) {       // the last group of block b - 1: its stores were issued before the loop-top barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!solo && lane == 0) __hip_atomic_store(f_pub, 4 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (nb < nblk) {
        if (x.gate > 0 && nb % x.gate == 0 && nb / x.gate >= x.gate0) {   // first block of a group of rows that arrive while we run
          bool ok = true;
          if (lane == 0) ok = spin_until(x.flags + kSpRowFlag + nb / x.gate, 1, f_abort, kGateSpinLimit);
          if (__builtin_amdgcn_readfirstlane((int)ok) == 0) sh_abort = 1;   // (a_staged still advances: nobody hangs)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        stage_a(nb, par ^ 1, tdyn & 63, I64{});
        a_staged = nb;                                       // (behind the tile's writes: LDS keeps a wave's order)
      }
      SP_STAMP(6);
      const float* const src = dDl + par * JB * kSpLdB;
      for (int g = 0; g < 4; ++g) {
        while (chain_prog < JB * b + 8 * (g + 1)) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int a = 8 * g + i, c4 = (tdyn & 63) * 4;
          const f32x4 v = *(const f32x4*)(src + a * kSpLdB + c4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), drsrc,
                                                 (unsigned)(((JB * b + a) * DP + c4) * 4), 0, 16);
        }
        if (g == 3) break;   // (nobody in this workgroup waits for these stores: drained and flagged in the next trip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // written through
        if (!solo && lane == 0) __hip_atomic_store(f_pub, 4 * b + g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      SP_STAMP(5);
    } else {
      // ---- waves 2-3: old atoms and U rows of block b + 1 ---------------------------------------
      const int ht = tdyn - 128, nb = b + 1;
      if (nb < nblk) {
        stage_old(nb, ht, I128{}, true);
        while (rows_taken < nb) __builtin_amdgcn_s_sleep(1); // wave 0 holds the rows of block b in registers
        if (w == 2) SP_STAMP(7);
        bool ok = true;
        const bool from_worker = !solo && nb > kSpSelf;
        if (!from_worker && x.gate > 0 && nb / x.gate >= x.gate0) {
          // (gated solo sweep: U rows straight from the product -- wave 1 has seen their group's word)
          while (a_staged < nb) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (sh_abort) ok = false;
        }
        if (from_worker) {
          if (lane == 0) ok = spin_until(f_rows + nb, 1, f_abort);
          ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
        }
        if (w == 2) SP_STAMP(8);
        if (!ok) sh_abort = 1;
        else stage_rows(nb, from_worker, ht, I128{});
        if (w == 2) SP_STAMP(9);
        if (ok && !solo) {
          // U_{b+1} -= A[b+1][b] dD_b for the first 24 atoms of THIS block, behind the chain (the half of the
          // columns this wave has just staged); the last 8 atoms' share is what is left at the loop top
          const int h = w - 2;
          f32x4 acc2[2][8];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
              acc2[mt][nt] = *(const f32x4*)(Ub + sp_ub(16 * mt + 4 * q, 128 * h + 16 * nt + l15));
          while (a_staged < nb) __builtin_amdgcn_s_sleep(1);
          for (int g = 0; g < 3; ++g) {
            while (chain_prog < JB * b + 8 * (g + 1)) __builtin_amdgcn_s_sleep(1);
            sp_mma_group<8>(acc2, sAp + (2 * (par ^ 1)) * JB * kSpLdA,
                            [&](int kk, int nt) { return dDl[(par * JB + kk) * kSpLdB + 128 * h + 16 * nt + l15]; },
                            g, l15, q);
          }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
              *(f32x4*)(Ub + sp_ub(16 * mt + 4 * q, 128 * h + 16 * nt + l15)) = acc2[mt][nt];
          if (w == 2) SP_STAMP(10);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small dictionaries (d <= 64, k <= 256: the 8 x 8 patches of BASELINE config 5): the whole sweep in ONE workgroup
// with every U row in LDS (k x 64 floats) -- no workers, no hand-offs through memory, no stand-by, no repair launch.
// Wave 0 runs the chain of block b (one column per lane; the in-block rank-one updates on v_mfma_f32_4x4x1 as in
// sweep_persist_kernel).  Waves 1-3 own the 32-row blocks r = 1, 2, ... round robin and keep them up to date behind
// the chain: while the chain of block b runs they apply the finished deltas of block b - 1 to every row block
// r > b (U_r -= A[r][b-1] dD_{b-1}, 64 v_mfma_f32_16x16x4 per product, the A operands straight from global memory,
// fetched one product ahead) and the first 24 deltas of block b to row block b + 1 as the chain publishes them; the
// last 8 are applied by all four waves at the loop top.  They also stage the next block's old atoms and A[b][b] and
// write the previous block's new atoms into the dictionary (in place: those columns were read a block earlier).
// Per row block the updates arrive in the order of the other forms of the sweep (blocks ascending, atoms ascending
// in groups of four on the MFMA) and the chain is sweep_persist_kernel's: bitwise the same dictionary.
// Degenerate atoms are repaired at the end as degenerate_fixup_kernel does (same reduction order).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSsLdD = 80;      // row stride of the [32][64] delta tiles: MFMA B-operand reads conflict-free
constexpr int kSsMaxBlk = 8;    // k <= 256
__device__ __forceinline__ int ss_ub(int row, int col) { return (((row >> 2) * 64 + col) << 2) + (row & 3); }

__global__ __launch_bounds__(256) void sweep_small_kernel(const SweepParams p) {
  constexpr int JB = kSweepBlock;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nblk = (p.k + JB - 1) / JB, kl = p.k - 1;
  float* const Ur = smem;                                   // [nblk][8][64][4]  U rows, quad interleaved per block
  float* const sA = Ur + nblk * JB * 64;                    // [2][32][32]       A[b][b]
  float* const sAn = sA + 2 * JB * JB;                      // [2][32][34]      -A[b][b]
  float* const dD = sAn + 2 * JB * kSpLdA;                  // [2][32][80]       deltas of block b
  float* const oldA = dD + 2 * JB * kSsLdD;                 // [2][32][64]       old atoms of block b
  __shared__ volatile int chain_prog;
  __shared__ float sh[256];
  __shared__ int s_idx[JB * kSsMaxBlk], s_deg[JB * kSsMaxBlk];
  __shared__ int s_count;
  const float lo = p.positive ? 0.0f : -INFINITY;
  const float eps2 = p.eps * p.eps;
  using I1 = std::integral_constant<int, 1>;
  using I4 = std::integral_constant<int, 4>;

  // -A[r][bb] operands of one product for this lane: rows 32 r + l15 (+16), columns 32 bb + 4 ks + q.  Through a
  // buffer descriptor: one multiply per product, the column block as a scalar offset, rows >= k read as zero (beyond
  // the descriptor); bb < r, so the 32 columns always exist.  (With 64-bit address arithmetic per element, requesting
  // the operands cost a helper wave more issue time than the products they feed.)
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((int64_t)p.k * p.lda * 4), 0x00020000);
  auto load_a = [&](int r, int bb, float (&a0)[8], float (&a1)[8]) {
    const unsigned v0 = (unsigned)(((JB * r + l15) * (int)p.lda + q) * 4), v1 = v0 + (unsigned)(16 * (int)p.lda * 4);
    const unsigned so = (unsigned)(JB * bb * 4);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      a0[ks] = -__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(arsrc, v0, so + 16 * ks, 0));
      a1[ks] = -__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(arsrc, v1, so + 16 * ks, 0));
    }
  };
  // U_r (columns 16 nt0 .. of NT 16-column tiles) += (-A[r][bb]) dD_bb over the k-steps KS0 .. KS0 + NKS - 1
  auto product = [&](float* ur, const float* dd, const float (&a0)[8], const float (&a1)[8], int ks0, int nt0,
                     auto nt_c, auto nks_c) {
    constexpr int NT = decltype(nt_c)::value, NKS = decltype(nks_c)::value;
    f32x4 acc[2][NT];
    float bv[NKS][NT];
#pragma unroll
    for (int s2 = 0; s2 < NKS; ++s2)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[s2][nt] = dd[(4 * (ks0 + s2) + q) * kSsLdD + 16 * (nt0 + nt) + l15];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = *(const f32x4*)(ur + ss_ub(16 * mt + 4 * q, 16 * (nt0 + nt) + l15));
#pragma unroll
    for (int s2 = 0; s2 < NKS; ++s2) {
      float x0 = a0[0], x1 = a1[0];                         // a0[ks0 + s2] without a dynamically indexed register array
#pragma unroll
      for (int i = 1; i < 8; ++i) { x0 = (ks0 + s2 == i) ? a0[i] : x0; x1 = (ks0 + s2 == i) ? a1[i] : x1; }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, bv[s2][nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, bv[s2][nt], acc[1][nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *(f32x4*)(ur + ss_ub(16 * mt + 4 * q, 16 * (nt0 + nt) + l15)) = acc[mt][nt];
  };
  using K2 = std::integral_constant<int, 2>;
  using K8 = std::integral_constant<int, 8>;
  auto owner = [](int r) { return 1 + r % 3; };             // the helper wave that keeps row block r up to date

  // ---- everything in: U rows (columns >= d and rows >= k are zero), block 0's old atoms and A[0][0]
  {
    // lane = column, wave = quads {w, w + 4} of every block: four coalesced row loads make one conflict-free 16-byte
    // LDS store (the row-major mapping -- 16-byte loads, four scalar stores each -- was an 8-way bank conflict)
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(p.U, 0, (int)((int64_t)p.k * p.ldu * 4), 0x00020000);
    const int col = tid & 63;
    const bool colok = col < p.d;
    f32x4 rv[kSsMaxBlk][2];                                 // (all loads first: one memory latency, not one per block)
#pragma unroll
    for (int blk = 0; blk < kSsMaxBlk; ++blk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                       // rows >= k (and blocks >= nblk): beyond the descriptor, zero
          const unsigned off = (unsigned)(((JB * blk + 4 * (w + 4 * j) + r) * (int)p.ldu + col) * 4);
          rv[blk][j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ursrc, off, 0, 0));
        }
#pragma unroll
    for (int blk = 0; blk < kSsMaxBlk; ++blk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (blk < nblk)
          *(f32x4*)(Ur + blk * JB * 64 + ss_ub(4 * (w + 4 * j), col)) = colok ? rv[blk][j] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  {
    float rv[4];                                            // A[0][0] and its negation
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, a = e >> 5, c = e & 31;
      const float v = p.A[(int64_t)min(a, kl) * p.lda + min(c, kl)];
      rv[i] = (a >= p.k || c >= p.k) ? 0.0f : v;
    }
    f32x4 ro[2];                                            // columns 0 .. 31 of D -> rows of oldA[0] (zero padded)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, c = u & 63, jj = 4 * (u >> 6);
      ro[i] = *(const f32x4*)(p.Dsrc + (int64_t)min(c, p.d - 1) * p.ldd + min(jj, p.k - 4));
      if (c >= p.d || jj >= p.k) ro[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, a = e >> 5, c = e & 31;
      sA[e] = rv[i];
      sAn[a * kSpLdA + c] = -rv[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, c = u & 63;
#pragma unroll
      for (int f = 0; f < 4; ++f) oldA[(4 * (u >> 6) + f) * 64 + c] = ro[i][f];
    }
  }
  if (tid == 0) chain_prog = 0;
  // operands of this wave's products at the current block: pa* of U_r -= A[r][b-1] dD_{b-1} for its row blocks
  // r = r0, r0 + 3, r0 + 6 > b; pf* of the chain-following U_{b+1} -= A[b+1][b] dD_b.  They are requested one block
  // ahead (na*, nf*): no product waits for memory.
  float pa0[3][8], pa1[3][8], pf0[8], pf1[8], na0[3][8], na1[3][8], nf0[8], nf1[8];
  if (w > 0 && 1 < nblk && owner(1) == w) load_a(1, 0, pf0, pf1);
  float top_a[4] = {0.f, 0.f, 0.f, 0.f};
  // new atoms go straight to the dictionary (lane = feature; the columns of block b were read a block ago)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.Dout, 0, (int)((int64_t)p.d * p.ldo * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Dsrc), 0, (int)((int64_t)p.d * p.ldd * 4), 0x00020000);
  const unsigned ovoff = lane < p.d ? (unsigned)((int64_t)lane * p.ldo * 4) : 0xFFFFFF00u;   // (beyond the buffer: dropped)
#ifdef LASSO_SWEEP_TIMING
  long long* const tlog = (long long*)p.dD;                 // [nblk + 1][8] wall_clock64 stamps (debug builds)
#define SS_STAMP(slot) do { if (lane == 0) tlog[b * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define SS_STAMP(slot) do {} while (0)
#endif
  for (int b = 0; b < nblk; ++b) {
    const int par = b & 1;
    __syncthreads();                                        // helpers done with block b - 1's work; block b staged
    if (w == 0) SS_STAMP(0);
    if (b > 0) {                                            // the last 8 deltas of block b - 1 -> rows of block b
      float a0[8] = {}, a1[8] = {};                         // (its A operands were fetched a block ago: no memory wait here)
      a0[6] = top_a[0]; a0[7] = top_a[1]; a1[6] = top_a[2]; a1[7] = top_a[3];
      product(Ur + b * JB * 64, dD + (par ^ 1) * JB * kSsLdD, a0, a1, 6, w, I1{}, K2{});
      __syncthreads();
    }
    if (b + 1 < nblk) {                                     // -A[b+1][b] at k-steps 6, 7 for the next loop top
      const unsigned v0 = (unsigned)(((JB * (b + 1) + l15) * (int)p.lda + q) * 4), so = (unsigned)((JB * b + 24) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        top_a[i] = -__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(arsrc, v0 + (i >> 1) * 16 * (int)p.lda * 4,
                                                                           so + 16 * (i & 1), 0));
    }
    if (w == 0) {
      SS_STAMP(1);
      // ---- the chain of block b (sweep_persist_kernel's, one column per lane)
      const int j0 = JB * b, nb_at = min(JB, p.k - j0);
      const float* const cA = sA + par * JB * JB;
      const float* const cAn = sAn + par * JB * kSpLdA + (lane & 3);
      const float* const ub = Ur + b * JB * 64;
      const float* const oa = oldA + par * JB * 64;
      float* const dOut = dD + par * JB * kSsLdD;
      f32x4 u[JB / 4];
#pragma unroll
      for (int Q = 0; Q < JB / 4; ++Q) u[Q] = *(const f32x4*)(ub + ss_ub(4 * Q, lane));
      unsigned degmask = 0;
      float cqn[JB / 4], caan, dcn;
      auto fetch = [&](auto a_c) {
        constexpr int a = decltype(a_c)::value;
#pragma unroll
        for (int Q = (a + 1) / 4; Q < JB / 4; ++Q) cqn[Q] = cAn[a * kSpLdA + 4 * Q];
        caan = cA[a * JB + a];
        dcn = oa[a * 64 + lane];
      };
      fetch(std::integral_constant<int, 0>{});
      auto chain = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;        // all 32 atoms exist: no per-atom branch
      static_for<JB>([&](auto a_c) {
        constexpr int a = decltype(a_c)::value, Q0 = a / 4, r0 = a % 4, Qs = (a + 1) / 4;
        float cq[JB / 4];
#pragma unroll
        for (int Q = Qs; Q < JB / 4; ++Q) cq[Q] = cqn[Q];
        const float caa = caan, dc = dcn;
        if constexpr (a + 1 < JB) fetch(std::integral_constant<int, a + 1>{});
        const float v = fmaxf(fmaf(caa, dc, u[Q0][r0]), lo);              // u_j = U_j + A_jj d_j   (:85-88)
        float ss = fmaf(v, v, 0.0f);
        ss = wave_sum_dpp_bcast(ss);
        const bool deg = ss < eps2;
        const float inv = deg ? 0.0f : __builtin_amdgcn_rsqf(ss);
        const float dnew = v * inv;
        const float delta = (FULL || a < nb_at) ? dnew - dc : 0.0f;
        if (FULL || a < nb_at) {
          degmask |= (deg ? 1u : 0u) << a;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dnew), orsrc, ovoff, (unsigned)((j0 + a) * 4), 0);
        }
        dOut[a * kSsLdD + lane] = delta;
        if constexpr (a % 8 == 7) chain_prog = j0 + a + 1;                 // (LDS keeps a wave's writes in order)
#pragma unroll
        for (int Q = Qs; Q < JB / 4; ++Q) u[Q] = __builtin_amdgcn_mfma_f32_4x4x1f32(cq[Q], delta, u[Q], 0, 0, 0);
      });
      };
      if (nb_at == JB) chain(std::true_type{});
      else chain(std::false_type{});
      SS_STAMP(2);
      if (lane < JB) s_deg[j0 + lane] = lane < nb_at ? (int)((degmask >> lane) & 1u) : 0;
      if (lane < nb_at) p.degenerate[j0 + lane] = (int)((degmask >> lane) & 1u);
    } else {
      // ---- waves 1-3: this wave's row blocks, then the staging for block b + 1.  Order of the memory traffic: the
      // staging loads and the next block's A operands are requested first, the products run on operands requested
      // a block ago, the staged values go to LDS last.
      const float* const ddp = dD + (par ^ 1) * JB * kSsLdD;              // deltas of block b - 1 (complete)
      const float* const ddc = dD + par * JB * kSsLdD;                    // deltas of block b (being written)
      const int nb = b + 1;
      float sd[16];
      f32x4 so[8];
      // (the wave that follows the chain this block -- the owner of row block nb -- stages nothing)
      const bool st_diag = nb < nblk && w == owner(nb + 1), st_old = nb < nblk && w == owner(nb + 2);
      if (st_diag) {                                                      // A[nb][nb]
        // (rows beyond k read as zero through the descriptor; columns beyond k are masked: lane & 31 is the column)
        const unsigned v0 = (unsigned)((((JB * nb + (lane >> 5)) * (int)p.lda) + JB * nb + (lane & 31)) * 4);
        const bool colok = JB * nb + (lane & 31) < p.k;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(arsrc, v0 + (unsigned)(2 * i * (int)p.lda * 4), 0, 0));   // (the row in the CHECKED part of the offset)
          sd[i] = colok ? v : 0.0f;
        }
      }
      if (st_old) {                                                       // old atoms of block nb (feature = lane)
        const unsigned v0 = (unsigned)((lane * (int)p.ldd + JB * nb) * 4);    // (features >= d: beyond the descriptor, zero)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          so[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srsrc, v0, 16 * i, 0));
          if (JB * nb + 4 * i >= p.k) so[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      int r1 = b + 2;                                                     // the next block's products
      while (r1 < nblk && owner(r1) != w) ++r1;
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
        if (r1 + 3 * s2 < nblk) load_a(r1 + 3 * s2, b, na0[s2], na1[s2]);
      const bool follows_next = b + 2 < nblk && owner(b + 2) == w;
      if (follows_next) load_a(b + 2, b + 1, nf0, nf1);
      SS_STAMP(2 + w);
      int r0 = b + 1;
      while (r0 < nblk && owner(r0) != w) ++r0;
      if (b > 0) {
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
          if (r0 + 3 * s2 < nblk) product(Ur + (r0 + 3 * s2) * JB * 64, ddp, pa0[s2], pa1[s2], 0, 0, I4{}, K8{});
      }
      if (nb < nblk && owner(nb) == w) {
        for (int g = 0; g < 3; ++g) {
          while (chain_prog < JB * b + 8 * (g + 1)) __builtin_amdgcn_s_sleep(1);
          product(Ur + nb * JB * 64, ddc, pf0, pf1, 2 * g, 0, I4{}, K2{});
        }
      }
      if (st_diag) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = lane + 64 * i, a = e >> 5, c = e & 31;
          sA[(nb & 1) * JB * JB + e] = sd[i];
          sAn[(nb & 1) * JB * kSpLdA + a * kSpLdA + c] = -sd[i];
        }
      }
      if (st_old) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int f = 0; f < 4; ++f) oldA[((nb & 1) * JB + 4 * i + f) * 64 + lane] = so[i][f];
      }
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
        for (int i = 0; i < 8; ++i) { pa0[s2][i] = na0[s2][i]; pa1[s2][i] = na1[s2][i]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) { pf0[i] = nf0[i]; pf1[i] = nf1[i]; }
      if (w == 1) SS_STAMP(6);                              // this wave's work done
      if (w == 2) SS_STAMP(7);
    }
  }
#ifdef LASSO_SWEEP_TIMING
  { const int b = nblk; if (w == 0) SS_STAMP(0); }
#endif
  // ---- degenerate atoms (rare), as degenerate_fixup_kernel: the i-th in atom order takes pool row i
  __syncthreads();
  const int ndeg = __syncthreads_count(tid < p.k && s_deg[tid] != 0);     // (k <= 256 threads: one flag each)
  if (tid == 0) {
    int c = 0;
    if (ndeg)                                               // rare: the ordered list, serially
      for (int jj = 0; jj < p.k; ++jj)
        if (s_deg[jj]) s_idx[c++] = jj;
    s_count = c;
    p.ndeg_in_out[0] = c;
    if (p.ndeg_mirror) {                                   // {count, valid}: the host may poll word 1 (zeroed before the call)
      __hip_atomic_store(p.ndeg_mirror, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(p.ndeg_mirror + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (s_count) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < s_count; ++i) {
    const int jj = s_idx[i];
    float g = 0.0f;
    if (tid < p.d) {
      if (p.pool && p.pool_rows > 0) g = p.pool[(int64_t)min(i, p.pool_rows - 1) * p.pool_ld + tid];
      else g = counter_normal(p.seed, (unsigned)jj, (unsigned)tid);
      if (p.positive) g = fmaxf(g, 0.0f);
    }
    sh[tid] = fmaf(g, g, 0.0f);
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if (tid < s2) sh[tid] += sh[tid + s2];
      __syncthreads();
    }
    const float inv = 1.0f / sqrtf(sh[0]);
    if (tid < p.d) p.Dout[(int64_t)tid * p.ldo + jj] = g * inv;
    __syncthreads();
  }
}
size_t sweep_small_lds(int k) {
  const int nblk = (k + kSweepBlock - 1) / kSweepBlock;
  return (size_t)(nblk * kSweepBlock * 64 + 2 * kSweepBlock * kSweepBlock + 2 * kSweepBlock * kSpLdA +
                  2 * kSweepBlock * kSsLdD + 2 * kSweepBlock * 64) * 4;
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
    if (p.ndeg_mirror) {                                   // {count, valid}: the host may poll word 1 (zeroed before the call)
      __hip_atomic_store(p.ndeg_mirror, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(p.ndeg_mirror + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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

// Last launch of the single-launch sweep: workgroup g takes the 32 new atoms of block g (rows of DtN), replaces the
// degenerate ones among them exactly as degenerate_fixup_kernel does (the i-th degenerate atom in atom order takes
// pool row i; same reduction order for its norm), and writes them as columns of the dictionary: repair and
// transposition in one launch.  Workgroup 0 also leaves the count of degenerate atoms.
__global__ __launch_bounds__(256) void fixup_transpose_kernel(const SweepParams p) {
  constexpr int JB = kSweepBlock;
  __shared__ float tile[JB][257];
  __shared__ int s_list[JB], s_bsum, s_tsum;
  __shared__ float sh[256];
  __shared__ int s_n, s_first;
  const int tid = threadIdx.x, j0 = JB * blockIdx.x, nb_at = min(JB, p.k - j0);
  const int per = (p.k + 255) / 256;
  const int lo = min(tid * per, p.k), hi = min(lo + per, p.k);
  __shared__ int s_late;
  if (tid == 0) {
    s_bsum = 0; s_tsum = 0; s_late = 0;
    if (p.wait_word) {
      // the other stream of the pipelined M-step (objective, U rows of the later stages) still reads the OLD dictionary:
      // wait for its word instead of a cross-stream event in front of this launch (~6-10 us on the step's chain)
      int spins = 0;
      while (__hip_atomic_load(p.wait_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.wait_value &&
             ++spins < kGateSpinLimit)
        __builtin_amdgcn_s_sleep(16);
      s_late = spins >= kGateSpinLimit;  // ~30 s (the other stream's work includes the caller's collectives): reported
                                         // through the count of degenerate atoms (-1), never silently
    }
  }
  __syncthreads();
  int before = 0, total = 0;
  for (int j = lo; j < hi; ++j) {
    const int f = p.degenerate[j] != 0;
    total += f;
    before += (j < j0) ? f : 0;
  }
  if (total) { atomicAdd(&s_tsum, total); atomicAdd(&s_bsum, before); }   // (integer sums: any order; no degenerate atom, no atomic)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 256 * i, a = e >> 6, c4 = (e & 63) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a < nb_at) v = *(const f32x4*)(p.Dt + (int64_t)(j0 + a) * 256 + c4);
#pragma unroll
    for (int f = 0; f < 4; ++f) tile[a][c4 + f] = v[f];
  }
  __syncthreads();
  if (tid == 0) {
    const int b = s_bsum, t = s_tsum;
    if (blockIdx.x == 0) {
      p.ndeg_in_out[0] = s_late ? -1 : t;
      if (p.ndeg_mirror) {
        __hip_atomic_store(p.ndeg_mirror, s_late ? -1 : t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.ndeg_mirror + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    int n = 0;
    if (t)
      for (int a = 0; a < nb_at; ++a)
        if (p.degenerate[j0 + a]) s_list[n++] = a;
    s_n = n; s_first = b;
  }
  __syncthreads();
  for (int i = 0; i < s_n; ++i) {
    const int a = s_list[i], j = j0 + a, idx = s_first + i;
    float g = 0.0f;
    if (tid < p.d) {
      if (p.pool && p.pool_rows > 0) g = p.pool[(int64_t)min(idx, p.pool_rows - 1) * p.pool_ld + tid];
      else g = counter_normal(p.seed, (unsigned)j, (unsigned)tid);
      if (p.positive) g = fmaxf(g, 0.0f);
    }
    sh[tid] = fmaf(g, g, 0.0f);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) sh[tid] += sh[tid + s];
      __syncthreads();
    }
    const float inv = 1.0f / sqrtf(sh[0]);
    tile[a][tid] = g * inv;
    __syncthreads();
  }
  const int a = tid & 31;
  if (a < nb_at)
    for (int c = tid >> 5; c < p.d; c += 8) p.Dout[(int64_t)c * p.ldo + j0 + a] = tile[a][c];
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
// [A | B] = Z^T [Z | X] in one launch on 256 x 256 blocks (see gram_ab256_kernel): k, d multiples of 256,
// aligned operands, a batch that gives every split at least 512 rows, scratch for the partials.
// Returns false when the shape does not qualify (the caller takes the two-launch path).
size_t gram_ab_scratch_bytes(int64_t d, int64_t k) {
  if (k < kG3B || d < kG3B || k % kG3B || d % kG3B) return 0;
  const int64_t nb = k / kG3B, blocks = nb * (nb + 1) / 2 + nb * (d / kG3B);
  const size_t one = (size_t)k * (size_t)(k + d) * 4;
  // up to four rounds of workgroups on an MI355X's 256 CUs, at most 512 MB
  const int64_t splits = std::min<int64_t>(std::min<int64_t>(kGramAbMaxSplits, std::max<int64_t>(1, 1024 / blocks)),
                                           std::max<int64_t>(1, (int64_t)(((size_t)512 << 20) / one)));
  return (size_t)splits * one;
}

bool launch_gram_ab(const float* Z, int64_t ldz, int k, const float* X, int64_t ldx, int d, int n, float* A, float* B,
                    float* scratch, size_t scratch_bytes, int cus, hipStream_t stream, hipError_t* err, int* raise,
                    int raise_value) {
  *err = hipSuccess;
  if (k < kG3B || d < kG3B || k % kG3B || d % kG3B || (ldz & 3) || (ldx & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)X & 15) ||
      !scratch || n < 8 * 512)
    return false;
  const int nb = k / kG3B, blocks = nb * (nb + 1) / 2 + nb * (d / kG3B);
  // Sample splits: one workgroup per CU at a time (LDS), so the launch runs in ceil(blocks * s / cus) rounds; every
  // split also writes and re-reads k (k + d) partial sums.  Pick the s with the least modelled time.
#ifndef LASSO_GRAM_AB_MIN_ROWS
#define LASSO_GRAM_AB_MIN_ROWS 256   // (round 4: it was 512; an 8192-row shard then had at most 16 splits = 224 workgroups)
#endif
  const int smax = (int)std::min<size_t>(std::min(n / LASSO_GRAM_AB_MIN_ROWS, kGramAbMaxSplits), scratch_bytes / ((size_t)k * (k + d) * 4));
  if (smax < 1) return false;
  const double work = (double)blocks * kG3B * kG3B * 2.0 * n / 130e12, fold = (double)k * (k + d) * 8.0 / 3e12;
  int splits = 1;
  double best = 1e30, best_eff = 0.0;
  for (int sidx = 1; sidx <= smax && blocks * sidx <= 4 * cus + blocks; ++sidx) {
    const int wgs = blocks * sidx;
    const double eff = (double)wgs / ((double)((wgs + cus - 1) / cus) * cus), t = work / eff + sidx * fold;
    if (t < best) { best = t; best_eff = eff; splits = sidx; }
  }
  if (best_eff < 0.75) return false;                       // would leave over a quarter of the chip idle
  const int rps = ((n + splits - 1) / splits + kG3S - 1) / kG3S * kG3S;
  const int sp = (n + rps - 1) / rps;
  const size_t lds = (size_t)4 * kG3S * kG3Ld * 4;
  if ((*err = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_ab256_kernel), lds)) != hipSuccess) return true;
  hipLaunchKernelGGL(gram_ab256_kernel, dim3(blocks, 1, sp), dim3(512), lds, stream, Z, ldz, k, X, ldx, d, n, scratch, rps,
                     GramRows{-1, -1, nullptr, 0, nullptr, nullptr, 0, raise, raise_value});
  const int64_t stride = (int64_t)k * (k + d);
  const int nt = (k + 31) / 32;
  const int nsym = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(sum_splits_ab_kernel, dim3(nsym + (unsigned)(((int64_t)k * d + 255) / 256)), dim3(256), 0, stream,
                     scratch, sp, stride, (int64_t)(k + d), k, d, nsym, A, B);
  *err = hipGetLastError();
  return true;
}

#ifndef LASSO_GRAM_MIN_ROWS
#define LASSO_GRAM_MIN_ROWS 128
#endif
constexpr int kGramMinRowsC = LASSO_GRAM_MIN_ROWS;      // (== kGramMinRows below)
// A = Z^T Z (k x k, sym) and B = Z^T X (k x d) of a SMALL dictionary in one product launch + one fold launch (round 6:
// the EM step of 8 x 8 patches -- d = 64, k = 256 -- spent four launches here).  false: not applicable (the caller takes
// the two products one after the other).  Scratch: [splits][k][k] for A, then [splits][k][d] for B.
bool launch_gram_ab128(const float* Z, int64_t ldz, int k, const float* X, int64_t ldx, int d, int n, float* A, float* B,
                       float* scratch, size_t scratch_bytes, int cus, int max_splits, hipStream_t stream, hipError_t* err,
                       int* raise, int raise_value) {
  *err = hipSuccess;
  if (k < kG2B || d > kG2B || (k & 3) || (d & 3) || (ldz & 3) || (ldx & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)X & 15) ||
      !scratch || n <= 0)
    return false;
  const int nbp = (k + kG2B - 1) / kG2B, nsym = nbp * (nbp + 1) / 2, nb2 = nbp;         // B: one column block per row block
  if (nsym >= 128) return false;                                                          // (the few-tiles fold below)
  int splits = std::max(1, 2 * cus / std::max(nsym + nb2, 1));
  splits = std::min(splits, std::max(n / kGramMinRowsC, 1));
  splits = std::max(1, std::min(splits, max_splits));
  const int rps = ((n + splits - 1) / splits + kG2S - 1) / kG2S * kG2S;
  const int sp = std::max(1, (n + rps - 1) / std::max(rps, 1));
  const int64_t strideA = (int64_t)k * k, strideB = (int64_t)k * d;
  if ((size_t)sp * (size_t)(strideA + strideB) * 4 > scratch_bytes) return false;
  float* const partA = scratch;
  float* const partB = scratch + (size_t)sp * strideA;
  const size_t lds = (size_t)4 * kG2S * kG2Ld * 4;
  if ((*err = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_tn128_kernel), lds)) != hipSuccess) return true;
  hipLaunchKernelGGL(gram_tn128_kernel, dim3(nsym + nb2, 1, sp), dim3(256), lds, stream, Z, ldz, k, Z, ldz, k, n, partA,
                     (int64_t)k, 1, rps, strideA, Gram2{X, ldx, d, partB, (int64_t)d, strideB, nsym, raise, raise_value});
  const int nt = (k + 31) / 32, ntiles = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(sum_splits_sym4_b_kernel, dim3(4 * ntiles + (unsigned)((strideB + 255) / 256)), dim3(256), 0, stream,
                     partA, sp, strideA, (int64_t)k, k, A, (int64_t)k, 4 * ntiles, partB, strideB, k, d, B, (int64_t)d);
  *err = hipGetLastError();
  return true;
}

static bool gram_use128(const float* P, int64_t ldp, int pc, const float* Q, int64_t ldq, int qc) {
  return pc >= kG2B && qc >= kG2B && pc % 4 == 0 && qc % 4 == 0 && ldp % 4 == 0 && ldq % 4 == 0 &&
         ((uintptr_t)P & 15) == 0 && ((uintptr_t)Q & 15) == 0;
}

#ifndef LASSO_GRAM_MIN_ROWS
#define LASSO_GRAM_MIN_ROWS 128
#endif
// samples a split keeps at least (round 4: 128, it was 512 -- an 8192-row shard of a k = 256 dictionary ran its three
// 128 x 128 blocks of Z^T Z on 48 workgroups, 16 chunks of samples each: 48 us)
constexpr int kGramMinRows = LASSO_GRAM_MIN_ROWS;
int gram_splits(int pc, int qc, int n, int sym, int cus, int max_splits) {
  int blocks = ((qc + kGB - 1) / kGB) * ((pc + kGB - 1) / kGB);
  if (sym) blocks = blocks / 2 + (pc + kGB - 1) / kGB / 2 + 1;
  if (pc >= kG2B && qc >= kG2B) {               // 128 x 128 blocks, two workgroups per CU
    const int nbp = (pc + kG2B - 1) / kG2B, nbq = (qc + kG2B - 1) / kG2B;
    blocks = sym ? nbp * (nbp + 1) / 2 : nbp * nbq;
    int s2 = std::max(1, 2 * cus / std::max(blocks, 1));   // one round of resident workgroups, no tail
    s2 = std::min(s2, std::max(n / kGramMinRows, 1));
    return std::max(1, std::min(s2, max_splits));
  }
  int s = (3 * cus + blocks - 1) / std::max(blocks, 1);
  s = std::min(s, std::max(n / kGramMinRows, 1));
  return std::max(1, std::min(s, max_splits));
}

hipError_t launch_gram_tn(const float* P, int64_t ldp, int pc, const float* Q, int64_t ldq, int qc,
                          int n, float* C, int64_t ldc, int sym, float* scratch, int splits,
                          hipStream_t stream) {
  if (!scratch) splits = 1;
  if (gram_use128(P, ldp, pc, Q, ldq, qc) && (!sym || scratch)) {
    const int rps = ((n + splits - 1) / splits + kG2S - 1) / kG2S * kG2S;
    const int sp = std::max(1, (n + rps - 1) / std::max(rps, 1));
    const int nbp = (pc + kG2B - 1) / kG2B, nbq = (qc + kG2B - 1) / kG2B;
    const int blocks = sym ? nbp * (nbp + 1) / 2 : nbp * nbq;
    const size_t lds = (size_t)4 * kG2S * kG2Ld * 4;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_tn128_kernel), lds); e != hipSuccess) return e;
    const bool to_scratch = sym || sp > 1;
    float* const out = to_scratch ? scratch : C;
    const int64_t old_ = to_scratch ? qc : ldc, stride128 = (int64_t)pc * qc;
    hipLaunchKernelGGL(gram_tn128_kernel, dim3(blocks, 1, sp), dim3(256), lds, stream, P, ldp, pc, Q, ldq, qc, n, out,
                       old_, sym, rps, stride128, Gram2{nullptr, 0, 0, nullptr, 0, 0, 0});
    if (sym) {
      const int nt = (pc + 31) / 32, ntiles = nt * (nt + 1) / 2;
      if (ntiles < 128)
        hipLaunchKernelGGL(sum_splits_sym4_kernel, dim3(4 * ntiles), dim3(256), 0, stream, scratch, sp, stride128,
                           (int64_t)qc, pc, C, ldc);
      else
        hipLaunchKernelGGL(sum_splits_sym_kernel, dim3(ntiles), dim3(256), 0, stream, scratch, sp, stride128,
                           (int64_t)qc, pc, C, ldc);
    } else if (sp > 1) {
      hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)((stride128 + 255) / 256)), dim3(256), 0, stream, scratch, sp,
                         stride128, pc, qc, C, ldc);
    }
    return hipGetLastError();
  }
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

template <int NW, int F>
static hipError_t sweep_blocks(const SweepParams& p, hipStream_t stream) {
  const size_t lds = (size_t)kSweepBlock * 64 * NW * F * 4;
  if (lds > 32 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&sweep_block_kernel<true, NW, F>), lds);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void*>(&sweep_block_kernel<false, NW, F>), lds);
    if (e != hipSuccess) return e;
  }
  for (int j0 = 0; j0 < p.k; j0 += kSweepBlock) {
    if (j0 + kSweepBlock <= p.k)
      hipLaunchKernelGGL((sweep_block_kernel<true, NW, F>), dim3(1), dim3(64 * NW), lds, stream, p, j0);
    else
      hipLaunchKernelGGL((sweep_block_kernel<false, NW, F>), dim3(1), dim3(64 * NW), lds, stream, p, j0);
    if (j0 + kSweepBlock < p.k) {
      const int rows = p.k - j0 - kSweepBlock;
      hipLaunchKernelGGL(trailing_update_kernel, dim3(std::min(rows, 256), p.dp / 256), dim3(256), 0, stream, p, j0);
    }
  }
  return hipGetLastError();
}

size_t sweep_persist_extra_bytes(int k) {
  const size_t nblk = (size_t)(k + kSweepBlock - 1) / kSweepBlock;
  return 3 * nblk * kSweepBlock * 256 * 4 + 4096 + nblk * 128 + 256;   // DtN, dDg, Uw, flags (+ debug time stamps), pipeline words
}

int* sweep_pipe_words(void* persist_extra, int k) {
  const size_t nblk = (size_t)(k + kSweepBlock - 1) / kSweepBlock;
  return (int*)((char*)persist_extra + 3 * nblk * kSweepBlock * 256 * 4 + 4096 + nblk * 128);
}

int* sweep_persist_flags(void* persist_extra, int k) {
  const size_t rows = (size_t)((k + kSweepBlock - 1) / kSweepBlock) * kSweepBlock * 256;
  return (int*)((float*)persist_extra + 3 * rows);
}

// dp == 256: the single-launch sweep (+ its stand-by).  `extra` = sweep_persist_extra_bytes(k)
// bytes; on return *dt_out is where the new atoms are (rows of length 256).
static hipError_t sweep_persistent(const SweepParams& p, void* extra, float** dt_out, hipStream_t stream, int gate = 0,
                                   int gate0 = 1) {
  const int nblk = (p.k + kSweepBlock - 1) / kSweepBlock;
  const size_t rows = (size_t)nblk * kSweepBlock * 256;
  SweepPersist x;
  x.DtN = (float*)extra;
  x.dDg = x.DtN + rows;
  x.Uw = x.dDg + rows;
  x.flags = (int*)(x.Uw + rows);
  x.solo = nblk <= kSpSelf + 1;
  if (g_force_standby) x.solo = 1;   // (test hook, lasso_debug_force_standby)
  x.run_if = nullptr;
  const size_t lds = (size_t)(2 * kSweepBlock * kSweepBlock + 4 * kSweepBlock * kSpLdA + 3 * kSweepBlock * kSpLdB) * 4;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&sweep_persist_kernel), lds);
  if (e != hipSuccess) return e;
  if (!p.flags_cleared && (e = hipMemsetAsync(x.flags, 0, 4096, stream)) != hipSuccess) return e;
  // Workgroups go round the 8 XCDs in launch order: worker i is workgroup 8 i, on the sweeper's XCD -- deltas and
  // rows change hands through that XCD's L2.  The workgroups in between return at once.
#ifndef LASSO_SWEEP_WG_STRIDE
#define LASSO_SWEEP_WG_STRIDE 8
#endif
  x.wg_stride = LASSO_SWEEP_WG_STRIDE;
  x.gate = gate;
  x.gate0 = gate0;
  // gated (pipelined M-step): other launches run beside the sweep and must find room on EVERY XCD (a launch's
  // workgroups go round the XCDs; a workgroup of the sweep takes a CU's whole LDS): workers on consecutive
  // workgroups -- four CUs per XCD instead of all 32 of one
  if (gate > 0) x.wg_stride = 1;
  const int grid = x.solo ? 1 : (nblk - kSpSelf - 1) * x.wg_stride + 1;
  hipLaunchKernelGGL(sweep_persist_kernel, dim3(grid), dim3(256), lds, stream, p, x);
  if (!x.solo) {            // stand-by: runs only if the grid gave up (a workgroup was not resident)
    SweepPersist y = x;
    y.solo = 1;
    y.run_if = x.flags + 1;
    hipLaunchKernelGGL(sweep_persist_kernel, dim3(1), dim3(256), lds, stream, p, y);
  }
  *dt_out = x.DtN;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined M-step (round 6).  The sweep of atom block b needs rows b of A and of U = B - A D^T only, and it walks
// the blocks at ~8 us each -- far slower than the chip produces them.  So [A | B] is produced by BLOCK ROWS of 256
// atoms (gram_ab256_kernel's row mode: every block row with as many sample splits as fill the chip by itself, folded
// by fold_rows_kernel, mirrored into the rows below), the sweep is launched as soon as block row 0's U rows exist, and
// the later block rows are produced on a second stream while it runs, each announced by one flag word the sweep's
// workers / staging wave wait for.  The caller (lasso_amd/parallel.py) owns the two streams and the all-reduce of a
// block row between its fold and its U rows.
// ---------------------------------------------------------------------------------------------------------------
static_assert(kSweepRowFlag == kSpRowFlag, "one definition of the row flags' place");
MstepPipePlan mstep_pipe_plan(int64_t n, int64_t d, int64_t k, int cus) {
  MstepPipePlan pl;
  pl.nstages = 0; pl.scratch_bytes = 0;
  if (d != kG3B || k % kG3B || k < 2 * kG3B || k > kSweepMaxK || k / kG3B > kPipeMaxBlocks || n < 1 || n > INT32_MAX / 2 ||
      cus < 64)
    return pl;
  // Stages: the HEAD (the first half of the block rows, produced before the sweep starts) and then one block row per
  // stage beside the sweep.  Why not the first block row alone: a worker of the sweep cannot start catching up on the
  // deltas of the blocks in front of it before its rows of A and U exist (~2.3 us per block of deltas, measured), so the
  // rows of block r are needed ~6 r us after the sweep starts, and row-mode production (more splits, more partial sums,
  // three launches per stage) delivers 256 rows per ~75 us: with a one-block-row head the sweep stalled 43 + 36 us at
  // the group boundaries (profiles/r06/pipe_stamps_4stage.txt).
  const int nb = (int)(k / kG3B), workers = (int)(k / kSweepBlock);
  const int head = (nb + 1) / 2;
  size_t off = 0;
  int lo = 0;
  while (lo < nb) {
    const int hi = lo == 0 ? head : lo + 1, s = pl.nstages++;
    // the head runs before the sweep; the others beside it (the sweep holds one CU per worker), with a margin, so that
    // every workgroup of a launch finds a CU at once (one workgroup per CU: LDS and registers)
    int avail = lo == 0 ? cus : std::max(cus - workers - 8, cus / 2);
    static const int avail_knob = [] { const char* g = getenv("LASSO_PIPE_AVAIL"); return g ? atoi(g) : 0; }();   // A/B knob, read once
    if (lo > 0 && avail_knob > 0) avail = std::max(16, std::min(avail, avail_knob));
    int blocks = 0;
    for (int r = lo; r < hi; ++r) blocks += nb - r + (int)(d / kG3B);
    int splits = std::max(1, std::min(avail / blocks, kGramAbMaxSplits));
    int rps = (int)(((n + splits - 1) / splits + kG3S - 1) / kG3S * kG3S);
    rps = std::max(rps, 2 * kG3S);
    splits = (int)((n + rps - 1) / rps);
    pl.lo[s] = lo; pl.hi[s] = hi; pl.blocks[s] = blocks;
    pl.splits[s] = splits; pl.rps[s] = rps; pl.scratch_off[s] = off;
    off += ((size_t)splits * (hi - lo) * kG3B * (size_t)(k + d) * 4 + 255) & ~(size_t)255;
    lo = hi;
  }
  pl.scratch_bytes = off;
  return pl;
}

hipError_t launch_gram_rows(const float* Z, int64_t ldz, int k, const float* X, int64_t ldx, int d, int n, float* AB,
                            int64_t ldab, int stage, const MstepPipePlan& plan, float* scratch, int* clear_words, int nclear,
                            hipStream_t stream) {
  if (stage < 0 || stage >= plan.nstages || (ldz & 3) || (ldx & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)X & 15)) return hipErrorInvalidValue;
  const int lo = plan.lo[stage], hi = plan.hi[stage];
  float* const part = (float*)((char*)scratch + plan.scratch_off[stage]);
  const size_t lds = (size_t)4 * kG3S * kG3Ld * 4;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_ab256_kernel), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(gram_ab256_kernel, dim3(plan.blocks[stage], 1, plan.splits[stage]), dim3(512), lds, stream, Z, ldz, k, X,
                     ldx, d, n, part, plan.rps[stage], GramRows{lo, hi, clear_words, nclear, nullptr, nullptr, 0});
  int tiles = 0;
  for (int r = lo; r < hi; ++r) tiles += (kG3B / 32) * ((k - kG3B * r + d) / 32);
  hipLaunchKernelGGL(fold_rows_kernel, dim3(tiles), dim3(256), 0, stream, part, plan.splits[stage],
                     (int64_t)(hi - lo) * kG3B * (k + d), (int64_t)(k + d), k, d, lo, AB, ldab);
  return hipGetLastError();
}

hipError_t launch_set_flag(int* flag, int value, hipStream_t stream) {
  hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, stream, flag, value);
  return hipGetLastError();
}

hipError_t launch_wait_word(const int* word, int seq, int host_memory, hipStream_t stream) {
  hipLaunchKernelGGL(wait_word_kernel, dim3(1), dim3(1), 0, stream, word, seq, host_memory);
  return hipGetLastError();
}

hipError_t launch_uprod_rows(const float* A, int64_t lda, const float* Bm, int64_t ldb, const float* C0, int64_t ldc0,
                             float* U, int64_t ldu, int rows, int kk, int* ticket, int* flag, int nflags, int flag_value,
                             hipStream_t stream) {
  if (rows % 16 || kk % 256 || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)Bm & 15)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(uprod_rows_kernel, dim3(rows), dim3(64), 0, stream, A, lda, Bm, ldb, C0, ldc0, U, ldu, kk, ticket, flag,
                     nflags, flag_value);
  return hipGetLastError();
}

hipError_t launch_sweep_gated(const SweepParams& p, void* persist_extra, int gate, int gate0, float** dt_out, hipStream_t stream) {
  if (p.dp != 256 || !persist_extra || !dt_out || !p.Dsrc || !p.Dout) return hipErrorInvalidValue;
  return sweep_persistent(p, persist_extra, dt_out, stream, gate, gate0);
}

#ifndef LASSO_SWEEP_WAVES256
#define LASSO_SWEEP_WAVES256 4
#endif
hipError_t launch_dict_sweep(const SweepParams& p, hipStream_t stream, void* persist_extra, float** dt_out) {
  hipError_t e;
  if (dt_out) *dt_out = p.Dt;
#ifndef LASSO_SWEEP_NO_SMALL
  if (p.d <= 64 && p.k <= kSweepBlock * kSsMaxBlk && p.Dsrc && p.Dout) {     // the one-workgroup sweep, nothing else
    const size_t lds = sweep_small_lds(p.k);
    if ((e = ensure_dynamic_lds(reinterpret_cast<const void*>(&sweep_small_kernel), lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(sweep_small_kernel, dim3(1), dim3(256), lds, stream, p);
    return hipGetLastError();
  }
#endif
  if (p.dp == 256 && persist_extra && dt_out) {
    if ((e = sweep_persistent(p, persist_extra, dt_out, stream)) != hipSuccess) return e;
    SweepParams f = p;
    f.Dt = *dt_out;
    if (p.Dout) hipLaunchKernelGGL(fixup_transpose_kernel, dim3((p.k + kSweepBlock - 1) / kSweepBlock), dim3(256), 0, stream, f);
    else hipLaunchKernelGGL(degenerate_fixup_kernel, dim3(1), dim3(256), 0, stream, f);
    return hipGetLastError();
  }
  switch (p.dp) {
    case 256: e = sweep_blocks<LASSO_SWEEP_WAVES256, 4 / LASSO_SWEEP_WAVES256>(p, stream); break;
    case 512: e = sweep_blocks<2, 4>(p, stream); break;
    case 768: e = sweep_blocks<3, 4>(p, stream); break;
    case 1024: e = sweep_blocks<4, 4>(p, stream); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(degenerate_fixup_kernel, dim3(1), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// last launch of the single-launch sweep, on its own (pipelined M-step: the caller decides when D may change):
// repairs the degenerate atoms and writes the dictionary from the new atoms at p.Dt
hipError_t launch_sweep_fixup(const SweepParams& p, hipStream_t stream) {
  if (!p.Dout || p.dp != 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fixup_transpose_kernel, dim3((p.k + kSweepBlock - 1) / kSweepBlock), dim3(256), 0, stream, p);
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
