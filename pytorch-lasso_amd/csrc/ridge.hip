// Unconstrained M-step: V = ((Z^T Z + lambd n I)^-1 Z^T X)^T, reference
// lasso/linear/dict_learning.py:106-123 (torch.linalg.cholesky + cholesky_solve).
// Given A = Z^T Z [k][k] and B = Z^T X [k][d] (lasso_gram_accumulate) the work is a k x k
// Cholesky factorisation and two triangular solves with d right-hand sides -- k^3/3 + 2 k^2 d
// flop (0.9 GFLOP at k = 1024, d = 256): nothing for the chip, but a chain of k dependent
// pivots, so the time is latency (reported, no roofline).  rocSOLVER's small-matrix kernels
// take 3.4 ms on it; the blocked form below ~0.4 ms.
//
// Working matrix S [kp + dp][kp] (kp, dp: k, d rounded up to 64), rows 0..kp-1 = M = A + lam I
// (lower triangle used, padded diagonal = 1), rows kp.. = B^T; the factor goes to a second matrix
// F of the same shape.  Right-looking, 64-wide block columns, ONE launch per column:
//   chol_first_kernel  factors the first diagonal block: L_00 in LDS and its inverse (diag_factor);
//   chol_step_kernel   step j, one workgroup per 64 x 64 block (i, c) of the trailing part,
//                      i >= c > j, rows of B^T included: P_i = S_ij Linv_j^T, P_c = S_cj Linv_j^T
//                      (the panel, recomputed per block -- 64^3 flop -- instead of a launch of its
//                      own), S_ic -= P_i P_c^T; the blocks of column c = j + 1 also store
//                      P_i = L_ij into F.  The workgroup of block (j+1, j+1) then factors it
//                      (look-ahead), so the chain of dependent pivots overlaps the update.
//   chol_panel_kernel  the last column's panel (rows of B^T only).
// Carrying B^T along makes the forward substitution part of the factorisation: afterwards rows
// kp.. of F hold Y^T = B^T L^-T.  The backward substitution V L = Y^T is independent per row of V:
//   ridge_backward_kernel   one workgroup per 16 rows of V (the whole 16 x kp strip in LDS),
//                      block columns from last to first, V_j = (Y_j - sum_{i>j} V_i L_ij) Linv_j,
//                      the L / Linv tiles streamed through an LDS-DMA ring.
// All products on v_mfma_f32_16x16x4_f32 from LDS tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"
#include "tile_device.hpp"

namespace lasso {
namespace {

constexpr int kRB = 64;                       // block size
constexpr int kRidgeMaxK = 4096;

// S[r][c] for r < kp: A + lam on the diagonal (identity on the padding); r >= kp: B^T
__global__ __launch_bounds__(256) void ridge_setup_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ S, int k, int d, int kp, int dp, float lam) {
  const int64_t total = (int64_t)(kp + dp) * kp;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / kp), c = (int)(idx % kp);
    float v = 0.0f;
    if (r < kp) {
      if (r < k && c < k) v = A[(int64_t)r * k + c] + (r == c ? lam : 0.0f);
      else if (r == c) v = 1.0f;
    } else {
      const int dd = r - kp;
      if (dd < d && c < k) v = B[(int64_t)c * d + dd];
    }
    S[idx] = v;
  }
}

// C[64][64] (+)= sign * A[64][64] * B[64][64]^T on LDS tiles with row stride 65 floats; 256
// threads = 4 waves, wave w owns rows 16w..16w+15 (4 column blocks).  Plain per-lane operand
// reads (stride-65 rows: conflict-free for the 16 rows x 4 k of one MFMA).
__device__ __forceinline__ void mma64_nt(const float* __restrict__ a, const float* __restrict__ b, f32x4 (&acc)[4],
                                         int lane, int w) {
  const int l15 = lane & 15, q = lane >> 4;
#pragma unroll 4
  for (int k0 = 0; k0 < kRB; k0 += 4) {
    const float av = a[(16 * w + l15) * 65 + k0 + q];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const float bv = b[(16 * nb + l15) * 65 + k0 + q];
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nb], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void load_block(const float* __restrict__ src, int64_t ld, float* __restrict__ t, int tid) {
  for (int e = tid; e < kRB * kRB / 4; e += 256) {
    const int r = e / 16, c4 = (e % 16) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + c4);
    t[r * 65 + c4] = v[0]; t[r * 65 + c4 + 1] = v[1]; t[r * 65 + c4 + 2] = v[2]; t[r * 65 + c4 + 3] = v[3];
  }
}

// accumulator layout -> LDS tile / global block: lane holds rows 16w + 4q + rg, column 16 nb + l15
__device__ __forceinline__ void acc_to_tile(const f32x4 (&acc)[4], float* __restrict__ t, int lane, int w) {
  const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) t[(16 * w + 4 * q + rg) * 65 + 16 * nb + l15] = acc[nb][rg];
}

// ---------------------------------------------------------------------------
// factorisation of one 64 x 64 diagonal block held in the LDS tile t (row stride 65): on
// return t's lower triangle = L, x = L^-1 (lower, upper part zero).  Called by all 256 threads.
//   Cholesky: four 16-wide panels.  Panel b lives in the registers of wave 0, lane r = row r
//   (16 values): 16 unrolled column steps, the pivot row's entries broadcast with v_readlane
//   (120 readlane + fma pairs, no barrier, no LDS); the rank-16 update of the columns to its
//   right on MFMA by the other waves.  8 barriers instead of 128.
//   Inverse: the four 16 x 16 diagonal sub-blocks by substitution (one thread per column),
//   the off-diagonal ones by block distance.
// bad pivot (<= 0 or NaN): *info = 1 + its global index (first one wins), inverse pivot 0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

__device__ __forceinline__ void diag_factor(float* __restrict__ t, float* __restrict__ x, int row_base,
                                            int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  for (int b = 0; b < 4; ++b) {
    if (w == 0) {
      float a[16];
#pragma unroll
      for (int c2 = 0; c2 < 16; ++c2) a[c2] = t[lane * 65 + 16 * b + c2];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const int pl = 16 * b + cc;                         // pivot row = lane pl (wave-uniform)
        const float pv = a[cc];
        float y = __builtin_amdgcn_rsqf(pv);
        y = y * (1.5f - 0.5f * pv * y * y);                 // one Newton step on v_rsq_f32
        const float piv = lane_bcast(pv, pl);
        float inv = lane_bcast(y, pl);
        if (!(piv > 0.0f)) {
          inv = 0.0f;
          if (lane == 0) atomicCAS(info, 0, row_base + pl + 1);
        }
        const float l = pv * inv;                           // lane pl: sqrt(piv)
        a[cc] = l;
#pragma unroll
        for (int c2 = cc + 1; c2 < 16; ++c2) a[c2] = fmaf(-l, lane_bcast(l, pl + (c2 - cc)), a[c2]);
      }
      if (lane >= 16 * b) {
#pragma unroll
        for (int c2 = 0; c2 < 16; ++c2) t[lane * 65 + 16 * b + c2] = (16 * b + c2 <= lane) ? a[c2] : 0.0f;
      }
    }
    __syncthreads();
    if (w > b) {                                            // rows 16w.., column blocks b < cb <= w
      for (int cb = b + 1; cb <= w; ++cb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t[(16 * w + l15) * 65 + 16 * b + s4 + q],
                                                     t[(16 * cb + l15) * 65 + 16 * b + s4 + q], acc, 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) t[(16 * w + 4 * q + rg) * 65 + 16 * cb + l15] -= acc[rg];
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < kRB * kRB; e += 256) x[(e >> 6) * 65 + (e & 63)] = 0.0f;
  __syncthreads();
  // diagonal sub-blocks: thread (b, cc) solves column cc of sub-block b by forward substitution
  if (tid < 64) {
    const int b = tid >> 4, cc = tid & 15, o = 16 * b;
    float col[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s = (i == cc) ? 1.0f : 0.0f;
#pragma unroll
      for (int m = 0; m < 16; ++m)
        if (m < i) s = fmaf(-t[(o + i) * 65 + o + m], col[m], s);      // col[m] = 0 for m < cc
      const float dg = t[(o + i) * 65 + o + i];
      col[i] = (i >= cc && dg > 0.0f) ? s / dg : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[(o + i) * 65 + o + cc] = col[i];
  }
  __syncthreads();
  // off-diagonal sub-blocks by block distance: X_ab = -X_aa (sum_{b <= m < a} L_am X_mb); the two
  // products are parked in the (unused) upper triangles of x and t
  for (int dist = 1; dist < 4; ++dist) {
    for (int e = tid; e < (4 - dist) * 256; e += 256) {
      const int pair = e >> 8, el = e & 255, r = el >> 4, cc = el & 15;
      const int a = pair + dist, b = pair;
      float s = 0.0f;
      for (int mb = b; mb < a; ++mb)
#pragma unroll
        for (int m = 0; m < 16; ++m) s = fmaf(t[(16 * a + r) * 65 + 16 * mb + m], x[(16 * mb + m) * 65 + 16 * b + cc], s);
      x[(16 * b + r) * 65 + 16 * a + cc] = s;
    }
    __syncthreads();
    for (int e = tid; e < (4 - dist) * 256; e += 256) {
      const int pair = e >> 8, el = e & 255, r = el >> 4, cc = el & 15;
      const int a = pair + dist, b = pair;
      float s = 0.0f;
#pragma unroll
      for (int m = 0; m < 16; ++m) s = fmaf(x[(16 * a + r) * 65 + 16 * a + m], x[(16 * b + m) * 65 + 16 * a + cc], s);
      t[(16 * b + r) * 65 + 16 * a + cc] = -s;
    }
    __syncthreads();
    for (int e = tid; e < (4 - dist) * 256; e += 256) {
      const int pair = e >> 8, el = e & 255, r = el >> 4, cc = el & 15;
      const int a = pair + dist, b = pair;
      x[(16 * a + r) * 65 + 16 * b + cc] = t[(16 * b + r) * 65 + 16 * a + cc];
      x[(16 * b + r) * 65 + 16 * a + cc] = 0.0f;
    }
    __syncthreads();
  }
}

// L_jj (upper part zero) -> F, Linv_j -> linv + j * 4096
__device__ __forceinline__ void store_factor(const float* __restrict__ t, const float* __restrict__ x, float* __restrict__ F,
                                             int kp, int j, float* __restrict__ linv) {
  float* const blk = F + (int64_t)(kRB * j) * kp + kRB * j;
  float* const out = linv + (int64_t)j * kRB * kRB;
  for (int e = threadIdx.x; e < kRB * kRB; e += 256) {
    const int r = e >> 6, cc = e & 63;
    blk[(int64_t)r * kp + cc] = cc <= r ? t[r * 65 + cc] : 0.0f;
    out[e] = x[r * 65 + cc];
  }
}

// the first diagonal block (the later ones are factored by the trailing launch that completes them)
__global__ __launch_bounds__(256) void chol_first_kernel(const float* __restrict__ S, float* __restrict__ F, int kp,
                                                         float* __restrict__ linv, int* __restrict__ info) {
  __shared__ float t[kRB * 65];
  __shared__ float x[kRB * 65];
  load_block(S, kp, t, threadIdx.x);
  __syncthreads();
  diag_factor(t, x, 0, info);
  store_factor(t, x, F, kp, 0, linv);
}

// ---------------------------------------------------------------------------
// step j: blocks (i, c), i >= c > j over the kp/64 + dp/64 row blocks.  The workgroup of block
// (j+1, j+1) goes on to factor it (look-ahead: the 64 dependent pivots of the next diagonal
// block overlap the rest of this step's trailing update instead of being a launch of their own).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_step_kernel(float* __restrict__ S, float* __restrict__ F, int kp, int nrb,
                                                        int ncb, int j, float* __restrict__ linv, int* __restrict__ info) {
  // enumerate the lower-trapezoid blocks: c in (j, ncb), i in [c, nrb); block 0 = (j+1, j+1)
  int bid = blockIdx.x, c = j + 1;
  while (bid >= nrb - c) { bid -= nrb - c; ++c; }
  const int i = c + bid;
  __shared__ float ta[kRB * 65], tb[kRB * 65], tl[kRB * 65];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  load_block(linv + (int64_t)j * kRB * kRB, kRB, tl, tid);
  load_block(S + (int64_t)(kRB * i) * kp + kRB * j, kp, ta, tid);
  if (c != i) load_block(S + (int64_t)(kRB * c) * kp + kRB * j, kp, tb, tid);
  __syncthreads();
  f32x4 pi[4] = {}, pc[4] = {};
  mma64_nt(ta, tl, pi, lane, w);                           // P_i = S_ij Linv^T
  if (c != i) mma64_nt(tb, tl, pc, lane, w);               // P_c = S_cj Linv^T
  __syncthreads();
  acc_to_tile(pi, ta, lane, w);
  if (c != i) acc_to_tile(pc, tb, lane, w);
  __syncthreads();
  const int l15 = lane & 15, q = lane >> 4;
  if (c == j + 1) {                                        // this block column also stores the panel L_ij
    float* const dst = F + (int64_t)(kRB * i) * kp + kRB * j;
    for (int e = tid; e < kRB * kRB; e += 256) dst[(int64_t)(e >> 6) * kp + (e & 63)] = ta[(e >> 6) * 65 + (e & 63)];
  }
  f32x4 upd[4] = {};
  mma64_nt(ta, c != i ? tb : ta, upd, lane, w);            // P_i P_c^T
  float* const blk = S + (int64_t)(kRB * i) * kp + kRB * c;
  if (i == j + 1) {                                        // (j+1, j+1): finish the block in LDS and factor it
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 16 * w + 4 * q + rg, cc = 16 * nb + l15;
        tl[r * 65 + cc] = blk[(int64_t)r * kp + cc] - upd[nb][rg];
      }
    __syncthreads();
    diag_factor(tl, tb, kRB * (j + 1), info);
    store_factor(tl, tb, F, kp, j + 1, linv);
    return;
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float* const pp = blk + (int64_t)(16 * w + 4 * q + rg) * kp + 16 * nb + l15;
      *pp -= upd[nb][rg];
    }
}

// the last block column has no trailing part: only its panel L_ij = S_ij Linv^T for i > j
__global__ __launch_bounds__(256) void chol_panel_kernel(const float* __restrict__ S, float* __restrict__ F, int kp, int j,
                                                         const float* __restrict__ linv) {
  const int i = j + 1 + blockIdx.x;
  __shared__ float ta[kRB * 65], tl[kRB * 65];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  load_block(linv + (int64_t)j * kRB * kRB, kRB, tl, tid);
  load_block(S + (int64_t)(kRB * i) * kp + kRB * j, kp, ta, tid);
  __syncthreads();
  f32x4 pi[4] = {};
  mma64_nt(ta, tl, pi, lane, w);
  const int l15 = lane & 15, q = lane >> 4;
  float* const blk = F + (int64_t)(kRB * i) * kp + kRB * j;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) blk[(int64_t)(16 * w + 4 * q + rg) * kp + 16 * nb + l15] = pi[nb][rg];
}

// ---------------------------------------------------------------------------
// backward substitution V L = Y^T for 16 rows of V per workgroup; Y^T = rows kp.. of F.
// Left-looking over the block columns j = NB-1 .. 0:  V_j = (Y_j - sum_{i>j} V_i L_ij) Linv_j.
// The NB (NB+1) / 2 tiles L_ij / Linv_j do not depend on V, so they stream through a ring of
// LDS buffers by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 4 tile rows per instruction, each
// wave fetches its 16 rows), `ring` tiles deep, one barrier per tile.  A 1 KiB piece lands
// lane-linear; pieces are 1040 B apart and the contraction index of MFMA step s is 16 q + s
// (q = lane / 16), which makes the B-operand reads conflict-free without a transpose.
// ---------------------------------------------------------------------------
constexpr int kPiece = 260;                                // floats between 4-row pieces (1024 B + 16 B pad)
constexpr int kTileFloats = 16 * kPiece;                   // 16,640 B per ring slot

__global__ __launch_bounds__(256) void ridge_backward_kernel(const float* __restrict__ F, const float* __restrict__ linv,
                                                             float* __restrict__ V, int64_t ldv, int k, int d, int KP,
                                                             int ring) {
  const int NB = KP / kRB, LDV = KP + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const tiles = sm;                                 // [ring][16][260]
  float* const v = tiles + ring * kTileFloats;             // [16][KP + 1]  the strip of V (starts as Y^T)
  float* const tt = v + 16 * LDV;                          // [16][65]      Y_j - sum ...
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int r0 = 16 * blockIdx.x;
  for (int e = tid; e < 16 * (KP / 4); e += 256) {
    const int r = e / (KP / 4), c4 = (e % (KP / 4)) * 4;
    const f32x4 x4 = *reinterpret_cast<const f32x4*>(F + (int64_t)(KP + r0 + r) * KP + c4);
    float* const dst = v + r * LDV + c4;
    dst[0] = x4[0]; dst[1] = x4[1]; dst[2] = x4[2]; dst[3] = x4[3];
  }
  // this wave's four pieces of tile (j, i): rows 16w + 4p .. +3 of L_ij (i > j) or Linv_j (i == j)
  auto issue = [&](int j, int i, int slot) {
    const float* src;
    unsigned ld;
    if (i == j) { src = linv + (int64_t)j * kRB * kRB; ld = kRB; }
    else { src = F + (int64_t)(kRB * i) * KP + kRB * j; ld = (unsigned)KP; }
    lds_char* const dst = (lds_char*)sm + slot * (kTileFloats * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned voff = ((unsigned)(16 * w + 4 * p + (lane >> 4)) * ld + 4u * (lane & 15)) * 4u;
      sp::dma_piece(src, voff, dst + (4 * w + p) * (kPiece * 4));
    }
  };
  auto next = [&](int& j, int& i) {                        // order: (NB-1, NB-1), then per j: i = NB-1 .. j+1, j
    if (i > j) --i;
    else { --j; i = NB - 1; }
  };
  const int total = NB * (NB + 1) / 2;
  int ij = NB - 1, ii = NB - 1, issued = 0;                // issue cursor
  int cj = NB - 1, ci = NB - 1;                            // consume cursor
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the strip's loads are the compiler's; DMA counts from here
  for (; issued < ring - 1 && issued < total; ++issued) { issue(ij, ii, issued % ring); next(ij, ii); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int n = 0; n < total; ++n) {
    if (ring == 1) {
      __syncthreads();                                     // tile n - 1 consumed
      issue(cj, ci, 0);
      LASSO_WAIT_VMCNT(0);
      __syncthreads();
    } else {
      const int ahead = issued - n - 1;                    // tiles in flight behind tile n: 0 .. ring - 2
      if (ahead >= 2) LASSO_WAIT_VMCNT(8);
      else if (ahead == 1) LASSO_WAIT_VMCNT(4);
      else LASSO_WAIT_VMCNT(0);
      __syncthreads();                                     // tile n landed (all waves' pieces), tile n - 1 consumed
      if (issued < total) { issue(ij, ii, issued % ring); next(ij, ii); ++issued; }
    }
    const float* const tl = tiles + (n % ring) * kTileFloats;
    if (ci > cj) {                                         // acc += V_i L_ij
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float av = v[l15 * LDV + kRB * ci + 16 * q + s];
        const float bv = tl[(4 * q + (s >> 2)) * kPiece + (s & 3) * kRB + 16 * w + l15];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
      }
    } else {                                               // V_j = (Y_j - acc) Linv_j
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, c = 16 * w + l15;
        tt[r * 65 + c] = v[r * LDV + kRB * cj + c] - acc[rg];
      }
      __syncthreads();
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float av = tt[l15 * 65 + 16 * q + s];
        const float bv = tl[(4 * q + (s >> 2)) * kPiece + (s & 3) * kRB + 16 * w + l15];
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, o, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) v[(4 * q + rg) * LDV + kRB * cj + 16 * w + l15] = o[rg];
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    next(cj, ci);
  }
  __syncthreads();
  for (int e = tid; e < 16 * KP; e += 256) {
    const int r = e / KP, c = e % KP;
    if (r0 + r < d && c < k) V[(int64_t)(r0 + r) * ldv + c] = v[r * LDV + c];
  }
}

// ---------------------------------------------------------------------------
// kp > 2048 (the 16-row strip of V no longer fits LDS): right-looking back substitution, one
// launch per block column j from last to first, blocks (rb, c), c <= j:  V_j = Y_j Linv_j
// (recomputed per block); c == j stores it, c < j applies Y_c -= V_j L_jc in place in F.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_block_t(const float* __restrict__ src, int64_t ld, float* __restrict__ t, int tid) {
  for (int e = tid; e < kRB * kRB / 4; e += 256) {         // t[c][r] = src[r][c]
    const int r = e / 16, c4 = (e % 16) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + c4);
    t[(c4 + 0) * 65 + r] = v[0]; t[(c4 + 1) * 65 + r] = v[1]; t[(c4 + 2) * 65 + r] = v[2]; t[(c4 + 3) * 65 + r] = v[3];
  }
}

__global__ __launch_bounds__(256) void ridge_backstep_kernel(float* __restrict__ F, const float* __restrict__ linv,
                                                             float* __restrict__ V, int64_t ldv, int k, int d, int kp, int j) {
  const int rb = blockIdx.x, c = blockIdx.y;
  __shared__ float ta[kRB * 65], tb[kRB * 65], tl[kRB * 65];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  float* const yrow = F + (int64_t)(kp + kRB * rb) * kp;
  load_block(yrow + kRB * j, kp, ta, tid);
  load_block_t(linv + (int64_t)j * kRB * kRB, kRB, tl, tid);
  if (c < j) load_block_t(F + (int64_t)(kRB * j) * kp + kRB * c, kp, tb, tid);
  __syncthreads();
  f32x4 vj[4] = {};
  mma64_nt(ta, tl, vj, lane, w);                           // V_j = Y_j Linv_j
  if (c == j) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = kRB * rb + 16 * w + 4 * q + rg, cc = kRB * j + 16 * nb + l15;
        if (r < d && cc < k) V[(int64_t)r * ldv + cc] = vj[nb][rg];
      }
    return;
  }
  __syncthreads();
  acc_to_tile(vj, ta, lane, w);
  __syncthreads();
  f32x4 upd[4] = {};
  mma64_nt(ta, tb, upd, lane, w);                          // V_j L_jc
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) yrow[(int64_t)(16 * w + 4 * q + rg) * kp + kRB * c + 16 * nb + l15] -= upd[nb][rg];
}

hipError_t backward_k(const float* F, const float* linv, float* V, int64_t ldv, int k, int d, int kp, int dp, hipStream_t st) {
  if (kp > 2048) {
    for (int j = kp / kRB - 1; j >= 0; --j)
      hipLaunchKernelGGL(ridge_backstep_kernel, dim3(dp / kRB, j + 1), dim3(256), 0, st, const_cast<float*>(F), linv, V, ldv, k,
                         d, kp, j);
    return hipGetLastError();
  }
  const size_t fixed = (size_t)(16 * (kp + 1) + 16 * 65) * 4, budget = 160 * 1024 - 512;
  int ring = (int)std::min<size_t>(4, (budget - fixed) / (kTileFloats * 4));
  if (ring < 1) return hipErrorInvalidValue;
  const size_t lds = fixed + (size_t)ring * kTileFloats * 4;
  const void* fn = reinterpret_cast<const void*>(&ridge_backward_kernel);
  if (hipError_t e = ensure_dynamic_lds(fn, 160 * 1024 - 256); e != hipSuccess) return e;
  hipLaunchKernelGGL(ridge_backward_kernel, dim3(dp / 16), dim3(256), lds, st, F, linv, V, ldv, k, d, kp, ring);
  return hipGetLastError();
}

}  // namespace

size_t ridge_workspace_bytes(int64_t d, int64_t k) {
  const size_t kp = (size_t)(k + kRB - 1) / kRB * kRB, dp = (size_t)(d + kRB - 1) / kRB * kRB;
  return 2 * (kp + dp) * kp * 4 + kp * kRB * 4 + 256;
}

// V [d][k] (ldv) = ((A + lam I)^-1 B)^T;  info_dev: 0, or 1 + the index of the first non-positive pivot
hipError_t launch_ridge_solve(const float* A, const float* B, float* V, int64_t ldv, int d, int k, float lam,
                              void* workspace, int* info_dev, hipStream_t st) {
  const int kp = (k + kRB - 1) / kRB * kRB, dp = (d + kRB - 1) / kRB * kRB;
  if (kp > kRidgeMaxK) return hipErrorInvalidValue;
  float* const S = (float*)workspace;
  float* const F = S + (size_t)(kp + dp) * kp;             // the factor: L (lower blocks) over Y^T = B^T L^-T
  float* const linv = F + (size_t)(kp + dp) * kp;
  hipError_t e = hipMemsetAsync(info_dev, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ridge_setup_kernel, dim3(1024), dim3(256), 0, st, A, B, S, k, d, kp, dp, lam);
  const int ncb = kp / kRB, nrb = (kp + dp) / kRB;
  hipLaunchKernelGGL(chol_first_kernel, dim3(1), dim3(256), 0, st, S, F, kp, linv, info_dev);
  for (int j = 0; j < ncb; ++j) {
    if (j + 1 < ncb) {
      int blocks = 0;
      for (int c = j + 1; c < ncb; ++c) blocks += nrb - c;
      hipLaunchKernelGGL(chol_step_kernel, dim3(blocks), dim3(256), 0, st, S, F, kp, nrb, ncb, j, linv, info_dev);
    } else {
      hipLaunchKernelGGL(chol_panel_kernel, dim3(nrb - ncb), dim3(256), 0, st, S, F, kp, j, linv);
    }
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return backward_k(F, linv, V, ldv, k, d, kp, dp, st);
}

}  // namespace lasso
