// Device-side building blocks shared by the tile kernels (fista_tile_sp.hip, fista_splitk.hip,
// objective.hip, backtrack.hip): LDS layouts, the per-wave LDS-DMA ring that streams
// W / W^T from L2, and the MFMA GEMM-1 loop  acc = A_tile * W^T.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "lasso_kernels.h"
#include "static_for.hpp"

namespace lasso {

typedef float f32x4 __attribute__((ext_vector_type(4)));
using lds_void_ptr = __attribute__((address_space(3))) void*;
using lds_char = __attribute__((address_space(3))) char;
using lds_f32 = __attribute__((address_space(3))) float;
using lds_f32x4 = __attribute__((address_space(3))) f32x4;

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0]|[15:14], expcnt[6:4], lgkmcnt[11:8])
#ifdef LASSO_ABL_NOVMWAIT   // timing ablation only (results invalid)
#define LASSO_WAIT_VMCNT(n)
#else
#define LASSO_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#endif
#define LASSO_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
// Raise the wave's issue priority around an MFMA cluster: the two waves sharing a SIMD
// then alternate (one bursts MFMAs while the other does its LDS reads / DMA issue)
// instead of converging to lockstep and idling the matrix pipe together.
#ifndef LASSO_SETPRIO
#define LASSO_SETPRIO 1
#endif
#if LASSO_SETPRIO
#define LASSO_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define LASSO_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define LASSO_PRIO_HI()
#define LASSO_PRIO_LO()
#endif

constexpr int kStepBytes = 4096;               // 32 rows x 128 B
constexpr int kRingBytesPerWave = 2 * kStepBytes;

// byte offset of element (row, col) inside a swizzled [16][LD] fp32 LDS tile:
// 16-byte chunk index is XORed with the row in its low 4 bits.
template <int LD>
__device__ __forceinline__ int tile_off(int row, int col) {
  const int chunk = col >> 2;
  return row * (LD * 4) + ((chunk ^ (row & 15)) << 4) + ((col & 3) << 2);
}

// One ring step = 4 LDS-DMA instructions (1 KiB each, lane-linear in LDS), issued
// from inline asm so the address form is exactly  saddr(SGPR pair) + voffset(VGPR,
// unsigned bytes) + imm  and nothing 64-bit is precomputed per step.  hipcc does not
// count these in its own s_waitcnt bookkeeping; every consumer below waits with an
// explicit counted vmcnt (DMA returns in issue order).  M0 (the LDS destination) is
// saved/restored inside the statement.  NOTE: the instruction's immediate offset is
// added to the LDS address as well as to the global address, so it stays 0 and the
// per-step advance goes into the SGPR base.
__device__ __forceinline__ void dma_step(const float* src, const unsigned (&voff)[4],
                                         lds_char* slot) {
#ifdef LASSO_ABL_NODMA      // timing ablation only (results invalid)
  return;
#endif
  const unsigned lds_addr = (unsigned)(uintptr_t)slot;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %5 offset:0\n\t"
      "s_add_u32 m0, %6, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5 offset:0\n\t"
      "s_add_u32 m0, %6, 0x800\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %5 offset:0\n\t"
      "s_add_u32 m0, %6, 0xc00\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %5 offset:0\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(src), "s"(lds_addr)
      : "memory", "scc");
}

// Visit a 16-row tile of a row-major [n][k] matrix in groups of 4 consecutive columns:
// f(r, c, v) with c a multiple of 4 and v = src[row0+r][c..c+3] (0 beyond n / k).  Uses
// 16-byte loads when the layout allows (k, ld multiples of 4 and a 16-B aligned base),
// else masked scalar loads.  K is the padded tile width, NT the workgroup size.
template <int K, int NT, int ROWS = kTileM, bool BATCH = true, typename F>
__device__ __forceinline__ void visit_tile4(const float* __restrict__ src, int64_t ld, int row0, int n,
                                            int k, F&& f) {
  const bool vec = src && (k & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)src) & 15) == 0;
  if constexpr (BATCH && (ROWS * (K / 4)) % NT == 0 && ROWS * (K / 4) / NT <= 16) {
    // The usual case: ALL of the thread's 16-byte loads are issued before the first one is used (clamped
    // addresses, values masked afterwards).  As a rolled loop with the loads under their bounds checks this was
    // one memory round trip per piece -- eight in a row per 16 x 1024 tile.
    if (vec && n > 0 && k >= 4) {
      constexpr int ITER = ROWS * (K / 4) / NT;
      f32x4 v[ITER];
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int idx = threadIdx.x + NT * i, r = idx / (K / 4), c = (idx - r * (K / 4)) * 4;
        v[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(row0 + r, n - 1) * ld + min(c, k - 4));
      }
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int idx = threadIdx.x + NT * i, r = idx / (K / 4), c = (idx - r * (K / 4)) * 4;
        const bool ok = (row0 + r) < n && c < k;
        f(r, c, ok ? v[i] : (f32x4){0.f, 0.f, 0.f, 0.f});
      }
      return;
    }
  }
  for (int idx = threadIdx.x; idx < ROWS * (K / 4); idx += NT) {
    const int r = idx / (K / 4), c = (idx - r * (K / 4)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src && (row0 + r) < n) {
      const float* rp = src + (int64_t)(row0 + r) * ld;
      if (vec) {
        if (c < k) v = *reinterpret_cast<const f32x4*>(rp + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < k) v[e] = rp[c + e];
      }
    }
    f(r, c, v);
  }
}

// The same over TWO matrices with the same tile geometry (the line search's point p and gradient g):
// f(r, c, a, b).  All loads of both are in flight together when both layouts allow 16-byte loads.
template <int K, int NT, int ROWS = kTileM, typename F>
__device__ __forceinline__ void visit_tile4x2(const float* __restrict__ sa, int64_t lda, const float* __restrict__ sb,
                                              int64_t ldb, int row0, int n, int k, F&& f) {
  constexpr int ITER = ROWS * (K / 4) / NT;
  static_assert((ROWS * (K / 4)) % NT == 0 && ITER <= 16, "tile geometry");
  const bool va = sa && (k & 3) == 0 && (lda & 3) == 0 && (((uintptr_t)sa) & 15) == 0;
  const bool vb = sb && (k & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)sb) & 15) == 0;
  if (va && vb && n > 0 && k >= 4) {
    f32x4 a[ITER], b[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = threadIdx.x + NT * i, r = idx / (K / 4), c = (idx - r * (K / 4)) * 4;
      const int64_t rr = min(row0 + r, n - 1);
      const int cc = min(c, k - 4);
      a[i] = *reinterpret_cast<const f32x4*>(sa + rr * lda + cc);
      b[i] = *reinterpret_cast<const f32x4*>(sb + rr * ldb + cc);
    }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = threadIdx.x + NT * i, r = idx / (K / 4), c = (idx - r * (K / 4)) * 4;
      const bool ok = (row0 + r) < n && c < k;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      f(r, c, ok ? a[i] : zero, ok ? b[i] : zero);
    }
    return;
  }
  for (int idx = threadIdx.x; idx < ROWS * (K / 4); idx += NT) {
    const int r = idx / (K / 4), c = (idx - r * (K / 4)) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if ((row0 + r) < n) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < k) {
          if (sa) a[e] = sa[(int64_t)(row0 + r) * lda + c + e];
          if (sb) b[e] = sb[(int64_t)(row0 + r) * ldb + c + e];
        }
    }
    f(r, c, a, b);
  }
}

// store 4 consecutive columns of a row (masked), 16-B store when allowed
__device__ __forceinline__ void store_row4(float* __restrict__ dst, int64_t ld, int row, int n, int k,
                                           int c, const f32x4& v, bool vec) {
  if (row >= n) return;
  float* rp = dst + (int64_t)row * ld;
  if (vec) {
    if (c < k) *reinterpret_cast<f32x4*>(rp + c) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < k) rp[c + e] = v[e];
  }
}
__device__ __forceinline__ bool vec4_ok(const float* p, int64_t ld, int k) {
  return p && (k & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)p) & 15) == 0;
}

// LDS address of the 16-byte chunk holding columns c..c+3 (c % 4 == 0) of row r
template <int LD>
__device__ __forceinline__ int tile_chunk_off(int row, int c) {
  return row * (LD * 4) + (((c >> 2) ^ (row & 15)) << 4);
}

__device__ __forceinline__ float soft_threshold(float v, float lam) {
  // ATen softshrink: v>lam ? v-lam : (v<-lam ? v+lam : 0).  Evaluated as v - clamp(v,-lam,lam)
  // (v_med3_f32 + v_sub_f32): bit-identical for every non-NaN v -- v-lam and v-(-lam) are the
  // same IEEE operations and v-v is +0 -- in 2 VALU ops instead of 6.
  return __fsub_rn(v, __builtin_amdgcn_fmed3f(v, -lam, lam));
}


// Per-thread constants of the streaming scheme for a tile of M = 4096/D rows x D padded
// features (D = 256: 16 rows, the flagship shape; D = 128 / 64: 32 / 64 rows for small
// feature counts).  The 16 output blocks of GEMM-1 (M/16 row blocks x D/16 column blocks)
// are dealt two per wave: wave w works on row block rb = w / (D/32) and on the column
// pair cw = w % (D/32), i.e. it streams rows [32cw, 32cw+32) of Wp and -- for GEMM-2, same
// row block -- rows [cw*KW, (cw+1)*KW) of Wtp with KW = K/(D/32).
template <int K, int D = kFistaD>
struct TileCtx {
  static constexpr int WPR = D / 32;          // waves per row block
  static constexpr int KW = K / WPR;          // GEMM-2 output columns per wave
  int lane, wid, n, q;
  int rb, cw;                    // row block, column index inside the row block
  unsigned voff1[4], voff2[4];   // per-lane DMA source byte offsets (ld = K / ld = 256)
  int boff[2];                   // B-fragment byte offsets inside a ring slot
  int aoff[2][2];                // A-fragment byte offsets inside a [16][*] tile row
  lds_char* ring;                // this wave's 2-slot ring
  const float* w1;               // Wp  + 32*wid*K
  const float* w2;               // Wtp + (K/8)*wid*256

  __device__ __forceinline__ void init(const float* Wp, const float* Wtp, lds_char* rings) {
    const int tid = threadIdx.x;
    lane = tid & 63;
    wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    n = lane & 15;
    q = lane >> 4;
    // DMA instruction j writes LDS bytes [j*1024, j*1024+1024) lane-linearly = rows
    // 8j..8j+7 of the step tile, 8 lanes per 128-B row; the XOR swizzle is applied on
    // the SOURCE chunk.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      voff1[j] = (unsigned)(row * K + 4 * c) * 4u;
      voff2[j] = (unsigned)(row * D + 4 * c) * 4u;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) boff[ss] = n * 128 + (((4 * ss + q) ^ ((n >> 1) & 7)) << 4);
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) aoff[par][ss] = ((8 * par + 4 * ss + q) ^ n) << 4;
    ring = rings + wid * kRingBytesPerWave;
    rb = wid / WPR;
    cw = wid - rb * WPR;
    w1 = Wp + (size_t)(32 * cw) * K;
    w2 = Wtp + (size_t)(KW * cw) * D;
  }
};

// GEMM-2, pass PS:  g2[cb] = r_tile[16][256] * Wtp[wid*K/8 + 32*PS + 16*cb .. +16][256]^T.
// `rf` are the r fragments (A layout) of the whole tile; the ring is refilled two steps
// ahead from the W^T stream and, past its end, with steps 0/1 of the next GEMM-1.
template <int K, int PS>
__device__ __forceinline__ void gemm2_pass(const TileCtx<K>& c, const f32x4 (&rf)[kFistaD / 32][2],
                                           f32x4 (&g2)[2]) {
  constexpr int D = kFistaD;
  constexpr int T2 = D / 32;
  constexpr int NP = (K / kFistaWaves) / 32;
  static_for<T2>([&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    constexpr int U = PS * T2 + t;            // step index inside GEMM-2
    lds_char* const slot = c.ring + (U & 1) * kStepBytes;
    LASSO_WAIT_VMCNT(4);
    f32x4 b[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
        b[cb][ss] = *(const lds_f32x4*)(slot + cb * 2048 + c.boff[ss]);
    LASSO_WAIT_LGKM0();
    if constexpr (U + 2 < NP * T2) {
      constexpr int pn = (U + 2) / T2, tn = (U + 2) % T2;
      dma_step(c.w2 + (size_t)(32 * pn) * D + 32 * tn, c.voff2, slot);
    } else {
      dma_step(c.w1 + 32 * (U + 2 - NP * T2), c.voff1, slot);   // next GEMM-1, steps 0/1
    }
    LASSO_PRIO_HI();
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[t][ss][j], b[0][ss][j], g2[0], 0, 0, 0);
        g2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[t][ss][j], b[1][ss][j], g2[1], 0, 0, 0);
      }
    LASSO_PRIO_LO();
  });
}

// load all r fragments (A layout) of a swizzled [16][256] tile
template <int K, int D = kFistaD>
__device__ __forceinline__ void load_r_frags(const TileCtx<K, D>& c, lds_char* rt,
                                             f32x4 (&rf)[D / 32][2]) {
#pragma unroll
  for (int t = 0; t < D / 32; ++t)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
      rf[t][ss] = *(const lds_f32x4*)(rt + (16 * c.rb + c.n) * (D * 4) + (t >> 1) * 256 + c.aoff[t & 1][ss]);
}


// ---------------------------------------------------------------------------
// Software-pipelined step machinery (see fista_tile_sp.hip for the rationale)
// ---------------------------------------------------------------------------
namespace sp {


struct Frag {
  f32x4 b[2][2];   // [col-block][k-half]
  f32x4 a[2];      // [k-half]       (GEMM-1 only; GEMM-2 uses the r fragments)
};

template <class Ctx>
__device__ __forceinline__ void load_b(const Ctx& c, Frag& f, const lds_char* slot) {
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) f.b[cb][ss] = *(const lds_f32x4*)(slot + cb * 2048 + c.boff[ss]);
}

template <class Ctx>
__device__ __forceinline__ void load_a(const Ctx& c, Frag& f, const lds_char* row_chunk, int par) {
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) f.a[ss] = *(const lds_f32x4*)(row_chunk + c.aoff[par][ss]);
}

// MFMAs number [LO, HI) of the 16 of a step (order: k-half, j, col-block)
template <int LO, int HI>
__device__ __forceinline__ void mfma_range(f32x4 (&acc)[2], const f32x4 (&a)[2], const Frag& f) {
  static_for<HI - LO>([&](auto i_c) {
    constexpr int i = LO + decltype(i_c)::value;
    constexpr int ss = i / 8, j = (i % 8) / 2, cb = i % 2;
    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][j], f.b[cb][ss][j], acc[cb], 0, 0, 0);
  });
}

// one LDS-DMA instruction (1 KiB = rows 8j..8j+7 of a step tile); see dma_step()
__device__ __forceinline__ void dma_piece(const float* src, unsigned voff, lds_char* dst) {
#ifdef LASSO_ABL_NODMA      // timing ablation only (results invalid)
  return;
#endif
  const unsigned lds_addr = (unsigned)(uintptr_t)dst;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:0\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(src), "s"(lds_addr)
      : "memory");
}

// pin the issue order at this point (hipcc otherwise sinks the prefetching ds_reads down
// to their first use and re-exposes the LDS latency the pipeline is there to hide)
#define LASSO_PIN() __builtin_amdgcn_sched_barrier(0)


constexpr int kHead = 4;   // MFMAs issued before the mid-step "slot free -> DMA refill" point

// MFMAs 0..15 of one step on fragments `f` with the refill of `slot` (4 LDS-DMA pieces)
// spread between them; e0..e3 are optional epilogue stages run in the four gaps.
template <typename E0, typename E1, typename E2, typename E3>
__device__ __forceinline__ void step_body(f32x4 (&acc)[2], const f32x4 (&a)[2], const Frag& f,
                                          const float* src, const unsigned (&voff)[4], lds_char* slot,
                                          E0&& e0, E1&& e1, E2&& e2, E3&& e3) {
  LASSO_PIN();
  mfma_range<0, kHead>(acc, a, f);
  LASSO_PIN();
  LASSO_WAIT_LGKM0();                 // the prefetching ds_reads have returned: slot is free
  dma_piece(src, voff[0], slot);
  LASSO_PIN();
  e0();
  mfma_range<4, 7>(acc, a, f);
  LASSO_PIN();
  dma_piece(src, voff[1], slot + 1024);
  LASSO_PIN();
  e1();
  mfma_range<7, 10>(acc, a, f);
  LASSO_PIN();
  dma_piece(src, voff[2], slot + 2048);
  LASSO_PIN();
  e2();
  mfma_range<10, 13>(acc, a, f);
  LASSO_PIN();
  dma_piece(src, voff[3], slot + 3072);
  LASSO_PIN();
  e3();
  mfma_range<13, 16>(acc, a, f);
  LASSO_PIN();
}
__device__ __forceinline__ void no_stage() {}


}  // namespace sp

// Pipelined GEMM-1:  acc[cb] += A_tile[16][K] * Wp[32*wid + 16*cb .. +16][K]^T (cb = 0,1).
// Pre: ring slots 0/1 hold (or have in flight) W steps 0/1 of this wave; post: they have
// `tail0` / `tail1` (+ tail_voff) in flight.
//   acc[cb] += A_tile[16][K] * Wp[32*wid + 16*cb .. +16][K]^T
// Pre:  ring slots 0/1 hold (or have in flight) W steps 0/1 of this wave.
// Post: ring slots 0/1 have `tail0` / `tail1` (+ tail_voff) in flight.
// EXTRA: vector-memory operations (loads, stores) the caller issued AFTER the ring's steps 0/1 went out and before this
// call.  vmcnt retires in order, so "step 0 / step 1 have landed" is then vmcnt(4 + EXTRA), not vmcnt(4): the first
// two steps run while those operations are still in flight; from step 2 on the waits are the usual ones (and do
// imply that the caller's operations have completed).
template <int K, int EXTRA = 0>
__device__ __forceinline__ void gemm1_stream_sp(const TileCtx<K>& c, lds_char* at, f32x4 (&acc)[2],
                                                const float* tail0, const float* tail1,
                                                const unsigned (&tail_voff)[4]) {
  constexpr int S1 = K / 32;
  static_assert(S1 % 2 == 0 && S1 >= 6, "geometry");
  static_assert(EXTRA >= 0 && 8 + EXTRA <= 63, "vmcnt is a 6-bit counter");
  lds_char* const slot0 = c.ring;
  lds_char* const slot1 = c.ring + kStepBytes;
  const lds_char* const arow = at + c.n * (K * 4);
  sp::Frag X, Y;
  // prologue: fragments of step 0 into X, slot0 refilled with step 2
  LASSO_WAIT_VMCNT(4 + EXTRA);
  sp::load_b(c, X, slot0);
  sp::load_a(c, X, arow, 0);
  LASSO_WAIT_LGKM0();
  dma_step(c.w1 + 64, c.voff1, slot0);
  auto nothing = [] {};
  // MODE 0: both refills from given sources; 1: only the even step refills (odd = last step)
  auto trip = [&](int s2, const float* srcE, const unsigned (&voffE)[4], const float* srcO,
                  const unsigned (&voffO)[4], auto last_c, auto extra_c) {
    constexpr bool last = decltype(last_c)::value;
    LASSO_WAIT_VMCNT(4 + decltype(extra_c)::value);
    sp::load_b(c, Y, slot1);
    sp::load_a(c, Y, arow + s2 * 256, 1);
    sp::step_body(acc, X.a, X, srcE, voffE, slot1, nothing, nothing, nothing, nothing);
    if constexpr (!last) {
      LASSO_WAIT_VMCNT(4);
      sp::load_b(c, X, slot0);
      sp::load_a(c, X, arow + (s2 + 1) * 256, 0);
      sp::step_body(acc, Y.a, Y, srcO, voffO, slot0, nothing, nothing, nothing, nothing);
    } else {
      LASSO_PIN();
      sp::mfma_range<0, 16>(acc, Y.a, Y);
      LASSO_PIN();
    }
  };
  using F = std::false_type;
  using T = std::true_type;
  using E0 = std::integral_constant<int, 0>;
  if constexpr (EXTRA > 0) {               // the first trip on its own: its even step still has the caller's operations in front
    trip(0, c.w1 + 96, c.voff1, c.w1 + 128, c.voff1, F{}, std::integral_constant<int, EXTRA>{});
#pragma unroll 1
    for (int s2 = 1; s2 < S1 / 2 - 2; ++s2)
      trip(s2, c.w1 + 64 * s2 + 96, c.voff1, c.w1 + 64 * s2 + 128, c.voff1, F{}, E0{});
  } else {
#pragma unroll 1
    for (int s2 = 0; s2 < S1 / 2 - 2; ++s2)
      trip(s2, c.w1 + 64 * s2 + 96, c.voff1, c.w1 + 64 * s2 + 128, c.voff1, F{}, E0{});
  }
  // steps S1-4 / S1-3: refills = W step S1-1 and tail0;  steps S1-2 / S1-1: refill = tail1 / none
  trip(S1 / 2 - 2, c.w1 + 32 * (S1 - 1), c.voff1, tail0, tail_voff, F{}, E0{});
  trip(S1 / 2 - 1, tail1, tail_voff, tail1, tail_voff, T{}, E0{});
}

// gemm1_stream_sp with all S1 steps written out and a caller's stage `hook(integral_constant<int, i>)`, i = 0 .. 4 S1 - 1,
// run in the gaps between the MFMAs of a step (four per step, the slots of the FISTA kernel's staged epilogue):
// element-wise work that is independent of this product -- the NEXT line-search candidate -- issued where the matrix
// pipe is busy anyway instead of in front of a barrier where it idles.  Same MFMA order, same ring protocol.
template <int K, typename Hook>
__device__ __forceinline__ void gemm1_stream_hooked(const TileCtx<K>& c, lds_char* at, f32x4 (&acc)[2],
                                                    const float* tail0, const float* tail1,
                                                    const unsigned (&tail_voff)[4], Hook&& hook) {
  constexpr int S1 = K / 32;
  static_assert(S1 % 2 == 0 && S1 >= 6, "geometry");
  lds_char* const slot0 = c.ring;
  lds_char* const slot1 = c.ring + kStepBytes;
  const lds_char* const arow = at + c.n * (K * 4);
  sp::Frag X, Y;
  LASSO_WAIT_VMCNT(4);
  sp::load_b(c, X, slot0);
  sp::load_a(c, X, arow, 0);
  LASSO_WAIT_LGKM0();
  dma_step(c.w1 + 64, c.voff1, slot0);
  static_for<S1 / 2>([&](auto s2_c) {
    constexpr int s2 = decltype(s2_c)::value;
    constexpr bool last = s2 == S1 / 2 - 1;
    auto h = [&](auto i_c) { return [&] { hook(std::integral_constant<int, 8 * s2 + decltype(i_c)::value>{}); }; };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
    LASSO_WAIT_VMCNT(4);
    sp::load_b(c, Y, slot1);
    sp::load_a(c, Y, arow + s2 * 256, 1);
    if constexpr (s2 < S1 / 2 - 2)
      sp::step_body(acc, X.a, X, c.w1 + 64 * s2 + 96, c.voff1, slot1, h(I0{}), h(I1{}), h(I2{}), h(I3{}));
    else if constexpr (s2 == S1 / 2 - 2)
      sp::step_body(acc, X.a, X, c.w1 + 32 * (S1 - 1), c.voff1, slot1, h(I0{}), h(I1{}), h(I2{}), h(I3{}));
    else
      sp::step_body(acc, X.a, X, tail1, tail_voff, slot1, h(I0{}), h(I1{}), h(I2{}), h(I3{}));
    if constexpr (!last) {
      LASSO_WAIT_VMCNT(4);
      sp::load_b(c, X, slot0);
      sp::load_a(c, X, arow + (s2 + 1) * 256, 0);
      if constexpr (s2 < S1 / 2 - 2)
        sp::step_body(acc, Y.a, Y, c.w1 + 64 * s2 + 128, c.voff1, slot0, h(I4{}), h(I5{}), h(I6{}), h(I7{}));
      else
        sp::step_body(acc, Y.a, Y, tail0, tail_voff, slot0, h(I4{}), h(I5{}), h(I6{}), h(I7{}));
    } else {
      LASSO_PIN();
      sp::mfma_range<0, 8>(acc, Y.a, Y);
      LASSO_PIN();
      h(I4{})(); h(I5{})();
      sp::mfma_range<8, 16>(acc, Y.a, Y);
      LASSO_PIN();
      h(I6{})(); h(I7{})();
    }
  });
}

// Pipelined GEMM-2 without an epilogue (round 5, bt_iter.hip): all NP passes of
//   g[ps][cb] = r_tile[16][256] * Wtp[wid*K/8 + 32*ps + 16*cb .. +16][256]^T
// with the fragment reads of step U+1 under the MFMAs of step U and the ring refill spread between them -- the step
// machinery of GEMM-1.  Same MFMA order per accumulator as gemm2_pass (bitwise the same g).
// Pre:  ring slots 0/1 have W^T steps 0/1 of this wave in flight (the tails of gemm1_stream_sp).
// Post: ring slots 0/1 have steps 0/1 of the next GEMM-1 (c.w1) in flight.
// EXTRA: as in gemm1_stream_sp (operations issued between the ring's steps 0/1 and this call).
template <int K, int EXTRA = 0>
__device__ __forceinline__ void gemm2_stream_sp(const TileCtx<K>& c, const f32x4 (&rf)[kFistaD / 32][2],
                                                f32x4 (&g)[(K / kFistaWaves) / 32][2]) {
  constexpr int D = kFistaD;
  constexpr int T2 = D / 32;
  constexpr int NP = (K / kFistaWaves) / 32;
  constexpr int S2 = NP * T2;
  static_assert(S2 % 2 == 0 && S2 >= 4, "geometry");
  lds_char* const slot0 = c.ring;
  lds_char* const slot1 = c.ring + kStepBytes;
  sp::Frag X, Y;
  LASSO_WAIT_VMCNT(4 + EXTRA);
  sp::load_b(c, X, slot0);
  LASSO_WAIT_LGKM0();
  dma_step(c.w2 + (size_t)(32 * (2 / T2)) * D + 32 * (2 % T2), c.voff2, slot0);
  auto nothing = [] {};
  static_for<S2>([&](auto u_c) {
    constexpr int U = decltype(u_c)::value;
    constexpr int ps = U / T2, t = U % T2;
    sp::Frag& cur = (U & 1) ? Y : X;
    sp::Frag& nxt = (U & 1) ? X : Y;
    lds_char* const nslot = (U & 1) ? slot0 : slot1;     // slot of step U+1
    if constexpr (t == 0) {
      g[ps][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      g[ps][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (U + 1 < S2) {
      LASSO_WAIT_VMCNT(4 + (U == 0 ? EXTRA : 0));
      sp::load_b(c, nxt, nslot);
      if constexpr (U + 3 < S2) {
        constexpr int pn = (U + 3) / T2, tn = (U + 3) % T2;
        sp::step_body(g[ps], rf[t], cur, c.w2 + (size_t)(32 * pn) * D + 32 * tn, c.voff2, nslot, nothing, nothing,
                      nothing, nothing);
      } else {
        sp::step_body(g[ps], rf[t], cur, c.w1 + 32 * (U + 3 - S2), c.voff1, nslot, nothing, nothing, nothing, nothing);
      }
    } else {
      LASSO_PIN();
      sp::mfma_range<0, 16>(g[ps], rf[t], cur);
      LASSO_PIN();
    }
    // Anchor: both accumulator chains of a pass are "read" where the pass ends.  Without it instruction selection
    // (which sched_barrier does not bind) let the second chain float to its first real use, behind ALL passes, with
    // every B fragment it needs spilled on the way (392 scratch registers).
    if constexpr (t == T2 - 1) asm volatile("" :: "v"(g[ps][0]), "v"(g[ps][1]));
  });
}

// wave-wide sum (wave-uniform result), fixed order, on the ALU path: DPP row_shr 1,2,4,8
// leaves each 16-lane row's total in its last lane; four readlanes finish the job (the
// ds_bpermute route of __shfl_xor costs ~6 LDS round trips instead).
__device__ __forceinline__ float wave_sum(float x) {
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

}  // namespace lasso
