// General fp32 MFMA GEMM of the support paths:  C[m x nn] = C0 -/+ A B^T  with
// A [m x kk] (lda) and B [nn x kk] (ldb), both contiguous along the contraction.
// Users: the unfused FISTA path for shapes beyond the fused tile kernel (ista.py:72-73:
// r = y W^T - x and g = r W), U = B - A D^T of the Gram-form M-step (dict_learning.py:82),
// b = x W and S = I - W^T W of coordinate descent (coordinate_descent.py:19,22-23).
//
// BM x BN block per workgroup of 4 waves (2 x 2), each wave a (BM/2) x (BN/2) patch of
// 16x16 MFMA accumulators; the contraction runs in chunks of 32 floats that are staged
// global -> registers -> LDS (double buffered: the loads of chunk c+1 are in flight while
// chunk c feeds the matrix pipe) as [rows][128 B] with the 16-byte-chunk XOR swizzle of the
// FISTA ring, so every operand fragment is one conflict-free ds_read_b128 that serves four
// v_mfma_f32_16x16x4_f32.  128 x 128 blocks need 4 flop per byte of L2->LDS traffic less
// than 64 x 64 ones and are used whenever they still fill the chip.
// Roofline: MFMA-bound, 2*m*nn*kk flop (measured numbers: DESIGN.md 3.3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"
#include "static_for.hpp"

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz_off(int row, int chunk) {   // bytes inside a [rows][128 B] tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// rows of `src` [rows x kk] (ld) starting at r0, columns k0 + 4*chunk .. +3 -> v (zero outside)
template <bool VEC>
__device__ __forceinline__ f32x4 load_chunk4(const float* __restrict__ src, int64_t ld, int row, int rows,
                                             int kcol, int kk) {
  f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
  if (row < rows) {
    const float* p = src + (int64_t)row * ld + kcol;
    if constexpr (VEC) {
      if (kcol < kk) v = *(const f32x4*)p;          // kk % 4 == 0: the chunk is all in or all out
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (kcol + e < kk) v[e] = p[e];
    }
  }
  return v;
}

// Epilogue of the unfused FISTA path's second GEMM (EPI = true): the block of g = -(A B^T) never goes to
// memory -- the proximal step runs on the accumulators: z_next = softshrink(y - lr*g, lam), the block's
// sum |z - z_next|, y = z_next + c (z_next - z), z = z_next (ista.py:90,93,98-102; each product and sum rounded
// on its own like the reference's separate ATen ops, the arithmetic of generic_prox_kernel).  Z and Y are
// updated in place: every element belongs to exactly one block and neither is an operand of this GEMM.
struct ProxEpilogue {
  float* Z; int64_t ldz;
  float* Y; int64_t ldy;
  float lr, lam, coef;
  float* dpart;                 // [gridDim.y * gridDim.x] per-block sums of |z - z_next|
};

// one LDS-DMA instruction: 64 lanes x 16 bytes from src + voff (per lane) to the 1 KiB at LDS address `lds_addr`
// (lane-linear), no registers in between (see tile_device.hpp: dma_step)
__device__ __forceinline__ void gemm_dma_piece(const float* src, unsigned voff, unsigned lds_addr) {
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

// DMA = true (kk % 32 == 0, 16-byte aligned operands, row offsets of a block below 2 GiB): the chunks go global -> LDS
// by LDS-DMA -- each lane fetches the 16 bytes whose swizzled home is its lane-linear slot -- instead of through 8
// registers and 8 ds_writes per thread and chunk; same LDS image, same fragment reads, bitwise the same product.
template <int BM, int BN, bool VEC, bool EPI = false, bool DMA = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ B, int64_t ldb,
                                                         const float* __restrict__ C0, int64_t ldc0,
                                                         float* __restrict__ C, int64_t ldc, int m, int nn,
                                                         int kk, int add, ProxEpilogue ep = ProxEpilogue(),
                                                         int* __restrict__ zero_words = nullptr, int nzero = 0) {
  // (a caller's flag words cleared on the way: saves the launch behind this one a fill of its own)
  if (zero_words && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < nzero; i += 256) zero_words[i] = 0;
  constexpr int MI = BM / 32, NJ = BN / 32;          // 16x16 blocks per wave: MI x NJ
  constexpr int PA = BM / 32, PB = BN / 32;          // staged 16-byte chunks per thread and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sa = smem;                             // [2][BM * 128]
  char* const sb = smem + 2 * BM * 128;              // [2][BN * 128]
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, q = lane >> 4;
  f32x4 acc[MI][NJ] = {};
  // staging map: thread -> chunk (tid & 7) of rows (tid >> 3) + 32 h
  const int srow = tid >> 3, sch = tid & 7;
  f32x4 ga[PA], gb[PB];
  // (VEC: 16-byte buffer loads from descriptors of the block's rows of A and B, the offset out of range beyond the rows /
  // beyond kk -- reads 0 --, offsets opaque: no branch per chunk, the batch leaves as a batch; the launcher checks
  // ld BM 4 < 2^31)
  const int rows_a = min(BM, m - i0), rows_b = min(BN, nn - j0);
  auto rows_rsrc = [&](const float* base, int64_t ld, int r0, int rows) {
    const int64_t bytes = (int64_t)rows * ld * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (int64_t)r0 * ld), 0,
                                             (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ars = rows_rsrc(A, lda, i0, rows_a), brs = rows_rsrc(B, ldb, j0, rows_b);
  auto fetch = [&](int k0) {
    if constexpr (VEC) {
      const int kc = k0 + 4 * sch;
#pragma unroll
      for (int h = 0; h < PA; ++h) {
        unsigned o = (srow + 32 * h < rows_a && kc < kk) ? (unsigned)((srow + 32 * h) * (int)lda + kc) * 4u : 0xfffffff0u;
        asm volatile("" : "+v"(o));
        ga[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, o, 0, 0));
      }
#pragma unroll
      for (int h = 0; h < PB; ++h) {
        unsigned o = (srow + 32 * h < rows_b && kc < kk) ? (unsigned)((srow + 32 * h) * (int)ldb + kc) * 4u : 0xfffffff0u;
        asm volatile("" : "+v"(o));
        gb[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, o, 0, 0));
      }
    } else {
#pragma unroll
      for (int h = 0; h < PA; ++h) ga[h] = load_chunk4<VEC>(A, lda, i0 + srow + 32 * h, m, k0 + 4 * sch, kk);
#pragma unroll
      for (int h = 0; h < PB; ++h) gb[h] = load_chunk4<VEC>(B, ldb, j0 + srow + 32 * h, nn, k0 + 4 * sch, kk);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int h = 0; h < PA; ++h) *(f32x4*)(sa + buf * BM * 128 + swz_off(srow + 32 * h, sch)) = ga[h];
#pragma unroll
    for (int h = 0; h < PB; ++h) *(f32x4*)(sb + buf * BN * 128 + swz_off(srow + 32 * h, sch)) = gb[h];
  };
  // DMA: per-lane byte offsets inside the block's rows (row clamped to the last valid one: its results are dropped)
  unsigned va[PA], vb[PB];
  const int wdma = __builtin_amdgcn_readfirstlane(w);
  if constexpr (DMA) {
#pragma unroll
    for (int h = 0; h < PA; ++h) {
      const int r = srow + 32 * h, c = sch ^ ((r >> 1) & 7);
      va[h] = (unsigned)((int64_t)min(r, m - 1 - i0) * lda * 4 + c * 16);
    }
#pragma unroll
    for (int h = 0; h < PB; ++h) {
      const int r = srow + 32 * h, c = sch ^ ((r >> 1) & 7);
      vb[h] = (unsigned)((int64_t)min(r, nn - 1 - j0) * ldb * 4 + c * 16);
    }
  }
  auto dma = [&](int k0, int buf) {
    const float* const abase = A + (int64_t)i0 * lda + k0;
    const float* const bbase = B + (int64_t)j0 * ldb + k0;
    const unsigned la = (unsigned)(uintptr_t)(sa + buf * BM * 128) + (unsigned)(8 * wdma) * 128u;
    const unsigned lb = (unsigned)(uintptr_t)(sb + buf * BN * 128) + (unsigned)(8 * wdma) * 128u;
#pragma unroll
    for (int h = 0; h < PA; ++h) gemm_dma_piece(abase, va[h], la + 32 * 128 * h);
#pragma unroll
    for (int h = 0; h < PB; ++h) gemm_dma_piece(bbase, vb[h], lb + 32 * 128 * h);
  };
  if constexpr (DMA) {
    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's pieces are in LDS
  } else {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < kk; k0 += 32) {
    const bool more = k0 + 32 < kk;
    if (more) {
      if constexpr (DMA) dma(k0 + 32, buf ^ 1);
      else fetch(k0 + 32);
    }
    const char* const ta = sa + buf * BM * 128;
    const char* const tb = sb + buf * BN * 128;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      f32x4 a[MI], b[NJ];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)(ta + swz_off((BM / 2) * wr + 16 * mi + l15, 4 * ss + q));
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) b[nj] = *(const f32x4*)(tb + swz_off((BN / 2) * wc + 16 * nj + l15, 4 * ss + q));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], b[nj][j], acc[mi][nj], 0, 0, 0);
    }
    if constexpr (DMA) {
      __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0)
    } else {
      if (more) stash(buf ^ 1);
    }
    __syncthreads();
    buf ^= 1;
  }
  // Both epilogues through buffer descriptors of the tile's rows (offset out of range beyond m / nn: reads 0, store
  // dropped; offsets opaque): with the bounds as branches hipcc gave every element a branch of its own, each load its
  // s_waitcnt vmcnt(0) and -- at the joins of those lane-masked branches -- every STORE one too, so that a lane's 64 to
  // 128 stores each waited for the previous one's acknowledgement (round 5; the launcher checks ld BM 4 < 2^31).
  const int rows_valid = min(BM, m - i0);
  auto tile_rsrc = [&](const float* base, int64_t ld) {
    const int64_t bytes = base ? (int64_t)rows_valid * ld * 4 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base + (int64_t)i0 * ld : C), 0,
                                             (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
  };
  auto tile_off = [&](int rl, int cc, int64_t ld) {
    unsigned o = (rl < rows_valid && cc < nn) ? (unsigned)(rl * (int)ld + cc) * 4u : 0xfffffff0u;
    asm volatile("" : "+v"(o));
    return o;
  };
  if constexpr (EPI) {
    const __amdgpu_buffer_rsrc_t zrs = tile_rsrc(ep.Z, ep.ldz), yrs = tile_rsrc(ep.Y, ep.ldy);
    float dsum = 0.0f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      // all of a row block's z, y values in flight before the first is used
      float zo[NJ][4], yo[NJ][4];
      unsigned oz[NJ][4], oy[NJ][4];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int rl = (BM / 2) * wr + 16 * mi + 4 * q + rg, cc = j0 + (BN / 2) * wc + 16 * nj + l15;
          oz[nj][rg] = tile_off(rl, cc, ep.ldz);
          oy[nj][rg] = tile_off(rl, cc, ep.ldy);
          zo[nj][rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zrs, oz[nj][rg], 0, 0));
          yo[nj][rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, oy[nj][rg], 0, 0));
        }
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float g = 0.0f - acc[mi][nj][rg];
          const float v = __fsub_rn(yo[nj][rg], __fmul_rn(ep.lr, g));
          const float zn = __fsub_rn(v, __builtin_amdgcn_fmed3f(v, -ep.lam, ep.lam));
          dsum += oz[nj][rg] != 0xfffffff0u ? __builtin_fabsf(__fsub_rn(zo[nj][rg], zn)) : 0.0f;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__fadd_rn(zn, __fmul_rn(ep.coef, __fsub_rn(zn, zo[nj][rg])))),
                                                yrs, oy[nj][rg], 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(zn), zrs, oz[nj][rg], 0, 0);
        }
    }
    // block sum in a fixed order: lanes (butterfly over the wave), then the four waves
    __syncthreads();                                   // the staging buffers are free
    float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
    if (lane == 0) red[w] = dsum;
    __syncthreads();
    if (tid == 0) ep.dpart[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  } else {
    const __amdgpu_buffer_rsrc_t c0rs = tile_rsrc(C0, ldc0), crs = tile_rsrc(C, ldc);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float c0[NJ][4];
      unsigned oc[NJ][4];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int rl = (BM / 2) * wr + 16 * mi + 4 * q + rg, cc = j0 + (BN / 2) * wc + 16 * nj + l15;
          oc[nj][rg] = tile_off(rl, cc, ldc);
          c0[nj][rg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, tile_off(rl, cc, ldc0), 0, 0));   // (no C0: 0 records, reads 0)
        }
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(add ? c0[nj][rg] + acc[mi][nj][rg] : c0[nj][rg] - acc[mi][nj][rg]),
                                                crs, oc[nj][rg], 0, 0);
    }
  }
}

// the LDS-DMA form's preconditions beyond VEC (16-byte aligned rows): whole 32-float chunks and 32-bit lane offsets
#ifndef LASSO_GEMM_NODMA
static bool gemm_dma_ok(int64_t lda, int64_t ldb, int kk, int bm, int bn) {
  return kk % 32 == 0 && kk > 0 && lda * bm * 4 < ((int64_t)1 << 31) && ldb * bn * 4 < ((int64_t)1 << 31);
}
#else
static bool gemm_dma_ok(int64_t, int64_t, int, int, int) { return false; }
#endif

template <int BM, int BN, bool VEC>
hipError_t launch_tile(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C0,
                       int64_t ldc0, float* C, int64_t ldc, int m, int nn, int kk, int add,
                       hipStream_t stream, int* zero_words, int nzero) {
  constexpr int lds = 2 * (BM + BN) * 128;
  if (lds > 48 * 1024)
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, VEC>), lds);
        e != hipSuccess)
      return e;
  const dim3 grid((nn + BN - 1) / BN, (m + BM - 1) / BM);
  if constexpr (VEC) {
    if (gemm_dma_ok(lda, ldb, kk, BM, BN)) {
      if (lds > 48 * 1024)
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, true, false, true>), lds);
            e != hipSuccess)
          return e;
      hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, true, false, true>), grid, dim3(256), lds, stream, A, lda, B, ldb, C0,
                         ldc0, C, ldc, m, nn, kk, add, ProxEpilogue(), zero_words, nzero);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, VEC>), grid, dim3(256), lds, stream, A, lda, B, ldb, C0, ldc0, C,
                     ldc, m, nn, kk, add, ProxEpilogue(), zero_words, nzero);
  return hipGetLastError();
}

template <int BM, int BN, bool VEC>
hipError_t launch_tile_prox(const float* A, int64_t lda, const float* B, int64_t ldb, int m, int nn, int kk,
                            const ProxEpilogue& ep, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * 128;
  if (lds > 48 * 1024)
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, VEC, true>), lds);
        e != hipSuccess)
      return e;
  const dim3 grid((nn + BN - 1) / BN, (m + BM - 1) / BM);
  if constexpr (VEC) {
    if (gemm_dma_ok(lda, ldb, kk, BM, BN)) {
      if (lds > 48 * 1024)
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, true, true, true>), lds);
            e != hipSuccess)
          return e;
      hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, true, true, true>), grid, dim3(256), lds, stream, A, lda, B, ldb,
                         nullptr, 0, nullptr, 0, m, nn, kk, 0, ep);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, VEC, true>), grid, dim3(256), lds, stream, A, lda, B, ldb, nullptr, 0,
                     nullptr, 0, m, nn, kk, 0, ep);
  return hipGetLastError();
}

}  // namespace

static int gemm_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    cus = 1;
  return cus;
}

// Block shape of an [m x nn] product (sides 128 or 64): the least modelled time on `cus` compute units --
// blocks per CU (the padding of short sides is in the block count) x block area / (rate of the shape x a
// co-residency factor): 128 x 128 blocks need the fewest LDS reads per MFMA but only pay off while every CU still
// holds two workgroups at a time (one alone leaves its barriers uncovered); 784 output columns, for instance, are
// 224 blocks of 128 x 128 on 256 CUs -- 416 of 128 x 64 run 8 % faster.  The accumulation order of an output
// element does not depend on the shape: bitwise the same product either way.
static void gemm_blocks(int m, int nn, int* bm_out, int* bn_out) {
  const int cus = gemm_cus();
  double best = 1e300;
  int64_t best_blocks = 0;
  for (int bm : {128, 64})
    for (int bn : {128, 64}) {
      const int64_t blocks = (int64_t)((m + bm - 1) / bm) * ((nn + bn - 1) / bn);
      const int64_t per_cu = (blocks + cus - 1) / cus;
      const double rate = (bm == 128 && bn == 128) ? 1.0 : (bm == 64 && bn == 64) ? 0.78 : 0.9;
      const double t = (double)per_cu * bm * bn / (rate * (per_cu >= 2 ? 1.0 : 0.8));
      if (t < best * (1.0 - 1e-9) || (t <= best * (1.0 + 1e-9) && blocks < best_blocks)) {
        best = t; best_blocks = blocks; *bm_out = bm; *bn_out = bn;
      }
    }
}

// blocks the fused GEMM-2 + prox launch uses for an [m x nn] result (= partial sums it writes)
// 64 x 64: the epilogue of a block is a memory phase (its z, y: 4 x 64 x 64 floats in, as many out) during which the
// block's waves leave the matrix pipe idle, and the contraction in front of it is only d long -- five small workgroups
// per CU overlap those phases where two 128 x 128 ones could not (PMC: 62 % MFMA busy against 85 % for the plain
// product; 113 -> 123 TFLOP/s at n=16384, d=512, k=4096, 126 -> 130 at d=1024, equal at d=2048)
static void prox_blocks(int m, int nn, int* bm_out, int* bn_out) {
  (void)m; (void)nn;
  *bm_out = 64;
  *bn_out = 64;
  if (const char* e = getenv("LASSO_PROX_BLOCKS")) {        // A/B knob: "128x128", "128x64", "64x128", "64x64", "auto"
    int a = 0, b = 0;
    if (sscanf(e, "%dx%d", &a, &b) == 2 && (a == 64 || a == 128) && (b == 64 || b == 128)) { *bm_out = a; *bn_out = b; }
    else if (e[0] == 'a') gemm_blocks(m, nn, bm_out, bn_out);
  }
}

int gemm_nt_prox_parts(int m, int nn) {
  int bm, bn;
  prox_blocks(m, nn, &bm, &bn);
  return ((m + bm - 1) / bm) * ((nn + bn - 1) / bn);
}

// G = -(A B^T) consumed in the epilogue: Z, Y [m x nn] updated in place, dpart[gemm_nt_prox_parts(m, nn)] written
hipError_t launch_gemm_nt_prox(const float* A, int64_t lda, const float* B, int64_t ldb, float* Z, int64_t ldz,
                               float* Y, int64_t ldy, int m, int nn, int kk, float lr, float lam, float coef,
                               float* dpart, hipStream_t stream) {
  if (m <= 0 || nn <= 0) return hipSuccess;
  const bool vec = kk % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                   ((uintptr_t)B & 15) == 0;
  int bm, bn;
  prox_blocks(m, nn, &bm, &bn);
  // (32-bit buffer offsets inside a block's rows of every operand: a very wide row pitch takes the smaller blocks --
  // 64 rows reach a pitch of 2^23 floats -- before the launch is refused)
  const int64_t ldmax = std::max(std::max(lda, ldb), std::max(ldz, ldy));
  if (ldmax * 128 * 4 >= ((int64_t)1 << 31)) { bm = 64; bn = 64; }
  if (ldmax * 64 * 4 >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
  const ProxEpilogue ep = {Z, ldz, Y, ldy, lr, lam, coef, dpart};
#define LASSO_PROX_CASE(BM_, BN_)                                                                 \
  if (bm == BM_ && bn == BN_)                                                                      \
    return vec ? launch_tile_prox<BM_, BN_, true>(A, lda, B, ldb, m, nn, kk, ep, stream)          \
               : launch_tile_prox<BM_, BN_, false>(A, lda, B, ldb, m, nn, kk, ep, stream)
  LASSO_PROX_CASE(128, 128);
  LASSO_PROX_CASE(128, 64);
  LASSO_PROX_CASE(64, 128);
  LASSO_PROX_CASE(64, 64);
#undef LASSO_PROX_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_gemm_nt_sub(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C0,
                              int64_t ldc0, float* C, int64_t ldc, int m, int nn, int kk,
                              hipStream_t stream, int add, int* zero_words, int nzero) {
  if (m <= 0 || nn <= 0) return hipSuccess;
  const bool vec = kk % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                   ((uintptr_t)B & 15) == 0;
  int bm, bn;
  gemm_blocks(m, nn, &bm, &bn);
  // (32-bit buffer offsets inside a block's rows: a very wide row pitch takes blocks of fewer rows -- 32 rows reach a
  // pitch of 2^24 floats -- before the launch is refused)
  const int64_t ldmax = std::max(std::max(lda, ldb), std::max(ldc, C0 ? ldc0 : (int64_t)0));
  if (ldmax * 128 * 4 >= ((int64_t)1 << 31)) { bm = std::min(bm, 64); bn = std::min(bn, 64); }
  if (ldmax * 64 * 4 >= ((int64_t)1 << 31)) { bm = 32; bn = 32; }
  if (ldmax * 32 * 4 >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
  // Small products (U = B - A D^T of the M-step: 1024 x 256 outputs = 64 blocks of 64 x 64 on 256 CUs): 32-wide
  // sides until every CU has a workgroup.
  if (bm == 64 && bn == 64) {
    const int cus = gemm_cus();
    auto blocks = [&](int a, int b) { return (int64_t)((m + a - 1) / a) * ((nn + b - 1) / b); };
    if (blocks(64, 64) < cus && m > 32) bm = 32;
    if (blocks(bm, 64) < cus && nn > 32) bn = 32;
  }
#define LASSO_GEMM_CASE(BM_, BN_)                                                                              \
  if (bm == BM_ && bn == BN_)                                                                                   \
    return vec ? launch_tile<BM_, BN_, true>(A, lda, B, ldb, C0, ldc0, C, ldc, m, nn, kk, add, stream, zero_words, nzero)         \
               : launch_tile<BM_, BN_, false>(A, lda, B, ldb, C0, ldc0, C, ldc, m, nn, kk, add, stream, zero_words, nzero)
  LASSO_GEMM_CASE(128, 128);
  LASSO_GEMM_CASE(128, 64);
  LASSO_GEMM_CASE(64, 128);
  LASSO_GEMM_CASE(64, 64);
  LASSO_GEMM_CASE(32, 64);
  LASSO_GEMM_CASE(64, 32);
  LASSO_GEMM_CASE(32, 32);
#undef LASSO_GEMM_CASE
  return hipErrorInvalidValue;
}

}  // namespace lasso
