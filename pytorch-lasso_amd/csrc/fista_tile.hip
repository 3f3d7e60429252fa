// Fused persistent FISTA/ISTA kernel for gfx950 (MI355X), fp32.
//
// Replaces the hot loop of lasso/linear/solvers/ista.py:79-102 (fixed step):
//     r      = p W^T - x                      (ista.py:72)   GEMM-1  [16 x K] x [K x D]
//     g      = r W                            (ista.py:73)   GEMM-2  [16 x D] x [D x K]
//     z_next = softshrink(p - lr*g, alpha*lr) (ista.py:90)
//     delta  = sum |z - z_next|               (ista.py:93)   -> per-tile partial
//     y      = z_next + c_i (z_next - z)      (ista.py:98-100)
// One workgroup (8 waves) owns a 16-row tile of the batch and runs ALL
// requested iterations for it without touching HBM for state:
//   * y tile   [16][K]  fp32 lives in LDS (A operand of GEMM-1, XOR-swizzled)
//   * z tile            lives in VGPRs in MFMA C-layout (K/32 regs per lane)
//   * r tile   [16][D]  is exchanged through LDS once per iteration
//   * W / W^T are streamed from L2 by LDS-DMA (global_load_lds_dwordx4) into a
//     private 2-slot ring per wave -> no workgroup barrier in the GEMM loops,
//     two s_barrier per iteration (r exchange, y complete).
// Both GEMMs run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
//
// MFMA operand mapping used throughout (lane l, n=l&15, q=l>>4):
//   one ds_read_b128 gives 4 consecutive K-values {4q..4q+3} of a 16-wide
//   K-group; MFMA number j of the group consumes element j, i.e. it contracts
//   over k = kbase + 4q + j, q=0..3.  A and B use the same convention, so the
//   sum over j,q covers the 16 K-values exactly once.
#include "tile_device.hpp"

namespace lasso {

template <int K>
__global__ __launch_bounds__(kFistaThreads, 2) void fista_tile_kernel(const FistaTileParams p) {
  constexpr int D = kFistaD;
  constexpr int NW = kFistaWaves;
  constexpr int KW = K / NW;        // GEMM-2 output columns per wave
  constexpr int NP = KW / 32;       // GEMM-2 passes (2 col-blocks each)
  constexpr int T2 = D / 32;        // GEMM-2 steps per pass
  constexpr int YT_BYTES = kTileM * K * 4;
  constexpr int RT_BYTES = kTileM * D * 4;
  static_assert(D == 32 * NW, "each wave owns two GEMM-1 column blocks");
  static_assert((K / 32) % 2 == 0 && (NP * T2) % 2 == 0, "ring parity");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS map: [rings 64 KiB | y tile | r tile | reduction scratch].  The DMA rings sit
  // in the low 64 KiB so their M0 destination offsets fit 16 bits.
  lds_char* const rings = (lds_char*)smem;
  lds_char* const yt = rings + NW * kRingBytesPerWave;
  lds_char* const rt = yt + YT_BYTES;
  lds_f32* const red = (lds_f32*)(rt + RT_BYTES);

  TileCtx<K> c;
  c.init(p.Wp, p.Wtp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;

  // prologue: first two W steps of GEMM-1 are always in flight on entry
  dma_step(c.w1, c.voff1, c.ring);
  dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;

    // ---- load state -------------------------------------------------------
    // y tile -> LDS (swizzled).  y_in == nullptr means y = z_in; z_in == nullptr means 0.
    {
      const float* ysrc = p.y_in ? p.y_in : p.z_in;
      const int64_t ldy = p.y_in ? p.ldy_in : p.ldz_in;
      for (int idx = tid; idx < kTileM * K; idx += kFistaThreads) {
        const int r = idx / K, c = idx - r * K;
        float v = 0.0f;
        if (ysrc && (row0 + r) < p.n && c < p.k) v = ysrc[(int64_t)(row0 + r) * ldy + c];
        *(lds_f32*)(yt + tile_off<K>(r, c)) = v;
      }
    }
    // z in C-layout registers: zreg[pass][cb][reg] <-> row 4q+reg, col wid*KW+32*pass+16*cb+n
    f32x4 zreg[NP][2];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 4 * q + rg, c = wid * KW + 32 * ps + 16 * cb + n;
          float v = 0.0f;
          if (p.z_in && (row0 + r) < p.n && c < p.k)
            v = (p.z_in + (int64_t)row0 * p.ldz_in)[r * (int)p.ldz_in + c];
          zreg[ps][cb][rg] = v;
        }
    // -x in the C layout of GEMM-1's output (cols 32*wid + 16*cb + n)
    f32x4 xneg[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, c = 32 * wid + 16 * cb + n;
        float v = 0.0f;
        if ((row0 + r) < p.n && c < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + c];
        xneg[cb][rg] = -v;
      }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < p.iters; ++it) {
      const float coef = p.coef[it];
      float dsum = 0.0f;
      // opaque copies of the lane coordinates: keeps the (cheap) epilogue address
      // arithmetic inside the iteration instead of 32 hoisted-and-spilled addresses
      int no = n, qo = q;
      asm volatile("" : "+v"(no), "+v"(qo));

      // ================= GEMM-1: r = y W^T - x ==========================
      f32x4 acc[2] = {xneg[0], xneg[1]};
      // (its last two steps refill the ring with GEMM-2's first two W^T steps)
      gemm1_stream<K>(c, yt, acc, c.w2, c.w2 + 32, c.voff2);

      // r tile -> LDS (C layout -> swizzled row-major), then everyone reads all of it
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *(lds_f32*)(rt + tile_off<D>(4 * qo + rg, 32 * wid + 16 * cb + no)) = acc[cb][rg];
      LASSO_WAIT_LGKM0();
#ifndef LASSO_ABL_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      f32x4 rf[T2][2];
      load_r_frags<K>(c, rt, rf);

      // ================= GEMM-2 + prox/momentum epilogue ================
      static_for<NP>([&](auto ps_c) {
        constexpr int ps = decltype(ps_c)::value;
        f32x4 g2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        gemm2_pass<K, ps>(c, rf, g2);
        // epilogue for the 2 finished column blocks (in-place y update is safe:
        // GEMM-1 of this iteration is complete for every wave)
#ifdef LASSO_ABL_NOEPI   // timing ablation only (results invalid)
        asm volatile("" :: "v"(g2[0]), "v"(g2[1]));
        if (false)
#endif
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            lds_f32* const yp = (lds_f32*)(
                yt + tile_off<K>(4 * qo + rg, wid * KW + 32 * ps + 16 * cb + no));
            const float yv = *yp;
            const float zo = zreg[ps][cb][rg];
            const float step = __fmul_rn(p.lr, g2[cb][rg]);           // lr * grad
            const float zn = soft_threshold(__fsub_rn(yv, step), p.lam);
            dsum += __builtin_fabsf(__fsub_rn(zo, zn));                // |z - z_next|
            const float mom = __fmul_rn(coef, __fsub_rn(zn, zo));      // c (z_next - z)
            *yp = __fadd_rn(zn, mom);
            zreg[ps][cb][rg] = zn;
          }
      });

      // ---- per-tile sum |z - z_next| (deterministic order) ---------------
      dsum = wave_sum(dsum);
      if (lane == 0) red[wid] = dsum;
      LASSO_WAIT_LGKM0();
#ifndef LASSO_ABL_NOBAR
      __builtin_amdgcn_s_barrier();   // y tile complete; red[] complete
#endif
      if (p.partials && tid == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += red[w];
        p.partials[(int64_t)it * p.ntiles + tile] = tsum;
      }
    }

    // ---- store state --------------------------------------------------------
    {
      int no = n, qo = q;
      asm volatile("" : "+v"(no), "+v"(qo));
      float* const zo_base = p.z_out + (int64_t)row0 * p.ldz_out;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int r = 4 * qo + rg, c = wid * KW + 32 * ps + 16 * cb + no;
            if ((row0 + r) < p.n && c < p.k) zo_base[r * (int)p.ldz_out + c] = zreg[ps][cb][rg];
          }
    }
    if (p.y_out) {
      for (int idx = tid; idx < kTileM * K; idx += kFistaThreads) {
        const int r = idx / K, c = idx - r * K;
        if ((row0 + r) < p.n && c < p.k)
          p.y_out[(int64_t)(row0 + r) * p.ldy_out + c] = *(const lds_f32*)(yt + tile_off<K>(r, c));
      }
    }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();   // tile LDS may be overwritten by the next tile
  }
  LASSO_WAIT_VMCNT(0);  // drain the prefetched DMA before the LDS is released
}

size_t fista_tile_lds_bytes(int K) {
  return (size_t)kTileM * K * 4 + (size_t)kTileM * kFistaD * 4 + (size_t)kFistaWaves * kRingBytesPerWave + 64;
}

template <int K>
static hipError_t launch_k(const FistaTileParams& p, int grid, hipStream_t stream) {
  const size_t lds = fista_tile_lds_bytes(K);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fista_tile_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(fista_tile_kernel<K>, dim3(grid), dim3(kFistaThreads), lds, stream, p);
  return hipGetLastError();
}

hipError_t launch_fista_tile(const FistaTileParams& p, int kpad, int grid, hipStream_t stream) {
  switch (kpad) {
    case 256: return launch_k<256>(p, grid, stream);
    case 512: return launch_k<512>(p, grid, stream);
    case 1024: return launch_k<1024>(p, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lasso
