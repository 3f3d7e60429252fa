// 16-wave variant of the fused persistent FISTA kernel (see fista_tile.hip for the
// algorithm, the LDS layouts and the MFMA operand convention -- they are identical).
//
// Difference: the 16-row tile is worked on by 16 waves (4 per SIMD) instead of 8.
// Wave w owns ONE 16-column block of GEMM-1's output (r columns [16w,16w+16)) and
// K/256 blocks of GEMM-2's output; a ring step is 16 rows x 128 B = 2 KiB = 2 LDS-DMA
// instructions.  With four waves per SIMD the LDS-read / DMA-issue / epilogue phases of
// one wave are covered by the MFMAs of the other three, which the 8-wave kernel (two
// waves per SIMD, 199 VGPRs) cannot do: ablations on MI355X attribute ~6 % to DMA issue
// and ~7 % to the prox epilogue there.  The r fragments are re-read from LDS per step
// instead of being held in 64 VGPRs, so the kernel fits the 128-VGPR budget.
#include "tile_device.hpp"

namespace lasso {
namespace w16 {

constexpr int kWaves = 16;
constexpr int kThreads = kWaves * 64;
constexpr int kStep = 2048;                // 16 rows x 128 B
constexpr int kRing = 2 * kStep;

__device__ __forceinline__ void dma2(const float* src, const unsigned (&voff)[2], lds_char* slot) {
#ifdef LASSO_ABL_NODMA      // timing ablation only (results invalid)
  return;
#endif
  const unsigned lds_addr = (unsigned)(uintptr_t)slot;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:0\n\t"
      "s_add_u32 m0, %4, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3 offset:0\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff[0]), "v"(voff[1]), "s"(src), "s"(lds_addr)
      : "memory", "scc");
}

template <int K>
struct Ctx {
  int lane, wid, n, q;
  unsigned voff1[2], voff2[2];
  int boff[2];
  int aoff[2][2];
  lds_char* ring;
  const float* w1;
  const float* w2;
  __device__ __forceinline__ void init(const float* Wp, const float* Wtp, lds_char* rings) {
    const int tid = threadIdx.x;
    lane = tid & 63;
    wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    n = lane & 15;
    q = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      voff1[j] = (unsigned)(row * K + 4 * c) * 4u;
      voff2[j] = (unsigned)(row * kFistaD + 4 * c) * 4u;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) boff[ss] = n * 128 + (((4 * ss + q) ^ ((n >> 1) & 7)) << 4);
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) aoff[par][ss] = ((8 * par + 4 * ss + q) ^ n) << 4;
    ring = rings + wid * kRing;
    w1 = Wp + (size_t)(16 * wid) * K;
    w2 = Wtp + (size_t)((K / kWaves) * wid) * kFistaD;
  }
};

// one streamed step: 8 MFMAs on two independent accumulators (k-halves of the step)
template <int K, int PAR>
__device__ __forceinline__ void step(const Ctx<K>& c, const lds_char* atile_row, f32x4 (&acc)[2],
                                     const float* pf_src, const unsigned (&pf_voff)[2]) {
  lds_char* const slot = c.ring + PAR * kStep;
  LASSO_WAIT_VMCNT(2);
  f32x4 b[2], a[2];
#ifdef LASSO_ABL_NOLDS      // timing ablation only (results invalid)
  b[0] = acc[0]; b[1] = acc[1]; a[0] = acc[1]; a[1] = acc[0];
  asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(a[0]), "+v"(a[1]));
#else
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) b[ss] = *(const lds_f32x4*)(slot + c.boff[ss]);
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) a[ss] = *(const lds_f32x4*)(atile_row + c.aoff[PAR][ss]);
  LASSO_WAIT_LGKM0();
#endif
  dma2(pf_src, pf_voff, slot);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][j], b[0][j], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][j], b[1][j], acc[1], 0, 0, 0);
  }
}

template <int K>
__global__ __launch_bounds__(kThreads, 4) void fista_tile16_kernel(const FistaTileParams p) {
  constexpr int D = kFistaD;
  constexpr int S1 = K / 32;
  constexpr int KW = K / kWaves;      // GEMM-2 output columns per wave
  constexpr int NP = KW / 16;         // passes of one column block
  constexpr int T2 = D / 32;
  constexpr int YT_BYTES = kTileM * K * 4;
  constexpr int RT_BYTES = kTileM * D * 4;
  static_assert(D == 16 * kWaves, "one GEMM-1 column block per wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const yt = rings + kWaves * kRing;
  lds_char* const rt = yt + YT_BYTES;
  lds_f32* const red = (lds_f32*)(rt + RT_BYTES);

  Ctx<K> c;
  c.init(p.Wp, p.Wtp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;

  dma2(c.w1, c.voff1, c.ring);
  dma2(c.w1 + 32, c.voff1, c.ring + kStep);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    {
      const float* ysrc = p.y_in ? p.y_in : p.z_in;
      const int64_t ldy = p.y_in ? p.ldy_in : p.ldz_in;
      for (int idx = tid; idx < kTileM * K; idx += kThreads) {
        const int r = idx / K, cc = idx - r * K;
        float v = 0.0f;
        if (ysrc && (row0 + r) < p.n && cc < p.k) v = ysrc[(int64_t)(row0 + r) * ldy + cc];
        *(lds_f32*)(yt + tile_off<K>(r, cc)) = v;
      }
    }
    f32x4 zreg[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = wid * KW + 16 * ps + n;
        float v = 0.0f;
        if (p.z_in && (row0 + r) < p.n && cc < p.k)
          v = (p.z_in + (int64_t)row0 * p.ldz_in)[r * (int)p.ldz_in + cc];
        zreg[ps][rg] = v;
      }
    f32x4 xneg;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int r = 4 * q + rg, cc = 16 * wid + n;
      float v = 0.0f;
      if ((row0 + r) < p.n && cc < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + cc];
      xneg[rg] = -v;
    }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < p.iters; ++it) {
      const float coef = p.coef[it];
      float dsum = 0.0f;
      int no = n, qo = q;
      asm volatile("" : "+v"(no), "+v"(qo));

      // ---- GEMM-1: r = y W^T - x --------------------------------------------
      f32x4 acc[2] = {xneg, {0.f, 0.f, 0.f, 0.f}};
      const lds_char* const yrow = yt + n * (K * 4);
#pragma unroll 1
      for (int s2 = 0; s2 < S1 / 2 - 1; ++s2) {
        step<K, 0>(c, yrow + s2 * 256, acc, c.w1 + 64 * s2 + 64, c.voff1);
        step<K, 1>(c, yrow + s2 * 256, acc, c.w1 + 64 * s2 + 96, c.voff1);
      }
      step<K, 0>(c, yrow + (S1 / 2 - 1) * 256, acc, c.w2, c.voff2);
      step<K, 1>(c, yrow + (S1 / 2 - 1) * 256, acc, c.w2 + 32, c.voff2);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        *(lds_f32*)(rt + tile_off<D>(4 * qo + rg, 16 * wid + no)) = acc[0][rg] + acc[1][rg];
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();

      // ---- GEMM-2 + prox/momentum epilogue -----------------------------------
      const lds_char* const rrow = rt + n * (D * 4);
      static_for<NP>([&](auto ps_c) {
        constexpr int ps = decltype(ps_c)::value;
        f32x4 g2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        static_for<T2>([&](auto t_c) {
          constexpr int t = decltype(t_c)::value;
          constexpr int U = ps * T2 + t;
          if constexpr (U + 2 < NP * T2) {
            constexpr int pn = (U + 2) / T2, tn = (U + 2) % T2;
            step<K, (U & 1)>(c, rrow + (t >> 1) * 256, g2, c.w2 + (size_t)(16 * pn) * D + 32 * tn, c.voff2);
          } else {
            step<K, (U & 1)>(c, rrow + (t >> 1) * 256, g2, c.w1 + 32 * (U + 2 - NP * T2), c.voff1);
          }
        });
#ifdef LASSO_ABL_NOEPI   // timing ablation only (results invalid)
        asm volatile("" :: "v"(g2[0]), "v"(g2[1]));
        if (false)
#endif
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          lds_f32* const yp = (lds_f32*)(yt + tile_off<K>(4 * qo + rg, wid * KW + 16 * ps + no));
          const float yv = *yp;
          const float zo = zreg[ps][rg];
          const float g = g2[0][rg] + g2[1][rg];
          const float stp = __fmul_rn(p.lr, g);
          const float zn = soft_threshold(__fsub_rn(yv, stp), p.lam);
          dsum += __builtin_fabsf(__fsub_rn(zo, zn));
          const float mom = __fmul_rn(coef, __fsub_rn(zn, zo));
          *yp = __fadd_rn(zn, mom);
          zreg[ps][rg] = zn;
        }
      });

      dsum = wave_sum(dsum);
      if (lane == 0) red[wid] = dsum;
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      if (p.partials && tid == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tsum += red[w];
        p.partials[(int64_t)it * p.ntiles + tile] = tsum;
      }
    }

    {
      int no = n, qo = q;
      asm volatile("" : "+v"(no), "+v"(qo));
      float* const zo_base = p.z_out + (int64_t)row0 * p.ldz_out;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 4 * qo + rg, cc = wid * KW + 16 * ps + no;
          if ((row0 + r) < p.n && cc < p.k) zo_base[r * (int)p.ldz_out + cc] = zreg[ps][rg];
        }
    }
    if (p.y_out) {
      for (int idx = tid; idx < kTileM * K; idx += kThreads) {
        const int r = idx / K, cc = idx - r * K;
        if ((row0 + r) < p.n && cc < p.k)
          p.y_out[(int64_t)(row0 + r) * p.ldy_out + cc] = *(const lds_f32*)(yt + tile_off<K>(r, cc));
      }
    }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
  }
  LASSO_WAIT_VMCNT(0);
}

template <int K>
static hipError_t launch_k(const FistaTileParams& p, int grid, hipStream_t stream) {
  const size_t lds = (size_t)kTileM * K * 4 + (size_t)kTileM * kFistaD * 4 + (size_t)kWaves * kRing + 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fista_tile16_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(fista_tile16_kernel<K>, dim3(grid), dim3(kThreads), lds, stream, p);
  return hipGetLastError();
}

}  // namespace w16

hipError_t launch_fista_tile16(const FistaTileParams& p, int kpad, int grid, hipStream_t stream) {
  switch (kpad) {
    case 256: return w16::launch_k<256>(p, grid, stream);
    case 512: return w16::launch_k<512>(p, grid, stream);
    case 1024: return w16::launch_k<1024>(p, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lasso
