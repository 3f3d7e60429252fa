// Synthesis side of the convolutional solver as ONE implicit-GEMM kernel (stride 1):
//   R[n][c][i][j] = sum_{a,b,k} Ym[(n, i+ph-a, j+pw-b)][k] W[k][c][a][b]  -  x[n][c][i][j]
// (reference lasso/conv2d/ista.py:19, conv_transpose2d(z, W) - x, with the code held as the
// matrix Ym [N*Hz*Wz][K], one row per code pixel).  The explicit path (conv.hip) forms
// COLSt = Wt Ym^T with the general GEMM -- a [C kh kw][M] matrix written to and read back
// from HBM -- and then overlap-adds it; here neither exists.
//
// GEMM view: rows = image pixels, columns = the C <= 16 image channels (one 16-wide MFMA
// column block), contraction over (tap, atom) = kh kw K.  With so few columns nothing is
// reused across column blocks, so the split is over the CONTRACTION: a workgroup (8 waves)
// owns a 4 x 16 pixel tile, wave w owns the atoms [KC w/8, KC (w+1)/8) of every K chunk and
// keeps its B fragments (W for those atoms, every tap) in registers for the whole launch
// (persistent over tiles: W is read once per workgroup).  Per tile and K chunk (KC <= 128
// atoms) the code rows of the tile's halo ((4+kh-1) x (16+kw-1) code pixels) are staged in LDS
// once -- 16-byte row-contiguous global loads -- and the A operand of a (tap, k-step) MFMA is
// one ds_read_b32 at (lane part) + (compile-time offset): a tile row is 16 consecutive halo
// pixels, row pitch KC+2 floats, so each half wave hits the 32 banks once.  The halo buffer is
// double-buffered (next chunk / next tile in flight during the MFMAs); the eight partial
// accumulators meet in LDS, are added in wave order, x is subtracted and the residual leaves
// in 64-byte row pieces.
// Roofline: MFMA (2 M' C16 kh kw K flop, C padded to 16) -- 9.7 GFLOP at N=32, 16x64x64, 256
// 3x3 atoms = 62 us; HBM: Ym once plus the halo overlap (served by L2), R and x once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include "lasso_kernels.h"
#include "static_for.hpp"

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

constexpr int kSyTH = 4, kSyTW = 16, kSyWaves = 8, kSyThreads = 64 * kSyWaves;

struct ConvSynth {
  const float* Ym;     // [N*Hz*Wz][K]
  const float* W;      // [K][C][kh][kw]
  const float* x;      // [N][C][H][W] or null
  float* R;            // [N][C][H][W]
  ConvGeom g;
  int tiles_i, tiles_j, ntiles;
};

// KSC: k-steps (of 4 atoms) per wave and chunk -> chunk KC = 32 KSC atoms; CH chunks cover K
template <int KH, int KW, int KSC, int CH>
__global__ __launch_bounds__(kSyThreads, 2) void conv_synth_kernel(const ConvSynth p) {
  constexpr int KC = 32 * KSC, KCP = KC + 2;                 // row pitch = 2 (mod 32) floats: see the A operand reads
  constexpr int HH = kSyTH + KH - 1, HW = kSyTW + KW - 1, HP = HH * HW;
  constexpr int F4 = KC / 4;                                 // float4 per halo pixel and chunk
  constexpr int NST = (HP * F4 + kSyThreads - 1) / kSyThreads;
  constexpr int kBuf = HP * KCP;                             // floats per halo buffer
  extern __shared__ __attribute__((aligned(16))) float sy_smem[];
  lds_f32* const halo = (lds_f32*)sy_smem;                   // [2][HP][KCP]
  lds_f32* const red = (lds_f32*)sy_smem + (2 * kBuf + 3) / 4 * 4;   // [8 waves][4 rows][64 lanes][4]
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cl = lane & 15, kq = lane >> 4;

  // ---- this wave's B fragments: W[k][c = cl][a][b], k = KC ch + 4 KSC wid + 4 ks + kq ---------
  float B[CH][KH * KW][KSC];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch)
#pragma unroll
    for (int t = 0; t < KH * KW; ++t)
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        const int k = KC * ch + 4 * KSC * wid + 4 * ks + kq;
        const bool ok = k < g.K && cl < g.C;
        const float v = p.W[((int64_t)(ok ? k : 0) * g.C + (ok ? cl : 0)) * (KH * KW) + t];
        B[ch][t][ks] = ok ? v : 0.0f;
      }

  // staging map: round r, element e = tid + 512 r -> halo pixel e / F4, float4 e % F4.  The tile-independent part
  // is kept per round (packed: halo row, column, float4), the loads go through a buffer descriptor over Ym whose
  // range check returns the zeros of pixels outside the code grid (offset ~0u) -- all waves stage in lockstep with
  // the MFMA pipe idle, so the address arithmetic per item is kept to a handful of instructions per load.
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Ym), 0, (int)((int64_t)g.N * g.Hz * g.Wz * g.K * 4), 0x00020000);
  int st_map[NST];
#pragma unroll
  for (int r = 0; r < NST; ++r) {
    const int e = tid + kSyThreads * r;
    const int hp = min(e / F4, HP - 1);
    st_map[r] = e < HP * F4 ? ((hp / HW) << 20) | ((hp % HW) << 10) | (e % F4) : -1;
  }
  f32x4 stg[NST];
  auto load_item = [&](int tile, int ch) {
    const int tj = tile % p.tiles_j, ti = (tile / p.tiles_j) % p.tiles_i, n = tile / (p.tiles_j * p.tiles_i);
    const int u0 = ti * kSyTH + g.ph - (KH - 1), v0 = tj * kSyTW + g.pw - (KW - 1);
    const int pix0 = n * g.Hz * g.Wz;
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      const int m = st_map[r];
      const int u = u0 + (m >> 20), v = v0 + ((m >> 10) & 1023), k = KC * ch + 4 * (m & 1023);
      const bool ok = m >= 0 && (unsigned)u < (unsigned)g.Hz && (unsigned)v < (unsigned)g.Wz && k < g.K;
      const unsigned off = ok ? (unsigned)(((pix0 + u * g.Wz + v) * g.K + k) * 4) : ~0u;
      stg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, off, 0, 0));
    }
  };
  auto put_item = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      const int e = tid + kSyThreads * r;
      if (e < HP * F4) {                                       // (rows are 8-byte aligned only)
        lds_f32x2* const dst = (lds_f32x2*)(halo + buf * kBuf + (e / F4) * KCP + 4 * (e % F4));
        dst[0] = (f32x2){stg[r][0], stg[r][1]};
        dst[1] = (f32x2){stg[r][2], stg[r][3]};
      }
    }
  };

  const int first = blockIdx.x;
  if (first >= p.ntiles) return;
  load_item(first, 0);
  put_item(0);
  __syncthreads();
  int buf = 0;
  // A operand: lane part (pixel column cl of the tile row, this wave's atoms, k offset kq)
  const int a_lane = cl * KCP + 4 * KSC * wid + kq;
  for (int tile = first; tile < p.ntiles; tile += gridDim.x) {
    f32x4 acc[kSyTH];
#pragma unroll
    for (int rb = 0; rb < kSyTH; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this thread's outputs of the tile and their x (fetched now: the round trip hides behind the MFMAs)
    const int tj = tile % p.tiles_j, ti = (tile / p.tiles_j) % p.tiles_i, n = tile / (p.tiles_j * p.tiles_i);
    constexpr int NO = kSyTH * 16 * 16 / kSyThreads;
    float xv[NO];
    int64_t oidx[NO];
#pragma unroll
    for (int h = 0; h < NO; ++h) {
      const int o = tid + kSyThreads * h;
      const int i = ti * kSyTH + ((o >> 4) & 3), j = tj * kSyTW + (o & 15), c = o >> 6;
      const bool ok = c < g.C && i < g.H && j < g.W;
      oidx[h] = ok ? (((int64_t)n * g.C + c) * g.H + i) * g.W + j : -1;
      xv[h] = (ok && p.x) ? p.x[oidx[h]] : 0.0f;
    }
    static_for<CH>([&](auto ch_c) {
      constexpr int ch = decltype(ch_c)::value;
      const int ntile = ch + 1 < CH ? tile : tile + (int)gridDim.x;
      const bool more = ntile < p.ntiles;
#ifndef LASSO_SY_ABL_NOSTAGE
      if (more) load_item(ntile, ch + 1 < CH ? ch + 1 : 0);          // in flight during the MFMAs
#endif
      const lds_f32* const hb = halo + buf * kBuf + a_lane;
      // the A values of tap t+1 are read while tap t's MFMAs run (pinned: left alone, the reads sit right in
      // front of their MFMAs and every eighth MFMA waits out an LDS round trip)
      float av[2][KSC][kSyTH];
      auto read_tap = [&](auto t_c, int par) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value, a = t / KW, b = t % KW;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
          for (int rb = 0; rb < kSyTH; ++rb) av[par][ks][rb] = hb[((rb + KH - 1 - a) * HW + (KW - 1 - b)) * KCP + 4 * ks];
      };
#ifndef LASSO_SY_ABL_NOMFMA
      read_tap(std::integral_constant<int, 0>{}, 0);
      static_for<KH * KW>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (t + 1 < KH * KW) read_tap(std::integral_constant<int, t + 1>{}, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
          for (int rb = 0; rb < kSyTH; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1][ks][rb], B[ch][t][ks], acc[rb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
#endif
#ifndef LASSO_SY_ABL_NOSTAGE
      if (more) put_item(buf ^ 1);
#endif
      __syncthreads();                                               // other buffer complete, this one free
      buf ^= 1;
    });
    // ---- the eight partial sums meet; x is subtracted; 64-byte row pieces leave ------------------
#ifdef LASSO_SY_ABL_NOEPI
    if (p.ntiles < 0)
#endif
    {
#pragma unroll
    for (int rb = 0; rb < kSyTH; ++rb) *(lds_f32x4*)(red + ((wid * kSyTH + rb) * 64 + lane) * 4) = acc[rb];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NO; ++h) {
      const int o = tid + kSyThreads * h;
      const int i16 = o & 15, rb = (o >> 4) & 3, c = o >> 6;
      const int src = ((i16 >> 2) * 16 + c) * 4 + (i16 & 3);          // MFMA C layout: row 4 (lane >> 4) + r, column lane & 15
      float sum = 0.0f;
#pragma unroll
      for (int w = 0; w < kSyWaves; ++w) sum += red[(w * kSyTH + rb) * 256 + src];
      if (oidx[h] >= 0) p.R[oidx[h]] = sum - xv[h];
    }
    }
#ifdef LASSO_SY_ABL_NOEPI
#pragma unroll
    for (int rb = 0; rb < kSyTH; ++rb) asm volatile("" :: "v"(acc[rb]));
#endif
    // (red is rewritten only after the next tile's chunk barriers)
  }
}

template <int KH, int KW, int KSC, int CH>
hipError_t synth_launch(const ConvSynth& p, int cus, hipStream_t stream) {
  constexpr int KCP = 32 * KSC + 2, HP = (kSyTH + KH - 1) * (kSyTW + KW - 1);
  const size_t lds = (size_t)((2 * HP * KCP + 3) / 4 * 4 + kSyWaves * kSyTH * 256) * 4;
  const void* fn = reinterpret_cast<const void*>(&conv_synth_kernel<KH, KW, KSC, CH>);
  if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const int grid = std::min(p.ntiles, per_cu * cus);
  hipLaunchKernelGGL((conv_synth_kernel<KH, KW, KSC, CH>), dim3(grid), dim3(kSyThreads), lds, stream, p);
  return hipGetLastError();
}

}  // namespace

// *done = false when the geometry is not covered (stride > 1, C > 16, K not a multiple of 4, a
// kernel size / atom count without an instantiation): the caller takes the explicit path.
// dry != 0: no launch -- *done says whether this geometry is covered (lasso_conv_ista_kernel_name)
hipError_t launch_conv_synth(const float* Ym, const float* w, const float* x, float* r, const ConvGeom& g, int cus,
                             bool* done, hipStream_t stream, int dry) {
  *done = false;
  // (C is padded to the 16 columns of an MFMA block: below 8 channels the explicit path does less work)
  if (g.sh != 1 || g.sw != 1 || g.C > 16 || g.C < 8 || g.K < 4 || (g.K & 3) || g.kh != g.kw || (((uintptr_t)Ym) & 15)) return hipSuccess;
  ConvSynth p;
  p.Ym = Ym; p.W = w; p.x = x; p.R = r; p.g = g;
  p.tiles_i = (g.H + kSyTH - 1) / kSyTH;
  p.tiles_j = (g.W + kSyTW - 1) / kSyTW;
  const int64_t nt = (int64_t)g.N * p.tiles_i * p.tiles_j;
  if (nt <= 0 || nt > INT32_MAX || (int64_t)g.N * g.Hz * g.Wz * g.K * 4 > INT32_MAX) return hipSuccess;   // 32-bit buffer offsets
  p.ntiles = (int)nt;
  const int ks = g.kh, K = g.K;
  hipError_t e = hipSuccess;
  if (dry) {
    *done = (ks == 3 && K <= 256) || (ks == 5 && K <= 128) || (ks == 7 && K <= 64);
    return hipSuccess;
  }
  *done = true;
  if (ks == 3 && K <= 32) e = synth_launch<3, 3, 1, 1>(p, cus, stream);
  else if (ks == 3 && K <= 64) e = synth_launch<3, 3, 2, 1>(p, cus, stream);
  else if (ks == 3 && K <= 128) e = synth_launch<3, 3, 4, 1>(p, cus, stream);
  else if (ks == 3 && K <= 256) e = synth_launch<3, 3, 4, 2>(p, cus, stream);
  else if (ks == 5 && K <= 32) e = synth_launch<5, 5, 1, 1>(p, cus, stream);
  else if (ks == 5 && K <= 64) e = synth_launch<5, 5, 2, 1>(p, cus, stream);
  else if (ks == 5 && K <= 128) e = synth_launch<5, 5, 2, 2>(p, cus, stream);
  else if (ks == 7 && K <= 32) e = synth_launch<7, 7, 1, 1>(p, cus, stream);
  else if (ks == 7 && K <= 64) e = synth_launch<7, 7, 2, 1>(p, cus, stream);
  else *done = false;
  return e;
}

}  // namespace lasso
