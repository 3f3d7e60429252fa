// Persistent single-launch ISTA/FISTA solve on bf16 tensors, with or without the backtracking
// line search (BASELINE config 3: n=16384, d=256, k=1024, bf16; reference
// lasso/linear/solvers/ista.py:17-54,57-104 run on bf16 tensors).
//
// One workgroup (8 waves) owns one 64-row tile for the WHOLE solve; all tiles are resident at
// once (n <= 64 x #CUs -- config 3 is exactly 256 tiles on 256 CUs).  On chip, per tile:
//   * the point p (= y for FISTA, z for ISTA) as a 64 x K bf16 LDS tile in GEMM A-operand layout
//     (128 KiB at K = 1024) -- it IS the A operand of the gradient's GEMM-1 and is read 16 bytes
//     per thread by every trial; updated in place at the end of an outer iteration;
//   * x (bf16) in registers; a 32 KiB LDS region that is the residual tile of GEMM-2 during the
//     gradient and a double-buffered 64-atom staging tile for candidates during the trials.
// The gradient g and the iterate z live in memory as bf16 -- the precision the reference's own
// bf16 tensors hold them in: g (32 MB at config 3: it stays in the 256 MB Infinity Cache) is
// written once per outer iteration and read once per trial, z is read and written once per
// outer iteration (in place in the caller's z_out); 16-byte, row-contiguous reads.  The multi-launch path of bt_bf16.hip moved p, g AND z
// through HBM in fp32 on every gradient, trial and finish (11 GB per config-3 solve) with a
// launch per phase and a host round trip per outer iteration; here: one launch, none, and only
// bf16 g / z traffic (~2.6 GB, most of it served by the Infinity Cache).
//
// A trial never materialises its candidate z+ = S(p - lr g): the K atoms are walked in passes
// of 64; per pass every thread forms 8 candidate values (one row, 8 atoms) from p (LDS) and g
// (prefetched two passes ahead), accumulates the three element sums of ista.py:30-35 and stores the
// bf16 candidate into the staging tile, and GEMM-1 consumes that tile (2 MFMA steps) while the
// next pass is being formed (MFMAs and element-wise VALU work interleaved in one instruction
// stream) -- one barrier per pass.
//
// GEMM-1  r = A W^T - x     : A fragments from LDS, W fragments fragment-major from L2 (Wq1).
// GEMM-2  g^T = W^T r^T     : the TRANSPOSED product (MFMA A operand = W fragment, B operand =
//     residual fragment), so that a lane ends up with 4 CONSECUTIVE atoms of one row and stores
//     8 bytes of bf16 g.  (Wq2's fragment-major pack is the same bytes in either role.)
//
// Global decisions without leaving the kernel (ista.py:26-35,45 sum over the WHOLE batch):
// every workgroup publishes its partial sums of a trial as two tagged 16-byte granules
// (write-through), every workgroup sweeps all granules of that trial and reduces them in the
// same fixed order -- identical verdict everywhere, no host round trip, no extra launch.
// The sweep of trial t is taken AFTER trial t+1 has been computed speculatively (step/eta), so
// the granules' flight time hides behind a GEMM; an accepted trial t discards t+1's work.
// The stop rule (ista.py:93) uses the same exchange with one granule per workgroup.
// Every spin is bounded; on a timeout the grid aborts (out[2]) and the host falls back to the
// multi-launch kernels.
#include "bf16_device.hpp"

#ifndef LASSO_BT16_G2RING
#define LASSO_BT16_G2RING 4   // GEMM-2 of the gradient: W fragment ring depth in steps
#endif
#ifndef LASSO_BT16_ACCEPT_BATCH
#define LASSO_BT16_ACCEPT_BATCH 4   // passes of the accept step per register set (two sets: 2 x this many passes in flight)
#endif
// timing-only ablations of the trial batch (results invalid): -DLASSO_BT16_ABL_NOMFMA / _NOCAND / _NOBAR
#ifdef LASSO_BT16_ABL_NOMFMA
#define BT16_MFMA(acc, a, b) asm volatile("" : "+v"(acc) : "v"(a), "v"(b))
#else
#define BT16_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
#endif
#ifdef LASSO_BT16_ABL_NOBAR
#define BT16_PASS_BARRIER() do { } while (0)
#else
#define BT16_PASS_BARRIER() __syncthreads()
#endif
#ifndef LASSO_BT16_LEAD
#define LASSO_BT16_LEAD 8      // candidate operations in front of the first MFMA of the step behind a pass barrier
#endif
#ifndef LASSO_BT16_SCHED
#define LASSO_BT16_SCHED 0     // how a trial step's MFMAs and candidate arithmetic are scheduled (see `step`)
#endif
#ifndef LASSO_BT16_VPM
#define LASSO_BT16_VPM 9       // VALU instructions per MFMA in the prescribed mix
#endif
#ifndef LASSO_BT16_CHECK
#define LASSO_BT16_CHECK 2     // double passes of a speculative trial before its predecessor's verdict is read
#endif

#ifdef LASSO_BT16_TIMING
// debug build (tools/bt16_timeline.py): wall-clock stamps of ONE outer iteration, 16 per workgroup
__device__ unsigned long long lasso_bt16_stamps[1024 * 16];
extern "C" int lasso_debug_bt16_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lasso_bt16_stamps), sizeof(lasso_bt16_stamps));
}
#define BT16_STAMP(slot) do { if (it == LASSO_BT16_TIMING && tid == 0) lasso_bt16_stamps[blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define BT16_STAMP(slot) do { } while (0)
#endif

namespace lasso {
namespace {

using namespace bf16dev;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;

constexpr int kRing = 8;             // granule ring (epochs in flight never span more than 4)
constexpr int kMaxTrials = 1000;     // ista.py:17 (maxiter=1000)
constexpr int kPass = 64;            // atoms per trial pass
constexpr int kStageBytes = kRows * kPass * 2;       // 8 KiB
constexpr int TB = 5;                // trials of the line search computed per batch (one cross-workgroup decision each)
constexpr int kCandOps = 28;         // plain VALU operations of one thread's candidate (4 elements x 7 stages)
constexpr int kSlotBytes = 32 * kPass * 2;           // a staging tile of the line search: [32 rows][64 atoms] bf16
// staging tile of (trial t, pass parity): trials 0-2 double-buffered, 3-4 single (8 tiles = the 32 KiB scratch)
__host__ __device__ constexpr int slot_of(int t, int par) { return t < 3 ? 2 * t + par : 3 + t; }
static_assert(slot_of(TB - 1, 1) * kSlotBytes + kSlotBytes <= kRows * kFistaD * 2, "staging tiles exceed the scratch");
constexpr int kScratchBytes = kRows * kFistaD * 2;   // 32 KiB: residual tile | 2 staging tiles + reductions

__device__ __forceinline__ u32x2 pack4(const float (&v)[4]) {
  bf16x4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = (__bf16)v[e];
  return __builtin_bit_cast(u32x2, b);
}
__device__ __forceinline__ void unpack4(u32x2 u, float (&v)[4]) {
  const bf16x4 b = __builtin_bit_cast(bf16x4, u);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (float)b[e];
}
__device__ __forceinline__ void unpack8(u32x4 u, float (&v)[8]) {
  const bf16x8 b = __builtin_bit_cast(bf16x8, u);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)b[e];
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  bf16x8 b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[e];
  return __builtin_bit_cast(u32x4, b);
}
__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }
// Unpack a register-resident entry HERE: the volatile no-op keeps the compiler from hoisting the
// bf16 -> fp32 conversions of the (loop-invariant) g out of the trial loop, which would hold it
// a second time in fp32 -- 128 more VGPRs than there are.
__device__ __forceinline__ void unpack8_here(u32x4 u, float (&v)[8]) {
  asm volatile("" : "+v"(u));
  unpack8(u, v);
}

// wave-wide sum in double, fixed (butterfly) order, same value in every lane
__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

// the same on the ALU path (DPP row shifts + four readlanes, like wave_sum): a fixed order, no LDS round trips --
// the line search's decision sums 4 TB + 1 values per wave this way
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

// byte offset of 16-byte chunk c8 (0..7) of row `row` in a staging tile (128 B per row)
__device__ __forceinline__ int stage_off(int row, int c8) { return row * (kPass * 2) + ((c8 ^ (row & 7)) << 4); }

// GEMM-1 of the gradient (A = the whole p tile): like bf16dev::gemm1_bf16, but the W fragments run
// SIX steps ahead of their use through an 8-deep register ring (the gradient phase has the
// registers to spare and one L2 round trip is worth ~4 steps of MFMAs), addressed as
// descriptor + lane offset + compile-time offset.
template <int K>
__device__ __forceinline__ void gemm1_bf16_deep(const lds_char* at, __amdgpu_buffer_rsrc_t wrsrc, unsigned lane16,
                                                int lane, f32x4 (&acc)[4][2]) {
  constexpr int S1 = K / 32;
  static_assert(S1 % 8 == 0, "K must be a multiple of 256");
  const int i = lane & 15, kg = lane >> 4;
  bf16x8 b[8][2];
  auto frag = [&](unsigned soff, int f) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, soff + f * 1024, 0));
  };
#pragma unroll
  for (int s = 0; s < 6; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) b[s][cb] = frag(0, s * 2 + cb);
#pragma unroll 1
  for (int j = 0; j < S1 / 8; ++j) {
    const unsigned soff = (unsigned)j * 8 * 2 * 1024;
    static_for<8>([&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      const int s = 8 * j + u;
      // step s + 6 (the last six prefetches re-read the final step)
      const unsigned so = (s + 6 < S1) ? soff : (unsigned)(S1 - 1 - u - 6) * 2 * 1024;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) b[(u + 6) % 8][cb] = frag(so, (u + 6) * 2 + cb);
      bf16x8 a[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[rb] = *(const lds_bf16x8*)(at + tile16_off<K * 2>(16 * rb + i, 4 * s + kg));
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b[u][cb], acc[rb][cb], 0, 0, 0);
    });
  }
}

template <int K>
__global__ __launch_bounds__(kThreads, 2) void bt16_persist_kernel(const Bt16PersistParams p) {
  constexpr int NAB = K / 128;                 // 16-atom blocks per wave (GEMM-2)
  constexpr int NP = K / kPass;                // trial passes
  constexpr int S2 = kFistaD / 32;             // contraction steps of GEMM-2
  constexpr int PT_BYTES = kRows * K * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const pt = (lds_char*)smem;        // the point p, [64][K] bf16, A-operand layout
  lds_char* const st = pt + PT_BYTES;          // 32 KiB scratch (see above)
  lds_f32* const red = (lds_f32*)(st + 2 * kStageBytes);   // valid while `st` is not the residual tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cl = lane & 15, q = lane >> 4;
  const int erow = tid >> 3, ec8 = tid & 7;    // element-wise role: one row, 8 consecutive atoms per pass
  const int tile = blockIdx.x, row0 = tile * kRows;
  const __bf16* const Xg = (const __bf16*)p.X;
  const bf16x8* const wq1 = (const bf16x8*)p.Wq1 + (int64_t)wid * (K / 32) * 2 * 64;
  const bf16x8* const wq2 = (const bf16x8*)p.Wq2 + (int64_t)wid * S2 * NAB * 64;
  __bf16* const Zg = (__bf16*)p.Z;
  __bf16* const Gg = (__bf16*)p.G;             // [ntiles * 64][K] bf16, row stride K
  const __bf16* const Z0g = (const __bf16*)p.Z0;
  // r0 of this wave's 64 x 32 block in GEMM-1's accumulator layout: [tile][wave][rb][cb][lane] f32x4 (16-byte, coalesced)
  f32x4* const R0g = (f32x4*)p.R0 + ((int64_t)blockIdx.x * kWaves + wid) * 8 * 64 + lane;
  const bool zvec = (p.ldz & 7) == 0 && (((uintptr_t)p.Z) & 15) == 0 && (p.k & 7) == 0;
  const bool z0vec = Z0g && (p.ldz0 & 7) == 0 && (((uintptr_t)p.Z0) & 15) == 0 && (p.k & 7) == 0;
  const bool erow_ok = (row0 + erow) < p.n;
  const __amdgpu_buffer_rsrc_t grsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.gran, 0, kRing * p.ntiles * 32 * TB, 0x00020000);
  // The unrolled trial passes address W fragments, the p tile and the staging tiles as
  // (one lane-dependent register) + (compile-time offset): a buffer descriptor for the wave's
  // Wq1 pack, and even/odd-pass lane constants for the swizzled LDS tiles -- otherwise the
  // compiler materialises ~100 loop-invariant addresses and spills them.
  const __amdgpu_buffer_rsrc_t w1rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(wq1), 0, (K / 32) * 2 * 64 * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(wq2), 0, S2 * NAB * 64 * 16, 0x00020000);
  const unsigned lane16 = lane * 16;
  auto wfrag1 = [&](int frag) {                 // fragment index (step * 2 + col block) of this wave's pack
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1rsrc, lane16, frag * 1024, 0));
  };
  // p tile chunk of (erow, pass j): erow * 2K + 256 (j >> 1) + ((((j & 1) << 3 | ec8) ^ (erow & 15)) << 4)
  const int ptE = erow * (2 * K) + ((ec8 ^ (erow & 15)) << 4), ptO = erow * (2 * K) + (((8 | ec8) ^ (erow & 15)) << 4);
  auto pt_off = [&](int j) { return ((j & 1) ? ptO : ptE) + 256 * (j >> 1); };
  // staging: this thread's chunk, and the A fragments (row 16 rb + cl, chunk 4 u + q)
  const int stW = stage_off(erow, ec8);
  const int stA0 = stage_off(cl, q), stA1 = stage_off(cl, 4 + q);   // + 2048 rb  (row & 7 == cl & 7 for every rb)

  // 8 consecutive z values (bf16) of this thread's row, pass j
  auto load_z8 = [&](const __bf16* base, int64_t ld, bool vec, int j, float (&v)[8]) {
    const int a0 = kPass * j + 8 * ec8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    if (!base || !erow_ok || a0 >= p.k) return;
    const __bf16* src = base + (int64_t)(row0 + erow) * ld + a0;
    if (vec) {
      unpack8(*reinterpret_cast<const u32x4*>(src), v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (a0 + e < p.k) v[e] = (float)src[e];
    }
  };
  // the same 8 values, still packed (exact: they are bf16 in memory)
  auto load_z8p = [&](const __bf16* base, int64_t ld, bool vec, int j) -> u32x4 {
    const int a0 = kPass * j + 8 * ec8;
    if (!base || !erow_ok || a0 >= p.k) return (u32x4){0u, 0u, 0u, 0u};
    const __bf16* src = base + (int64_t)(row0 + erow) * ld + a0;
    if (vec) return *reinterpret_cast<const u32x4*>(src);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (a0 + e < p.k) ? (float)src[e] : 0.0f;
    return pack8(v);
  };
  auto store_z8 = [&](int j, const float (&v)[8]) {
    const int a0 = kPass * j + 8 * ec8;
    if (!erow_ok || a0 >= p.k) return;
    __bf16* dst = Zg + (int64_t)(row0 + erow) * p.ldz + a0;
    if (zvec) {
      *reinterpret_cast<u32x4*>(dst) = pack8(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (a0 + e < p.k) dst[e] = (__bf16)v[e];
    }
  };
  // g rows are padded to whole tiles in the workspace, so every thread's row exists
  const __bf16* const grow = Gg + (int64_t)(row0 + erow) * K + 8 * ec8;   // + kPass * j
  auto load_g8 = [&](int j) { return *reinterpret_cast<const u32x4*>(grow + kPass * j); };

  // ---- x in GEMM-1's accumulator layout (4 rows per packed entry) -> workspace -------------------
  // (16 registers for the whole solve were what the compiler spilled first once the line search held five
  // accumulator sets: 21 scratch loads in front of every residual.  Now it is fetched where it is used, 4 x 16
  // bytes per lane, issued ahead of the GEMM whose residual needs it.)
  u32x4* const XrG = (u32x4*)p.XR + ((int64_t)blockIdx.x * kWaves + wid) * 4 * 64 + lane;     // [rb] stride 64
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    u32x2 xr2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float v[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = row0 + 16 * rb + 4 * q + rg, cc = 32 * wid + 16 * cb + cl;
        v[rg] = (r < p.n && cc < p.d) ? (float)Xg[(int64_t)r * p.ldx + cc] : 0.0f;
      }
      xr2[cb] = pack4(v);
    }
    XrG[rb * 64] = (u32x4){xr2[0][0], xr2[0][1], xr2[1][0], xr2[1][1]};
  }
  // ---- y_0 = z_0 (ista.py:76-78) -> the p tile -------------------------------------------------
#pragma unroll 1
  for (int j = 0; j < NP; ++j) {
    float v[8];
    load_z8(Z0g, p.ldz0, z0vec, j, v);
    *(lds_u32x4*)(pt + tile16_off<K * 2>(erow, 8 * j + ec8)) = pack8(v);
  }
  __syncthreads();

  // residual of an accumulator set: acc <- acc - x (xq: this lane's x of the four row blocks), returns its sum r^2
  auto residual = [&](f32x4 (&acc)[4][2], const u32x4 (&xq)[4]) {
    float rss = 0.0f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        float xv[4];
        unpack4((u32x2){xq[rb][2 * cb], xq[rb][2 * cb + 1]}, xv);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float res = acc[rb][cb][rg] - xv[rg];
          acc[rb][cb][rg] = res;
          rss = fmaf(res, res, rss);
        }
      }
    return rss;
  };
  auto abort_now = [&]() { __hip_atomic_store(p.out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  unsigned epoch = 0;                          // trials published so far (tags are epoch numbers, never 0)
  int iterations = 0;
  float last_delta = __builtin_nanf("");
  bool warned = false, aborted = false;

  for (int it = 0; it < p.maxiter && !aborted; ++it) {
    const float coef = p.fast ? p.coef[it] : 0.0f;
    BT16_STAMP(0);
    // ================================ gradient at p (ista.py:22-24 / 72-73) =================
    float rss0;
    {
      f32x4 acc[4][2];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      u32x4 xq[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) xq[rb] = XrG[rb * 64];
      gemm1_bf16_deep<K>(pt, w1rsrc, lane16, lane, acc);
      BT16_STAMP(11);
      rss0 = residual(acc, xq);
      if (p.backtrack) {                           // r0 in the accumulator layout, for the line search's sum dz g
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) R0g[(rb * 2 + cb) * 64] = acc[rb][cb];
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int r = 16 * rb + 4 * q + rg, cc = 32 * wid + 16 * cb + cl;
            *(lds_bf16*)(st + tile16_off<kFistaD * 2>(r, cc >> 3) + 2 * (cc & 7)) = (__bf16)acc[rb][cb][rg];
          }
    }
    __syncthreads();                            // residual tile complete
    BT16_STAMP(12);
    // g^T = W^T r^T for the wave's K/8 atoms, NB atom blocks (16 NB atoms) at a time, W fragments
    // three steps ahead through a 4-deep ring; stored as bf16.  Which atom an MFMA output row
    // stands for is free: block a, row i is atom 4 NB (i >> 2) + 4 a + (i & 3) of the group, so
    // that a lane's 4 rows x NB blocks are 4 NB CONSECUTIVE atoms -- g leaves in 16-byte stores,
    // 8 NB bytes per lane, whole 128-byte lines per row (8-byte pieces took 3x as long).  The W
    // fragment of such a block is a per-lane gather from the group's NB fragments of the pack.
    constexpr int NB = NAB < 4 ? NAB : 4;
    static_for<NAB / NB>([&](auto h_c) {
      constexpr int h = decltype(h_c)::value;
      f32x4 gt[NB][4];
      unsigned gofs[NB];                         // lane part of the gather (bytes); step and group go into the scalar offset
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const int ag = 4 * NB * (cl >> 2) + 4 * a + (cl & 3);            // atom of (block a, row cl) inside the group
        gofs[a] = (unsigned)(((ag >> 4) * 64 + (ag & 15) + 16 * q) * 16);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) gt[a][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // (descriptor + lane offset + uniform offset: with 64-bit per-fragment addresses the compiler sank every
      // load to just before its MFMAs to save the address registers -- the ring was gone, 13.6 us)
      auto wfrag2 = [&](int step, int a) __attribute__((always_inline)) {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                              w2rsrc, gofs[a], (unsigned)((step * NAB + NB * h) * 1024), 0));
      };
      constexpr int RD = LASSO_BT16_G2RING;      // ring depth (steps): RD - 1 steps of W fragments in flight
      static_assert(S2 % RD == 0, "ring depth must divide the steps");
      bf16x8 wf[RD][NB];
#pragma unroll
      for (int s = 0; s < RD - 1; ++s)
#pragma unroll
        for (int a = 0; a < NB; ++a) wf[s][a] = wfrag2(s, a);
      static_assert(RD % 2 == 0, "even ring depth (residual fragment parity)");
      bf16x8 rf[2][4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) rf[0][rb] = *(const lds_bf16x8*)(st + tile16_off<kFistaD * 2>(16 * rb + cl, q));
#pragma unroll 1
      for (int j = 0; j < S2 / RD; ++j) {
        static_for<RD>([&](auto u_c) {
          constexpr int u = decltype(u_c)::value;
          const int s = RD * j + u;
          const int sp = min(s + RD - 1, S2 - 1);
#pragma unroll
#ifndef LASSO_BT16_ABL_NOW2
          for (int a = 0; a < NB; ++a) wf[(u + RD - 1) % RD][a] = wfrag2(sp, a);
#else
          for (int a = 0; a < NB; ++a) asm volatile("" : "+v"(wf[(u + RD - 1) % RD][a]));
#endif
          const int sn = min(s + 1, S2 - 1);       // the residual fragments run one step ahead
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
            rf[(u + 1) & 1][rb] = *(const lds_bf16x8*)(st + tile16_off<kFistaD * 2>(16 * rb + cl, 4 * sn + q));
          __builtin_amdgcn_sched_barrier(0);       // prefetches first, in this order (left alone, the loads sink to their uses)
#pragma unroll
          for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
              gt[a][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][a], rf[u & 1][rb], gt[a][rb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      if (h == NAB / NB - 1) BT16_STAMP(13);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        // lane: atoms (K/8) w + 16 NB h + 4 NB q + (0 .. 4 NB - 1) of row 16 rb + cl (padded atoms are exact zeros)
        __bf16* const dst = Gg + (int64_t)(row0 + 16 * rb + cl) * K + (K / 8) * wid + 16 * NB * h + 4 * NB * q;
#pragma unroll
        for (int a2 = 0; a2 < NB / 2; ++a2) {
          const float v[8] = {gt[2 * a2][rb][0], gt[2 * a2][rb][1], gt[2 * a2][rb][2], gt[2 * a2][rb][3],
                              gt[2 * a2 + 1][rb][0], gt[2 * a2 + 1][rb][1], gt[2 * a2 + 1][rb][2], gt[2 * a2 + 1][rb][3]};
#ifndef LASSO_BT16_ABL_NOGST
          *reinterpret_cast<u32x4*>(dst + 8 * a2) = pack8(v);
#else
          { u32x4 pk = pack8(v); asm volatile("" :: "v"(pk), "v"(dst)); }
#endif
        }
      }
    });
    __syncthreads();                            // g is in memory (this CU reads it back); the residual tile is dead
    BT16_STAMP(1);
    {                                           // sum r0^2 of the tile -> red[48] (kept until the next gradient)
      const float r0w = wave_sum(rss0);
      if (lane == 0) red[40 + wid] = r0w;
      __syncthreads();
      if (tid == 0) {
        float a = 0.0f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) a += red[40 + w];
        red[48] = a;
      }
    }

    // ================================ line search: TB trials per batch ========================
    // ista.py:38-47: every trial t of an outer iteration uses the SAME point p, the same gradient g and the step
    // lr0 / eta^t -- the candidates are independent.  Round 4: the trials t = 0 .. TB-1 of a batch are computed
    // TOGETHER, one batch-wide decision picks the first t with F <= Q (ista.py:45; a second batch only if none
    // passes).  A config-3 solve takes 10 cross-workgroup decisions instead of 42, p / g / the W fragments are read
    // once per batch instead of once per trial, and the wasted work is TB - (accepted t + 1) trials of MFMAs
    // instead of a quarter trial plus a decision round trip per trial.
    //
    // TB accumulator sets of a 64-row tile would be 160 registers, so a batch walks the tile as two halves of
    // 32 rows (acc: TB x 2 row blocks x 2 column blocks = 80 registers), each half in passes of 64 atoms:
    //   element-wise role: thread = (row tid >> 4 of the half, 4 atoms tid & 15): p (LDS, 8 bytes), g (memory,
    //     8 bytes, two passes ahead), TB candidates z+ = S_{alpha lr_t}(p - lr_t g) in bf16 -> TB staging tiles
    //     [32][64] bf16 (4 KiB), the three element sums of ista.py:30-35 per trial in registers;
    //   MFMA role: wave w, r columns [32w, 32w + 32): per trial 8 MFMAs (2 steps x 2 row blocks x 2 column blocks)
    //     on A fragments from that trial's staging tile and W fragments that are shared by the TB trials.
    // Candidates of pass j + 1 are formed between the MFMAs of pass j (two independent instruction streams,
    // interleaved by hand and pinned with sched_barrier like the single-trial kernel did).  The 32 KiB scratch holds
    // 8 staging tiles: trials 0-2 are double-buffered by pass parity, trials 3-4 single-buffered -- a pass
    // therefore runs {MFMAs of trials 3, 4 | candidates j+1 of trials 0, 1}, barrier, {MFMAs of trials 0, 1, 2 |
    // candidates j+1 of trials 2, 3, 4}, barrier: the single-buffered tiles are consumed before they are refilled.
    static_assert(TB == 5, "the staging-tile schedule below is written for 5 trials per batch");
    __syncthreads();
    const float rss0_tile = red[48];             // (the scratch is about to become staging tiles)
    const int hrow = tid >> 4, e4 = tid & 15;    // element-wise role inside a half: row, 4-atom piece
    const int stW4 = hrow * (kPass * 2) + ((((e4 >> 1) ^ (hrow & 7)) << 4) | ((e4 & 1) << 3));
    const int ptE4 = hrow * (2 * K) + ((((e4 >> 1) ^ (hrow & 15)) << 4) | ((e4 & 1) << 3));
    const int ptO4 = hrow * (2 * K) + ((((8 | (e4 >> 1)) ^ (hrow & 15)) << 4) | ((e4 & 1) << 3));
    const unsigned gran_lane = (unsigned)min(tid, p.ntiles - 1) * (unsigned)(TB * 32);   // clamped: spare lanes add 0
    const int sweep_waves = (p.ntiles + 63) >> 6;
    lds_f32* const vote = red + 56;                // [8] per-wave "all granules arrived"
    double* const dpart = (double*)(st + 2 * kStageBytes + 2048);  // [8][4 TB + 1] (behind red[0 .. 224))

    // sums of batch epoch e (steps lr_t, trials t < nvalid count) -> index of the first accepted trial, -1 none,
    // -2 handshake timed out.  Every workgroup: same data, same order, same verdict.  Thread w reads workgroup w's
    // 2 TB granules; sums in double: per wave in butterfly order, then the waves' partial sums in index order.
    auto decide_batch = [&](unsigned e, const float (&hol)[TB], int nvalid, float& f_out) -> int {
      constexpr int NV = 4 * TB + 1;
      if (wid < sweep_waves) {
        const unsigned off = (unsigned)((e % kRing) * p.ntiles * (TB * 32)) + gran_lane;
        u32x4 ga[TB], gb[TB];
        bool ok = true;
        int spins = 0;
        // poll ONE granule per workgroup (the last trial's second one: 16 bytes) until it carries this epoch, then
        // fetch all 2 TB and check every tag -- the granules of a batch leave together, a straggler among them
        // (a store overtaken by a later one) only repeats the fetch.  Polling all of them moved 40 KiB per workgroup
        // and round through the fabric while the late workgroups were still computing.
        for (;;) {
          const u32x4 last = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off + 32 * (TB - 1) + 16, 0, 16);
          bool all = last[0] == e;
          if (__all(all)) {
#pragma unroll
            for (int t = 0; t < TB; ++t) {
              ga[t] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off + 32 * t, 0, 16);
              gb[t] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off + 32 * t + 16, 0, 16);
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) all = all && ga[t][0] == e && gb[t][0] == e;
            if (__all(all)) break;
          }
          if (++spins >= kStopSpinLimit ||
              ((spins & 63) == 63 && __hip_atomic_load(p.out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
        if (!ok) {
#pragma unroll
          for (int t = 0; t < TB; ++t) { ga[t] = (u32x4){0u, 0u, 0u, 0u}; gb[t] = (u32x4){0u, 0u, 0u, 0u}; }
        }
        const bool mine = tid < p.ntiles;
        ok = __all(ok);
        double sv[NV];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          sv[4 * t] = (double)__uint_as_float(ga[t][1]);      // sum r1^2
          sv[4 * t + 1] = (double)__uint_as_float(ga[t][2]);  // sum |z+|
          sv[4 * t + 2] = (double)__uint_as_float(ga[t][3]);  // sum dz g
          sv[4 * t + 3] = (double)__uint_as_float(gb[t][1]);  // sum dz^2
        }
        sv[4 * TB] = (double)__uint_as_float(gb[0][2]);       // sum r0^2
#pragma unroll
        for (int jj = 0; jj < NV; ++jj) sv[jj] = wave_sum_f64_dpp(mine ? sv[jj] : 0.0);
        if (lane == 0) {
#pragma unroll
          for (int jj = 0; jj < NV; ++jj) dpart[NV * wid + jj] = sv[jj];
          vote[wid] = ok ? 1.0f : 0.0f;
        }
      }
      __syncthreads();
      if (tid < TB) {
        const int t = tid;
        double sv[5] = {0., 0., 0., 0., 0.};
        bool ok = true;
        for (int w = 0; w < sweep_waves; ++w) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) sv[jj] += dpart[NV * w + 4 * t + jj];
          sv[4] += dpart[NV * w + 4 * TB];
          ok = ok && vote[w] != 0.0f;
        }
        const float rss1 = (float)sv[0], l1 = (float)sv[1], dzg = (float)sv[2], dz2 = (float)sv[3], rss0t = (float)sv[4];
        float holt = hol[0];
#pragma unroll
        for (int u = 1; u < TB; ++u) holt = t == u ? hol[u] : holt;
        const float f0 = __fmul_rn(0.5f, rss0t);                                     // ista.py:23
        const float al1 = __fmul_rn((float)p.alpha, l1);
        const float F = __fadd_rn(__fmul_rn(0.5f, rss1), al1);                      // :28
        const float Q = __fadd_rn(__fadd_rn(__fadd_rn(f0, dzg), __fmul_rn(holt, dz2)), al1);   // :32-35
        red[32 + t] = !ok ? 2.0f : (F <= Q ? 1.0f : 0.0f);                          // :45
        red[40 + t] = F;
        if (!ok && t == 0) abort_now();
      }
      __syncthreads();
      int acc_t = -1;
      f_out = 0.0f;
#pragma unroll
      for (int t = TB - 1; t >= 0; --t) {
        const float v = red[32 + t];
        if (v == 2.0f) acc_t = -2;
        else if (acc_t != -2 && t < nvalid && v == 1.0f) { acc_t = t; f_out = red[40 + t]; }
      }
      if (red[32] == 2.0f) acc_t = -2;
      __syncthreads();                             // red[] / dpart are free again (staging tiles, accept)
      return acc_t;
    };

    // the TB trials with steps (lrs[t], lams[t]) on this tile, sums published as epoch e
    auto run_batch = [&](const float (&lrs)[TB], const float (&lams)[TB], unsigned e) {
      float l1[TB], rss1[TB], dzg[TB];
      f32x2 dz22[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) { l1[t] = 0.f; rss1[t] = 0.f; dzg[t] = 0.f; dz22[t] = (f32x2){0.f, 0.f}; }
      BT16_STAMP(2);
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        f32x4 acc[TB][2][2];
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[t][rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const lds_char* const ph = pt + h * (32 * 2 * K);                      // this half's rows of the p tile
        const lds_char* const sth = st + 0;                                    // staging tiles (A rows = half rows)
        const __bf16* const gph = Gg + (int64_t)(row0 + 32 * h + hrow) * K + 4 * e4;      // + kPass * j
        auto load_g4 = [&](int j) { return *reinterpret_cast<const u32x2*>(gph + kPass * j); };
        using std::integral_constant;
        bf16x8 b[2][2][2];                         // [pass parity][step][col block]: W fragments, one pass ahead
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) b[0][u][cb] = wfrag1(u * 2 + cb);
        u32x2 gq[2];
        float pv[4], gv[4];
        unpack4(load_g4(0), gv);
        gq[1] = load_g4(1);
        gq[0] = load_g4(NP > 2 ? 2 : NP - 1);
        unpack4(*(const lds_u32x2*)(ph + ptE4), pv);
        // one packed pair (elements 2 e2, 2 e2 + 1) of trial TG's candidate for the pass whose p, g are in pv, gv
        auto cand_pair = [&](auto tg_c, int e2, float (&zn)[4]) __attribute__((always_inline)) {
          constexpr int TG = decltype(tg_c)::value;
          // plain fp32 VALU operations (this translation unit is built with -fno-slp-vectorize: a packed
          // v_pk_*_f32 re-formed by the SLP vectoriser costs 2.6 plain operations on gfx950)
#pragma unroll
          for (int e = 2 * e2; e < 2 * e2 + 2; ++e) {
            const float pe = pv[e], ge = gv[e];
#ifdef LASSO_BT16_ABL_NOCAND
            const float ze = pe; (void)ge;
#else
            const float ze = soft_threshold(fmaf(-lrs[TG], ge, pe), lams[TG]);    // ista.py:40 (rounded to bf16 by pack4)
#endif
            const float de = __fsub_rn(ze, pe);                                                            // :31
            l1[TG] += __builtin_fabsf(ze);
            dz22[TG][e & 1] = fmaf(de, de, dz22[TG][e & 1]);
            zn[e] = ze;
          }
        };
        auto load_a = [&](int slot, bf16x8 (&a)[2][2]) __attribute__((always_inline)) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
              a[u][rb] = *(const lds_bf16x8*)(sth + slot * kSlotBytes + (u ? stA1 : stA0) + 2048 * rb);
        };
        // 8 MFMAs of trial TM (pass parity PAR, A fragments in `a`) with -- TG >= 0 -- the candidate of trial TG for
        // the NEXT pass formed between them, and -- NXT >= 0 -- the A fragments of staging tile NXT fetched into `an`.
        // A wave issues in order: while it issues back-to-back MFMAs (one per ~17 cycles and SIMD) it issues nothing
        // else, and the element-wise work of a candidate is a chain of ~10 dependent VALU operations per element.
        // Written as {4 MFMAs, one pair's arithmetic} per scheduling region the two streams simply took turns
        // (ablations: MFMAs 9.6 us + candidates 8 us + barriers 6 us + LDS/unpack 8 us = the 31.5 us of a half; neither
        // sched_group_barrier mixes nor larger regions changed what hipcc emitted: MFMAs first, VALU behind).  So the
        // interleave is spelled out: the candidate's 40 operations are laid out stage by stage ACROSS its 4 elements
        // (consecutive operations are independent) and cut into 8 groups, group i = {5 VALU, MFMA i}, each pinned by
        // sched_barrier; LEAD > 0 moves that many operations of the later groups in front of the first MFMA (the
        // step behind a barrier: its A fragments are still in flight).
        auto step = [&](auto tm_c, auto par_c, auto tg_c, auto nxt_c, auto lead_c, bf16x8 (&a)[2][2], bf16x8 (&an)[2][2])
                        __attribute__((always_inline)) {
          constexpr int TM = decltype(tm_c)::value, PAR = decltype(par_c)::value, TG = decltype(tg_c)::value,
                        NXT = decltype(nxt_c)::value, LEAD = decltype(lead_c)::value;
          float cm[4], cv[4];
          u32x2 packed = {0u, 0u};
          // operation `op` (0 .. kCandOps - 1) of the candidate: stage op / 4 on element op % 4.  Per element 6.5 plain
          // fp32 VALU operations (round 4 started at 11: p - lr g is ONE fma -- the candidate is rounded to bf16
          // right after, the reference's own bf16 run rounds each of its three ATen results --, and sum dz g is not
          // formed here at all: with g = r0 W it equals sum r0 (r1 - r0), which the MFMA role evaluates on the two
          // residuals it holds anyway -- see the residuals below).  VALU and bf16 MFMAs share a SIMD almost
          // exclusively on gfx950 (tools/ubench/mix.hip: {1 MFMA + 5 v_fma} x 2 waves = 28.3 ns against 16.9 + 17.8
          // apart; a v_pk_fma_f32 costs 2.6 plain ones), so every operation saved here is time saved.
          auto cand_op = [&](auto op_c) __attribute__((always_inline)) {
            constexpr int op = decltype(op_c)::value, stg = op >> 2, e = op & 3;
#ifndef LASSO_BT16_ABL_NOCAND
            if constexpr (TG >= 0 && op < kCandOps) {
              constexpr int T_ = TG >= 0 ? TG : 0;
              if constexpr (stg == 0) cv[e] = fmaf(-lrs[T_], gv[e], pv[e]);                       // ista.py:40
              else if constexpr (stg == 1) cm[e] = __builtin_amdgcn_fmed3f(cv[e], -lams[T_], lams[T_]);
              else if constexpr (stg == 2) cv[e] = __fsub_rn(cv[e], cm[e]);                       // soft threshold
              else if constexpr (stg == 3) {                                                      // -> bf16 (MFMA operand): pairs
                if constexpr ((e & 1) == 0) {
                  const float vv[4] = {cv[e], cv[e + 1], 0.f, 0.f};
                  const u32x2 pk = pack4(vv);
                  packed[e >> 1] = pk[0];
                }
              }
              // the element sums of ista.py:30-35 take the candidate BEFORE its bf16 rounding (no unpack: one
              // operation per element less); the rounding errors are unbiased and <= 2^-9 |z| each -- ~1e-6 of a
              // sum over a batch, below the fp32 accumulation error of the sums themselves
              else if constexpr (stg == 4) cm[e] = __fsub_rn(cv[e], pv[e]);                       // dz, :31
              else if constexpr (stg == 5) l1[T_] += __builtin_fabsf(cv[e]);
              else dz22[T_][e & 1] = fmaf(cm[e], cm[e], dz22[T_][e & 1]);
            }
#endif
          };
          if constexpr (NXT >= 0) load_a(NXT, an);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int PER = (kCandOps - LEAD + 7) / 8;  // operations per group behind the lead
          static_for<LEAD>([&](auto o_c) { cand_op(o_c); });
          static_for<8>([&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            static_for<PER>([&](auto o_c) { cand_op(integral_constant<int, LEAD + PER * g + decltype(o_c)::value>{}); });
            if constexpr (TG >= 0 && g == 4)             // the candidate is complete once stage 3 is through (op 15)
              *(lds_u32x2*)(st + slot_of(TG >= 0 ? TG : 0, PAR ^ 1) * kSlotBytes + stW4) = packed;
            {
              constexpr int u = g >> 2, rb = (g >> 1) & 1, cb = g & 1;
              BT16_MFMA(acc[TM][rb][cb], a[u][rb], b[PAR][u][cb]);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        };
        bf16x8 aA[2][2], aB[2][2];
        __syncthreads();                           // the scratch is free (reductions of the previous phase / half)
        {                                          // candidates of pass 0, every trial
          static_for<TB>([&](auto t_c) {
            float zn[4];
            cand_pair(t_c, 0, zn);
            cand_pair(t_c, 1, zn);
            *(lds_u32x2*)(st + slot_of(decltype(t_c)::value, 0) * kSlotBytes + stW4) = pack4(zn);
          });
        }
        // pass j (parity PAR) that also forms the candidates of pass j + 1
        auto pass = [&](int j, auto par_c) __attribute__((always_inline)) {
          constexpr int PAR = decltype(par_c)::value;
          constexpr integral_constant<int, 0> Z0c{};
          BT16_PASS_BARRIER();                     // the candidates of pass j are complete; tiles of parity PAR^1 are free
          const unsigned fn = (unsigned)((j + 1) * 4) * 1024u;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              b[PAR ^ 1][u][cb] = __builtin_bit_cast(
                  bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1rsrc, lane16, fn + (u * 2 + cb) * 1024, 0));
          load_a(6, aA);                           // (j; trial 3)
          unpack4(*(const lds_u32x2*)(ph + ((PAR ^ 1) ? ptO4 : ptE4) + 256 * ((j + 1) >> 1)), pv);
          unpack4(gq[PAR ^ 1], gv);
          if (j + 3 < NP) gq[PAR ^ 1] = load_g4(j + 3);      // (no load that nobody waits for: a barrier's vmcnt(0) would)
          step(integral_constant<int, 3>{}, par_c, integral_constant<int, 0>{}, integral_constant<int, 7>{}, integral_constant<int, LASSO_BT16_LEAD>{}, aA, aB);
          step(integral_constant<int, 4>{}, par_c, integral_constant<int, 1>{}, integral_constant<int, slot_of(0, PAR)>{}, Z0c, aB, aA);
          BT16_PASS_BARRIER();                     // the single-buffered tiles (trials 3, 4) are consumed
          step(integral_constant<int, 0>{}, par_c, integral_constant<int, 2>{}, integral_constant<int, slot_of(1, PAR)>{}, Z0c, aA, aB);
          step(integral_constant<int, 1>{}, par_c, integral_constant<int, 3>{}, integral_constant<int, slot_of(2, PAR)>{}, Z0c, aB, aA);
          step(integral_constant<int, 2>{}, par_c, integral_constant<int, 4>{}, integral_constant<int, -1>{}, Z0c, aA, aB);
        };
#pragma unroll 1
        for (int j2 = 0; j2 < NP / 2 - 1; ++j2) {
          pass(2 * j2, integral_constant<int, 0>{});
          pass(2 * j2 + 1, integral_constant<int, 1>{});
        }
        pass(NP - 2, integral_constant<int, 0>{});
        f32x4 r0frag[2][2];                        // r0 of this half's rows (the gradient phase wrote it), for the residuals
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) r0frag[rb][cb] = R0g[((2 * h + rb) * 2 + cb) * 64];
        u32x4 xqh[2];                              // ... and x of these rows
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) xqh[rb] = XrG[(2 * h + rb) * 64];
        __syncthreads();
        {                                          // last pass (parity 1): MFMAs only
          constexpr integral_constant<int, 1> P1{};
          constexpr integral_constant<int, -1> NONE{};
          load_a(6, aA);
          constexpr integral_constant<int, 0> Z0c{};
          step(integral_constant<int, 3>{}, P1, NONE, integral_constant<int, 7>{}, Z0c, aA, aB);
          step(integral_constant<int, 4>{}, P1, NONE, integral_constant<int, slot_of(0, 1)>{}, Z0c, aB, aA);
          step(integral_constant<int, 0>{}, P1, NONE, integral_constant<int, slot_of(1, 1)>{}, Z0c, aA, aB);
          step(integral_constant<int, 1>{}, P1, NONE, integral_constant<int, slot_of(2, 1)>{}, Z0c, aB, aA);
          step(integral_constant<int, 2>{}, P1, NONE, NONE, Z0c, aA, aB);
        }
        // residuals of the half: r1 = acc - x, this lane's sum r1^2 per trial -- and sum dz g, evaluated as
        // sum r0 (r1 - r0): g = r0 W, so <g, dz> = <r0, dz W^T> = <r0, r1 - r0> (the gradient phase left r0 in the
        // workspace in this very layout; fp32)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            float xv[4];
            unpack4((u32x2){xqh[rb][2 * cb], xqh[rb][2 * cb + 1]}, xv);
            const f32x4 r0v = r0frag[rb][cb];
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const float res = acc[t][rb][cb][rg] - xv[rg];
                rss1[t] = fmaf(res, res, rss1[t]);
                dzg[t] = fmaf(r0v[rg], res - r0v[rg], dzg[t]);
              }
          }
        if (h == 0) BT16_STAMP(3);
      }
      BT16_STAMP(6);
      __syncthreads();                             // every wave is done with the staging tiles: red[] may be written
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const float s0 = wave_sum(rss1[t]), s1 = wave_sum(l1[t]), s2 = wave_sum(dzg[t]),
                    s3 = wave_sum(dz22[t][0] + dz22[t][1]);
        if (lane == 0) {
          red[64 + (4 * t) * kWaves + wid] = s0;
          red[64 + (4 * t + 1) * kWaves + wid] = s1;
          red[64 + (4 * t + 2) * kWaves + wid] = s2;
          red[64 + (4 * t + 3) * kWaves + wid] = s3;
        }
      }
      __syncthreads();
      if (tid < TB) {
        const int t = tid;
        float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int w = 0; w < kWaves; ++w) sm[jj] += red[64 + (4 * t + jj) * kWaves + w];
        const unsigned off = (unsigned)(((e % kRing) * p.ntiles + tile) * (TB * 32) + 32 * t);
        const u32x4 g0 = {e, __float_as_uint(sm[0]), __float_as_uint(sm[1]), __float_as_uint(sm[2])};
        const u32x4 g1 = {e, __float_as_uint(sm[3]), __float_as_uint(rss0_tile), 0u};
        __builtin_amdgcn_raw_buffer_store_b128(g0, grsrc, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(g1, grsrc, off + 16, 0, 16);
      }
      BT16_STAMP(7);
    };

    // ================================ step size (ista.py:86-90) =============================
    float lr_acc = (float)p.lr0, lam_acc = (float)(p.alpha * p.lr0), f_acc = __builtin_nanf("");
    int t_acc = 0;
    // accept step, AB passes at a time with ALL their g and z loads in flight together (kept packed: 8 registers per
    // pass): the phase is a chain of memory round trips (four passes per trip took 14.8 us at K = 1024).  Round 4:
    // two register sets -- a batch's loads fly under the previous batch's arithmetic, and the first batch's are
    // issued BEFORE the line search's decision is awaited (they do not depend on it).
    constexpr int AB = NP < LASSO_BT16_ACCEPT_BATCH ? NP : LASSO_BT16_ACCEPT_BATCH;
    static_assert(NP % AB == 0, "accept batch must divide the passes");
    u32x4 gA[AB], zA[AB], gB[AB], zB[AB];
    auto accept_load = [&](int jb, u32x4 (&gq)[AB], u32x4 (&zq)[AB]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < AB; ++u) gq[u] = load_g8(jb + u);
#pragma unroll
      for (int u = 0; u < AB; ++u) zq[u] = it == 0 ? load_z8p(Z0g, p.ldz0, z0vec, jb + u) : load_z8p(Zg, p.ldz, zvec, jb + u);
    };
    auto accept_compute = [&](int jb, const u32x4 (&gq)[AB], const u32x4 (&zq)[AB], float& dsum) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < AB; ++u) {
        const int j = jb + u;
        float pv[8], gv[8], zn[8], yn[8], zo1[8];
        lds_u32x4* const pp = (lds_u32x4*)(pt + pt_off(j));
        unpack8(*pp, pv);
        unpack8(gq[u], gv);
        unpack8(zq[u], zo1);
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
          zn[e8] = bf16_round(soft_threshold(fmaf(-lr_acc, gv[e8], pv[e8]), lam_acc));   // the trial's candidate, ista.py:40
          dsum += __builtin_fabsf(__fsub_rn(zo1[e8], zn[e8]));                        // :93
          yn[e8] = __fadd_rn(zn[e8], __fmul_rn(coef, __fsub_rn(zn[e8], zo1[e8])));    // :99-100
        }
        store_z8(j, zn);                                                              // :102
        *pp = pack8(yn);                                                              // next point, in place
      }
    };
    if (p.backtrack) {
      // batch b holds the trials s0 .. s0 + TB - 1 with the steps lr0 / eta^s (in double like the reference's
      // python floats, ista.py:47)
      double lr_d = p.lr0;
      for (int s0 = 0;; s0 += TB) {
        float lrs[TB], lams[TB], hols[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          lrs[t] = (float)lr_d; lams[t] = (float)(p.alpha * lr_d); hols[t] = (float)(0.5 / lr_d);
          lr_d = lr_d / p.eta;
        }
        run_batch(lrs, lams, ++epoch);
        accept_load(0, gA, zA);                    // (independent of the decision awaited next)
        BT16_STAMP(4);
        float fv = 0.f;
        const int nvalid = min(TB, kMaxTrials - s0);
        const int v = decide_batch(epoch, hols, nvalid, fv);
        BT16_STAMP(5);
        if (v == -2) { aborted = true; break; }
        if (v >= 0) {
          lr_acc = lrs[0]; lam_acc = lams[0];
#pragma unroll
          for (int t = 1; t < TB; ++t) { lr_acc = v == t ? lrs[t] : lr_acc; lam_acc = v == t ? lams[t] : lam_acc; }
          t_acc = s0 + v; f_acc = fv;
          break;
        }
        if (s0 + TB >= kMaxTrials) { warned = true; t_acc = kMaxTrials - 1; break; }   // :48-52: revert to lr0
      }
      if (aborted) break;
    }
    if (blockIdx.x == 0 && tid == 0) {
      if (p.trials) p.trials[it] = t_acc + 1;
      if (p.lrs) p.lrs[it] = lr_acc;
      if (p.fvals) p.fvals[it] = f_acc;
    }

    // ================================ accept: z+, |z - z+|, momentum (ista.py:93-102) ========
    BT16_STAMP(9);
    float dsum = 0.0f;
    if (!p.backtrack) accept_load(0, gA, zA);
#pragma unroll 1
    for (int jb = 0; jb < NP; jb += 2 * AB) {
      if (jb + AB < NP) accept_load(jb + AB, gB, zB);        // the next batch's round trip under this batch's arithmetic
      accept_compute(jb, gA, zA, dsum);
      if (jb + AB < NP) {
        if (jb + 2 * AB < NP) accept_load(jb + 2 * AB, gA, zA);
        accept_compute(jb + AB, gB, zB, dsum);
      }
    }
    iterations = it + 1;
    __syncthreads();                            // the p tile is complete for the next gradient
    BT16_STAMP(10);
    if (p.budget >= 0.0f) {
      // global stop rule: one 8-byte granule per workgroup, swept by wave 0 of every workgroup
      dsum = wave_sum(dsum);
      if (lane == 0) red[wid] = dsum;
      __syncthreads();
      if (tid == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tsum += red[w];
        __hip_atomic_store(p.dgran + (size_t)(it % kRing) * p.ntiles + tile,
                           ((unsigned long long)(unsigned)(it + 1) << 32) | __float_as_uint(tsum),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (wid == 0) {
        const unsigned want = (unsigned)(it + 1);
        const unsigned long long* const row = p.dgran + (size_t)(it % kRing) * p.ntiles;
        double part = 0.0;
        bool ok = true;
        for (int wg = lane; wg < p.ntiles; wg += 64) {
          unsigned long long gv;
          int spins = 0;
          for (;;) {
            gv = __hip_atomic_load(row + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(gv >> 32) == want) break;
            if (++spins >= kStopSpinLimit ||
                ((spins & 63) == 63 &&
                 __hip_atomic_load(p.out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              ok = false;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
          part += __uint_as_float((unsigned)gv);
        }
        ok = __all(ok);
        const float total = (float)wave_sum_f64(part);
        if (lane == 0) {
          red[32] = !ok ? 2.0f : (total <= p.budget ? 1.0f : 0.0f);
          red[34] = total;
          if (!ok) abort_now();
        }
      }
      __syncthreads();
      const float v = red[32];
      last_delta = red[34];
      __syncthreads();
      if (v == 2.0f) { aborted = true; break; }
      if (v == 1.0f) break;                     // :93-95
    }
  }
  if (blockIdx.x == 0 && tid == 0 && !aborted) {
    p.out[0] = iterations;
    p.out[1] = __float_as_int(last_delta);
    p.out[3] = warned ? 1 : 0;
  }
}

template <int K>
hipError_t persist_k(const Bt16PersistParams& p, hipStream_t stream) {
  const size_t lds = (size_t)kRows * K * 2 + kScratchBytes;
  const void* fn = reinterpret_cast<const void*>(&bt16_persist_kernel<K>);
  if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt16_persist_kernel<K>, dim3(p.ntiles), dim3(kThreads), lds, stream, p);
  return hipGetLastError();
}

template <int K>
hipError_t persist_occ(int* per_cu) {
  const size_t lds = (size_t)kRows * K * 2 + kScratchBytes;
  const void* fn = reinterpret_cast<const void*>(&bt16_persist_kernel<K>);
  if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, fn, kThreads, lds);
}

}  // namespace

size_t bt16_persist_trial_granule_bytes(int ntiles) { return (size_t)kRing * ntiles * 32 * TB; }
size_t bt16_persist_granule_bytes(int ntiles) {
  return bt16_persist_trial_granule_bytes(ntiles) + (size_t)kRing * ntiles * 8;
}

hipError_t bt16_persist_occupancy(int kpad, int* per_cu) {
  switch (kpad) {
    case 256: return persist_occ<256>(per_cu);
    case 512: return persist_occ<512>(per_cu);
    case 1024: return persist_occ<1024>(per_cu);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_bt16_persist(const Bt16PersistParams& p, int kpad, hipStream_t stream) {
  switch (kpad) {
    case 256: return persist_k<256>(p, stream);
    case 512: return persist_k<512>(p, stream);
    case 1024: return persist_k<1024>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lasso
