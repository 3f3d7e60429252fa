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
#define LASSO_BT16_ACCEPT_BATCH 8   // passes of the accept step whose loads are in flight together
#endif
#ifndef LASSO_BT16_ACCEPT_PIPE
#define LASSO_BT16_ACCEPT_PIPE 0
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
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

constexpr int kRing = 8;             // granule ring (epochs in flight never span more than 4)
constexpr int kMaxTrials = 1000;     // ista.py:17 (maxiter=1000)
constexpr int kPass = 64;            // atoms per trial pass
constexpr int kStageBytes = kRows * kPass * 2;       // 8 KiB
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
  const bool zvec = (p.ldz & 7) == 0 && (((uintptr_t)p.Z) & 15) == 0 && (p.k & 7) == 0;
  const bool z0vec = Z0g && (p.ldz0 & 7) == 0 && (((uintptr_t)p.Z0) & 15) == 0 && (p.k & 7) == 0;
  const bool erow_ok = (row0 + erow) < p.n;
  const __amdgpu_buffer_rsrc_t grsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.gran, 0, kRing * p.ntiles * 32, 0x00020000);
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
  // the same 8 values, still packed (exact: they are bf16 in memory).  Aligned rows (vec) go through a buffer descriptor
  // of the tile's rows with the offset out of range where the thread has nothing to read (reads 0), the offset opaque:
  // with `if (row in range) load` hipcc gave every one of these loads a branch of its own and an s_waitcnt vmcnt(0)
  // behind it -- the accept step's "all loads in flight together" was a chain of 8 memory round trips per batch (round 5)
  const int rows_here = min((int)kRows, p.n - row0);
  auto z_rsrc = [&](const __bf16* base, int64_t ld) {
    const int64_t bytes = base ? (int64_t)rows_here * ld * 2 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(base ? base + (int64_t)row0 * ld : Zg), 0,
                                             (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t z0rs = z_rsrc(Z0g, p.ldz0), zrs = z_rsrc(Zg, p.ldz);
  auto z_off = [&](int64_t ld, int j, int tok) {
    const int a0 = kPass * j + 8 * ec8;
    unsigned o = (erow_ok && a0 < p.k) ? (unsigned)((erow * (int)ld + a0) * 2 + tok) : 0xfffffff0u;
    asm volatile("" : "+v"(o));
    return o;
  };
  auto load_z8p = [&](const __bf16* base, const __amdgpu_buffer_rsrc_t rs, int64_t ld, bool vec, int j, int tok = 0) -> u32x4 {
    if (vec) return __builtin_amdgcn_raw_buffer_load_b128(rs, z_off(ld, j, tok), 0, 0);
    const int a0 = kPass * j + 8 * ec8;
    if (!base || !erow_ok || a0 >= p.k) return (u32x4){0u, 0u, 0u, 0u};
    const __bf16* src = base + (int64_t)(row0 + erow) * ld + a0;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (a0 + e < p.k) ? (float)src[e] : 0.0f;
    return pack8(v);
  };
  auto store_z8 = [&](int j, const float (&v)[8]) {
    if (zvec) {
      __builtin_amdgcn_raw_buffer_store_b128(pack8(v), zrs, z_off(p.ldz, j, 0), 0, 0);
      return;
    }
    const int a0 = kPass * j + 8 * ec8;
    if (!erow_ok || a0 >= p.k) return;
    __bf16* dst = Zg + (int64_t)(row0 + erow) * p.ldz + a0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (a0 + e < p.k) dst[e] = (__bf16)v[e];
  };
  // g rows are padded to whole tiles in the workspace, so every thread's row exists
  const __bf16* const grow = Gg + (int64_t)(row0 + erow) * K + 8 * ec8;   // + kPass * j
  auto load_g8 = [&](int j) { return *reinterpret_cast<const u32x4*>(grow + kPass * j); };

  // ---- x in GEMM-1's accumulator layout (4 rows per packed entry), resident ------------------
  u32x2 Xr[4][2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float v[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = row0 + 16 * rb + 4 * q + rg, cc = 32 * wid + 16 * cb + cl;
        v[rg] = (r < p.n && cc < p.d) ? (float)Xg[(int64_t)r * p.ldx + cc] : 0.0f;
      }
      Xr[rb][cb] = pack4(v);
    }
  // ---- y_0 = z_0 (ista.py:76-78) -> the p tile -------------------------------------------------
#pragma unroll 1
  for (int j = 0; j < NP; ++j) {
    float v[8];
    load_z8(Z0g, p.ldz0, z0vec, j, v);
    *(lds_u32x4*)(pt + tile16_off<K * 2>(erow, 8 * j + ec8)) = pack8(v);
  }
  __syncthreads();

  // residual of an accumulator set: acc <- acc - x, returns this lane's sum r^2
  auto residual = [&](f32x4 (&acc)[4][2]) {
    float rss = 0.0f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        float xv[4];
        u32x2 xr = Xr[rb][cb];
        asm volatile("" : "+v"(xr));             // unpack here, not hoisted out of the solve (see unpack8_here)
        unpack4(xr, xv);
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
      gemm1_bf16_deep<K>(pt, w1rsrc, lane16, lane, acc);
      BT16_STAMP(11);
      rss0 = residual(acc);
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

    // Decision of the trial published as epoch e with 0.5 / step (every workgroup: same data, same
    // order, same verdict): 0 = rejected, 1 = accepted, 2 = handshake timed out.  Thread t reads
    // workgroup t's two granules; the loads are issued one double pass before the verdict is
    // taken (sweep_issue), so their round trip hides behind that pass; a granule that had not
    // arrived by then is polled (bounded).  Sums in double: per wave in butterfly order, then
    // the waves' partial sums in index order.
    const unsigned gran_lane = (unsigned)min(tid, p.ntiles - 1) * 32u;   // clamped: every lane loads, spare lanes add 0
    const int sweep_waves = (p.ntiles + 63) >> 6;
    lds_f32* const vote = red + 56;                // [8] per-wave "all granules arrived"
    double* const dpart = (double*)(st + 2 * kStageBytes + 256);   // [8][5]
    auto sweep_issue = [&](unsigned e, u32x4& ga, u32x4& gb) {
      const unsigned off = (unsigned)((e % kRing) * p.ntiles * 32) + gran_lane;
      ga = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off, 0, 16);
      gb = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off + 16, 0, 16);
    };
    auto decide = [&](unsigned e, u32x4 ga, u32x4 gb, float half_over_lr, float& f_out) {
      if (wid < sweep_waves) {
        const unsigned off = (unsigned)((e % kRing) * p.ntiles * 32) + gran_lane;
        bool ok = true;
        int spins = 0;
        while (ga[0] != e || gb[0] != e) {
          if (++spins >= kStopSpinLimit ||
              ((spins & 63) == 63 && __hip_atomic_load(p.out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          ga = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off, 0, 16);
          gb = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off + 16, 0, 16);
        }
        const bool mine = tid < p.ntiles;
        double sv[5] = {(double)__uint_as_float(ga[1]), (double)__uint_as_float(ga[2]), (double)__uint_as_float(ga[3]),
                        (double)__uint_as_float(gb[1]), (double)__uint_as_float(gb[2])};
        ok = __all(ok);
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) sv[jj] = wave_sum_f64(mine ? sv[jj] : 0.0);
        if (lane == 0) {
#pragma unroll
          for (int jj = 0; jj < 5; ++jj) dpart[5 * wid + jj] = sv[jj];
          vote[wid] = ok ? 1.0f : 0.0f;
        }
      }
      __syncthreads();
      if (tid == 0) {
        double sv[5] = {0., 0., 0., 0., 0.};
        bool ok = true;
        for (int w = 0; w < sweep_waves; ++w) {
#pragma unroll
          for (int jj = 0; jj < 5; ++jj) sv[jj] += dpart[5 * w + jj];
          ok = ok && vote[w] != 0.0f;
        }
        const float rss1 = (float)sv[0], l1 = (float)sv[1], dzg = (float)sv[2], dz2 = (float)sv[3], rss0t = (float)sv[4];
        const float f0 = __fmul_rn(0.5f, rss0t);                                     // ista.py:23
        const float al1 = __fmul_rn((float)p.alpha, l1);
        const float F = __fadd_rn(__fmul_rn(0.5f, rss1), al1);                      // :28
        const float Q = __fadd_rn(__fadd_rn(__fadd_rn(f0, dzg), __fmul_rn(half_over_lr, dz2)), al1);   // :32-35
        red[32] = !ok ? 2.0f : (F <= Q ? 1.0f : 0.0f);                              // :45
        red[34] = F;
        if (!ok) abort_now();
      }
      __syncthreads();
      f_out = red[34];
      return red[32];                               // rewritten at the earliest a barrier later (the next pass / accept)
    };

    [[maybe_unused]] const unsigned trial_e0 = epoch + 1;           // epoch of this outer iteration's first trial
    // trial with step (lr, lam): candidate passes -> staging -> GEMM-1, sums published as epoch e.
    // A trial after the first is speculative -- it is needed only if the previous one (epoch
    // check_e, 0 = none) gets rejected: that verdict is taken kCheck double passes into this
    // trial (its granules have had that long to arrive) and an accepted predecessor ends the
    // trial there instead of after all NP passes.  Returns the verdict (-1: none taken).
    constexpr int kIssue = (NP / 2 - 1) < LASSO_BT16_CHECK ? (NP / 2 - 1) : LASSO_BT16_CHECK;
    constexpr int kCheck = (NP / 2 - 1) < LASSO_BT16_CHECK + 1 ? (NP / 2 - 1) : LASSO_BT16_CHECK + 1;
    auto run_trial = [&](float lr, float lam, unsigned e, unsigned check_e, float hol_check, float& f_check) -> float {
      float l1 = 0.0f;
      f32x2 dzg2 = {0.f, 0.f}, dz22 = {0.f, 0.f};
      f32x4 acc[4][2];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 b[2][2][2];                        // [pass parity][step][col block] W fragments, two passes ahead
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) b[par][u][cb] = wfrag1((2 * par + u) * 2 + cb);
      u32x4 gq[2] = {load_g8(0), load_g8(1)};   // g of the next two passes (bf16, 16 B each)
      // two elements of a candidate and their share of the three element sums -- PLAIN fp32 VALU operations: beside
      // MFMAs a packed v_pk_mul/add/fma_f32 costs more than the two plain operations it replaces on gfx950
      // (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; measured here: 1.047 -> 1.011 ms per config-3 solve
      // with the packed form of rounds 2-3 replaced); this file is built with -fno-slp-vectorize so that the compiler
      // does not re-pack them.  The two sums accumulate by fma.
      // (round 5: the pair is rounded to bf16 by ONE v_cvt_pk_bf16_f32 whose result IS the staged operand; the rounded
      // values the sums need are its two halves shifted / masked back to fp32 -- before, every element was converted on
      // its own and the eight of a pass converted again for the store: 12 conversions per pass instead of 4)
      auto cand2 = [&](const float (&pv)[8], const float (&gv)[8], u32x4& zn, int e2) __attribute__((always_inline)) {
        const int e = 2 * e2;
        const float v0 = soft_threshold(__fsub_rn(pv[e], __fmul_rn(lr, gv[e])), lam);                    // ista.py:40
        const float v1 = soft_threshold(__fsub_rn(pv[e + 1], __fmul_rn(lr, gv[e + 1])), lam);
        unsigned pk;                             // {bf16(v0), bf16(v1)}, round to nearest even -- one instruction for the pair
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v0), "v"(v1));
        const float ze[2] = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float de = __fsub_rn(ze[h], pv[e + h]);                                                  // :31
          l1 += __builtin_fabsf(ze[h]);
          dzg2[h] = fmaf(de, gv[e + h], dzg2[h]);
          dz22[h] = fmaf(de, de, dz22[h]);
        }
        zn[e2] = pk;
      };
      // candidate values of pass j (this thread: one row, 8 atoms) -> staging tile j & 1
      auto candidates = [&](int j, int par) __attribute__((always_inline)) {
        float pv[8], gv[8];
        u32x4 zn;
        unpack8(*(const lds_u32x4*)(pt + (par ? ptO : ptE) + 256 * (j >> 1)), pv);
        unpack8(gq[par], gv);
        gq[par] = load_g8(min(j + 2, NP - 1));
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) cand2(pv, gv, zn, e2);
        *(lds_u32x4*)(st + par * kStageBytes + stW) = zn;
      };
      // Between two barriers a wave holds the MFMAs of pass jm (staging tile pm) and the
      // element-wise work of pass jc (-> staging tile pm ^ 1): independent streams, written out
      // interleaved -- two MFMAs, then one element's VALU work while the matrix pipe is busy --
      // and pinned in that order (left to itself the scheduler runs them one after the other and
      // the two pipes take turns idling).  Afterwards pass jm's W fragments are replaced by pass
      // jm + 2's.
      auto fused_pass = [&](int jm, int pm, int jc) __attribute__((always_inline)) {
        const lds_char* const sb = st + pm * kStageBytes;
        const int pc = pm ^ 1;
        bf16x8 a[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) a[u][rb] = *(const lds_bf16x8*)(sb + (u ? stA1 : stA0) + 2048 * rb);
        float pv[8], gv[8];
        u32x4 zn;
        unpack8(*(const lds_u32x4*)(pt + (pc ? ptO : ptE) + 256 * (jc >> 1)), pv);
        unpack8(gq[pc], gv);
        gq[pc] = load_g8(min(jc + 2, NP - 1));
        __builtin_amdgcn_sched_barrier(0);
        // two elements at a time on the packed fp32 VALU (v_pk_mul_f32 / v_pk_add_f32: the same
        // IEEE operations, two lanes of work per instruction); the sums run as two partial sums
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {          // 4 of the pass's 16 MFMAs
            const int i = 4 * e2 + m, u = i >> 3, rb = (i >> 1) & 3, cb = i & 1;
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][rb], b[pm][u][cb], acc[rb][cb], 0, 0, 0);
          }
          cand2(pv, gv, zn, e2);
          __builtin_amdgcn_sched_barrier(0);
        }
        *(lds_u32x4*)(st + pc * kStageBytes + stW) = zn;
        const unsigned fn = (unsigned)(min(jm + 2, NP - 1) * 4) * 1024u;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            b[pm][u][cb] = __builtin_bit_cast(
                bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1rsrc, lane16, fn + (u * 2 + cb) * 1024, 0));
      };
      if (e == trial_e0 + 1) BT16_STAMP(2);
      if (e == trial_e0 + 2) BT16_STAMP(8);
      candidates(0, 0);
      float verdict = -1.0f;
      u32x4 ga = {0u, 0u, 0u, 0u}, gb = {0u, 0u, 0u, 0u};
#pragma unroll 1
      for (int j2 = 0; j2 < kIssue; ++j2) {
        __syncthreads();                        // staging tile 0 complete, tile 1 free
        fused_pass(2 * j2, 0, 2 * j2 + 1);
        __syncthreads();                        // staging tile 1 complete, tile 0 free
        fused_pass(2 * j2 + 1, 1, 2 * j2 + 2);
      }
      if (check_e) sweep_issue(check_e, ga, gb);
      if (e == trial_e0 + 1) BT16_STAMP(3);
#pragma unroll 1
      for (int j2 = kIssue; j2 < kCheck; ++j2) {
        __syncthreads();
        fused_pass(2 * j2, 0, 2 * j2 + 1);
        __syncthreads();
        fused_pass(2 * j2 + 1, 1, 2 * j2 + 2);
      }
      if (check_e) {
        if (e == trial_e0 + 1) BT16_STAMP(4);
        verdict = decide(check_e, ga, gb, hol_check, f_check);
        if (e == trial_e0 + 1) BT16_STAMP(5);
        if (verdict != 0.0f) return verdict;    // accepted (or timed out): nobody will ask for this trial
      }
#pragma unroll 1
      for (int j2 = kCheck; j2 < NP / 2 - 1; ++j2) {
        __syncthreads();                        // staging tile 0 complete, tile 1 free
        fused_pass(2 * j2, 0, 2 * j2 + 1);
        __syncthreads();                        // staging tile 1 complete, tile 0 free
        fused_pass(2 * j2 + 1, 1, 2 * j2 + 2);
      }
      __syncthreads();
      fused_pass(NP - 2, 0, NP - 1);
      __syncthreads();
      {                                         // last pass: MFMAs only
        const lds_char* const sb = st + kStageBytes;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x8 a[4];
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) a[rb] = *(const lds_bf16x8*)(sb + (u ? stA1 : stA0) + 2048 * rb);
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b[1][u][cb], acc[rb][cb], 0, 0, 0);
        }
      }
      if (e == trial_e0 + 1) BT16_STAMP(6);
      float dzg = dzg2[0] + dzg2[1], dz2 = dz22[0] + dz22[1];
      float rss1 = residual(acc);
      rss1 = wave_sum(rss1); l1 = wave_sum(l1); dzg = wave_sum(dzg); dz2 = wave_sum(dz2);
      if (lane == 0) { red[4 * wid] = rss1; red[4 * wid + 1] = l1; red[4 * wid + 2] = dzg; red[4 * wid + 3] = dz2; }
      __syncthreads();                          // red[] complete; both staging tiles are free
      if (tid == 0) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) s[jj] += red[4 * w + jj];
        const unsigned off = (unsigned)(((e % kRing) * p.ntiles + tile) * 32);
        const u32x4 g0 = {e, __float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2])};
        const u32x4 g1 = {e, __float_as_uint(s[3]), __float_as_uint(red[48]), 0u};
        __builtin_amdgcn_raw_buffer_store_b128(g0, grsrc, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(g1, grsrc, off + 16, 0, 16);
      }
      if (e == trial_e0 + 1) BT16_STAMP(7);
      return verdict;
    };
    // ================================ step size (ista.py:86-90) =============================
    float lr_acc = (float)p.lr0, lam_acc = (float)(p.alpha * p.lr0), f_acc = __builtin_nanf("");
    int t_acc = 0;
    if (p.backtrack) {
      // step s computes trial s (step / eta^s, in double like the reference's python floats, :47)
      // and, a few passes into it, decides trial s-1
      double lr_d = p.lr0;
      float lr_prev = 0.f, lam_prev = 0.f, hol_prev = 0.f;
      for (int s = 0;; ++s) {
        const float lr_s = (float)lr_d, lam_s = (float)(p.alpha * lr_d), hol_s = (float)(0.5 / lr_d);
        float fv = 0.f, verdict;
        if (s < kMaxTrials) {
          const unsigned e_prev = s >= 1 ? epoch : 0u;
          verdict = run_trial(lr_s, lam_s, ++epoch, e_prev, hol_prev, fv);
        } else {
          u32x4 ga, gb;
          sweep_issue(epoch, ga, gb);
          verdict = decide(epoch, ga, gb, hol_prev, fv);
        }
        if (s >= 1) {
          if (verdict == 2.0f) { aborted = true; break; }
          if (verdict == 1.0f) { lr_acc = lr_prev; lam_acc = lam_prev; t_acc = s - 1; f_acc = fv; break; }
          if (s >= kMaxTrials) { warned = true; t_acc = kMaxTrials - 1; break; }   // :48-52: revert to lr0
        }
        lr_prev = lr_s; lam_prev = lam_s; hol_prev = hol_s;
        lr_d = lr_d / p.eta;
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
    // AB passes at a time, ALL their g and z loads in flight together (kept packed: 8 registers per pass): the
    // phase is a chain of memory round trips, and with four passes per trip it took 14.8 us at K = 1024
    auto accept_pass = [&](int j, u32x4 gq1, u32x4 zq1) __attribute__((always_inline)) {
      float pv[8], gv[8], zn[8], yn[8], zo1[8];
      lds_u32x4* const pp = (lds_u32x4*)(pt + pt_off(j));
      unpack8(*pp, pv);
      unpack8(gq1, gv);
      unpack8(zq1, zo1);
#pragma unroll
      for (int e8 = 0; e8 < 8; ++e8) {
        zn[e8] = bf16_round(soft_threshold(__fsub_rn(pv[e8], __fmul_rn(lr_acc, gv[e8])), lam_acc));
        dsum += __builtin_fabsf(__fsub_rn(zo1[e8], zn[e8]));                        // :93
        yn[e8] = __fadd_rn(zn[e8], __fmul_rn(coef, __fsub_rn(zn[e8], zo1[e8])));    // :99-100
      }
      store_z8(j, zn);                                                              // :102
      *pp = pack8(yn);                                                              // next point, in place
    };
#if LASSO_BT16_ACCEPT_PIPE
    // A/B knob: batches of 4 passes, the loads of batch b + 2 issued before batch b is worked on (two batches in flight
    // under every batch's arithmetic instead of two serial round trips of 8 passes)
    {
      constexpr int AB2 = NP < 4 ? NP : 4, NBT = NP / AB2;
      u32x4 gq[NBT][AB2], zq[NBT][AB2];
      int tok = 0;                               // an opaque zero in every load address: the loads of batch b + 2 cannot
                                                 // be moved in front of batch b - 1's arithmetic (hipcc hoists them all otherwise)
      auto issue = [&](auto b_c) __attribute__((always_inline)) {
        constexpr int b = decltype(b_c)::value;
#pragma unroll
        for (int u = 0; u < AB2; ++u) gq[b][u] = *reinterpret_cast<const u32x4*>(grow + kPass * (AB2 * b + u) + tok);
        {
          const bool first = it == 0;
          const __amdgpu_buffer_rsrc_t rsel = first ? z0rs : zrs;
          const int64_t ldsel = first ? p.ldz0 : p.ldz;
          if (first ? z0vec : zvec) {
#pragma unroll
            for (int u = 0; u < AB2; ++u) zq[b][u] = __builtin_amdgcn_raw_buffer_load_b128(rsel, z_off(ldsel, AB2 * b + u, tok), 0, 0);
          } else {
#pragma unroll
            for (int u = 0; u < AB2; ++u) zq[b][u] = load_z8p(first ? Z0g : Zg, rsel, ldsel, false, AB2 * b + u, tok);
          }
        }
      };
      issue(std::integral_constant<int, 0>{});
      if constexpr (NBT > 1) issue(std::integral_constant<int, 1>{});
      static_for<NBT>([&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
        if constexpr (b + 2 < NBT) issue(std::integral_constant<int, b + 2>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < AB2; ++u) accept_pass(AB2 * b + u, gq[b][u], zq[b][u]);
        asm volatile("" : "+v"(tok) : "v"(dsum));
        __builtin_amdgcn_sched_barrier(0);
      });
    }
#else
    constexpr int AB = NP < LASSO_BT16_ACCEPT_BATCH ? NP : LASSO_BT16_ACCEPT_BATCH;
    static_assert(NP % AB == 0, "accept batch must divide the passes");
#pragma unroll 1
    for (int jb = 0; jb < NP; jb += AB) {
      u32x4 gq[AB], zq[AB];
#pragma unroll
      for (int u = 0; u < AB; ++u) gq[u] = load_g8(jb + u);
      // (the uniform choices -- first iteration reads z0, aligned rows or not -- OUTSIDE the batch: a branch round each
      // load, uniform or not, ends in its own s_waitcnt vmcnt(0))
      {
        const bool first = it == 0;
        const __amdgpu_buffer_rsrc_t rsel = first ? z0rs : zrs;
        const int64_t ldsel = first ? p.ldz0 : p.ldz;
        if (first ? z0vec : zvec) {
#pragma unroll
          for (int u = 0; u < AB; ++u) zq[u] = __builtin_amdgcn_raw_buffer_load_b128(rsel, z_off(ldsel, jb + u, 0), 0, 0);
        } else {
#pragma unroll
          for (int u = 0; u < AB; ++u) zq[u] = load_z8p(first ? Z0g : Zg, rsel, ldsel, false, jb + u);
        }
      }
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
          zn[e8] = bf16_round(soft_threshold(__fsub_rn(pv[e8], __fmul_rn(lr_acc, gv[e8])), lam_acc));
          dsum += __builtin_fabsf(__fsub_rn(zo1[e8], zn[e8]));                        // :93
          yn[e8] = __fadd_rn(zn[e8], __fmul_rn(coef, __fsub_rn(zn[e8], zo1[e8])));    // :99-100
        }
        store_z8(j, zn);                                                              // :102
        *pp = pack8(yn);                                                              // next point, in place
      }
    }
#endif
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

size_t bt16_persist_granule_bytes(int ntiles) {
  return (size_t)kRing * ntiles * 32 + (size_t)kRing * ntiles * 8;
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
