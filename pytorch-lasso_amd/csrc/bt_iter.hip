// One launch per OUTER iteration of the backtracking line search on fp32 tensors (ista.py:17-54 inside the loop
// :79-102), round 5.  The multi-launch form (backtrack.hip) spent, per outer iteration at BASELINE config 3
// (n=16384, d=256, k=1024), 152 us in the gradient kernel, 353 us in the five-trial launch (27 of them the burst that
// re-loads p and g of every tile), 49 us in the HBM-bound accept pass and ~23 us in seven launch boundaries.  Every one
// of those steps is TILE-LOCAL except the decision F <= Q itself (sums over the whole batch), so one workgroup can take
// a 16-row tile through all of them back to back:
//
//   accept of iteration i-1   z_i = S(p - lr_acc g), sum |z - z_i|, y_i = z_i + c (z_i - z)       (:40,:93,:99-102)
//                             with the step the decision kernel of iteration i-1 left on the device; z_i -> Z,
//                             y_i -> Y and, as the A operand of what follows, into the LDS tile
//   gradient at p_i           r0 = p_i W^T - x, sum r0^2, g = r0 W                                 (:22-24)
//                             g stays in REGISTERS (MFMA C layout, 32 per lane) for the trials
//   trials t = 0 .. nt-1      z+ = S(p_i - lr_t g) into the LDS tile, r1 = z+ W^T - x,
//                             tile sums {sum r1^2, sum |z+|, sum dz g, sum dz^2}                   (:26-35,:40)
//   g -> G                    from the registers, behind the last trial's MFMAs (the next launch's accept reads it);
//                             the NEXT tile's z / g / y / x loads are issued at the same point
//
// bt_iter_decide_kernel then closes iteration i-1 (sum |z - z_i| over the tiles, the iteration's record, the stop
// rule :93-95) and takes the decision of iteration i (first trial with F <= Q, :45) -- TWO launches per outer
// iteration instead of seven, p / g / z of a tile cross HBM once per iteration in each direction (6 passes over [n,k]
// instead of 9: the trials' reload and the accept pass's own reads are gone), and the loads of a tile are exposed once
// per ~120 us of matrix work instead of once per launch.  The last iteration of a window is accepted by the same kernel
// in `tail` mode (accept only).  Arithmetic per element is the sequence of backtrack.hip's kernels (each product and
// sum rounded separately like the reference's ATen ops); the tile sums are taken in another (fixed) order.
#include <algorithm>
#include "tile_device.hpp"
#include "quad_transpose.hpp"

#ifdef LASSO_BTI_TIMING
// debug build (tools/bti_timeline.py): wall-clock stamps (100 MHz) of ONE tile of every workgroup in the last
// launch that ran the accept step and the trials, 16 per workgroup
__device__ unsigned long long lasso_bti_stamps[1024 * 16];
extern "C" int lasso_debug_bti_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lasso_bti_stamps), sizeof(lasso_bti_stamps));
}
#define BTI_STAMP(slot) do { if (stamp_on && threadIdx.x == 0) lasso_bti_stamps[blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define BTI_STAMP(slot) do { } while (0)
#endif
#ifndef LASSO_BTI_HOOK
#define LASSO_BTI_HOOK 0        // 1: the next candidate is formed in the MFMA gaps of the current trial's GEMM-1 (A/B knob)
#endif
#ifndef LASSO_BTI_STAGGER
#define LASSO_BTI_STAGGER 2     // measured on config 3: 5.208 (0) -> 5.129 (2) / 5.133 (4) ms per solve
#endif

namespace lasso {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ACCEPT: the launch opens with the accept step of the previous iteration (false: first iteration of a window -- the
// point is loaded as it is).  A compile-time switch, like everything else that decides WHICH registers a tile's
// prefetch defines: a load under a run-time condition keeps the old value alive across the whole tile (720 spilled
// registers in the first version of the prefetch).
//
// Memory traffic of a tile goes through BUFFER descriptors of the tile's valid rows: a piece outside the matrix
// (ragged last tile, padded columns) reads as zero and is dropped on store, so no load or store sits under a branch
// and every wave issues EXACTLY the same number of vector-memory operations per tile -- which is what lets the GEMMs
// behind them start with `vmcnt(4 + EXTRA)` waits (tile_device.hpp) instead of draining those operations first.
template <int K, bool ACCEPT>
__global__ __launch_bounds__(kFistaThreads, 2) void bt_iter_kernel(const BtIterParams p, const BtSteps s) {
  constexpr int D = kFistaD;
  constexpr int NW = kFistaWaves;
  constexpr int KW = K / NW;
  constexpr int NP = KW / 32;
  constexpr int ITER = kTileM * (K / 4) / kFistaThreads;
  constexpr unsigned kOOR = 0x80000000u;   // a byte offset beyond every tile descriptor
  static_assert(ITER >= 1 && (kTileM * (K / 4)) % kFistaThreads == 0, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const pt = rings + NW * kRingBytesPerWave;
  lds_char* const rt = pt + kTileM * K * 4;
  lds_f32* const red = (lds_f32*)(rt + kTileM * D * 4);
  if (p.skip && *p.skip != 0) return;
  constexpr bool accept = ACCEPT;
  float lr_a = 0.0f, lam_a = 0.0f;
  if (accept) {
    if (p.acc_flags[0] == 0) return;       // no trial of the previous iteration passed: the state stays as it is
    lr_a = p.acc_fvals[2];
    lam_a = p.acc_fvals[3];
  }

  TileCtx<K> c;
  c.init(p.Wp, p.Wtp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;
#if LASSO_BTI_STAGGER > 0
  // Workgroups enter in 8 groups LASSO_BTI_STAGGER x ~0.45 us apart, so that the per-tile bursts of the 256 workgroups
  // (50 MB of prefetch at the same instant: 8-10 us until the last piece is in) spread out; tiles take the same time in
  // every workgroup, so the offsets last for the launch
  for (int i = (int)((blockIdx.x >> 3) & 7) * LASSO_BTI_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(16);
#endif
  dma_step(c.w1, c.voff1, c.ring);
  dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);
  float* const P = p.fast ? p.Y : p.Z;     // the point: y (FISTA) or z (ISTA: the loads of p then repeat those of z), flat [n][k]

  // descriptor of rows [row0, row0 + 16) n [0, n) of a row-major matrix with row pitch ld (floats)
  auto tile_rsrc = [&](const float* base, int row0, int64_t ld) {
    const int rows = min(kTileM, p.n - row0);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (int64_t)row0 * ld), 0, rows * (int)ld * 4,
                                             0x00020000);
  };
  // The operands of a tile's accept step (z, g, p: ITER 16-byte pieces per thread each) and its block of x, all in
  // flight together.  Issued for tile j+1 in front of the LAST trial of tile j (see there).  Offsets: two thread
  // constants and one multiply-add per piece -- nothing worth hoisting out of the tile loop.  (The first version clamped
  // 64-bit addresses per piece; the compiler hoisted the per-piece parts, spilled them, and every scratch reload
  // between two prefetch loads waited for the loads in front of it: eight HBM round trips in a row, 10 us per tile.)
  constexpr int RSTEP = kFistaThreads / (K / 4);             // rows between a thread's consecutive pieces
  const int r0 = tid / (K / 4);
  const int cc0 = (tid - r0 * (K / 4)) * 4;                  // first column of the thread's pieces
  f32x4 pa[ITER], ga[ITER], za[ITER], xr[2];
  auto fetch_tile = [&](int tile) {
    const int row0 = tile * kTileM;
    if constexpr (ACCEPT) {
      const auto rz = tile_rsrc(p.Z, row0, p.k), rg = tile_rsrc(p.G, row0, p.k), rp = tile_rsrc(P, row0, p.k);
      const unsigned coff = cc0 < p.k ? (unsigned)cc0 * 4u : kOOR;
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const unsigned off = (unsigned)((r0 + RSTEP * i) * p.k) * 4u + coff;
        za[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, off, 0, 0));
        ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
        pa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, 0));
      }
    }
    const auto rx = tile_rsrc(p.X, row0, p.ldx);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = 32 * wid + 16 * cb + n;
      const unsigned coff = col < p.d ? (unsigned)col * 4u : kOOR;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        xr[cb][rg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (unsigned)((4 * q + rg) * (int)p.ldx) * 4u + coff, 0, 0));
    }
  };
  if ((int)blockIdx.x < p.ntiles) fetch_tile(blockIdx.x);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    const int next_tile = min(tile + (int)gridDim.x, p.ntiles - 1);   // (past the end: a harmless repeat of the last tile)
    float dsum = 0.0f;
#ifdef LASSO_BTI_TIMING
    const bool stamp_on = accept && tile == (int)(blockIdx.x + (LASSO_BTI_TIMING) * gridDim.x);   // -DLASSO_BTI_TIMING=<j>: the workgroup's j-th tile
#endif
    BTI_STAMP(0);
    // Per-tile copies of the lane coordinates the compiler cannot see through: everything derived from them (the LDS
    // addresses of the C layout, the offsets of the accept step's stores) is then computed where a tile uses it
    // instead of once per launch -- hoisted out of the tile loop those ~60 values were live across every GEMM and
    // spilled, and a scratch reload behind the prefetch waits for the prefetch (in-order vmcnt).
    int qo = q, no = n, r0o = r0, cco = cc0;
    asm volatile("" : "+v"(qo), "+v"(no), "+v"(r0o), "+v"(cco));
    // LDS byte offset of this lane's C-layout element (row 4q+rg, column colbase+n) of the [16][K] tile: tile_off()
    int ep_rg[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) ep_rg[rg] = (4 * qo + rg) * (K * 4) + (((no >> 2) ^ rg) << 4) + ((no & 3) << 2);
    auto ep_addr = [&](int ps, int cb, int rg) {
      const int colbase = wid * KW + 32 * ps + 16 * cb;           // wave-uniform
      return (lds_f32*)(pt + ep_rg[rg] + ((((colbase >> 4) & 3) ^ qo) << 6) + (colbase >> 6) * 256);
    };
    f32x4 negx[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) negx[cb][rg] = __fsub_rn(0.0f, xr[cb][rg]);      // (x beyond the matrix read as 0)
    const auto rz = tile_rsrc(p.Z, row0, p.k), ry = tile_rsrc(p.Y, row0, p.k);
    const unsigned soff = (unsigned)(r0o * p.k) * 4u + (cco < p.k ? (unsigned)cco * 4u : kOOR);   // the thread's piece 0 in Z / Y
    // z_i / y_i of the accept step stay in registers through the gradient's GEMM-1 and go to memory behind it: stores in
    // front of that GEMM sat in front of its hand-counted ring waits (in-order vmcnt) and stalled it for their 3-4 us
    // of drain; behind it they drain under the r-tile exchange and GEMM-2's first steps.
    f32x4 zs[ACCEPT ? ITER : 1], ys[ACCEPT ? ITER : 1];
    if constexpr (ACCEPT) {
      // ---- the accept step of the previous iteration on this tile (pieces outside the matrix are zeros and stay zeros)
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        f32x4 zn, yn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zo = za[i][e];
          const float pv = pa[i][e];
          zn[e] = soft_threshold(__fsub_rn(pv, __fmul_rn(lr_a, ga[i][e])), lam_a);          // ista.py:40
          dsum += __builtin_fabsf(__fsub_rn(zo, zn[e]));                                      // :93
          yn[e] = __fadd_rn(zn[e], __fmul_rn(p.coef, __fsub_rn(zn[e], zo)));                  // :99-100
        }
        zs[i] = zn; ys[i] = yn;
        *(lds_f32x4*)(pt + tile_chunk_off<K>(r0o + RSTEP * i, cco)) = p.fast ? yn : zn;
      }
    } else if (p.zero_start) {
      // the solve starts from z = y = 0 (z0 == NULL): the tile is zeros, and this launch writes them to Z and Y for
      // the accept step that follows -- instead of two fill launches over [n][k] in front of the solve
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const unsigned off = soff + (unsigned)(RSTEP * i * p.k) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(zero, rz, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(zero, ry, off, 0, 0);
        *(lds_f32x4*)(pt + tile_chunk_off<K>(r0o + RSTEP * i, cco)) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    } else {
      visit_tile4<K, kFistaThreads>(P, p.k, row0, p.n, p.k, [&](int r, int cc, const f32x4& v) {
        *(lds_f32x4*)(pt + tile_chunk_off<K>(r, cc)) = v;
      });
    }
    BTI_STAMP(1);
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    BTI_STAMP(2);

    // ---- gradient at the point: r0 = p W^T - x, g = r0 W (ista.py:22-24)
    f32x4 gk[NP][2];
    {
      f32x4 acc[2] = {negx[0], negx[1]};
      gemm1_stream_sp<K>(c, pt, acc, c.w2, c.w2 + 32, c.voff2);
      BTI_STAMP(3);
      if constexpr (ACCEPT) {              // z_i -> Z (:102), y_i -> Y: exactly 2 ITER stores per wave
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
          const unsigned off = soff + (unsigned)(RSTEP * i * p.k) * 4u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zs[i]), rz, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ys[i]), ry, off, 0, 0);
        }
      }
      float rss = 0.0f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          rss = fmaf(acc[cb][rg], acc[cb][rg], rss);
          *(lds_f32*)(rt + tile_off<D>(4 * qo + rg, 32 * wid + 16 * cb + no)) = acc[cb][rg];
        }
      rss = wave_sum(rss);
      dsum = wave_sum(dsum);
      if (lane == 0) { red[wid] = rss; red[NW + wid] = dsum; }
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      if (tid == 0) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a += red[w]; b += red[NW + w]; }
        p.partials[tile] = a;
        if (accept) p.dpart[tile] = b;
      }
      f32x4 rf[D / 32][2];
      load_r_frags<K>(c, rt, rf);
      BTI_STAMP(4);
      // (2 ITER stores -- wave 0: two more -- went out since the ring's steps 0/1: GEMM-2's first two steps do not wait for them)
      gemm2_stream_sp<K, ACCEPT ? 2 * ITER : 0>(c, rf, gk);
      BTI_STAMP(5);
    }
    // the point in the C layout (this lane's 32 elements of the tile; every lane reads and later overwrites only its
    // own, and every wave passed the r-tile barrier after its last read of the tile as GEMM-1's operand)
    f32x4 pk[NP][2];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) pk[ps][cb][rg] = *ep_addr(ps, cb, rg);
    BTI_STAMP(6);

    // ---- the trials of this iteration on the tile (ista.py:38-47: same p, same g, steps lr0 / eta^t).
    // The candidate of trial t+1 is formed in registers right behind the MFMAs of trial t -- while the other wave of the
    // SIMD still runs its own (the waves of a workgroup drift apart by ~3 us per GEMM) -- and goes into the LDS tile
    // once every wave is through GEMM-1 of trial t (the barrier that also completes that trial's tile sums).
    f32x4 zn[NP][2];
    auto cand = [&](int t, float& l1, float& dzg, float& dz2) {
      const float lr = s.lr[t], lam = s.lam[t];
      l1 = 0.0f; dzg = 0.0f; dz2 = 0.0f;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float pv = pk[ps][cb][rg], g = gk[ps][cb][rg];
            const float v = soft_threshold(__fsub_rn(pv, __fmul_rn(lr, g)), lam);              // ista.py:40
            const float dz = __fsub_rn(v, pv);                                                   // :31
            l1 += __builtin_fabsf(v);
            dzg = __fadd_rn(dzg, __fmul_rn(dz, g));
            dz2 = __fadd_rn(dz2, __fmul_rn(dz, dz));
            zn[ps][cb][rg] = v;
          }
    };
    f32x4 acc[2];
    // the candidate of trial t + 1, one element per four MFMA gaps of trial t's GEMM-1 (LASSO_BTI_HOOK, bt_iter.hip
    // header): stage 0 v = p - lr g, 1 softshrink, 2 dz and |z+|, 3 the two products.  hs_*: the sums it accumulates.
    float hs_lr = 0.0f, hs_lam = 0.0f, hs_l1 = 0.0f, hs_dzg = 0.0f, hs_dz2 = 0.0f, hs_v = 0.0f, hs_dz = 0.0f;
    auto cand_stage = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value, e = i >> 2, st = i & 3;
      constexpr int ps = e >> 3, cb = (e >> 2) & 1, rg = e & 3;
      if constexpr (ps < NP) {
        const float pv = pk[ps][cb][rg], g = gk[ps][cb][rg];
        if constexpr (st == 0) hs_v = __fsub_rn(pv, __fmul_rn(hs_lr, g));
        else if constexpr (st == 1) hs_v = soft_threshold(hs_v, hs_lam);                       // ista.py:40
        else if constexpr (st == 2) {
          hs_dz = __fsub_rn(hs_v, pv);                                                          // :31
          hs_l1 += __builtin_fabsf(hs_v);
          zn[ps][cb][rg] = hs_v;
        } else {
          hs_dzg = __fadd_rn(hs_dzg, __fmul_rn(hs_dz, g));
          hs_dz2 = __fadd_rn(hs_dz2, __fmul_rn(hs_dz, hs_dz));
        }
      }
    };
    auto trial_front = [&](auto extra_c) {     // candidate -> LDS tile, r1 = z+ W^T - x
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) *ep_addr(ps, cb, rg) = zn[ps][cb][rg];
      acc[0] = negx[0]; acc[1] = negx[1];
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      gemm1_stream_sp<K, decltype(extra_c)::value>(c, pt, acc, c.w1, c.w1 + 32, c.voff1);
    };
    auto trial_front_hooked = [&](int tnext) {   // the same, with the candidate of trial `tnext` formed inside the GEMM
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) *ep_addr(ps, cb, rg) = zn[ps][cb][rg];
      acc[0] = negx[0]; acc[1] = negx[1];
      hs_lr = s.lr[tnext]; hs_lam = s.lam[tnext]; hs_l1 = 0.0f; hs_dzg = 0.0f; hs_dz2 = 0.0f;
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      gemm1_stream_hooked<K>(c, pt, acc, c.w1, c.w1 + 32, c.voff1, cand_stage);
    };
    auto trial_back = [&](int t, float l1, float dzg, float dz2) {   // the trial's four tile sums
      float rss = 0.0f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rss = fmaf(acc[cb][rg], acc[cb][rg], rss);
      rss = wave_sum(rss); l1 = wave_sum(l1); dzg = wave_sum(dzg); dz2 = wave_sum(dz2);
      if (lane == 0) { red[4 * wid] = rss; red[4 * wid + 1] = l1; red[4 * wid + 2] = dzg; red[4 * wid + 3] = dz2; }
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();        // red[] complete; every wave is through this trial's reads of the LDS tile
      if (tid < 4) {
        float a = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[4 * w + tid];
        p.partsM[((int64_t)t * 4 + tid) * p.ntiles + tile] = a;
      }
      // (no barrier here: the next writes of red[] come after the next trial's candidate barrier, which wave 0 reaches
      //  only after these reads)
    };
    // This tile's gradient g -> G (the accept step of the NEXT launch reads it) and the next tile's operands.  In the C
    // layout a lane owns four rows of one column: a 4 x 4 transpose inside each lane quad turns that into four
    // consecutive columns of one row -- 2 NP stores of 16 bytes per lane instead of 8 NP of 4.  Issued BEHIND the last
    // trial's MFMAs: the loads travel under that trial's sums and its barrier.  (In front of the last trial -- with
    // vmcnt(4 + the operation count) for its first two steps -- was measured and is slower, 5.22 against 5.18 ms per
    // solve: from step 2 on the ring's in-order waits make the GEMM wait out the prefetch, and with all 256 workgroups
    // in step that burst of 50 MB takes 8-10 us.)
    auto tile_end = [&]() {
      const auto rgd = tile_rsrc(p.G, row0, p.k);
      const int j = no & 3;
      const int colq = wid * KW + (no & 12);                               // first of this lane's four columns
      const unsigned roff = (unsigned)((4 * qo + j) * p.k) * 4u;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x4 v = gk[ps][cb];
          quad_transpose(v, j);
          const int col = colq + 32 * ps + 16 * cb;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rgd,
                                                 roff + (col < p.k ? (unsigned)col * 4u : kOOR), 0, 0);
        }
      fetch_tile(next_tile);
    };
    if (p.ntrials > 0) {
      float l1, dzg, dz2;
      cand(0, l1, dzg, dz2);
#pragma unroll 1
      for (int t = 0; t + 1 < p.ntrials; ++t) {
        if (t == 0) BTI_STAMP(7);
#if LASSO_BTI_HOOK
        trial_front_hooked(t + 1);
        if (t == 0) BTI_STAMP(9);
        trial_back(t, l1, dzg, dz2);
        l1 = hs_l1; dzg = hs_dzg; dz2 = hs_dz2;
#else
        trial_front(std::integral_constant<int, 0>{});
        if (t == 0) BTI_STAMP(9);
        float l1n, dzgn, dz2n;
        cand(t + 1, l1n, dzgn, dz2n);
        trial_back(t, l1, dzg, dz2);
        l1 = l1n; dzg = dzgn; dz2 = dz2n;
#endif
        if (t == 0) BTI_STAMP(10);
      }
      BTI_STAMP(11);
      trial_front(std::integral_constant<int, 0>{});
      BTI_STAMP(14);
      tile_end();
      trial_back(p.ntrials - 1, l1, dzg, dz2);
    } else {
      tile_end();
    }
    // (no barrier: every wave is through its last read of the LDS tile at the last trial's barrier, and red[] is next
    //  written behind the next tile's own barriers, which wave 0 reaches only after its reads above)
    BTI_STAMP(12);
  }
#ifdef LASSO_BTI_TIMING
  if (accept && threadIdx.x == 0) lasso_bti_stamps[blockIdx.x * 16 + 13] = wall_clock64();
#endif
  LASSO_WAIT_VMCNT(0);
}

// One block, 1024 threads, behind every bt_iter_kernel launch.
//  (1) closes the iteration whose accept step that launch ran (prev_* != nullptr): sum |z - z_next| over the tiles in
//      a fixed order, the iteration's record (trials, accepted step, F), the stop rule (ista.py:93-95) -> ctl
//  (2) decides the trials of the current iteration like bt_decide_multi_kernel (same sums, tree and fp32 operation
//      order): the first trial with F <= Q (:45) -> cur_flags / cur_fvals; none and `last_batch` -> ctl[0] = 2
// ctl: [0] 0 = running, 1 = the stop rule fired, 2 = a search ran out of pre-enqueued trials (the host continues from
// iteration ctl[1] on the synchronous path: the state is the start of that iteration), [1] iterations completed,
// [2] last sum |z - z_next| (float bits).
__global__ __launch_bounds__(1024) void bt_iter_decide_kernel(const float* __restrict__ partials,
                                                              const float* __restrict__ partsM, int ntiles, float alpha,
                                                              const BtSteps s, int ntrials, int first_index,
                                                              int last_batch, int* __restrict__ cur_flags,
                                                              float* __restrict__ cur_fvals,
                                                              const int* __restrict__ prev_flags,
                                                              const float* __restrict__ prev_fvals,
                                                              const float* __restrict__ dpart, int it_prev, float budget,
                                                              int* __restrict__ ctl, float* __restrict__ rec) {
  if (ctl[0] != 0) return;
  // wave-level reductions (shuffles in a fixed order), ONE block barrier per part: as LDS trees with a barrier per
  // level this launch took ~10 us, x 11 per config-3 solve
  __shared__ double sh[kBtMultiMax][5][2];
  __shared__ float shd[4];
  __shared__ int stop;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (prev_flags) {
    if (threadIdx.x < 256) {
      float acc = 0.0f;
      for (int t = threadIdx.x; t < ntiles; t += 256) acc += dpart[t];
#pragma unroll
      for (int st = 32; st > 0; st >>= 1) acc += __shfl_down(acc, st, 64);
      if (lane == 0) shd[wave] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float delta = (shd[0] + shd[1]) + (shd[2] + shd[3]);
      int fired = 0;
      if (prev_flags[0] != 0) {
        rec[4 * it_prev] = __int_as_float(prev_flags[2] + 1);        // {trials, accepted step, F of the accepted trial}
        rec[4 * it_prev + 1] = prev_fvals[2];
        rec[4 * it_prev + 2] = prev_fvals[0];
        ctl[1] = it_prev + 1;
        ctl[2] = __float_as_int(delta);
        if (budget >= 0.0f && delta <= budget) { ctl[0] = 1; fired = 1; }                    // ista.py:93-95
      }
      stop = fired;
    }
    __syncthreads();
    if (stop) return;
  }
  if (ntrials <= 0) return;
  if (cur_flags[0] != 0) return;           // a trial of an earlier batch of this iteration was accepted
  const int t = threadIdx.x >> 7, l = threadIdx.x & 127;    // 128 threads = 2 waves per trial
  double acc[5] = {0, 0, 0, 0, 0};
  if (t < ntrials)
    for (int tl = l; tl < ntiles; tl += 128) {
      acc[0] += partials[tl];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[1 + q] += partsM[((size_t)t * 4 + q) * ntiles + tl];
    }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int st = 32; st > 0; st >>= 1) acc[q] += __shfl_down(acc[q], st, 64);
    if (lane == 0) sh[t][q][wave & 1] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int u = 0; u < ntrials; ++u) {
    const float rss0 = (float)(sh[u][0][0] + sh[u][0][1]), rss1 = (float)(sh[u][1][0] + sh[u][1][1]);
    const float l1 = (float)(sh[u][2][0] + sh[u][2][1]);
    const float dzg = (float)(sh[u][3][0] + sh[u][3][1]), dz2 = (float)(sh[u][4][0] + sh[u][4][1]);
    const float f0 = __fmul_rn(0.5f, rss0);                                        // ista.py:23
    const float al1 = __fmul_rn(alpha, l1);
    const float F = __fadd_rn(__fmul_rn(0.5f, rss1), al1);                         // :28
    const float Q = __fadd_rn(__fadd_rn(__fadd_rn(f0, dzg), __fmul_rn(s.hol[u], dz2)), al1);  // :32-35
    cur_fvals[0] = F; cur_fvals[1] = Q;
    cur_flags[1] = first_index + u + 1;
    if (F <= Q) {                                                                  // :45
      cur_flags[0] = 1; cur_flags[2] = first_index + u; cur_fvals[2] = s.lr[u]; cur_fvals[3] = s.lam[u];
      return;
    }
  }
  if (last_batch) ctl[0] = 2;
}

// The accept step alone, behind the last iteration of a window (ista.py:40,93,99-102): an element-wise pass over
// 16-row tiles with the tile sums of |z - z_next| in bt_iter_kernel's place (dpart[tile]); HBM-bound.
__global__ __launch_bounds__(256) void bt_accept_tail_kernel(const BtIterParams p) {
  __shared__ float sh[4];
  if (p.skip && *p.skip != 0) return;
  if (p.acc_flags[0] == 0) return;
  const float lr_a = p.acc_fvals[2], lam_a = p.acc_fvals[3];
  const float* const P = p.fast ? p.Y : p.Z;
  const int tid = threadIdx.x, k4 = p.k >> 2;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    float dsum = 0.0f;
    for (int idx = tid; idx < kTileM * k4; idx += 256) {
      const int r = idx / k4, cc = (idx - r * k4) * 4;
      if (row0 + r >= p.n) break;
      const int64_t off = (int64_t)(row0 + r) * p.k + cc;
      const f32x4 zo = *reinterpret_cast<const f32x4*>(p.Z + off);
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.G + off);
      const f32x4 pv = *reinterpret_cast<const f32x4*>(P + off);
      f32x4 zn, yn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        zn[e] = soft_threshold(__fsub_rn(pv[e], __fmul_rn(lr_a, g[e])), lam_a);               // ista.py:40
        dsum += __builtin_fabsf(__fsub_rn(zo[e], zn[e]));                                        // :93
        yn[e] = __fadd_rn(zn[e], __fmul_rn(p.coef, __fsub_rn(zn[e], zo[e])));                    // :99-100
      }
      *reinterpret_cast<f32x4*>(p.Z + off) = zn;                                                 // :102
      if (p.fast) *reinterpret_cast<f32x4*>(p.Y + off) = yn;
    }
    dsum = wave_sum(dsum);
    if ((tid & 63) == 0) sh[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) p.dpart[tile] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
  }
}

template <int K>
static hipError_t launch_iter_k(const BtIterParams& p, const BtSteps& s, int grid, hipStream_t stream) {
  const size_t lds = (size_t)kTileM * K * 4 + (size_t)kTileM * kFistaD * 4 + (size_t)kFistaWaves * kRingBytesPerWave + 256;
  if (p.acc_flags) {
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt_iter_kernel<K, true>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((bt_iter_kernel<K, true>), dim3(grid), dim3(kFistaThreads), lds, stream, p, s);
  } else {
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt_iter_kernel<K, false>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((bt_iter_kernel<K, false>), dim3(grid), dim3(kFistaThreads), lds, stream, p, s);
  }
  return hipGetLastError();
}

hipError_t launch_bt_iter(const BtIterParams& p, const BtSteps& s, int kpad, int grid, hipStream_t stream) {
  if (p.tail) {
    hipLaunchKernelGGL(bt_accept_tail_kernel, dim3(std::min(p.ntiles, 4096)), dim3(256), 0, stream, p);
    return hipGetLastError();
  }
  switch (kpad) {
    case 256: return launch_iter_k<256>(p, s, grid, stream);
    case 512: return launch_iter_k<512>(p, s, grid, stream);
    case 1024: return launch_iter_k<1024>(p, s, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_bt_iter_decide(const float* partials, const float* partsM, int ntiles, double alpha, const BtSteps& s,
                                 int ntrials, int first_index, int last_batch, int* cur_flags, float* cur_fvals,
                                 const int* prev_flags, const float* prev_fvals, const float* dpart, int it_prev,
                                 float budget, int* ctl, float* rec, hipStream_t stream) {
  hipLaunchKernelGGL(bt_iter_decide_kernel, dim3(1), dim3(1024), 0, stream, partials, partsM, ntiles, (float)alpha, s,
                     ntrials, first_index, last_batch, cur_flags, cur_fvals, prev_flags, prev_fvals, dpart, it_prev,
                     budget, ctl, rec);
  return hipGetLastError();
}

}  // namespace lasso
