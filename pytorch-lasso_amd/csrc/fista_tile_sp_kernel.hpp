// The fused persistent FISTA kernel (template; instantiated by fista_tile_sp.hip and fista_tile_sp_ds.hip): one 8-wave workgroup per 16-row tile, software-pipelined.
// Algorithm, LDS layouts, W streaming and MFMA operand convention: DESIGN.md section 3.1.
//
// Why: all waves of a workgroup run the same instruction stream in near lockstep (the
// two barriers per iteration re-align them), so any stretch in which a wave is NOT
// issuing MFMAs -- waiting for its ds_read fragments, issuing LDS-DMA, running the prox
// epilogue -- is a stretch in which EVERY wave on the SIMD is idle and the matrix pipe
// drains (ablations on MI355X: epilogue 7 %, DMA issue 5 %, fragment waits ~10 %).
// Here each wave hides those stretches behind its OWN MFMAs:
//   * B/A fragments of step g+1 are read into a second register set while the MFMAs of
//     step g run (the ring slot is released -- and refilled by LDS-DMA with step g+3 --
//     as soon as those reads have returned, a few MFMAs into step g);
//   * the prox/momentum epilogue of GEMM-2 pass p runs between the MFMAs of the first
//     step of pass p+1 (two alternating accumulator sets).
#pragma once
#include "tile_device.hpp"

#ifndef LASSO_SLICE_UNROLL
#define LASSO_SLICE_UNROLL 2
#endif

#ifdef LASSO_FISTA_TIMING
// debug build (tools/fista_timeline.py): wall-clock stamps (100 MHz) of workgroup tid 0: [0] kernel entry, [1] first
// tile staged, then one per iteration start (first tile only), 64 per workgroup
__device__ unsigned long long lasso_fista_stamps[1024 * 64];
extern "C" int lasso_debug_fista_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lasso_fista_stamps), sizeof(lasso_fista_stamps));
}
#define FISTA_STAMP(slot) do { if (threadIdx.x == 0 && (slot) < 64) lasso_fista_stamps[blockIdx.x * 64 + (slot)] = wall_clock64(); } while (0)
#else
#define FISTA_STAMP(slot) do { } while (0)
#endif

namespace lasso {
namespace sp {

// Waves w and w+4 share a SIMD and leave every barrier in lockstep, so their non-MFMA
// gaps coincide; delaying the second half by a fraction of a step staggers them.
#ifndef LASSO_DESYNC_SLEEP
#define LASSO_DESYNC_SLEEP 0
#endif
#if LASSO_DESYNC_SLEEP > 0
#define LASSO_DESYNC() do { if (wid >= 4) __builtin_amdgcn_s_sleep(LASSO_DESYNC_SLEEP); } while (0)
#else
#define LASSO_DESYNC()
#endif

// STOP: compile the in-kernel global stop rule in (separate instantiation so that the
// fixed-iteration kernel keeps its register allocation).
// M: rows per tile; the padded feature count is D = 4096 / M (M = 16 / D = 256 is the flagship
// shape; M = 32 / 64 serve dictionaries with d <= 128 / 64 without padding d up to 256).
// NW: waves per workgroup.  8 (two per SIMD) is the tuned form; 4 with M halved is the same
// per-wave work in a workgroup of half the height and half the LDS, two of which share a CU:
// small batches of short rows (d <= 128) then spread over twice as many CUs.
// DS (round 5): contraction steps of GEMM-2 that are NOT all padding -- ceil(d / 32) when the rows have fewer features
// than the tile's padded D (0 = all D / 32).  GEMM-2 contracts over the features in chunks of 32; a chunk beyond d
// multiplies residual columns that are exactly zero with rows of W^T that are exactly zero, so leaving it out drops
// only +-0 terms: d = 64 on the 128-wide tiles runs 2 of 4 chunks, d = 192 on the flagship tile 6 of 8.  The default
// instantiations (DS = 0) are the code they were.
// ZS ("zero start", round 6): instantiations for launches from an all-zero code (z_in == y_in == NULL) that leave out
// the first iteration's GEMM-1 -- see the loop below.  A separate instantiation because the branch costs the loop
// body ~1 % (measured on the 100-iteration headline solve: 31.73 against 31.95 k it/s although it does 0.5 % less
// work), which a short solve wins back five times over (10 iterations: 5 % of the MFMA work gone) and a long one does
// not: launch_k() takes it for zero starts of at most kZeroStartIters iterations.
constexpr int kZeroStartIters = 32;
template <int K, int M, bool STOP, int NW = kFistaWaves, int DS = 0, bool ZS = false>
__global__ __launch_bounds__(64 * NW, (NW == 4 && K > 512) ? 1 : 2) void fista_tile_sp_kernel(const FistaTileParams p) {
  // step size and threshold: launch arguments, or device memory (lr = LASSO_LR_AUTO)
  const float lr_ = p.lr_dev ? p.lr_dev[0] : p.lr, lam_ = p.lr_dev ? p.lr_dev[1] : p.lam;
  constexpr int D = 512 * NW / M;
  constexpr int NT = 64 * NW;
  constexpr int S1 = K / 32;
  constexpr int KW = TileCtx<K, D>::KW;
  constexpr int NP = KW / 32;
  constexpr int T2 = DS ? DS : D / 32;         // feature chunks GEMM-2 contracts over
  constexpr int S2 = NP * T2;
  static_assert(T2 >= 1 && T2 <= D / 32, "DS out of range");
  constexpr int YT_BYTES = M * K * 4;
  constexpr int RT_BYTES = M * D * 4;
  static_assert(M * D == 512 * NW && M % 16 == 0 && D % 32 == 0, "tile shape");
  static_assert(S1 % 2 == 0 && S2 % 2 == 0 && S1 >= 6 && S2 >= 4 && NP >= 1, "geometry");
  static_assert(YT_BYTES <= 65536, "y tile must fit beside the rings");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const yt = rings + NW * kRingBytesPerWave;
  lds_char* const rt = yt + YT_BYTES;
  lds_f32* const red = (lds_f32*)(rt + RT_BYTES);

  // stand-by launch behind a split-k launch: nothing to do unless that kernel gave up
  if (p.run_if != nullptr && __hip_atomic_load(p.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;

  FISTA_STAMP(0);
  TileCtx<K, D> c;
  c.init(p.Wp, p.Wtp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;
  const int rbase = 16 * c.rb;          // first tile row of this wave's row block
  const int cw = c.cw;                  // column index of the wave inside its row block
  lds_char* const slot0 = c.ring;
  lds_char* const slot1 = c.ring + kStepBytes;
  // y-tile byte offset of this lane's C-layout element (row 4q+rg, column colbase+n), see
  // tile_off():  ep_rg[rg] + (((colbase>>4)&3) ^ q) << 6) + (colbase>>6)*256
  int ep_rg[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
    ep_rg[rg] = (rbase + 4 * q + rg) * (K * 4) + (((n >> 2) ^ rg) << 4) + ((n & 3) << 2);

  // Ring invariant on entry of GEMM-1 (every iteration, every tile):
  //   X.b holds the B fragments of step 0; slot1 <- step 1, slot0 <- step 2 in flight.
  Frag X, Y;
  dma_step(c.w1, c.voff1, slot0);
  dma_step(c.w1 + 32, c.voff1, slot1);
  LASSO_WAIT_VMCNT(4);
  load_b(c, X, slot0);
  LASSO_WAIT_LGKM0();
  dma_step(c.w1 + 64, c.voff1, slot0);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * M;
    {
      const float* ysrc = p.y_in ? p.y_in : p.z_in;
      const int64_t ldy = p.y_in ? p.ldy_in : p.ldz_in;
      // (rolled loads: once per tile and launch, and the batched form costs this kernel registers it does not have)
      visit_tile4<K, NT, M, false>(ysrc, ldy, row0, p.n, p.k, [&](int r, int cc, const f32x4& v) {
        *(lds_f32x4*)(yt + tile_chunk_off<K>(r, cc)) = v;
      });
    }
    f32x4 zreg[NP][2];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = rbase + 4 * q + rg, cc = cw * KW + 32 * ps + 16 * cb + n;
          float v = 0.0f;
          if (p.z_in && (row0 + r) < p.n && cc < p.k)
            v = (p.z_in + (int64_t)row0 * p.ldz_in)[r * (int)p.ldz_in + cc];
          zreg[ps][cb][rg] = v;
        }
    f32x4 xneg[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = rbase + 4 * q + rg, cc = 32 * cw + 16 * cb + n;
        float v = 0.0f;
        if ((row0 + r) < p.n && cc < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + cc];
        xneg[cb][rg] = -v;
      }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    bool stopped = false, aborted = false;
    if (tile == (int)blockIdx.x) FISTA_STAMP(1);
    for (int it = 0; it < p.iters; ++it) {
      if (tile == (int)blockIdx.x) FISTA_STAMP(2 + it);
      const float coef = p.coef[it];
      float dsum = 0.0f;
      int no = n, qo = q;
      asm volatile("" : "+v"(no), "+v"(qo));
      const lds_char* const yrow = yt + (rbase + n) * (K * 4);

      // ======================= GEMM-1: r = y W^T - x =========================
      f32x4 acc[2] = {xneg[0], xneg[1]};
      // Round 6 (ZS instantiations): the first iteration of an ALL-ZERO start (z_in == y_in == NULL: the E-step of an
      // EM loop, sparse_encode's default) has y = 0, so r = -x exactly -- every product of this GEMM is 0 w and the chain
      // returns what it started from (the sign of a zero residual of a zero feature aside).  It is left out: the wave
      // drops the two W steps it has in flight and primes its ring with the first three steps of W^T instead (two DMA
      // latencies against 16 us of MFMAs: 5 % of a 10-iteration E-step).  Codes bitwise those of the full iteration.
      const bool skip1 = ZS && it == 0;
      unsigned long long gr[4] = {0ull, 0ull, 0ull, 0ull};
      const bool check = STOP && p.stop_on && it > 0;
      const unsigned long long* const grow =
          p.stop_gran ? p.stop_gran + (size_t)((it - 1) & (kStopRing - 1)) * p.ntiles : nullptr;
      if (skip1) {
        LASSO_WAIT_VMCNT(0);                         // W steps 1, 2 have landed (unused)
        dma_step(c.w2, c.voff2, slot0);              // W^T steps 0, 1 (step U = pass U/T2, d-chunk U%T2)
        dma_step(c.w2 + (size_t)(32 * (1 / T2)) * D + 32 * (1 % T2), c.voff2, slot1);
        LASSO_WAIT_VMCNT(4);
        load_b(c, X, slot0);
        LASSO_WAIT_LGKM0();
        dma_step(c.w2 + (size_t)(32 * (2 / T2)) * D + 32 * (2 % T2), c.voff2, slot0);
        // now X.b = B fragments of GEMM-2 step 0; slot1 <- W^T step 1, slot0 <- W^T step 2 (the invariant below)
      } else {
      load_a(c, X, yrow, 0);                       // A fragments of step 0 (y is final now)
      // one trip = steps s = 2*s2 (on X) and s+1 (on Y).  srcE/srcO: DMA refills issued
      // in the even/odd step (steps s+3 / s+4 of the stream).
      auto trip = [&](int s2, const float* srcE, const unsigned (&voffE)[4], const float* srcO,
                      const unsigned (&voffO)[4], auto last_c) {
        constexpr bool last = decltype(last_c)::value;
        // ---- even step: compute X, fetch step s+1 -> Y
        LASSO_WAIT_VMCNT(4);
        load_b(c, Y, slot1);
        load_a(c, Y, yrow + s2 * 256, 1);
        step_body(acc, X.a, X, srcE, voffE, slot1, no_stage, no_stage, no_stage, no_stage);
        // ---- odd step: compute Y, fetch step s+2 -> X (B only when it is GEMM-2's step 0)
        LASSO_WAIT_VMCNT(4);
        load_b(c, X, slot0);
        if constexpr (!last) load_a(c, X, yrow + (s2 + 1) * 256, 0);
        step_body(acc, Y.a, Y, srcO, voffO, slot0, no_stage, no_stage, no_stage, no_stage);
      };
      using F = std::false_type;
      using T = std::true_type;
      // in-kernel stop rule: wave 0 fetches the previous iteration's per-tile |dz| granules
      // a quarter into GEMM-1 (every workgroup has published them by then) and looks at
      // them when GEMM-1 is done -- the L2 round trip hides behind the MFMAs.
      // The contraction over the atoms is summed in SLICES of 128 atoms (two trips): the MFMA
      // chain restarts from 0 at every slice boundary and the slice totals are added left to
      // right, r = ((p_0 + p_1) + p_2) + ... with p_0's chain starting from -x.  This is the
      // order in which the split-k kernel (fista_splitk.hip, one slice per workgroup) can
      // form the same r -- a row's code is bitwise independent of the kernel that computed it.
      f32x4 run[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      auto fetch_granules = [&](int s2) {
        if (STOP && check && wid == 0 && s2 == S1 / 8) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (lane + 64 * e < p.ntiles)
              gr[e] = __hip_atomic_load(grow + lane + 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      // one loop trip = one slice = two trips of the step machinery; slice 0 is peeled (run = p_0), so the
      // loop body is straight-line code
      static_assert((S1 / 2 - 2) % 2 == 0 && S1 / 2 - 2 >= 2, "whole slices in the regular part");
      fetch_granules(0);
      trip(0, c.w1 + 96, c.voff1, c.w1 + 128, c.voff1, F{});
      fetch_granules(1);
      trip(1, c.w1 + 64 + 96, c.voff1, c.w1 + 64 + 128, c.voff1, F{});
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) { run[cb][rg] = acc[cb][rg]; acc[cb][rg] = 0.0f; }
#pragma unroll LASSO_SLICE_UNROLL
      for (int s2 = 2; s2 < S1 / 2 - 2; s2 += 2) {
        fetch_granules(s2);
        trip(s2, c.w1 + 64 * s2 + 96, c.voff1, c.w1 + 64 * s2 + 128, c.voff1, F{});
        fetch_granules(s2 + 1);
        trip(s2 + 1, c.w1 + 64 * (s2 + 1) + 96, c.voff1, c.w1 + 64 * (s2 + 1) + 128, c.voff1, F{});
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            run[cb][rg] = __fadd_rn(run[cb][rg], acc[cb][rg]);
            acc[cb][rg] = 0.0f;
          }
      }
      // steps S1-4, S1-3: refills are W step S1-1 and W^T step 0
      trip(S1 / 2 - 2, c.w1 + 32 * (S1 - 1), c.voff1, c.w2, c.voff2, F{});
      // steps S1-2, S1-1: refills are W^T steps 1 and 2 (step U = pass U/T2, d-chunk U%T2)
      trip(S1 / 2 - 1, c.w2 + (size_t)(32 * (1 / T2)) * D + 32 * (1 % T2), c.voff2,
           c.w2 + (size_t)(32 * (2 / T2)) * D + 32 * (2 % T2), c.voff2, T{});
      // now X.b = B fragments of GEMM-2 step 0; slot1 <- W^T step 1, slot0 <- W^T step 2
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[cb][rg] = __fadd_rn(run[cb][rg], acc[cb][rg]);   // + the last slice
      }   // (!skip1)

      if (STOP && check && wid == 0) {
        // every granule must carry tag == it (iteration it-1 published as it-1+1); re-poll the
        // (rare) late ones.  Sum in a fixed order: identical decision in every workgroup.
        const unsigned want = (unsigned)it;
        float part = 0.0f;
        int spins = 0;
        bool ok;
        do {
          ok = true;
          part = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (lane + 64 * e < p.ntiles) {
              if ((unsigned)(gr[e] >> 32) != want) {
                gr[e] = __hip_atomic_load(grow + lane + 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && ((unsigned)(gr[e] >> 32) == want);
              }
              part += __uint_as_float((unsigned)gr[e]);
            }
          ok = __all(ok);
          if (!ok) {
            __builtin_amdgcn_s_sleep(8);
            // another workgroup gave up (it is not co-resident with the rest): leave at once
            if ((spins & 63) == 63 &&
                __hip_atomic_load(p.stop_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
              break;
          }
        } while (!ok && ++spins < kStopSpinLimit);
        const float total = wave_sum(part);
        if (lane == 0) {
          // red[NW]: 0 = go on, 1 = iteration it-1 met the rule (ista.py:93), 2 = handshake
          // timed out -- some workgroup of the grid is not resident; EVERY workgroup aborts
          // and the host repeats the solve on the chunked path (lasso_hip.hip)
          red[NW] = !ok ? 2.0f : (total <= p.stop_budget ? 1.0f : 0.0f);
          red[NW + 1] = total;
          if (!ok) __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // r tile -> LDS, everyone reads all of it
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *(lds_f32*)(rt + tile_off<D>(rbase + 4 * qo + rg, 32 * cw + 16 * cb + no)) = acc[cb][rg];
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      LASSO_DESYNC();
      if (STOP && check && red[NW] != 0.0f) {      // iteration it-1 met the stop rule: z (registers) is its z_next
        aborted = red[NW] == 2.0f;
        if (blockIdx.x == 0 && tid == 0 && !aborted) {
          p.stop_out[0] = it;
          p.stop_out[1] = __float_as_int(red[NW + 1]);
        }
        stopped = true;
        break;
      }
      f32x4 rf[T2][2];
#pragma unroll
      for (int t = 0; t < T2; ++t)           // (load_r_frags of tile_device.hpp, the first T2 chunks)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
          rf[t][ss] = *(const lds_f32x4*)(rt + (16 * c.rb + c.n) * (D * 4) + (t >> 1) * 256 + c.aoff[t & 1][ss]);

      // ================= GEMM-2 + pipelined prox/momentum epilogue ==============
      f32x4 g2[2][2];   // [pass parity][col-block]
      f32x4 yv[2], yn[2];   // epilogue temporaries: y read from / written to the LDS tile
      // The prox/momentum epilogue of pass ps, cut into four stages that are issued in
      // the gaps between the MFMAs of the NEXT pass's first step.
      auto ep_addr = [&](auto ps_c, int cb, int rg) {
        constexpr int ps = decltype(ps_c)::value;
        const int colbase = cw * KW + 32 * ps + 16 * cb;           // wave-uniform
        return (lds_f32*)(yt + ep_rg[rg] + ((((colbase >> 4) & 3) ^ qo) << 6) + (colbase >> 6) * 256);
      };
      auto ep_read = [&](auto ps_c) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) yv[cb][rg] = *ep_addr(ps_c, cb, rg);
      };
      auto ep_math = [&](auto ps_c, auto cb_c) {
        constexpr int ps = decltype(ps_c)::value;
        constexpr int cb = decltype(cb_c)::value;
#ifdef LASSO_ABL_NOEPI   // timing ablation only (results invalid)
        asm volatile("" :: "v"(g2[ps & 1][cb]));
        if (false)
#endif
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float zo = zreg[ps][cb][rg];
          const float stp = __fmul_rn(lr_, g2[ps & 1][cb][rg]);              // lr * grad
          const float zn = soft_threshold(__fsub_rn(yv[cb][rg], stp), lam_);
          dsum += __builtin_fabsf(__fsub_rn(zo, zn));                          // |z - z_next|
          const float mom = __fmul_rn(coef, __fsub_rn(zn, zo));                // c (z_next - z)
          yn[cb][rg] = __fadd_rn(zn, mom);
          zreg[ps][cb][rg] = zn;
        }
      };
      auto ep_write = [&](auto ps_c) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) *ep_addr(ps_c, cb, rg) = yn[cb][rg];
      };
      using CB0 = std::integral_constant<int, 0>;
      using CB1 = std::integral_constant<int, 1>;
      static_for<S2>([&](auto u_c) {
        constexpr int U = decltype(u_c)::value;
        constexpr int ps = U / T2, t = U % T2;
        Frag& cur = (U & 1) ? Y : X;
        Frag& nxt = (U & 1) ? X : Y;
        lds_char* const nslot = (U & 1) ? slot0 : slot1;     // slot of step U+1
        if constexpr (t == 0) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) g2[ps & 1][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        LASSO_WAIT_VMCNT(4);
        load_b(c, nxt, nslot);            // step U+1 (for U = S2-1: next iteration's GEMM-1 step 0)
        const float* src;
        if constexpr (U + 3 < S2) {
          constexpr int pn = (U + 3) / T2, tn = (U + 3) % T2;
          src = c.w2 + (size_t)(32 * pn) * D + 32 * tn;
        } else {
          src = c.w1 + 32 * (U + 3 - S2);    // next GEMM-1, steps 0..2
        }
        const unsigned (&voff)[4] = (U + 3 < S2) ? c.voff2 : c.voff1;
        if constexpr (t == 0 && ps > 0) {
          using PP = std::integral_constant<int, ps - 1>;
          step_body(g2[ps & 1], rf[t], cur, src, voff, nslot,
                    [&] { ep_read(PP{}); }, [&] { ep_math(PP{}, CB0{}); },
                    [&] { ep_math(PP{}, CB1{}); }, [&] { ep_write(PP{}); });
        } else {
          step_body(g2[ps & 1], rf[t], cur, src, voff, nslot, no_stage, no_stage, no_stage, no_stage);
        }
      });
      {   // last pass: nothing left to hide behind
        using PL = std::integral_constant<int, NP - 1>;
        ep_read(PL{}); ep_math(PL{}, CB0{}); ep_math(PL{}, CB1{}); ep_write(PL{});
      }

      dsum = wave_sum(dsum);
      if (lane == 0) red[wid] = dsum;
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();   // y tile complete; red[] complete
      LASSO_DESYNC();
      if ((p.partials || (STOP && p.stop_on)) && tid == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += red[w];
        if (p.partials) p.partials[(int64_t)it * p.part_stride + tile] = tsum;
        if (STOP && p.stop_on)   // one 8-byte write-through store {tag = it+1, value}: the data is the flag
          __hip_atomic_store(p.stop_gran + (size_t)(it & (kStopRing - 1)) * p.ntiles + tile,
                             ((unsigned long long)(unsigned)(it + 1) << 32) | __float_as_uint(tsum),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (STOP && p.stop_on && !stopped && blockIdx.x == 0 && wid == 0 && p.iters > 0) {
      // ran to maxiter: report the last iteration's global delta (does not change z)
      const unsigned want = (unsigned)p.iters;
      const unsigned long long* const lrow = p.stop_gran + (size_t)((p.iters - 1) & (kStopRing - 1)) * p.ntiles;
      float part = 0.0f;
      int spins = 0;
      bool ok;
      do {
        ok = true;
        part = 0.0f;
        for (int e = lane; e < p.ntiles; e += 64) {
          const unsigned long long g = __hip_atomic_load(lrow + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(g >> 32) == want);
          part += __uint_as_float((unsigned)g);
        }
        ok = __all(ok);
        if (!ok) __builtin_amdgcn_s_sleep(8);
      } while (!ok && ++spins < kStopSpinLimit);
      const float total = wave_sum(part);
      if (lane == 0) {
        p.stop_out[0] = p.iters;
        p.stop_out[1] = __float_as_int(total);
        if (!ok) __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (STOP && aborted) break;     // z_out untouched: the host re-runs the solve
    if (tile == (int)blockIdx.x) FISTA_STAMP(2 + p.iters);

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
            const int r = rbase + 4 * qo + rg, cc = cw * KW + 32 * ps + 16 * cb + no;
            if ((row0 + r) < p.n && cc < p.k) zo_base[r * (int)p.ldz_out + cc] = zreg[ps][cb][rg];
          }
    }
    if (p.y_out) {
      const bool yvec = vec4_ok(p.y_out, p.ldy_out, p.k);
      for (int idx = tid; idx < M * (K / 4); idx += NT) {
        const int r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
        const f32x4 v = *(const lds_f32x4*)(yt + tile_chunk_off<K>(r, cc));
        store_row4(p.y_out, p.ldy_out, row0 + r, p.n, p.k, cc, v, yvec);
      }
    }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (tile == (int)blockIdx.x) FISTA_STAMP(3 + p.iters);
  }
  LASSO_WAIT_VMCNT(0);
}

template <int K, int M, bool STOP, int NW, int DS = 0, bool ZS = false>
static hipError_t launch_ks(const FistaTileParams& p, int grid, hipStream_t stream) {
  const size_t lds = (size_t)M * K * 4 + (size_t)512 * NW * 4 + (size_t)NW * kRingBytesPerWave + 64;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fista_tile_sp_kernel<K, M, STOP, NW, DS, ZS>), lds);
      e != hipSuccess)
    return e;
  hipLaunchKernelGGL((fista_tile_sp_kernel<K, M, STOP, NW, DS, ZS>), dim3(grid), dim3(64 * NW), lds, stream, p);
  return hipGetLastError();
}

// the ZS instantiation of the fixed-iteration kernel (defined and instantiated in fista_tile_sp_zs.hip: its own
// translation unit, so that it builds beside the others)
template <int K, int M, int NW>
hipError_t launch_zero_start(const FistaTileParams& p, int grid, hipStream_t stream);

template <int K, int M, int NW = kFistaWaves>
static hipError_t occupancy_k(int* blocks_per_cu) {
  const size_t lds = (size_t)M * K * 4 + (size_t)512 * NW * 4 + (size_t)NW * kRingBytesPerWave + 64;
  const void* fn = reinterpret_cast<const void*>(&fista_tile_sp_kernel<K, M, true, NW>);
  if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, 64 * NW, lds);
}

template <int K, int M, int NW = kFistaWaves>
static hipError_t launch_k(const FistaTileParams& p, int grid, hipStream_t stream) {
  if (p.stop_on) return launch_ks<K, M, true, NW>(p, grid, stream);
  static const bool zs_off = getenv("LASSO_NO_ZERO_START") != nullptr;        // (A/B)
  if (!p.z_in && !p.y_in && p.iters > 0 && p.iters <= kZeroStartIters && !zs_off)
    return launch_zero_start<K, M, NW>(p, grid, stream);
  return launch_ks<K, M, false, NW>(p, grid, stream);
}

}  // namespace sp
}  // namespace lasso
