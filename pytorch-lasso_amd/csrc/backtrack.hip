// Backtracking line search (Beck-Teboulle), replaces lasso/linear/solvers/ista.py:17-54.
// All sums are over the WHOLE batch (one step size for every sample), like the reference.
// Per outer iteration at the point p (y for FISTA, z for ISTA):
//   bt_grad_kernel    r0 = p W^T - x, g0 = r0 W -> HBM, rss0 partials        (:22-24)
//   bt_trial_kernel   z+ = S(p - lr g0) -> HBM candidate, r1 = z+ W^T - x,
//                     partials {sum r1^2, sum|z+|, sum dz*g0, sum dz^2}         (:40-42)
//   bt_decide_kernel  F <= Q ?  (fp32, operation order of :26-35,45)  -> accepted flag
//   bt_finish_kernel  sum|z - z+|, y = z+ + c (z+ - z), z = z+                   (:93-102)
// Trial kernels launched after an accepted trial exit immediately (they read the flag),
// so a batch of trials can be enqueued without a host round trip per trial.
// Rooflines: grad = 4ndk flop, each trial = 2ndk flop (MFMA-bound, same streams as the
// fused kernel); finish is an HBM-bound elementwise pass (5 n k floats moved).
#include "tile_device.hpp"

namespace lasso {

template <int K>
__global__ __launch_bounds__(kFistaThreads, 2) void bt_grad_kernel(const BtParams p) {
  constexpr int D = kFistaD;
  constexpr int NW = kFistaWaves;
  constexpr int KW = K / NW;
  constexpr int NP = KW / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const pt = rings + NW * kRingBytesPerWave;
  lds_char* const rt = pt + kTileM * K * 4;
  lds_f32* const red = (lds_f32*)(rt + kTileM * D * 4);
  if (p.skip && *p.skip != 0) return;

  TileCtx<K> c;
  c.init(p.Wp, p.Wtp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;
  dma_step(c.w1, c.voff1, c.ring);
  dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    visit_tile4<K, kFistaThreads>(p.P, p.ldp, row0, p.n, p.k, [&](int r, int cc, const f32x4& v) {
      *(lds_f32x4*)(pt + tile_chunk_off<K>(r, cc)) = v;
    });
    f32x4 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = 32 * wid + 16 * cb + n;
        float v = 0.0f;
        if ((row0 + r) < p.n && cc < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + cc];
        acc[cb][rg] = -v;
      }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    gemm1_stream_sp<K>(c, pt, acc, c.w2, c.w2 + 32, c.voff2);
    float rss = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        rss = fmaf(acc[cb][rg], acc[cb][rg], rss);
        *(lds_f32*)(rt + tile_off<D>(4 * q + rg, 32 * wid + 16 * cb + n)) = acc[cb][rg];
      }
    rss = wave_sum(rss);
    if (lane == 0) red[wid] = rss;
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (tid == 0) {
      float a = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += red[w];
      p.partials[tile] = a;
    }
    f32x4 rf[D / 32][2];
    load_r_frags<K>(c, rt, rf);
    float* const g_base = p.G + (int64_t)row0 * p.k;
    static_for<NP>([&](auto ps_c) {
      constexpr int ps = decltype(ps_c)::value;
      f32x4 g2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      gemm2_pass<K, ps>(c, rf, g2);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 4 * q + rg, cc = wid * KW + 32 * ps + 16 * cb + n;
          if ((row0 + r) < p.n && cc < p.k) g_base[r * p.k + cc] = g2[cb][rg];
        }
    });
    __builtin_amdgcn_s_barrier();   // pt / rt / red reuse by the next tile
  }
  LASSO_WAIT_VMCNT(0);
}

template <int K>
__global__ __launch_bounds__(kFistaThreads, 2) void bt_trial_kernel(const BtParams p, float lr,
                                                                    float lam, int force) {
  constexpr int NW = kFistaWaves;
  if (p.skip && *p.skip != 0) return;
  if (!force && p.flags[0] != 0) return;   // an earlier trial of this iteration was accepted
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const zt = rings + NW * kRingBytesPerWave;
  lds_f32* const red = (lds_f32*)(zt + kTileM * K * 4);

  TileCtx<K> c;
  c.init(p.Wp, p.Wp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;
  dma_step(c.w1, c.voff1, c.ring);
  dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    float l1 = 0.0f, dzg = 0.0f, dz2 = 0.0f;
    const bool gvec = vec4_ok(p.G, p.k, p.k);   // G and C are internal [n][k] buffers
    // p and g of the tile: all 16 pieces of a thread in flight together (one memory round trip per tile)
    visit_tile4x2<K, kFistaThreads>(p.P, p.ldp, p.G, p.k, row0, p.n, p.k, [&](int r, int cc, const f32x4& pv, const f32x4& g) {
      f32x4 zn = {0.f, 0.f, 0.f, 0.f};
      if ((row0 + r) < p.n && cc < p.k) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cc + e < p.k) {
            zn[e] = soft_threshold(__fsub_rn(pv[e], __fmul_rn(lr, g[e])), lam);     // ista.py:40
            const float dz = __fsub_rn(zn[e], pv[e]);                                 // :31
            l1 += __builtin_fabsf(zn[e]);
            dzg = __fadd_rn(dzg, __fmul_rn(dz, g[e]));
            dz2 = __fadd_rn(dz2, __fmul_rn(dz, dz));
          }
        if (p.C) store_row4(p.C, p.k, row0 + r, p.n, p.k, cc, zn, gvec);   // nullptr: the finish kernel recomputes it
      }
      *(lds_f32x4*)(zt + tile_chunk_off<K>(r, cc)) = zn;
    });
    f32x4 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = 32 * wid + 16 * cb + n;
        float v = 0.0f;
        if ((row0 + r) < p.n && cc < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + cc];
        acc[cb][rg] = -v;
      }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    gemm1_stream_sp<K>(c, zt, acc, c.w1, c.w1 + 32, c.voff1);
    float rss = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) rss = fmaf(acc[cb][rg], acc[cb][rg], rss);
    rss = wave_sum(rss); l1 = wave_sum(l1); dzg = wave_sum(dzg); dz2 = wave_sum(dz2);
    if (lane == 0) { red[4 * wid] = rss; red[4 * wid + 1] = l1; red[4 * wid + 2] = dzg; red[4 * wid + 3] = dz2; }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (tid < 4) {
      float a = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += red[4 * w + tid];
      p.partials[(int64_t)p.ntiles * (1 + tid) + tile] = a;
    }
    __builtin_amdgcn_s_barrier();
  }
  LASSO_WAIT_VMCNT(0);
}

// Several trials of one outer iteration in ONE launch (round 4).  Every trial t of an iteration uses the same point p,
// the same gradient g0 and the step lr0 / eta^t (ista.py:38-47): the single-trial kernel above re-read p and g of
// every tile for every trial (134 MB per trial at config 3, a burst of ~27 us in front of 61 us of GEMM) and cost a
// launch plus a decision launch each.  Here a workgroup keeps p and g of its tile in REGISTERS (8 + 8 16-byte pieces
// per thread), and per trial forms the candidate tile in LDS from them, runs GEMM-1 on it and writes the trial's four
// tile sums to `partsM[t]`; bt_decide_multi_kernel then takes the first trial with F <= Q.  Needs 16-byte-aligned
// flat p / g (k % 4 == 0; the driver falls back to the single-trial launches otherwise) and the recomputing accept
// step (no candidate goes to memory).
template <int K>
__global__ __launch_bounds__(kFistaThreads, 2) void bt_trials_kernel(const BtParams p, const BtSteps s, int ntrials,
                                                                     float* __restrict__ partsM) {
  constexpr int NW = kFistaWaves;
  constexpr int ITER = kTileM * (K / 4) / kFistaThreads;
  if (p.skip && *p.skip != 0) return;
  if (p.flags[0] != 0) return;             // a trial of an earlier batch of this iteration was accepted
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const zt = rings + NW * kRingBytesPerWave;
  lds_f32* const red = (lds_f32*)(zt + kTileM * K * 4);

  TileCtx<K> c;
  c.init(p.Wp, p.Wp, rings);
  const int tid = threadIdx.x;
  const int lane = c.lane, wid = c.wid, n = c.n, q = c.q;
  dma_step(c.w1, c.voff1, c.ring);
  dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    f32x4 pa[ITER], ga[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {       // all 2 ITER pieces in flight together: one memory round trip per tile
      const int idx = tid + kFistaThreads * i, r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
      const int64_t rr = min(row0 + r, p.n - 1);
      const int cl = min(cc, p.k - 4);
      pa[i] = *reinterpret_cast<const f32x4*>(p.P + rr * p.ldp + cl);
      ga[i] = *reinterpret_cast<const f32x4*>(p.G + rr * (int64_t)p.k + cl);
    }
    f32x4 negx[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = 32 * wid + 16 * cb + n;
        float v = 0.0f;
        if ((row0 + r) < p.n && cc < p.d) v = p.X[(int64_t)(row0 + r) * p.ldx + cc];
        negx[cb][rg] = -v;
      }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {       // rows / columns beyond the matrix read clamped addresses: zero them
      const int idx = tid + kFistaThreads * i, r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
      if (!((row0 + r) < p.n && cc < p.k)) { pa[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ga[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll 1
    for (int t = 0; t < ntrials; ++t) {
      const float lr = s.lr[t], lam = s.lam[t];
      float l1 = 0.0f, dzg = 0.0f, dz2 = 0.0f;
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int idx = tid + kFistaThreads * i, r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
        f32x4 zn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          zn[e] = soft_threshold(__fsub_rn(pa[i][e], __fmul_rn(lr, ga[i][e])), lam);     // ista.py:40
          const float dz = __fsub_rn(zn[e], pa[i][e]);                                    // :31
          l1 += __builtin_fabsf(zn[e]);
          dzg = __fadd_rn(dzg, __fmul_rn(dz, ga[i][e]));
          dz2 = __fadd_rn(dz2, __fmul_rn(dz, dz));
        }
        *(lds_f32x4*)(zt + tile_chunk_off<K>(r, cc)) = zn;
      }
      f32x4 acc[2] = {negx[0], negx[1]};
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      gemm1_stream_sp<K>(c, zt, acc, c.w1, c.w1 + 32, c.voff1);
      float rss = 0.0f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rss = fmaf(acc[cb][rg], acc[cb][rg], rss);
      rss = wave_sum(rss); l1 = wave_sum(l1); dzg = wave_sum(dzg); dz2 = wave_sum(dz2);
      if (lane == 0) { red[4 * wid] = rss; red[4 * wid + 1] = l1; red[4 * wid + 2] = dzg; red[4 * wid + 3] = dz2; }
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      if (tid < 4) {
        float a = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[4 * w + tid];
        partsM[((int64_t)t * 4 + tid) * p.ntiles + tile] = a;
      }
      __builtin_amdgcn_s_barrier();        // zt / red reuse by the next trial
    }
  }
  LASSO_WAIT_VMCNT(0);
}

// The decisions of a batch of trials by one block: 128 threads per trial add that trial's five sums (double, fixed
// strides and tree -- the single-trial decision's sums in another grouping), then thread 0 walks the trials in order
// with the fp32 operation order of bt_decide_kernel; the first trial with F <= Q is the accepted one.  (As a loop over
// the trials with one block-wide reduction each this launch took 13 us for five trials; the trials' sums are independent.)
// partials: [0][tile] = sum r0^2 of the gradient kernel.
__global__ __launch_bounds__(1024) void bt_decide_multi_kernel(const float* __restrict__ partials,
                                                               const float* __restrict__ partsM, int ntiles,
                                                               float alpha, const BtSteps s, int ntrials,
                                                               int first_index, int* __restrict__ flags,
                                                               float* __restrict__ fvals, const int* __restrict__ skip) {
  if (skip && *skip != 0) return;
  if (flags[0] != 0) return;
  __shared__ double sh[kBtMultiMax][5][128];
  const int t = threadIdx.x >> 7, l = threadIdx.x & 127;
  double acc[5] = {0, 0, 0, 0, 0};
  if (t < ntrials)
    for (int tl = l; tl < ntiles; tl += 128) {
      acc[0] += partials[tl];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[1 + q] += partsM[((size_t)t * 4 + q) * ntiles + tl];
    }
#pragma unroll
  for (int q = 0; q < 5; ++q) sh[t][q][l] = acc[q];
  __syncthreads();
  for (int st = 64; st > 0; st >>= 1) {
    if (l < st)
#pragma unroll
      for (int q = 0; q < 5; ++q) sh[t][q][l] += sh[t][q][l + st];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  for (int u = 0; u < ntrials; ++u) {
    const float rss0 = (float)sh[u][0][0], rss1 = (float)sh[u][1][0], l1 = (float)sh[u][2][0];
    const float dzg = (float)sh[u][3][0], dz2 = (float)sh[u][4][0];
    const float f0 = __fmul_rn(0.5f, rss0);                                        // ista.py:23
    const float al1 = __fmul_rn(alpha, l1);
    const float F = __fadd_rn(__fmul_rn(0.5f, rss1), al1);                         // :28
    const float Q = __fadd_rn(__fadd_rn(__fadd_rn(f0, dzg), __fmul_rn(s.hol[u], dz2)), al1);  // :32-35
    fvals[0] = F; fvals[1] = Q;
    flags[1] = first_index + u + 1;
    if (F <= Q) {                                                                  // :45
      flags[0] = 1; flags[2] = first_index + u; fvals[2] = s.lr[u]; fvals[3] = s.lam[u];
      return;
    }
  }
}

// One block.  partials layout: [0][t] rss0, [1][t] rss1, [2][t] l1, [3][t] dz.g0, [4][t] dz^2.
// flags: [0] accepted, [1] trials evaluated so far this iteration, [2] index of accepted trial.
// fvals: [0] F, [1] Q of the last evaluated trial (diagnostics), [2] accepted lr, [3] its alpha*lr
__global__ __launch_bounds__(256) void bt_decide_kernel(const float* __restrict__ partials, int ntiles,
                                                        float alpha, float half_over_lr, float lr, float lam,
                                                        int trial_index, int force,
                                                        int* __restrict__ flags, float* __restrict__ fvals,
                                                        double* __restrict__ sums_out, const int* __restrict__ skip) {
  if (skip && *skip != 0) return;
  if (!force && flags[0] != 0) return;
  __shared__ double sh[5][256];
  double acc[5] = {0, 0, 0, 0, 0};
  for (int t = threadIdx.x; t < ntiles; t += 256)
#pragma unroll
    for (int s = 0; s < 5; ++s) acc[s] += partials[(size_t)s * ntiles + t];
#pragma unroll
  for (int s = 0; s < 5; ++s) sh[s][threadIdx.x] = acc[s];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int s = 0; s < 5; ++s) sh[s][threadIdx.x] += sh[s][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sums_out) {
    // row-sharded solve: this rank's five sums only; the host adds the ranks' and decides
#pragma unroll
    for (int s = 0; s < 5; ++s) sums_out[s] = sh[s][0];
    return;
  }
  if (threadIdx.x == 0) {
    const float rss0 = (float)sh[0][0], rss1 = (float)sh[1][0], l1 = (float)sh[2][0];
    const float dzg = (float)sh[3][0], dz2 = (float)sh[4][0];
    const float f0 = __fmul_rn(0.5f, rss0);                                        // ista.py:23
    const float al1 = __fmul_rn(alpha, l1);
    const float F = __fadd_rn(__fmul_rn(0.5f, rss1), al1);                         // :28
    const float Q = __fadd_rn(__fadd_rn(__fadd_rn(f0, dzg), __fmul_rn(half_over_lr, dz2)), al1);  // :32-35
    fvals[0] = F; fvals[1] = Q;
    flags[1] = trial_index + 1;
    if (force || F <= Q) { flags[0] = 1; flags[2] = trial_index; fvals[2] = lr; fvals[3] = lam; }  // :45
  }
}

// z_next = C (accepted candidate): delta partials, momentum, state update.  Grid-stride
// over n*k with a FIXED grid so the reduction order is deterministic.
__global__ __launch_bounds__(256) void bt_finish_kernel(float* __restrict__ Z, int64_t ldz,
                                                        float* __restrict__ Y, const float* __restrict__ Cand,
                                                        int n, int k, float coef, const int* __restrict__ flags,
                                                        float* __restrict__ dpart) {
  __shared__ float sh[256];
  float acc = 0.0f;
  if (flags[0] != 0) {
    const int64_t total = (int64_t)n * k;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
      const int64_t r = idx / k;
      const int cc = (int)(idx - r * k);
      const float zo = Z[r * ldz + cc];
      const float zn = Cand[idx];
      acc += __builtin_fabsf(__fsub_rn(zo, zn));                                   // ista.py:93
      Y[idx] = __fadd_rn(zn, __fmul_rn(coef, __fsub_rn(zn, zo)));                  // :99-100
      Z[r * ldz + cc] = zn;                                                        // :102
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dpart[blockIdx.x] = sh[0];
}

// Same as bt_finish_kernel, but the accepted candidate is recomputed from the point and its
// gradient with the accepted step (fvals[2], fvals[3]) instead of being read back -- the
// bf16 trial kernels do not write their candidates to HBM.
__global__ __launch_bounds__(256) void bt_finish_recompute_kernel(float* __restrict__ Z, float* __restrict__ Y,
                                                                  const float* __restrict__ P,
                                                                  const float* __restrict__ G, int64_t total,
                                                                  float coef, const int* __restrict__ flags,
                                                                  const float* __restrict__ fvals,
                                                                  float* __restrict__ dpart,
                                                                  const int* __restrict__ skip) {
  __shared__ float sh[256];
  float acc = 0.0f;
  if (skip && *skip != 0) return;
  if (flags[0] != 0) {
    const float lr = fvals[2], lam = fvals[3];
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
      const float zo = Z[idx];
      const float zn = soft_threshold(__fsub_rn(P[idx], __fmul_rn(lr, G[idx])), lam);   // ista.py:40
      acc += __builtin_fabsf(__fsub_rn(zo, zn));                                          // :93
      Y[idx] = __fadd_rn(zn, __fmul_rn(coef, __fsub_rn(zn, zo)));                         // :99-100 (P may alias Y)
      Z[idx] = zn;                                                                        // :102
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dpart[blockIdx.x] = sh[0];
}

// LDS of the grad kernel: p tile + r tile + the per-wave DMA rings + partial sums
static size_t bt_grad_lds_bytes(int K) {
  return (size_t)kTileM * K * 4 + (size_t)kTileM * kFistaD * 4 + (size_t)kFistaWaves * kRingBytesPerWave + 64;
}

template <int K>
static hipError_t launch_grad_k(const BtParams& p, int grid, hipStream_t stream) {
  const size_t lds = bt_grad_lds_bytes(K);
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt_grad_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt_grad_kernel<K>, dim3(grid), dim3(kFistaThreads), lds, stream, p);
  return hipGetLastError();
}

template <int K>
static hipError_t launch_trial_k(const BtParams& p, float lr, float lam, int force, int grid,
                                 hipStream_t stream) {
  const size_t lds = (size_t)kFistaWaves * kRingBytesPerWave + (size_t)kTileM * K * 4 + 256;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt_trial_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt_trial_kernel<K>, dim3(grid), dim3(kFistaThreads), lds, stream, p, lr, lam, force);
  return hipGetLastError();
}

hipError_t launch_bt_grad(const BtParams& p, int kpad, int grid, hipStream_t stream) {
  switch (kpad) {
    case 256: return launch_grad_k<256>(p, grid, stream);
    case 512: return launch_grad_k<512>(p, grid, stream);
    case 1024: return launch_grad_k<1024>(p, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_bt_trial(const BtParams& p, int kpad, int grid, double alpha, double lr,
                           int trial_index, int force, hipStream_t stream, double* sums_out) {
  const float lr_f = (float)lr, lam = (float)(alpha * lr);
  hipError_t e;
  switch (kpad) {
    case 256: e = launch_trial_k<256>(p, lr_f, lam, force, grid, stream); break;
    case 512: e = launch_trial_k<512>(p, lr_f, lam, force, grid, stream); break;
    case 1024: e = launch_trial_k<1024>(p, lr_f, lam, force, grid, stream); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  return launch_bt_decide(p, alpha, lr, trial_index, force, stream, sums_out);
}

template <int K>
static hipError_t launch_trials_k(const BtParams& p, const BtSteps& s, int ntrials, float* partsM, int grid,
                                  hipStream_t stream) {
  const size_t lds = (size_t)kFistaWaves * kRingBytesPerWave + (size_t)kTileM * K * 4 + 256;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt_trials_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt_trials_kernel<K>, dim3(grid), dim3(kFistaThreads), lds, stream, p, s, ntrials, partsM);
  return hipGetLastError();
}

// `ntrials` <= kBtMultiMax trials with the steps of `s`: the tile sums only (the caller launches the decision)
hipError_t launch_bt_trials_only(const BtParams& p, int kpad, int grid, const BtSteps& s, int ntrials, float* partsM,
                                 hipStream_t stream) {
  switch (kpad) {
    case 256: return launch_trials_k<256>(p, s, ntrials, partsM, grid, stream);
    case 512: return launch_trials_k<512>(p, s, ntrials, partsM, grid, stream);
    case 1024: return launch_trials_k<1024>(p, s, ntrials, partsM, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

// `ntrials` <= kBtMultiMax trials with the steps of `s` (trial indices first_index ..), then their decisions
hipError_t launch_bt_trials(const BtParams& p, int kpad, int grid, double alpha, const BtSteps& s, int ntrials,
                            int first_index, float* partsM, hipStream_t stream) {
  if (hipError_t e = launch_bt_trials_only(p, kpad, grid, s, ntrials, partsM, stream); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt_decide_multi_kernel, dim3(1), dim3(1024), 0, stream, p.partials, partsM, p.ntiles, (float)alpha, s,
                     ntrials, first_index, p.flags, p.fvals, p.skip);
  return hipGetLastError();
}

hipError_t launch_bt_decide(const BtParams& p, double alpha, double lr, int trial_index, int force,
                            hipStream_t stream, double* sums_out) {
  hipLaunchKernelGGL(bt_decide_kernel, dim3(1), dim3(256), 0, stream, p.partials, p.ntiles,
                     (float)alpha, (float)(0.5 / lr), (float)lr, (float)(alpha * lr), trial_index, force, p.flags,
                     p.fvals, sums_out, p.skip);
  return hipGetLastError();
}

// ---- unfused line search (d > 256 or k > 1024): element-wise halves around the general GEMM ------
// part[b] = sum over this block's grid-stride elements of v^2 (fixed grid => deterministic)
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ v, int64_t total,
                                                             float* __restrict__ part) {
  __shared__ float sh[256];
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    acc = fmaf(v[i], v[i], acc);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// candidate z_next = S_lam(p - lr g) -> Cand, and the block's sums of |z_next|, dz.g, dz^2
// (dz = z_next - p; ista.py:31-35,40) -> l1[b], dzg[b], dz2[b]
__global__ __launch_bounds__(256) void generic_trial_kernel(const float* __restrict__ P, const float* __restrict__ G,
                                                            float* __restrict__ Cand, int64_t total, float lr, float lam,
                                                            float* __restrict__ l1p, float* __restrict__ dzgp,
                                                            float* __restrict__ dz2p) {
  __shared__ float sh[3][256];
  float l1 = 0.0f, dzg = 0.0f, dz2 = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float pv = P[i], g = G[i];
    const float zn = soft_threshold(__fsub_rn(pv, __fmul_rn(lr, g)), lam);
    const float dz = __fsub_rn(zn, pv);
    Cand[i] = zn;
    l1 += __builtin_fabsf(zn);
    dzg = __fadd_rn(dzg, __fmul_rn(dz, g));
    dz2 = __fadd_rn(dz2, __fmul_rn(dz, dz));
  }
  sh[0][threadIdx.x] = l1; sh[1][threadIdx.x] = dzg; sh[2][threadIdx.x] = dz2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int j = 0; j < 3; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) { l1p[blockIdx.x] = sh[0][0]; dzgp[blockIdx.x] = sh[1][0]; dz2p[blockIdx.x] = sh[2][0]; }
}

hipError_t launch_sumsq_partials(const float* v, int64_t total, float* part, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(grid), dim3(256), 0, stream, v, total, part);
  return hipGetLastError();
}

hipError_t launch_generic_trial(const float* P, const float* G, float* Cand, int64_t total, float lr, float lam,
                                float* partials, int grid, hipStream_t stream) {
  // partials layout of bt_decide_kernel: [0] rss0, [1] rss1, [2] l1, [3] dz.g, [4] dz^2, each [grid]
  hipLaunchKernelGGL(generic_trial_kernel, dim3(grid), dim3(256), 0, stream, P, G, Cand, total, lr, lam,
                     partials + 2 * (size_t)grid, partials + 3 * (size_t)grid, partials + 4 * (size_t)grid);
  return hipGetLastError();
}

hipError_t launch_bt_finish_recompute(float* Z, float* Y, const float* P, const float* G, int64_t total, float coef,
                                      const int* flags, const float* fvals, float* dpart, int grid,
                                      hipStream_t stream, const int* skip) {
  hipLaunchKernelGGL(bt_finish_recompute_kernel, dim3(grid), dim3(256), 0, stream, Z, Y, P, G, total, coef, flags,
                     fvals, dpart, skip);
  return hipGetLastError();
}

// End of outer iteration `it` of a solve enqueued without host waits (one block).  ctl: [0] 0 = running, 1 = the stop
// rule fired, 2 = no trial of the pre-enqueued batch was accepted (the host continues from iteration ctl[1] on the
// synchronous path: the state is untouched), [1] iterations completed, [2] last sum |z - z_next| (float bits).
__global__ __launch_bounds__(256) void bt_iter_end_kernel(const float* __restrict__ dpart, int nparts,
                                                          int* __restrict__ flags, const float* __restrict__ fvals,
                                                          int* __restrict__ ctl, int it, float budget,
                                                          int* __restrict__ trials, float* __restrict__ lrs,
                                                          float* __restrict__ fs) {
  if (ctl[0] != 0) return;
  __shared__ float sh[256];
  float acc = 0.0f;
  for (int t = threadIdx.x; t < nparts; t += 256) acc += dpart[t];     // the order of reduce_partials_kernel
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (flags[0] == 0) {
      ctl[0] = 2;
    } else {
      const float delta = sh[0];
      trials[it] = flags[2] + 1;
      lrs[it] = fvals[2];
      fs[it] = fvals[0];
      ctl[1] = it + 1;
      ctl[2] = __float_as_int(delta);
      if (budget >= 0.0f && delta <= budget) ctl[0] = 1;                 // ista.py:93-95
    }
    flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;             // the next iteration starts its search afresh
  }
}

hipError_t launch_bt_iter_end(const float* dpart, int nparts, int* flags, const float* fvals, int* ctl, int it,
                              float budget, int* trials, float* lrs, float* fs, hipStream_t stream) {
  hipLaunchKernelGGL(bt_iter_end_kernel, dim3(1), dim3(256), 0, stream, dpart, nparts, flags, fvals, ctl, it, budget,
                     trials, lrs, fs);
  return hipGetLastError();
}

hipError_t launch_bt_finish(float* Z, int64_t ldz, float* Y, const float* Cand, int n, int k,
                            float coef, const int* flags, float* dpart, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(bt_finish_kernel, dim3(grid), dim3(256), 0, stream, Z, ldz, Y, Cand, n, k, coef,
                     flags, dpart);
  return hipGetLastError();
}

}  // namespace lasso
