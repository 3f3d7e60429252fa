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
//   g -> G                    through the LDS tile as 16-byte row pieces (the next launch's accept reads it)
//
// bt_iter_decide_kernel then closes iteration i-1 (sum |z - z_i| over the tiles, the iteration's record, the stop
// rule :93-95) and takes the decision of iteration i (first trial with F <= Q, :45) -- TWO launches per outer
// iteration instead of seven, p / g / z of a tile cross HBM once per iteration in each direction (6 passes over [n,k]
// instead of 9: the trials' reload and the accept pass's own reads are gone), and the loads of a tile are exposed once
// per ~120 us of matrix work instead of once per launch.  The last iteration of a window is accepted by the same kernel
// in `tail` mode (accept only).  Arithmetic per element is the sequence of backtrack.hip's kernels (each product and
// sum rounded separately like the reference's ATen ops); the tile sums are taken in another (fixed) order.
#include "tile_device.hpp"

namespace lasso {

template <int K>
__global__ __launch_bounds__(kFistaThreads, 2) void bt_iter_kernel(const BtIterParams p, const BtSteps s) {
  constexpr int D = kFistaD;
  constexpr int NW = kFistaWaves;
  constexpr int KW = K / NW;
  constexpr int NP = KW / 32;
  constexpr int ITER = kTileM * (K / 4) / kFistaThreads;
  static_assert(ITER >= 1 && (kTileM * (K / 4)) % kFistaThreads == 0, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const rings = (lds_char*)smem;
  lds_char* const pt = rings + NW * kRingBytesPerWave;
  lds_char* const rt = pt + kTileM * K * 4;
  lds_f32* const red = (lds_f32*)(rt + kTileM * D * 4);
  if (p.skip && *p.skip != 0) return;
  const bool accept = p.acc_flags != nullptr;
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
  if (!p.tail) {
    dma_step(c.w1, c.voff1, c.ring);
    dma_step(c.w1 + 32, c.voff1, c.ring + kStepBytes);
  }
  float* const P = p.fast ? p.Y : p.Z;     // the point: y (FISTA) or z (ISTA), flat [n][k]
  // LDS byte offset of this lane's C-layout element (row 4q+rg, column colbase+n) of the [16][K] tile: tile_off()
  int ep_rg[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) ep_rg[rg] = (4 * q + rg) * (K * 4) + (((n >> 2) ^ rg) << 4) + ((n & 3) << 2);
  auto ep_addr = [&](int ps, int cb, int rg) {
    const int colbase = wid * KW + 32 * ps + 16 * cb;           // wave-uniform
    return (lds_f32*)(pt + ep_rg[rg] + ((((colbase >> 4) & 3) ^ q) << 6) + (colbase >> 6) * 256);
  };

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kTileM;
    float dsum = 0.0f;
    if (accept) {
      // ---- the accept step of the previous iteration on this tile; all pieces of a thread in flight together
      f32x4 pa[ITER], ga[ITER], za[ITER];
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int idx = tid + kFistaThreads * i, r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
        const int64_t off = (int64_t)min(row0 + r, p.n - 1) * p.k + min(cc, p.k - 4);
        za[i] = *reinterpret_cast<const f32x4*>(p.Z + off);
        ga[i] = *reinterpret_cast<const f32x4*>(p.G + off);
        if (p.fast) pa[i] = *reinterpret_cast<const f32x4*>(p.Y + off);
      }
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int idx = tid + kFistaThreads * i, r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
        const bool ok = (row0 + r) < p.n && cc < p.k;
        f32x4 zn = {0.f, 0.f, 0.f, 0.f}, yn = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zo = za[i][e];
            const float pv = p.fast ? pa[i][e] : zo;
            zn[e] = soft_threshold(__fsub_rn(pv, __fmul_rn(lr_a, ga[i][e])), lam_a);        // ista.py:40
            dsum += __builtin_fabsf(__fsub_rn(zo, zn[e]));                                    // :93
            yn[e] = __fadd_rn(zn[e], __fmul_rn(p.coef, __fsub_rn(zn[e], zo)));                // :99-100
          }
          const int64_t off = (int64_t)(row0 + r) * p.k + cc;
          *reinterpret_cast<f32x4*>(p.Z + off) = zn;                                          // :102
          if (p.fast) *reinterpret_cast<f32x4*>(p.Y + off) = yn;
        }
        *(lds_f32x4*)(pt + tile_chunk_off<K>(r, cc)) = p.fast ? yn : zn;
      }
    } else {
      visit_tile4<K, kFistaThreads>(P, p.k, row0, p.n, p.k, [&](int r, int cc, const f32x4& v) {
        *(lds_f32x4*)(pt + tile_chunk_off<K>(r, cc)) = v;
      });
    }
    if (p.tail) {
      dsum = wave_sum(dsum);
      if (lane == 0) red[NW + wid] = dsum;
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      if (tid == 0) {
        float a = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[NW + w];
        p.dpart[tile] = a;
      }
      __builtin_amdgcn_s_barrier();
      continue;
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
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    // ---- gradient at the point: r0 = p W^T - x, g = r0 W (ista.py:22-24)
    f32x4 gk[NP][2];
    {
      f32x4 acc[2] = {negx[0], negx[1]};
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
      static_for<NP>([&](auto ps_c) {
        constexpr int ps = decltype(ps_c)::value;
        gk[ps][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gk[ps][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gemm2_pass<K, ps>(c, rf, gk[ps]);
      });
    }
    // the point in the C layout (this lane's 32 elements of the tile; every lane reads and later overwrites only its own)
    f32x4 pk[NP][2];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) pk[ps][cb][rg] = *ep_addr(ps, cb, rg);
    __builtin_amdgcn_s_barrier();          // red[] of the gradient phase has been read

    // ---- the trials of this iteration on the tile (ista.py:38-47: same p, same g, steps lr0 / eta^t)
#pragma unroll 1
    for (int t = 0; t < p.ntrials; ++t) {
      const float lr = s.lr[t], lam = s.lam[t];
      float l1 = 0.0f, dzg = 0.0f, dz2 = 0.0f;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float pv = pk[ps][cb][rg], g = gk[ps][cb][rg];
            const float zn = soft_threshold(__fsub_rn(pv, __fmul_rn(lr, g)), lam);             // ista.py:40
            const float dz = __fsub_rn(zn, pv);                                                  // :31
            l1 += __builtin_fabsf(zn);
            dzg = __fadd_rn(dzg, __fmul_rn(dz, g));
            dz2 = __fadd_rn(dz2, __fmul_rn(dz, dz));
            *ep_addr(ps, cb, rg) = zn;
          }
      f32x4 acc[2] = {negx[0], negx[1]};
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      gemm1_stream_sp<K>(c, pt, acc, c.w1, c.w1 + 32, c.voff1);
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
        p.partsM[((int64_t)t * 4 + tid) * p.ntiles + tile] = a;
      }
      __builtin_amdgcn_s_barrier();        // the tile / red reuse by the next trial
    }

    // ---- g -> G through the LDS tile: 16-byte row pieces (the accept step of the next launch reads them)
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) *ep_addr(ps, cb, rg) = gk[ps][cb][rg];
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + kFistaThreads * i, r = idx / (K / 4), cc = (idx - r * (K / 4)) * 4;
      const f32x4 v = *(const lds_f32x4*)(pt + tile_chunk_off<K>(r, cc));
      if ((row0 + r) < p.n && cc < p.k) *reinterpret_cast<f32x4*>(p.G + (int64_t)(row0 + r) * p.k + cc) = v;
    }
    // (the next tile's prologue writes the chunks of the LDS tile this same thread has just read: program order)
  }
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
                                                              int* __restrict__ ctl, int* __restrict__ trials,
                                                              float* __restrict__ lrs, float* __restrict__ fs) {
  if (ctl[0] != 0) return;
  __shared__ double sh[kBtMultiMax][5][128];
  __shared__ float shd[256];
  __shared__ int stop;
  if (prev_flags) {
    if (threadIdx.x < 256) {
      float acc = 0.0f;
      for (int t = threadIdx.x; t < ntiles; t += 256) acc += dpart[t];
      shd[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) shd[threadIdx.x] += shd[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float delta = shd[0];
      int fired = 0;
      if (prev_flags[0] != 0) {
        trials[it_prev] = prev_flags[2] + 1;
        lrs[it_prev] = prev_fvals[2];
        fs[it_prev] = prev_fvals[0];
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
    cur_fvals[0] = F; cur_fvals[1] = Q;
    cur_flags[1] = first_index + u + 1;
    if (F <= Q) {                                                                  // :45
      cur_flags[0] = 1; cur_flags[2] = first_index + u; cur_fvals[2] = s.lr[u]; cur_fvals[3] = s.lam[u];
      return;
    }
  }
  if (last_batch) ctl[0] = 2;
}

template <int K>
static hipError_t launch_iter_k(const BtIterParams& p, const BtSteps& s, int grid, hipStream_t stream) {
  const size_t lds = (size_t)kTileM * K * 4 + (size_t)kTileM * kFistaD * 4 + (size_t)kFistaWaves * kRingBytesPerWave + 256;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt_iter_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt_iter_kernel<K>, dim3(grid), dim3(kFistaThreads), lds, stream, p, s);
  return hipGetLastError();
}

hipError_t launch_bt_iter(const BtIterParams& p, const BtSteps& s, int kpad, int grid, hipStream_t stream) {
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
                                 float budget, int* ctl, int* trials, float* lrs, float* fs, hipStream_t stream) {
  hipLaunchKernelGGL(bt_iter_decide_kernel, dim3(1), dim3(1024), 0, stream, partials, partsM, ntiles, (float)alpha, s,
                     ntrials, first_index, last_batch, cur_flags, cur_fvals, prev_flags, prev_fvals, dpart, it_prev,
                     budget, ctl, trials, lrs, fs);
  return hipGetLastError();
}

}  // namespace lasso
