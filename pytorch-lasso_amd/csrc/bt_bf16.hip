// bf16 backtracking line search (BASELINE config 3: FISTA with backtracking, n=16384,
// d=256, k=1024, bf16 tensors; reference lasso/linear/solvers/ista.py:17-54 run on bf16
// tensors).  Same protocol as backtrack.hip (grad / trial / decide / finish, whole-batch
// sums, trial kernels exit once a trial is accepted) with the two kernels that hold the
// GEMMs rebuilt for the bf16 matrix pipe:
//
//   * operands in bf16 (x and W arrive in bf16 and are used exactly; the point p and the
//     residual r are rounded to bf16 where the reference's bf16 tensors round them too),
//     accumulation and all scalar sums in fp32 (v_mfma_f32_16x16x32_bf16);
//   * 64-ROW tiles per workgroup (8 waves): a 16x16x32 bf16 MFMA is ~16x faster than the
//     fp32 16x16x4 one, so with 16-row tiles the W fragments would have to arrive 16x
//     faster than L2/LDS can deliver them; four row blocks share every W fragment;
//   * the p tile (64 x K bf16 = 128 KiB at K=1024) lives in LDS, 16-byte chunks XOR-
//     swizzled by the row so that the ds_read_b128 A fragments of 16 rows hit 16 different
//     bank groups; the r tile of the second GEMM reuses the same LDS;
//   * W is pre-packed fragment-major in bf16 (pack_w_bf16_kernel): every B fragment of a
//     wave is one contiguous, fully coalesced 1 KiB global load straight into VGPRs
//     (prefetched two steps ahead) -- no LDS bandwidth is spent on W at all.
// With the GEMMs ~10x shorter the kernels are HBM-bound: grad reads p and writes g
// (8*n*k bytes), a trial reads p and g (8*n*k bytes; the candidate never goes to HBM).
#include "bf16_device.hpp"

namespace lasso {
namespace {

using namespace bf16dev;

// ---------------------------------------------------------------------------
// grad: r0 = p W^T - x, g0 = r0 W -> G (fp32), partials[0][tile] = sum r0^2      (ista.py:22-24)
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(kThreads, 1) void bt16_grad_kernel(const BtParams p) {
  constexpr int CPR = K / 8;                  // 16-byte chunks per p-tile row
  constexpr int CB2 = K / 128;                // 16-column blocks of g per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const pt = (lds_char*)smem;       // [64][K] bf16; the r tile [64][256] reuses its start
  lds_f32* const red = (lds_f32*)(pt + kRows * K * 2);
  if (p.skip && *p.skip != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const __bf16* X = (const __bf16*)p.Xh;
  const bf16x8* wq1 = (const bf16x8*)p.Wq1 + (int64_t)wid * (K / 32) * 2 * 64;
  const bf16x8* wq2 = (const bf16x8*)p.Wq2 + (int64_t)wid * 8 * CB2 * 64;
  const bool pvec = ((uintptr_t)p.P & 15) == 0 && p.ldp % 4 == 0;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kRows;
    for (int c = tid; c < kRows * CPR; c += kThreads) {
      const int r = c / CPR, ch = c % CPR;
      float v[8];
      load8(p.P, p.ldp, row0 + r, p.n, 8 * ch, p.k, pvec, v);
      *(lds_bf16x8*)(pt + tile16_off<K * 2>(r, ch)) = to_bf16x8(v);
    }
    __syncthreads();
    f32x4 acc[4][2] = {};
    gemm1_bf16<K>(pt, wq1, lane, acc);
    __syncthreads();                          // every wave is done reading the p tile
    float rss = residual_epilogue<true>(acc, X, p.ldx, row0, p.n, p.d, wid, lane, pt);
    __syncthreads();
    // g (64 x K/8 per wave) = r tile (64 x 256) * W[:, this wave's K/8 atoms], at most 64 atoms
    // (4 column blocks) at a time so that accumulators + W fragments stay in registers
    constexpr int CBH = CB2 > 4 ? 4 : CB2;
    static_for<CB2 / CBH>([&](auto h_c) {
      constexpr int h = decltype(h_c)::value;
      f32x4 g[4][CBH] = {};
      const int i = lane & 15, kg = lane >> 4;
      bf16x8 b[2][CBH];
#pragma unroll
      for (int cb = 0; cb < CBH; ++cb) b[0][cb] = wq2[(0 * CB2 + h * CBH + cb) * 64 + lane];
#pragma unroll 1
      for (int j = 0; j < 4; ++j) {              // rolled: two steps per trip, fragments one step ahead
        static_for<2>([&](auto u_c) {
          constexpr int u = decltype(u_c)::value;
          const int s = 2 * j + u;
          const int sp = min(s + 1, 7);
#pragma unroll
          for (int cb = 0; cb < CBH; ++cb) b[(u + 1) & 1][cb] = wq2[(sp * CB2 + h * CBH + cb) * 64 + lane];
          bf16x8 a[4];
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) a[rb] = *(const lds_bf16x8*)(pt + tile16_off<kFistaD * 2>(16 * rb + i, 4 * s + kg));
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < CBH; ++cb)
              g[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b[u][cb], g[rb][cb], 0, 0, 0);
        });
      }
      const int cl = lane & 15, q = lane >> 4;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBH; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int r = row0 + 16 * rb + 4 * q + rg, cc = (K / 8) * wid + 16 * (h * CBH + cb) + cl;
            if (r < p.n && cc < p.k) p.G[(int64_t)r * p.k + cc] = g[rb][cb][rg];
          }
    });
    rss = wave_sum(rss);
    if (lane == 0) red[wid] = rss;
    __syncthreads();
    if (tid == 0) {
      float a = 0.0f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) a += red[w];
      p.partials[tile] = a;
    }
    __syncthreads();                          // red[] and the tile are reused by the next tile
  }
}

// ---------------------------------------------------------------------------
// trial: z+ = S(p - lr g0) (kept in LDS only: the finish kernel recomputes the accepted one),
// r1 = z+ W^T - x,
// partials {[1] sum r1^2, [2] sum|z+|, [3] sum dz*g0, [4] sum dz^2}                 (ista.py:26-42)
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(kThreads, 1) void bt16_trial_kernel(const BtParams p, float lr, float lam, int force) {
  constexpr int CPR = K / 8;
  if (p.skip && *p.skip != 0) return;
  if (!force && p.flags[0] != 0) return;      // an earlier trial of this iteration was accepted
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const zt = (lds_char*)smem;
  lds_f32* const red = (lds_f32*)(zt + kRows * K * 2);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const __bf16* X = (const __bf16*)p.Xh;
  const bf16x8* wq1 = (const bf16x8*)p.Wq1 + (int64_t)wid * (K / 32) * 2 * 64;
  const bool pvec = ((uintptr_t)p.P & 15) == 0 && p.ldp % 4 == 0;
  const bool gvec = p.k % 4 == 0;             // G and C are internal [n][k] buffers
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int row0 = tile * kRows;
    float l1 = 0.0f, dzg = 0.0f, dz2 = 0.0f;
    for (int c = tid; c < kRows * CPR; c += kThreads) {
      const int r = c / CPR, ch = c % CPR;
      float pv[8], gv[8], zn[8];
      load8(p.P, p.ldp, row0 + r, p.n, 8 * ch, p.k, pvec, pv);
      load8(p.G, p.k, row0 + r, p.n, 8 * ch, p.k, gvec, gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        zn[e] = soft_threshold(__fsub_rn(pv[e], __fmul_rn(lr, gv[e])), lam);         // ista.py:40
        const float dz = __fsub_rn(zn[e], pv[e]);                                     // :31
        l1 += __builtin_fabsf(zn[e]);
        dzg = __fadd_rn(dzg, __fmul_rn(dz, gv[e]));
        dz2 = __fadd_rn(dz2, __fmul_rn(dz, dz));
      }
      *(lds_bf16x8*)(zt + tile16_off<K * 2>(r, ch)) = to_bf16x8(zn);
    }
    __syncthreads();
    f32x4 acc[4][2] = {};
    gemm1_bf16<K>(zt, wq1, lane, acc);
    float rss = residual_epilogue<false>(acc, X, p.ldx, row0, p.n, p.d, wid, lane, nullptr);
    rss = wave_sum(rss); l1 = wave_sum(l1); dzg = wave_sum(dzg); dz2 = wave_sum(dz2);
    if (lane == 0) { red[4 * wid] = rss; red[4 * wid + 1] = l1; red[4 * wid + 2] = dzg; red[4 * wid + 3] = dz2; }
    __syncthreads();
    if (tid < 4) {
      float a = 0.0f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) a += red[4 * w + tid];
      p.partials[(int64_t)p.ntiles * (1 + tid) + tile] = a;
    }
    __syncthreads();
  }
}

// Fragment-major bf16 copies of W [d][k] (row stride ldw) for the two GEMMs.
//   Wq1[w][s][cb][lane][t] = W[32w + 16cb + lane%16][32s + 8(lane/16) + t]        (r = p W^T)
//   Wq2[w][s][cb][lane][t] = W[32s + 8(lane/16) + t][(Kp/8)w + 16cb + lane%16]    (g = r W)
template <typename TIn>
__global__ __launch_bounds__(256) void pack_w_bf16_kernel(const TIn* __restrict__ W, int64_t ldw, int d, int k, int kp,
                                                          __bf16* __restrict__ q1, __bf16* __restrict__ q2) {
  const int S1 = kp / 32, CB2 = kp / 128;
  const int total = kFistaD * kp / 8;         // 16-byte chunks per packed copy
  for (int c = blockIdx.x * 256 + threadIdx.x; c < total; c += gridDim.x * 256) {
    {
      const int lane = c % 64, cb = (c / 64) % 2, s = (c / 128) % S1, w = c / (128 * S1);
      const int dd = 32 * w + 16 * cb + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        q1[(int64_t)c * 8 + t] = (dd < d && k0 + t < k) ? (__bf16)(float)W[(int64_t)dd * ldw + k0 + t] : (__bf16)0.0f;
    }
    {
      const int lane = c % 64, cb = (c / 64) % CB2, s = (c / (64 * CB2)) % 8, w = c / (64 * CB2 * 8);
      const int kk = (kp / 8) * w + 16 * cb + (lane & 15), d0 = 32 * s + 8 * (lane >> 4);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        q2[(int64_t)c * 8 + t] = (d0 + t < d && kk < k) ? (__bf16)(float)W[(int64_t)(d0 + t) * ldw + kk] : (__bf16)0.0f;
    }
  }
}

// dst[r][c] (fp32, ldd) = src[r][c] (bf16, lds)   or the inverse
__global__ __launch_bounds__(256) void cvt_bf16_kernel(const void* __restrict__ src, int64_t lds_, void* __restrict__ dst,
                                                       int64_t ldd, int n, int k, int to_f32) {
  const int64_t total = (int64_t)n * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / k;
    const int c = (int)(i % k);
    if (to_f32) ((float*)dst)[r * ldd + c] = (float)((const __bf16*)src)[r * lds_ + c];
    else ((__bf16*)dst)[r * ldd + c] = (__bf16)((const float*)src)[r * lds_ + c];
  }
}

template <int K>
hipError_t grad_k(const BtParams& p, int grid, hipStream_t stream) {
  const size_t lds = (size_t)kRows * K * 2 + 256;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt16_grad_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt16_grad_kernel<K>, dim3(grid), dim3(kThreads), lds, stream, p);
  return hipGetLastError();
}

template <int K>
hipError_t trial_k(const BtParams& p, float lr, float lam, int force, int grid, hipStream_t stream) {
  const size_t lds = (size_t)kRows * K * 2 + 256;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bt16_trial_kernel<K>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(bt16_trial_kernel<K>, dim3(grid), dim3(kThreads), lds, stream, p, lr, lam, force);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_bt16_grad(const BtParams& p, int kpad, int grid, hipStream_t stream) {
  switch (kpad) {
    case 256: return grad_k<256>(p, grid, stream);
    case 512: return grad_k<512>(p, grid, stream);
    case 1024: return grad_k<1024>(p, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_bt16_trial(const BtParams& p, int kpad, int grid, float lr, float lam, int force,
                             hipStream_t stream) {
  switch (kpad) {
    case 256: return trial_k<256>(p, lr, lam, force, grid, stream);
    case 512: return trial_k<512>(p, lr, lam, force, grid, stream);
    case 1024: return trial_k<1024>(p, lr, lam, force, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pack_w_bf16(const void* W, int64_t ldw, int d, int k, int kp, int w_is_bf16, void* q1, void* q2,
                              hipStream_t stream) {
  const int total = kFistaD * kp / 8;
  if (w_is_bf16)
    hipLaunchKernelGGL(pack_w_bf16_kernel<__bf16>, dim3((total + 255) / 256), dim3(256), 0, stream, (const __bf16*)W,
                       ldw, d, k, kp, (__bf16*)q1, (__bf16*)q2);
  else
    hipLaunchKernelGGL(pack_w_bf16_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, stream, (const float*)W,
                       ldw, d, k, kp, (__bf16*)q1, (__bf16*)q2);
  return hipGetLastError();
}

hipError_t launch_cvt_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int n, int k, int to_f32,
                           hipStream_t stream) {
  const int64_t total = (int64_t)n * k;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0,
                     stream, src, ld_src, dst, ld_dst, n, k, to_f32);
  return hipGetLastError();
}

}  // namespace lasso
