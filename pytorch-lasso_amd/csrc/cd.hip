// Greedy coordinate descent (reference lasso/linear/solvers/coordinate_descent.py:5-54,
// SURVEY.md 8f row f2).
//
// Per sample row the reference keeps the correlation vector b (= x W at the start, :19)
// and the tracked code z, and per step (:31-39) proposes S_alpha(b), commits the ONE
// coordinate j whose proposal moved furthest from z (torch.argmax: first index on ties)
// and corrects b += S[:, j] * (committed change), S = I - W^T W (:22-23).  A row leaves
// the active set once its committed change is <= tol*k (:45-48).  Rows never interact,
// so the reference's batched loop over a shrinking index set is n independent loops.
//
// Kernel: ONE WAVE PER ROW, b and z resident in VGPRs for the whole solve (K/64 floats
// each per lane, lane l owns columns 256c + 4l .. +3: every global access is a coalesced
// 1 KiB dwordx4 segment).  Per step: 4 VALU per element for |S_alpha(b) - z| and its lane
// maximum, one DPP wave reduction (max of |move| as integer bits), one v_cmp per element
// whose lane masks give the first column holding the maximum on the SCALAR unit, an indexed
// register read of the winner's slot, two readlanes, one coalesced read of row j of
// S (S is bitwise symmetric, so row j is column j) from L2, and K multiply-adds (mul and
// add rounded separately like the reference's `b + S*dz`).  No LDS, no barriers, no
// atomics; rows are dealt out statically (wave w: rows w, w + #waves, ...).  The step
// count per row and its active flag persist in the workspace, so a solve can be resumed (the verbose mode of the reference prints a loss per step).
//
// Roofline: not a GEMM and not HBM streaming -- the 4*K-byte row of S per row-step comes
// from L2/MALL (S is 4 MiB at K=1024) and the step is a dependent chain (argmax ->
// address -> load -> update), so the kernel is bound by VALU issue + L2 latency at the
// occupancy the 2*K/64 resident VGPRs allow.  Algorithmic bytes per row-step: 4*K (L2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"
#include "static_for.hpp"

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float shrink(float v, float lam) {
  return v - __builtin_amdgcn_fmed3f(v, -lam, lam);    // == softshrink bit for bit (lam >= 0)
}

// wave max of unsigned keys: DPP row_shr 1,2,4,8 leaves each 16-lane row's max in its last
// lane; the four row results are folded on the scalar unit.  Result is wave-uniform.
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
  const unsigned r0 = __builtin_amdgcn_readlane((int)x, 15), r1 = __builtin_amdgcn_readlane((int)x, 31);
  const unsigned r2 = __builtin_amdgcn_readlane((int)x, 47), r3 = __builtin_amdgcn_readlane((int)x, 63);
  return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
  constexpr int kBig = 0x7fffffff;
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(kBig, (int)x, 0x111, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(kBig, (int)x, 0x112, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(kBig, (int)x, 0x114, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(kBig, (int)x, 0x118, 0xf, 0xf, false));
  const unsigned r0 = __builtin_amdgcn_readlane((int)x, 15), r1 = __builtin_amdgcn_readlane((int)x, 31);
  const unsigned r2 = __builtin_amdgcn_readlane((int)x, 47), r3 = __builtin_amdgcn_readlane((int)x, 63);
  return min(min(r0, r1), min(r2, r3));
}

// The per-row state lives in ONE vector value per quantity (4*NC floats per lane, split in
// halves of <= 32 so that it stays a legal register tuple): element 4c+e of the lane is
// column 256c + 4*lane + e.  A vector -- unlike an array -- can be indexed with the
// wave-uniform winner slot at run time without leaving the register file (the compiler
// emits s_set_gpr_idx / v_movrel for it; arrays would be demoted to scratch memory).
template <int N> using fvec = float __attribute__((ext_vector_type(N)));

template <int NC>
__global__ __launch_bounds__(256) void cd_rows_kernel(const CdParams p) {
  constexpr int KP = 256 * NC;
  constexpr int H = NC > 8 ? NC / 8 : 1;       // register tuples per quantity
  constexpr int HN = 4 * NC / H;               // floats per tuple (<= 32)
  typedef fvec<HN> vec;
  const int lane = threadIdx.x & 63;
  const float alpha = p.alpha, tol = p.tol;
  // rows are dealt out statically, wave w takes rows w, w + #waves, ...: every branch in
  // this kernel is wave-uniform (the DPP/readlane reductions below need all 64 lanes)
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = (int)gridDim.x * 4;
  for (int row = wave; row < p.n; row += nwaves) {
    if (__builtin_amdgcn_readfirstlane(p.active[row]) == 0) continue;
    float* const bp = p.B + (int64_t)row * KP + 4 * lane;
    float* const zp = p.Zt + (int64_t)row * KP + 4 * lane;
    vec b[H], z[H];
    static_for<NC>([&](auto c) {
      constexpr int C = decltype(c)::value;
      const f32x4 tb = *(const f32x4*)(bp + 256 * C), tz = *(const f32x4*)(zp + 256 * C);
      static_for<4>([&](auto e_) {
        constexpr int I = 4 * C + decltype(e_)::value;
        b[I / HN][I % HN] = tb[decltype(e_)::value];
        z[I / HN][I % HN] = tz[decltype(e_)::value];
      });
    });
    int steps = 0, still = 1;
    while (steps < p.iters) {
      // pass 1  (:32-33): |S_alpha(b) - z| of every coordinate and the lane's maximum
      vec a[H];
      float best = 0.0f;
      static_for<4 * NC>([&](auto i) {
        constexpr int I = decltype(i)::value;
        a[I / HN][I % HN] = __builtin_fabsf(shrink(b[I / HN][I % HN], alpha) - z[I / HN][I % HN]);
        best = __builtin_fmaxf(best, a[I / HN][I % HN]);
      });
      // :34  argmax over the row.  Non-negative floats order like their bit patterns: wave
      // max on the integer DPP path; then the smallest column that attains it (torch.argmax
      // returns the first maximum) -- one v_cmp per coordinate, the rest is scalar work.
      const unsigned mb = wave_max_u32(__float_as_uint(best));
      const float mbf = __uint_as_float(mb);
      unsigned j = 0x7fffffffu;
      static_for<4 * NC>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(a[I / HN][I % HN] == mbf);
        const unsigned cand = m ? (unsigned)(256 * (I / 4) + (I % 4) + 4 * __builtin_ctzll(m)) : 0x7fffffffu;
        j = min(j, cand);
      });
      if (j == 0x7fffffffu) j = 0;        // only if a NaN got in: argmax of all-NaN
      // column j of S == row j (coalesced)
      const float* const sp = p.S + (int64_t)j * KP + 4 * lane;
      f32x4 s[NC];
      static_for<NC>([&](auto c) { s[c] = *(const f32x4*)(sp + 256 * c); });
      const int owner = (j & 255) >> 2;
      const int slot = ((j >> 8) << 2) | (j & 3);            // wave-uniform
      float bs, zs;
      if constexpr (H == 1) {
        bs = b[0][slot];
        zs = z[0][slot];
      } else {
        if (slot < HN) { bs = b[0][slot]; zs = z[0][slot]; }
        else { bs = b[1][slot - HN]; zs = z[1][slot - HN]; }
      }
      const float pr = shrink(bs, alpha);
      const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr), owner));
      const float dz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr - zs), owner));
      // :37  commit z_j (owner lane only)
      const float zn = lane == owner ? pz : zs;
      if constexpr (H == 1) {
        z[0][slot] = zn;
      } else {
        if (slot < HN) z[0][slot] = zn;
        else z[1][slot - HN] = zn;
      }
      // :36  b += S[:, j] * dz   (product and sum rounded separately, as ATen does)
      static_for<4 * NC>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const float t = s[I / 4][I % 4] * dz;
        b[I / HN][I % HN] = b[I / HN][I % HN] + t;
      });
      ++steps;
      if (!(mbf > tol)) { still = 0; break; }   // :46-48
    }
    static_for<NC>([&](auto c) {
      constexpr int C = decltype(c)::value;
      f32x4 tb, tz;
      static_for<4>([&](auto e_) {
        constexpr int I = 4 * C + decltype(e_)::value;
        tb[decltype(e_)::value] = b[I / HN][I % HN];
        tz[decltype(e_)::value] = z[I / HN][I % HN];
      });
      *(f32x4*)(bp + 256 * C) = tb;
      *(f32x4*)(zp + 256 * C) = tz;
    });
    // every lane writes the same values to the same two words: no divergent tail
    p.active[row] = still;
    p.row_steps[row] = p.row_steps[row] + steps;
  }
}

// z_out = S_alpha(b) (:52); optionally hands the tracked z back (the reference updates a
// caller-supplied z0 in place, :14,47)
__global__ __launch_bounds__(256) void cd_finish_kernel(const float* __restrict__ B, const float* __restrict__ Zt,
                                                        int kp, float* __restrict__ z_out, int64_t ldz,
                                                        float* __restrict__ zt_out, int64_t ldzt, int n,
                                                        int k, float alpha) {
  const int64_t total = (int64_t)n * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / k), c = (int)(i % k);
    if (z_out) z_out[(int64_t)r * ldz + c] = shrink(B[(int64_t)r * kp + c], alpha);
    if (zt_out) zt_out[(int64_t)r * ldzt + c] = Zt[(int64_t)r * kp + c];
  }
}

// tracked z <- z0 (zero padded to kp columns) or 0; every row active, no steps yet
__global__ __launch_bounds__(256) void cd_init_kernel(const float* __restrict__ z0, int64_t ldz0,
                                                      float* __restrict__ Zt, int kp, int n, int k,
                                                      int* __restrict__ active, int* __restrict__ row_steps) {
  const int64_t total = (int64_t)n * kp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / kp), c = (int)(i % kp);
    Zt[i] = (z0 && c < k) ? z0[(int64_t)r * ldz0 + c] : 0.0f;
    if (c == 0) { active[r] = 1; row_steps[r] = 0; }
  }
}

__global__ void cd_add_identity_kernel(float* __restrict__ S, int kp, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) S[(int64_t)i * kp + i] += 1.0f;                 // :23
}

// info[0] = rows still active, info[1] = max steps taken by any row
__global__ __launch_bounds__(256) void cd_info_kernel(const int* __restrict__ active,
                                                      const int* __restrict__ row_steps, int n,
                                                      int* __restrict__ info) {
  __shared__ int sa[256], sm[256];
  int a = 0, m = 0;
  for (int i = threadIdx.x; i < n; i += 256) { a += active[i]; m = max(m, row_steps[i]); }
  sa[threadIdx.x] = a; sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sa[threadIdx.x] += sa[threadIdx.x + s];
      sm[threadIdx.x] = max(sm[threadIdx.x], sm[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { info[0] = sa[0]; info[1] = sm[0]; }
}

}  // namespace

hipError_t launch_cd_init(const float* z0, int64_t ldz0, float* Zt, int kp, int n, int k, int* active,
                          int* row_steps, float* S, hipStream_t stream) {
  const int64_t total = (int64_t)n * kp;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
  if (n > 0)
    hipLaunchKernelGGL(cd_init_kernel, dim3(grid), dim3(256), 0, stream, z0, ldz0, Zt, kp, n, k, active,
                       row_steps);
  hipLaunchKernelGGL(cd_add_identity_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, S, kp, k);
  return hipGetLastError();
}

hipError_t launch_cd_rows(const CdParams& p, int kp, int cus, int* info, hipStream_t stream) {
  // 4 rows (waves) per workgroup; enough workgroups to fill every SIMD at the kernel's occupancy
  const int wgs = std::max(1, std::min((p.n + 3) / 4, cus * 8));
  switch (kp) {
    case 256: hipLaunchKernelGGL(cd_rows_kernel<1>, dim3(wgs), dim3(256), 0, stream, p); break;
    case 512: hipLaunchKernelGGL(cd_rows_kernel<2>, dim3(wgs), dim3(256), 0, stream, p); break;
    case 1024: hipLaunchKernelGGL(cd_rows_kernel<4>, dim3(wgs), dim3(256), 0, stream, p); break;
    case 2048: hipLaunchKernelGGL(cd_rows_kernel<8>, dim3(wgs), dim3(256), 0, stream, p); break;
    case 4096: hipLaunchKernelGGL(cd_rows_kernel<16>, dim3(wgs), dim3(256), 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  if (info) hipLaunchKernelGGL(cd_info_kernel, dim3(1), dim3(256), 0, stream, p.active, p.row_steps, p.n, info);
  return hipGetLastError();
}

hipError_t launch_cd_finish(const float* B, const float* Zt, int kp, float* z_out, int64_t ldz,
                            float* zt_out, int64_t ldzt, int n, int k, float alpha, hipStream_t stream) {
  const int64_t total = (int64_t)n * k;
  if (total == 0) return hipSuccess;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(cd_finish_kernel, dim3(grid), dim3(256), 0, stream, B, Zt, kp, z_out, ldz, zt_out, ldzt,
                     n, k, alpha);
  return hipGetLastError();
}

}  // namespace lasso
