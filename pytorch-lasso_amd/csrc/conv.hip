// Convolutional ISTA/FISTA (reference lasso/conv2d/ista.py:7-49) and
// the Toeplitz Lipschitz bound (lasso/conv2d/lip_const.py:96-135).
//
//   x_hat = conv_transpose2d(z, W)      (synthesis, ista.py:19)
//   g     = conv2d(x_hat - x, W)        (its adjoint,  ista.py:20)
//   z+    = S_{alpha lr}(y - lr g), momentum and global stop rule as in the linear solver.
//
// Layout: the code z [N][K][Hz][Wz] is kept as a MATRIX Zm [M = N*Hz*Wz][K] ("one row per
// code pixel") for the whole solve, so both convolutions become the dense MFMA GEMMs the
// rest of the library already has, against ONE weight matrix Wt [C*kh*kw][K]:
//   COLSt [CKK][M] = Wt Ym^T                    (gemm_nt_kernel: every code pixel's patch)
//   R     [N][C][H][W] = overlap-add(COLSt) - x (conv_residual_kernel, gather form: no atomics)
//   RC    [M][CKK]     = patches of R            (conv_patches_kernel, im2col, pixel-major)
//   G     [M][K]       = RC W^T                  (gemm_nt_kernel, contraction over CKK)
//   prox / momentum / sum|z - z+|                (generic_prox_kernel on the matrices)
// COLSt is stored tap-major so that the overlap-add gather of neighbouring pixels touches
// neighbouring addresses; RC is pixel-major (rows = contraction-contiguous GEMM operand).  Stride and padding live
// only in the index arithmetic of those two kernels.
// Rooflines: the GEMMs are MFMA-bound (2*2*M*CKK*K flop per iteration); the two
// data-movement kernels are HBM-bound (each reads or writes the M*CKK patch matrix once).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include "lasso_kernels.h"

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Zm[(n,u,v)][k] <- z[n][k][u][v]   (to_rows != 0)   or the inverse: per image a K x P
// matrix transpose through a 64 x 65 LDS tile (256-byte pieces on both sides); blockIdx.z = image.  dst2 (may be null)
// receives the same rows: the solver's y starts as a copy of z.
__global__ __launch_bounds__(256) void conv_relayout_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            float* __restrict__ dst2, int K, int P, int to_rows) {
  __shared__ float t[64][65];
  const int64_t img = (int64_t)blockIdx.z * K * P;
  // source matrix of this image: rows x cols, destination: cols x rows
  const int rows = to_rows ? K : P, cols = to_rows ? P : K;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < rows && c < cols) ? src[img + (int64_t)r * cols + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;        // destination row = c, column = r
    if (c < cols && r < rows) {
      const float v = t[tx][i];
      dst[img + (int64_t)c * rows + r] = v;
      if (dst2) dst2[img + (int64_t)c * rows + r] = v;
    }
  }
}

// R[n][c][i][j] = sum_{a,b} COLSt[(c,a,b)][(n,u,v)] - x[n][c][i][j],
// u = (i + ph - a)/sh, v = (j + pw - b)/sw where they are integers inside the code grid
__global__ __launch_bounds__(256) void conv_residual_kernel(const float* __restrict__ colst, const float* __restrict__ x,
                                                            float* __restrict__ r, const ConvGeom g) {
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % g.W);
    const int i = (int)((e / g.W) % g.H);
    const int c = (int)((e / ((int64_t)g.W * g.H)) % g.C);
    const int n = (int)(e / ((int64_t)g.W * g.H * g.C));
    float acc = 0.0f;
    if (g.sh == 1 && g.sw == 1) {               // the usual case: no divisions in the tap loops
      const int a_lo = max(0, i + g.ph - (g.Hz - 1)), a_hi = min(g.kh - 1, i + g.ph);
      const int b_lo = max(0, j + g.pw - (g.Wz - 1)), b_hi = min(g.kw - 1, j + g.pw);
      for (int a = a_lo; a <= a_hi; ++a) {
        const int u = i + g.ph - a;
        const float* row = colst + ((int64_t)n * g.Hz + u) * g.Wz + (j + g.pw);
        const int64_t tb = ((int64_t)c * g.kh + a) * g.kw;
        for (int b = b_lo; b <= b_hi; ++b) acc += row[(tb + b) * M - b];
      }
    } else {
      for (int a = 0; a < g.kh; ++a) {
        const int ii = i + g.ph - a;
        if (ii < 0 || ii % g.sh != 0) continue;
        const int u = ii / g.sh;
        if (u >= g.Hz) continue;
        for (int b = 0; b < g.kw; ++b) {
          const int jj = j + g.pw - b;
          if (jj < 0 || jj % g.sw != 0) continue;
          const int v = jj / g.sw;
          if (v >= g.Wz) continue;
          const int64_t t = ((int64_t)c * g.kh + a) * g.kw + b;
          acc += colst[t * M + ((int64_t)n * g.Hz + u) * g.Wz + v];
        }
      }
    }
    r[e] = acc - (x ? x[e] : 0.0f);
  }
}

// RC[(n,u,v)][(c,a,b)] = R[n][c][u*sh - ph + a][v*sw - pw + b]   (0 outside the image), row
// stride ldr (the tap count rounded up to a multiple of 4 so that the GEMM can stage it with
// 16-byte loads; the padding columns are zero)
__global__ __launch_bounds__(256) void conv_patches_kernel(const float* __restrict__ r, float* __restrict__ rc,
                                                           int ldr, const ConvGeom g) {
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  const int ckk = g.C * g.kh * g.kw;
  const int q4 = ldr / 4;                        // float4 groups per patch row
  const int64_t total = M * q4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / q4;
    int t = (int)(e - m * q4) * 4;
    const int v = (int)(m % g.Wz), u = (int)((m / g.Wz) % g.Hz);
    const int n = (int)(m / ((int64_t)g.Wz * g.Hz));
    int b = t % g.kw, a = (t / g.kw) % g.kh, c = t / (g.kw * g.kh);     // decoded once, then stepped
    const int i0 = u * g.sh - g.ph, j0 = v * g.sw - g.pw;
    f32x4 out;
#pragma unroll
    for (int s = 0; s < 4; ++s, ++t) {
      float val = 0.0f;
      const int i = i0 + a, j = j0 + b;
      if (t < ckk && i >= 0 && i < g.H && j >= 0 && j < g.W) val = r[(((int64_t)n * g.C + c) * g.H + i) * g.W + j];
      out[s] = val;
      if (++b == g.kw) { b = 0; if (++a == g.kh) { a = 0; ++c; } }
    }
    *(f32x4*)(rc + m * ldr + (e - m * q4) * 4) = out;
  }
}

// ---------------------------------------------------------------------------
// Fused gradient + proximal step (ista.py:20,29,42,44):  g = conv2d(R, W) as an IMPLICIT GEMM --
// the patch matrix RC [M][C kh kw] of conv_patches_kernel is never formed, the G matrix never
// written -- and z, y updated in the epilogue.
// One workgroup = 4 waves = a tile of 64 code pixels (TU x TV of one image) x 128 atoms (or 128 x 64 / 256 x 32
// for small dictionaries: the launcher picks the shape with the fewest tiles, see ConvGradProx below); it is
// persistent over pixel tiles (blockIdx.y = block of KW atoms), so that each wave keeps the B
// fragments of its 32 atoms -- W [K][C kh kw], contraction order (c, a, b) like the reference --
// in registers for the whole launch.  Per tile the receptive field of the 64 pixels,
// C x ((TU-1) sh + kh) x ((TV-1) sw + kw) residual values (zero outside the image), is staged in
// LDS once; the A operand of MFMA step s is read from it at  tap_offset[4 s + q] + pixel_offset
// (two small tables: stride and padding live only in those).  Epilogue: the 64 x 128 block of g
// goes through LDS so that z, y are read and written as 16-byte row-contiguous pieces; sum|z - z+|
// per workgroup in a fixed order (deterministic).
// ---------------------------------------------------------------------------
struct ConvGradProx {
  const float* R; const float* Wp; int ldr;
  float* Zm; float* Ym;
  float lr, lam, coef;
  float* dpart;
  ConvGeom g;
  int TU, TV, tiles_u, tiles_v, RH, RW;
  int tv_shift;                 // TV = 1 << tv_shift
  float inv_plane, inv_rw;      // 1 / (RH RW), 1 / RW: exact index splits of e < 2^14 without integer division
};
// Dictionaries of at most 64 atoms (KW = 64; round 4): a 128-atom tile left two of the four waves multiplying
// zeros and half of the epilogue's 16-byte pieces out of range, so the tile is turned to 128 pixels x 64 atoms --
// waves 2 (pixel halves) x 2 (atom halves), the same 4 x 2 MFMA blocks per wave, every z / y piece of the epilogue real.
// The contraction order per element is unchanged (codes bitwise those of the 128-atom tile).  K <= 32 goes one step
// further the same way: 256 pixels x 32 atoms, the four waves side by side along the pixels (KW = 32).
#ifndef LASSO_CGP_OCC
#define LASSO_CGP_OCC 3      // workgroups (4 waves) per SIMD-quad the kernel is compiled for: 3 -> 168 registers
#endif
#ifndef LASSO_CGP_OCC16
#define LASSO_CGP_OCC16 LASSO_CGP_OCC
#endif
constexpr int cgp_occ(int s4) { return s4 == 16 ? LASSO_CGP_OCC16 : LASSO_CGP_OCC; }
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global
// load / store in flight (vmcnt(0)), which would serialise the HBM phases with the MFMA phase
#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)

// The receptive field of a tile, C x RH x RW residual values (zero outside the image), into LDS: H independent
// loads in flight per thread.  No lane-dependent control flow: the loads go through a buffer descriptor of image n's
// residual with the offset out of range outside the image (reads 0), surplus lanes store to the spare word S[region].
// With loads and stores under lane masks hipcc lost count of what was in flight at the join and closed the tile's
// prologue with s_waitcnt vmcnt(0) -- which also waited for the z / y pieces requested behind the field on purpose.
template <int H>
__device__ __forceinline__ void cgp_stage_field(float* __restrict__ S, const __amdgpu_buffer_rsrc_t rrs, const ConvGradProx& p,
                                                int t0, int region, int plane, int i0, int j0) {
  const ConvGeom& g = p.g;
  for (int base = 0; base < region; base += 256 * H) {
    float sv[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int e = min(base + t0 + 256 * h, region - 1);
      // e = (c, rr, cc): floor((e + 1/2) / d) in fp32 is exact for e < 2^14 (the distance to the next integer
      // is at least 1/(2d), the rounding error below 2e-3/d) -- two runtime integer divisions per element were
      // a third of this kernel's instructions
      const int c = (int)(((float)e + 0.5f) * p.inv_plane), rem = e - c * plane;
      const int rr = (int)(((float)rem + 0.5f) * p.inv_rw), cc = rem - rr * p.RW;
      const int i = i0 + rr, j = j0 + cc;
      unsigned o = (i >= 0 && i < g.H && j >= 0 && j < g.W) ? (unsigned)((c * g.H + i) * g.W + j) * 4u : 0xfffffff0u;
      asm volatile("" : "+v"(o));
      sv[h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, o, 0, 0));
    }
#pragma unroll
    for (int h = 0; h < H; ++h) S[min(base + t0 + 256 * h, region)] = sv[h];
  }
}

template <int S4, int KW, bool SKIP>     // SKIP: C kh kw <= 4 S4 - 4, the last MFMA steps would multiply zeros
__global__ __launch_bounds__(256, cgp_occ(S4)) void conv_grad_prox_kernel(const ConvGradProx p) {
  constexpr int TP = 8192 / KW;                             // code pixels of a tile: 64 (KW = 128), 128 (KW = 64), 256 (KW = 32)
  constexpr int kCgpGtLd = KW + 4;
  constexpr int NWA = KW / 32;                              // waves side by side along the atoms
  constexpr int C4 = KW / 4, C4_SHIFT = KW == 128 ? 5 : KW == 64 ? 4 : 3;   // 16-byte pieces per tile row
  extern __shared__ __attribute__((aligned(16))) float cg_smem[];
  float* const Gt = cg_smem;                                // [TP][KW + 4]
  int* const toff = (int*)(Gt + TP * kCgpGtLd);             // [4 * S4]
  float* const S = (float*)(toff + 4 * S4);                 // [C][RH][RW] + one spare word
  __shared__ float red[256];
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int ckk = g.C * g.kh * g.kw, K = g.K;
  const int wa = w % NWA, wp = w / NWA;                     // this wave's 32 atoms / 64 pixels of the tile
  const int kcol0 = KW * blockIdx.y + 32 * wa;
  // B fragments of this wave's 32 atoms: B[k][e], e = 4 s + q, zero beyond ckk / K
  // (buffer loads, offset out of range where the fragment is zero, offsets opaque: under a condition -- also as a
  // select behind a clamped address -- hipcc put each of these 2 S4 loads in a branch of its own with an
  // s_waitcnt vmcnt(0) behind it: up to 96 L2 round trips in a row at the head of every launch; round 5)
  float bf[S4][2];
  {
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wp), 0, K * p.ldr * 4, 0x00020000);
#pragma unroll
    for (int s = 0; s < S4; ++s)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int k = kcol0 + 16 * nt + l15, e = 4 * s + q;
        unsigned o = (k < K && e < ckk) ? (unsigned)(k * p.ldr + e) * 4u : 0xfffffff0u;
        asm volatile("" : "+v"(o));
        bf[s][nt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, o, 0, 0));
      }
  }
  for (int e = tid; e < 4 * S4; e += 256) {
    int off = 0;
    if (e < ckk) {
      const int b = e % g.kw, a = (e / g.kw) % g.kh, c = e / (g.kw * g.kh);
      off = (c * p.RH + a) * p.RW + b;
    }
    toff[e] = off;
  }
  __syncthreads();
  // (this lane's tap offset of MFMA step s is read from the table where it is used: 36 registers that a third
  // resident workgroup needs more)
  int base_p[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int pix = 64 * wp + 16 * mt + l15, tu = pix >> p.tv_shift, tv = pix & (p.TV - 1);
    base_p[mt] = tu * g.sh * p.RW + tv * g.sw;
  }
  const int tiles_img = p.tiles_u * p.tiles_v, ntiles = g.N * tiles_img;
  const int region = g.C * p.RH * p.RW, plane = p.RH * p.RW;
  const bool kvec = (K & 3) == 0;
  // z, y rows through buffer descriptors: 32-bit offsets (the launcher checks M K 4 < 2^32), out-of-range pieces
  // read as zero and are not written
  const unsigned zbytes = (unsigned)((int64_t)g.N * g.Hz * g.Wz * K * 4);
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(p.Zm, 0, (int)zbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.Ym, 0, (int)zbytes, 0x00020000);
  float dsum = 0.0f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / tiles_img, tt = tile - n * tiles_img;
    const int u0 = (tt / p.tiles_v) * p.TU, v0 = (tt % p.tiles_v) * p.TV;
    const int i0 = u0 * g.sh - g.ph, j0 = v0 * g.sw - g.pw;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R) + (int64_t)n * g.C * g.H * g.W, 0,
                                                                         g.C * g.H * g.W * 4, 0x00020000);
    // (opaque per trip: otherwise hipcc hoists the index arithmetic of the unrolled staging and
    // epilogue loops out of the tile loop and spills it)
    int tdyn = tid;
    asm volatile("" : "+v"(tdyn));
    int bp[4] = {base_p[0], base_p[1], base_p[2], base_p[3]};     // (likewise: 4 * S4 operand addresses)
    asm volatile("" : "+v"(bp[0]), "+v"(bp[1]), "+v"(bp[2]), "+v"(bp[3]));
    // receptive fields of at most 512 values (one image channel, small kernels) take two loads per thread instead
    // of eight: the index arithmetic of the six idle slots was a quarter of the kernel's vector instructions there
    if (S4 <= 24 && region <= 512) cgp_stage_field<2>(S, rrs, p, tdyn, region, plane, i0, j0);
    else cgp_stage_field<8>(S, rrs, p, tdyn, region, plane, i0, j0);
    // z, y of the tile do not depend on g: fetched now, so that the HBM latency runs under the MFMAs
    f32x4 yo[8];                                              // (z is fetched in the epilogue: registers)
    unsigned zoff[8];                                         // byte offset of this thread's pieces (~0u: outside -> reads 0, writes dropped)
    if (kvec) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int idx = tdyn + 256 * h, pix = idx >> C4_SHIFT, c4 = (idx & (C4 - 1)) * 4;
        const int u = u0 + (pix >> p.tv_shift), v = v0 + (pix & (p.TV - 1));
        const int col = KW * blockIdx.y + c4;
        const bool ok = u < g.Hz && v < g.Wz && col < K;
        zoff[h] = ok ? (unsigned)(((n * g.Hz + u) * g.Wz + v) * K + col) * 4u : ~0u;
#ifdef LASSO_ABL_CONV_NOMEM    // timing ablation only (results invalid)
        yo[h] = (f32x4){0.f, 0.f, 0.f, (float)zoff[h]};
#else
        yo[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, zoff[h], 0, 0));
#endif
      }
    }
    LDS_BARRIER();
    f32x4 acc[4][2] = {};
#ifdef LASSO_ABL_CONV_NOMFMA   // timing ablation only (results invalid)
    if (p.lr < -1e30f)
#endif
#pragma unroll
    for (int s = 0; s < S4; ++s) {
      if (SKIP && 4 * s >= ckk) break;                                // padded steps multiply zeros (uniform branch): 13 of 16 real at 1 x 7 x 7
      const int off = toff[4 * s + q];
      float a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = S[off + bp[mt]];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bf[s][nt], acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Gt[(64 * wp + 16 * mt + 4 * q + rg) * kCgpGtLd + 32 * wa + 16 * nt + l15] = acc[mt][nt][rg];
    LDS_BARRIER();
    if (kvec) {
      f32x4 zo[8];
#pragma unroll
      for (int h = 0; h < 8; ++h)
#ifdef LASSO_ABL_CONV_NOMEM
        zo[h] = yo[h];
#else
        zo[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrsrc, zoff[h], 0, 0));
#endif
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int idx = tdyn + 256 * h, pix = idx >> C4_SHIFT, c4 = (idx & (C4 - 1)) * 4;
        const bool ok = zoff[h] != ~0u;
        const f32x4 gv = *(const f32x4*)(Gt + pix * kCgpGtLd + c4);
        f32x4 zn, yn;
        float ds = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = __fsub_rn(yo[h][e], __fmul_rn(p.lr, gv[e]));
          zn[e] = __fsub_rn(t, __builtin_amdgcn_fmed3f(t, -p.lam, p.lam));
          ds += __builtin_fabsf(__fsub_rn(zo[h][e], zn[e]));
          yn[e] = __fadd_rn(zn[e], __fmul_rn(p.coef, __fsub_rn(zn[e], zo[h][e])));
        }
#ifdef LASSO_ABL_CONV_NOMEM
        if (ok) dsum += ds + yn[0];
#else
        if (ok) {
          dsum += ds;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zn), zrsrc, zoff[h], 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yn), yrsrc, zoff[h], 0, 0);
        }
#endif
      }
    } else {
      for (int idx = tid; idx < TP * C4; idx += 256) {
        const int pix = idx >> C4_SHIFT, c4 = (idx & (C4 - 1)) * 4;
        const int u = u0 + (pix >> p.tv_shift), v = v0 + (pix & (p.TV - 1));
        const int col = KW * blockIdx.y + c4;
        if (u >= g.Hz || v >= g.Wz || col >= K) continue;
        const int64_t m = ((int64_t)n * g.Hz + u) * g.Wz + v;
        float* const zp = p.Zm + m * K + col;
        float* const yp = p.Ym + m * K + col;
        const f32x4 gv = *(const f32x4*)(Gt + pix * kCgpGtLd + c4);
        for (int e = 0; e < 4 && col + e < K; ++e) {
          const float zo = zp[e];
          const float t = __fsub_rn(yp[e], __fmul_rn(p.lr, gv[e]));
          const float zn = __fsub_rn(t, __builtin_amdgcn_fmed3f(t, -p.lam, p.lam));
          dsum += __builtin_fabsf(__fsub_rn(zo, zn));
          yp[e] = __fadd_rn(zn, __fmul_rn(p.coef, __fsub_rn(zn, zo)));
          zp[e] = zn;
        }
      }
    }
    LDS_BARRIER();
  }
  red[tid] = dsum;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) p.dpart[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// Wp[k][t] = w[k][t] with the row stride padded like RC (zeros in the padding)
__global__ __launch_bounds__(256) void conv_pad_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                         int ckk, int ldr) {
  const int total = K * ldr;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int k = e / ldr, t = e % ldr;
    wp[e] = t < ckk ? w[(int64_t)k * ckk + t] : 0.0f;
  }
}

// Wt[(c,a,b)][k] = w[k][c][a][b]
__global__ __launch_bounds__(256) void conv_pack_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int K,
                                                          int ckk) {
  const int total = K * ckk;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int k = e % K, t = e / K;
    wt[e] = w[(int64_t)k * ckk + t];
  }
}

// lip_const.py:96-135.  taps [O][I][T] (T = ksize^2; O <= I after the reference's swap).
// One workgroup per o: power(o, f) = sum_i (sum_t tap*cos(phase[t][f]))^2 + (... sin ...)^2,
// out[o] = max_f power.  phase[t][f] = w0[f]*h0[t] + w1[f]*h1[t] (fp32, like the reference).
__global__ __launch_bounds__(256) void conv_lip_kernel(const float* __restrict__ taps, int64_t so, int64_t si, int I,
                                                       int ks, int padding, const float* __restrict__ freq,
                                                       int sample, float* __restrict__ out_max) {
  extern __shared__ float sh_taps[];          // [I][T]
  __shared__ float sred[256];
  const int T = ks * ks, o = blockIdx.x;
  for (int e = threadIdx.x; e < I * T; e += 256) sh_taps[e] = taps[(int64_t)o * so + (int64_t)(e / T) * si + e % T];
  __syncthreads();
  float best = 0.0f;
  for (int f = threadIdx.x; f < sample * sample; f += 256) {
    const float w0 = freq[f / sample], w1 = freq[f % sample];
    float power = 0.0f;
    for (int i = 0; i < I; ++i) {
      float re = 0.0f, im = 0.0f;
      for (int t = 0; t < T; ++t) {
        const float h0 = 1.0f + (float)(padding - ks + t / ks), h1 = 1.0f + (float)(padding - ks + t % ks);
        const float a0 = w0 * h0, a1 = w1 * h1;
        const float ph = a0 + a1;
        float sn, cs;
        sincosf(ph, &sn, &cs);
        re = fmaf(sh_taps[i * T + t], cs, re);
        im = fmaf(sh_taps[i * T + t], sn, im);
      }
      power += re * re + im * im;
    }
    best = fmaxf(best, power);
  }
  sred[threadIdx.x] = best;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sred[threadIdx.x] = fmaxf(sred[threadIdx.x], sred[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out_max[o] = sred[0];
}

__global__ void conv_lip_sum_kernel(const float* __restrict__ maxes, int O, int take_sqrt, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.0f;
    for (int o = 0; o < O; ++o) s += maxes[o];
    out[0] = take_sqrt ? (double)sqrtf(s) : (double)s;
  }
}


// ---- patch front end (image -> patches -> centre -> dict_learning ->
// reconstruct; the reference's Omniglot notebook that did this is not in the checkout, so
// the layout follows torch.nn.functional.unfold: row = (n, u, v), column = (c, a, b)) -------
// One wave per patch row: copy the patch, optionally subtract its mean (kept in means[]).
__global__ __launch_bounds__(256) void patches_extract_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                              int64_t ld, float* __restrict__ means,
                                                              const ConvGeom g, int center) {
  const int lane = threadIdx.x & 63;
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  const int d = g.C * g.kh * g.kw;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    const int v = (int)(m % g.Wz), u = (int)((m / g.Wz) % g.Hz);
    const int n = (int)(m / ((int64_t)g.Wz * g.Hz));
    float sum = 0.0f;
    for (int t = lane; t < d; t += 64) {
      const int b = t % g.kw, a = (t / g.kw) % g.kh, c = t / (g.kw * g.kh);
      const float val = img[(((int64_t)n * g.C + c) * g.H + u * g.sh + a) * g.W + v * g.sw + b];
      out[m * ld + t] = val;
      sum += val;
    }
    if (center) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float mean = sum / (float)d;
      for (int t = lane; t < d; t += 64) out[m * ld + t] -= mean;
      if (means && lane == 0) means[m] = mean;
    }
  }
}

// img[n][c][i][j] = average over the patches covering the pixel of (patch value + its mean);
// pixels no patch covers are 0
__global__ __launch_bounds__(256) void patches_reconstruct_kernel(const float* __restrict__ pat, int64_t ld,
                                                                  const float* __restrict__ means,
                                                                  float* __restrict__ img, const ConvGeom g) {
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % g.W), i = (int)((e / g.W) % g.H);
    const int c = (int)((e / ((int64_t)g.W * g.H)) % g.C);
    const int n = (int)(e / ((int64_t)g.W * g.H * g.C));
    float acc = 0.0f;
    int cnt = 0;
    for (int a = 0; a < g.kh; ++a) {
      const int ii = i - a;
      if (ii < 0 || ii % g.sh != 0 || ii / g.sh >= g.Hz) continue;
      for (int b = 0; b < g.kw; ++b) {
        const int jj = j - b;
        if (jj < 0 || jj % g.sw != 0 || jj / g.sw >= g.Wz) continue;
        const int64_t m = ((int64_t)n * g.Hz + ii / g.sh) * g.Wz + jj / g.sw;
        acc += pat[m * ld + ((int64_t)c * g.kh + a) * g.kw + b] + (means ? means[m] : 0.0f);
        ++cnt;
      }
    }
    img[e] = cnt ? acc / (float)cnt : 0.0f;
  }
}

inline int grid_for(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, 8192); }

}  // namespace

hipError_t launch_conv_relayout(const float* src, float* dst, float* dst2, int N, int K, int P, int to_rows,
                                hipStream_t stream) {
  if ((int64_t)N * K * P == 0) return hipSuccess;
  const int rows = to_rows ? K : P, cols = to_rows ? P : K;
  for (int n0 = 0; n0 < N; n0 += 65535) {        // gridDim.z limit
    const int nb = std::min(N - n0, 65535);
    const int64_t o = (int64_t)n0 * K * P;
    hipLaunchKernelGGL(conv_relayout_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, nb), dim3(256), 0, stream,
                       src + o, dst + o, dst2 ? dst2 + o : nullptr, K, P, to_rows);
  }
  return hipGetLastError();
}

hipError_t launch_conv_pack_w(const float* w, float* wt, float* wp, int K, int ckk, int ldr, hipStream_t stream) {
  hipLaunchKernelGGL(conv_pack_w_kernel, dim3(grid_for((int64_t)K * ckk)), dim3(256), 0, stream, w, wt, K, ckk);
  hipLaunchKernelGGL(conv_pad_w_kernel, dim3(grid_for((int64_t)K * ldr)), dim3(256), 0, stream, w, wp, K, ckk, ldr);
  return hipGetLastError();
}

// R = conv_transpose2d(rows Ym) - x :  one implicit-GEMM kernel, or GEMM into COLSt, then the gather
// (w = the caller's weight [K][C][kh][kw]; null forces the explicit path)
hipError_t launch_conv_residual(const float* Ym, const float* Wt, const float* w, const float* x, float* colst, float* r,
                                const ConvGeom& g, int cus, hipStream_t stream) {
  if (w) {                                      // the implicit-GEMM kernel (conv_synth.hip) when the geometry is covered
    bool done = false;
    if (hipError_t e = launch_conv_synth(Ym, w, x, r, g, cus, &done, stream); e != hipSuccess) return e;
    if (done) return hipSuccess;
    if (hipError_t e = launch_conv_synth_few(Ym, w, x, r, g, cus, &done, stream); e != hipSuccess) return e;
    if (done) return hipSuccess;
  }
  const int ckk = g.C * g.kh * g.kw;
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  hipError_t e = launch_gemm_nt_sub(Wt, g.K, Ym, g.K, nullptr, 0, colst, M, ckk, (int)M, g.K, stream, /*add=*/1);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  hipLaunchKernelGGL(conv_residual_kernel, dim3(grid_for(total)), dim3(256), 0, stream, colst, x, r, g);
  return hipGetLastError();
}

// G [M][K] = conv2d(R, W) in row layout: pixel-major patches, then G = RC Wp^T on the general
// MFMA GEMM (contraction over the taps)
hipError_t launch_conv_gradient(const float* r, const float* Wp, float* rc, int ldr, float* G, const ConvGeom& g,
                                hipStream_t stream) {
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  hipLaunchKernelGGL(conv_patches_kernel, dim3(grid_for(M * (ldr / 4))), dim3(256), 0, stream, r, rc, ldr, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_gemm_nt_sub(rc, ldr, Wp, ldr, nullptr, 0, G, g.K, (int)M, g.K, ldr, stream, /*add=*/1);
}

// The fused implicit-GEMM gradient + prox step, when the geometry fits (C kh kw <= 192 and the
// receptive field of a 64-pixel tile <= 64 KiB); *count = number of dpart entries written, 0 when
// the caller has to take the explicit path.
hipError_t launch_conv_grad_prox(const float* r, const float* Wp, int ldr, float* Zm, float* Ym, float lr, float lam,
                                 float coef, float* dpart, int dpart_cap, const ConvGeom& g, int cus, int* count,
                                 hipStream_t stream, int dry) {
  *count = 0;
  const int ckk = g.C * g.kh * g.kw;
  if (ckk > 192) return hipSuccess;
  ConvGradProx p;
  // Tile of a workgroup: kw atoms x tp = 8192 / kw code pixels (TU x TV of one image).  Every shape costs the same
  // matrix-pipe time per tile, so the shape with the fewest tiles wins -- narrow atom tiles only for dictionaries
  // they hold whole (K <= 64 / K <= 32), a receptive field of at most 64 KiB; ties go to the wider atom tile (fewer
  // pixels: the smaller field to stage), then to the wider pixel row.
  int kw = 0, tp = 0;
  int64_t best_tiles = INT64_MAX;
  for (int kwc : {128, 64, 32}) {
    if (kwc < 128 && g.K > kwc) continue;
    const int tpc = 8192 / kwc, gyc = (g.K + kwc - 1) / kwc;
    for (int tv : {64, 32, 16, 8}) {
      const int tu = tpc / tv;
      if (tu < 1) continue;
      const int rh = (tu - 1) * g.sh + g.kh, rw = (tv - 1) * g.sw + g.kw;
      if ((int64_t)g.C * rh * rw > 16384) continue;
      const int64_t tiles = (int64_t)((g.Hz + tu - 1) / tu) * ((g.Wz + tv - 1) / tv) * gyc;
      if (tiles < best_tiles) { best_tiles = tiles; kw = kwc; tp = tpc; p.TV = tv; p.TU = tu; p.RH = rh; p.RW = rw; }
    }
  }
  if (kw == 0) return hipSuccess;
  p.tv_shift = p.TV == 64 ? 6 : p.TV == 32 ? 5 : p.TV == 16 ? 4 : 3;
  p.inv_plane = 1.0f / (float)(p.RH * p.RW);
  p.inv_rw = 1.0f / (float)p.RW;
  if ((int64_t)g.N * g.Hz * g.Wz * g.K * 4 >= ((int64_t)1 << 31) || (int64_t)g.C * g.H * g.W * 4 >= ((int64_t)1 << 31))
    return hipSuccess;                                                                // 32-bit offsets in the kernel
  p.tiles_u = (g.Hz + p.TU - 1) / p.TU;
  p.tiles_v = (g.Wz + p.TV - 1) / p.TV;
  p.R = r; p.Wp = Wp; p.ldr = ldr; p.Zm = Zm; p.Ym = Ym; p.lr = lr; p.lam = lam; p.coef = coef; p.dpart = dpart; p.g = g;
  const int64_t ntiles = (int64_t)g.N * p.tiles_u * p.tiles_v;
  const int gy = (g.K + kw - 1) / kw;
  const int s4 = ckk <= 64 ? 16 : ckk <= 96 ? 24 : ckk <= 144 ? 36 : 48;
  if (gy > dpart_cap || ntiles <= 0 || ntiles > INT32_MAX) return hipSuccess;
  const int gx = (int)std::min<int64_t>(ntiles, std::min(dpart_cap / gy, std::max(1, cgp_occ(s4) * cus / gy)));
  const size_t lds = (size_t)(tp * (kw + 4) + 4 * s4 + g.C * p.RH * p.RW + 1) * 4;
  if (dry) {                                     // (no launch: lasso_conv_ista_kernel_name asks whether the geometry is covered)
    *count = gx * gy;
    return hipSuccess;
  }
  const dim3 grid(gx, gy);
  const bool skip = ckk <= 4 * s4 - 4;
#define LASSO_CGP_CASE2(S4_, KW_, SK_)                                                                       \
  if (s4 == S4_ && kw == KW_ && skip == SK_) {                                                               \
    if (hipError_t e = ensure_dynamic_lds((const void*)&conv_grad_prox_kernel<S4_, KW_, SK_>, lds); e != hipSuccess) return e; \
    hipLaunchKernelGGL((conv_grad_prox_kernel<S4_, KW_, SK_>), grid, dim3(256), lds, stream, p);             \
  }
#define LASSO_CGP_CASE(S4_, KW_) LASSO_CGP_CASE2(S4_, KW_, false) LASSO_CGP_CASE2(S4_, KW_, true)
  LASSO_CGP_CASE(16, 128) LASSO_CGP_CASE(24, 128) LASSO_CGP_CASE(36, 128) LASSO_CGP_CASE(48, 128)
  LASSO_CGP_CASE(16, 64) LASSO_CGP_CASE(24, 64) LASSO_CGP_CASE(36, 64) LASSO_CGP_CASE(48, 64)
  LASSO_CGP_CASE(16, 32) LASSO_CGP_CASE(24, 32) LASSO_CGP_CASE(36, 32) LASSO_CGP_CASE(48, 32)
#undef LASSO_CGP_CASE
#undef LASSO_CGP_CASE2
  *count = gx * gy;
  return hipGetLastError();
}

hipError_t launch_patches_extract(const float* img, float* out, int64_t ld, float* means, const ConvGeom& g,
                                  int center, hipStream_t stream) {
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz;
  if (M == 0) return hipSuccess;
  hipLaunchKernelGGL(patches_extract_kernel, dim3((unsigned)std::min<int64_t>((M + 3) / 4, 8192)), dim3(256), 0,
                     stream, img, out, ld, means, g, center);
  return hipGetLastError();
}

hipError_t launch_patches_reconstruct(const float* pat, int64_t ld, const float* means, float* img,
                                      const ConvGeom& g, hipStream_t stream) {
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(patches_reconstruct_kernel, dim3(grid_for(total)), dim3(256), 0, stream, pat, ld, means, img, g);
  return hipGetLastError();
}

hipError_t launch_conv_lip(const float* taps, int O, int I, int64_t so, int64_t si, int ks, int padding,
                           const float* freq, int sample, int take_sqrt, float* maxes, double* out,
                           hipStream_t stream) {
  const size_t lds = (size_t)I * ks * ks * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv_lip_kernel, dim3(O), dim3(256), lds, stream, taps, so, si, I, ks, padding, freq, sample,
                     maxes);
  hipLaunchKernelGGL(conv_lip_sum_kernel, dim3(1), dim3(64), 0, stream, maxes, O, take_sqrt, out);
  return hipGetLastError();
}

}  // namespace lasso
