// Whole iterations of the convolutional solver in ONE kernel for SMALL IMAGES WITH FEW CHANNELS (reference
// lasso/conv2d/ista.py:18-29,41-46): the synthesis x_hat = conv_transpose2d(y, W), the residual, its
// adjoint g = conv2d(x_hat - x, W), the proximal step, the momentum step and the iteration's sum |z - z+|, with ONE
// WORKGROUP PER IMAGE -- images are independent, so a workgroup carries its image through up to 64 iterations of a launch
// (the stop rule's per-iteration sums are written out per workgroup and added up afterwards: lasso_conv_ista_solve's
// speculate-and-replay scheme reads them once per chunk).  With fewer images than CUs (and K <= 64) an image may be cut
// into BANDS of code rows, one work item each (fused_plan: whichever of the two forms is cheaper): a band synthesises the kh - 1 code rows of halo on either side again (only
// while that is <= 60 % more synthesis), reads its neighbours' rows of the OLD y (so y goes from one buffer to another)
// and the launch boundary is the grid-wide barrier: one iteration per launch.
// What the two-kernel form (conv_synth_few.hip + conv.hip) pays for and this one does not: the residual never leaves the
// CU (it lives in LDS, zero-padded by the convolution's padding, and is the pixel operand of the gradient GEMM where it
// lies -- no receptive fields staged per tile), the gradient block never goes through LDS (the transposed product's
// accumulators ARE 16-byte row pieces), the overlap-add reads its taps with compile-time offsets from zero-padded rows (no
// bounds per tap), both W fragment tables are packed once per solve, one launch per <= 64 iterations instead of two per
// iteration.
//
//   phase A  (synthesis)  rows = code pixels, columns = the C kh kw taps, contraction over the K atoms (in two halves
//            when K > 64): COLS[pixel][tap] = sum_k Ym[pixel][k] W[k][tap] for a chunk of R whole code rows, each wave a
//            16-pixel MFMA row block at a time, the block stored to LDS at [row][v + kw - 1][tap] (kw - 1 zero columns
//            either side); then every thread adds the taps that reach its <= 16 output pixels, kept in registers across
//            chunks.
//   phase B  (gradient + prox)  rows = code pixels, columns = the K atoms, contraction over the taps: every wave takes
//            16-pixel blocks (x all atoms, or x a half / a quarter of them when K > 32) on its own (no barrier): the
//            pixel operand = one ds_read_b32 per MFMA step from the residual image, the atom operand = the W fragments in
//            registers; the product is issued TRANSPOSED (atoms x pixels), which leaves every lane with 16-byte row
//            pieces of g -- the epilogue works on the pieces of z, y fetched one block ahead.
//
// Bitwise the codes of the two-kernel form: the same lane -> atom / tap assignment in both GEMMs (every MFMA contracts the
// same four values in the same slots, steps in the same order; the transposed product swaps the operands, not the slots),
// the overlap-add in conv_synth_few_kernel's order -- code rows ascending, taps b ascending, and where that kernel's
// 128-pixel chunks (counted from the first code row of ITS band of image rows) cut a code row in two, the first part's
// taps before the second part's (the `split` case below) -- and the element-wise steps written with the same operations.
// Only the iteration's sum |z - z+| is added in another (fixed) order.
// Eligibility (launcher): stride 1, C < 8, K <= 128 a multiple of 4, C kh kw <= 80, kw <= 7, at most 8192 residual values
// per item (2, 8 or 16 per thread); whole images from a third of the CUs on, bands (K <= 64, halo <= 60 %) where they are the cheaper form.
// Roofline: HBM -- z, y read and written once, y read a second time by phase B (+ the bands' halo rows): 9.8 flop per
// byte at 1 x 7 x 7 taps and 64 atoms, below the fp32 MFMA ridge.
// Measured and not kept: the gradient blocks of the rows a chunk completes right behind that chunk,
// their z, y requested a chunk ahead (no separate memory-bound phase: 40.0 against 37.2 us per iteration).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "lasso_kernels.h"
#ifndef LASSO_CF_ABL
#define LASSO_CF_ABL 0      // timing ablations of tools/ab_conv_fused_phases.sh (results invalid when != 0)
#endif

#ifdef LASSO_CF_TIMING     // debug build (tools/conv_fused_timeline.py): wall-clock stamps of the first image's phases
__device__ unsigned long long lasso_cf_stamps[1024 * 64];
extern "C" int lasso_debug_cf_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lasso_cf_stamps), sizeof(lasso_cf_stamps));
}
#define CF_STAMP(slot) do { if (threadIdx.x == 0 && item == (int)blockIdx.x && it == p.iters - 1 && (slot) < 64) lasso_cf_stamps[blockIdx.x * 64 + (slot)] = wall_clock64(); } while (0)
#define CF_STAMP_WAVE(slot) do { if ((threadIdx.x & 63) == 0 && item == (int)blockIdx.x && it == p.iters - 1) lasso_cf_stamps[blockIdx.x * 64 + (slot)] = wall_clock64(); } while (0)
#else
#define CF_STAMP(slot) do { } while (0)
#define CF_STAMP_WAVE(slot) do { } while (0)
#endif

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load / store in flight
// (vmcnt(0)) -- the next chunk's A operands and the next gradient block's z, y pieces are requested early precisely so
// that they stay in flight across these barriers
#define CF_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr int kCfWaves = 8, kCfThreads = 64 * kCfWaves, kCfMaxOut = 16, kCfMaxKw = 7, kCfMaxIters = 64;
constexpr int kSynthFewOuts = 8 * 512;       // conv_synth_few_kernel's outputs per band (kSfMaxOut kSfThreads): its band height
constexpr unsigned kCfOor = 0xfffffff0u;       // buffer offset beyond every image: reads 0, stores dropped

struct ConvFused {
  const float* Wf1;    // synthesis B fragments [NT KQ][64 lanes][4]
  const float* Wf2;    // gradient B fragments  [4 NT][64 lanes][NTP]   (NTP = 4, or 8 when K > 64)
  const int* toff;     // [16 NT] tap offsets into the band's padded residual image
  const float* x;      // [N][C][H][W]
  float* Zm;
  const float* Yin; float* Yout;   // y read / written: the same buffer for whole images, two buffers when images are cut
                                   // into bands (a band's synthesis reads its neighbours' rows of the OLD y)
  float lr, lam;
  float* dpart;        // [iters][gridDim.x] sums |z - z+| of the workgroup's items, one row per iteration
  int iters;           // iterations of this launch (<= kCfMaxIters; 1 when images are cut into bands)
  float coef[kCfMaxIters];   // momentum factor (t_k - 1) / t_{k+1} of each iteration (ista.py:41-42); 0 = ISTA
  ConvGeom g;
  int R, WP, RHB, RW;  // code rows per chunk; COLS row width Wz + 2 (kw - 1); residual band BR + kh - 1 rows x W + 2 pw
  int BR, bands;       // code rows per band, bands per image (1: whole images)
  int old_rb;          // image rows per band of conv_synth_few_kernel's launch for this problem (its chunk origins)
  float inv_wz, inv_w, inv_old_rb;
};

// W[k][tap] -> the two fragment tables and the tap offsets (once per solve)
__global__ __launch_bounds__(256) void conv_fused_pack_kernel(const float* __restrict__ w, float* __restrict__ wf1,
                                                              float* __restrict__ wf2, int* __restrict__ toff,
                                                              const ConvGeom g, int NT, int KQ, int RHB, int RW) {
  const int ckk = g.C * g.kh * g.kw, ntp = KQ > 4 ? 8 : 4;
  const int n1 = NT * KQ * 256, n2 = 4 * NT * 64 * ntp, n3 = 16 * NT;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n1 + n2 + n3; idx += gridDim.x * 256) {
    if (idx < n1) {                    // bf1[c][4 t + e] of lane (l15, q): W[k = 16 t + 4 q + e][tap = 16 c + l15]
      const int e = idx & 3, lane = (idx >> 2) & 63, ct = idx >> 8, t = ct % KQ, c = ct / KQ;
      const int tap = 16 * c + (lane & 15), k = 16 * t + 4 * (lane >> 4) + e;
      wf1[idx] = (tap < ckk && k < g.K) ? w[(int64_t)k * ckk + tap] : 0.0f;
    } else if (idx < n1 + n2) {        // bf2[s][nt] of lane (l15, q): W[k = 16 nt + l15][tap = 4 s + q]
      const int i2 = idx - n1, nt = i2 % ntp, lane = (i2 / ntp) & 63, s = i2 / (ntp * 64);
      const int k = 16 * nt + (lane & 15), e = 4 * s + (lane >> 4);
      wf2[i2] = (nt < KQ && k < g.K && e < ckk) ? w[(int64_t)k * ckk + e] : 0.0f;
    } else {
      const int e = idx - n1 - n2;
      int off = 0;
      if (e < ckk) {
        const int b = e % g.kw, a = (e / g.kw) % g.kh, c = e / (g.kw * g.kh);
        off = (c * RHB + a) * RW + b;
      }
      toff[e] = off;
    }
  }
}

template <int NT, int KQ, int MC>      // MC: output pixels of the residual band per thread (C rows W <= 512 MC)
__global__ __launch_bounds__(kCfThreads) void conv_fused_kernel(const ConvFused p) {
  constexpr int PITCH = 16 * NT + 1, TS = PITCH - 1, S4 = 4 * NT;
  // K > 64: the synthesis contracts the atoms in two halves (its W fragments are re-read from LDS per block and half:
  // 160 registers otherwise).  K > 32: the gradient phase's waves split the atoms two or four ways (KQW = 2 groups of
  // 16 atoms per wave: the W fragments of all four / eight groups would be 64+ registers beside eight outputs' state)
  constexpr int KH = KQ > 4 ? 2 : 1, KQH = KQ / KH;
  constexpr int NTP = KQ > 4 ? 8 : 4, KQW = KQ >= 4 ? 2 : KQ, AG = KQ / KQW, WPG = kCfWaves / AG;
  extern __shared__ __attribute__((aligned(16))) float cf_smem[];
  lds_f32* const f1 = (lds_f32*)cf_smem;                     // [NT KQ][64][4]
  lds_f32* const f2 = f1 + NT * KQ * 256;                    // [S4][64][NTP]
  lds_i32* const tofl = (lds_i32*)(f2 + S4 * 64 * NTP);      // [4 S4]
  lds_f32* const rimg = (lds_f32*)(tofl + 4 * S4);           // [C][RHB][RW]
  const ConvGeom& g = p.g;
  const int rimg_words = (g.C * p.RHB * p.RW + 3) & ~3;
  lds_f32* const cols = rimg + rimg_words;                   // [R WP + kCfMaxKw][PITCH]
  __shared__ float wred[kCfMaxIters][kCfWaves];      // sums |z - z+| per iteration and wave (each wave adds to its own column)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int ckk = g.C * g.kh * g.kw, K = g.K, P = g.Hz * g.Wz, Wz = g.Wz;
  const int nsteps = (ckk + 3) >> 2;

  // ---- once per launch: fragment tables, tap offsets, zeros (the column padding of both LDS images is never written) ----
  for (int e = tid; e < NT * KQ * 64; e += kCfThreads)
    *(lds_f32x4*)(f1 + 4 * e) = *(const f32x4*)(p.Wf1 + 4 * e);
  for (int e = tid; e < S4 * 16 * NTP; e += kCfThreads)
    *(lds_f32x4*)(f2 + 4 * e) = *(const f32x4*)(p.Wf2 + 4 * e);
  for (int e = tid; e < 4 * S4; e += kCfThreads) tofl[e] = p.toff[e];
  {
    const int zero_words = rimg_words + (p.R * p.WP + kCfMaxKw) * PITCH;
    for (int e = tid; e < zero_words; e += kCfThreads) rimg[e] = 0.0f;
  }
  for (int e = tid; e < kCfMaxIters * kCfWaves; e += kCfThreads) (&wred[0][0])[e] = 0.0f;
  { const int item = blockIdx.x, it = 0; CF_STAMP(0); }
  __syncthreads();

  const int items = g.N * p.bands;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int n = item / p.bands, band = item - n * p.bands;
    // the band's code rows [c0, c1), the rows of the residual it needs (image rows [r0, r1): padded rows c0 .. c1 + kh - 2
    // minus the convolution's zero padding) and the code rows that reach into those, [s0, s1)
    const int c0 = band * p.BR, c1 = min(g.Hz, c0 + p.BR);
    const int r0 = max(0, c0 - g.ph), r1 = min(g.H, c1 + g.kh - 1 - g.ph), nrows = r1 - r0;
    const int s0 = max(0, c0 - g.kh + 1), s1 = min(g.Hz, c1 + g.kh - 1);
    const int outs = g.C * nrows * g.W, mcount = (outs + kCfThreads - 1) / kCfThreads;
    const int64_t img_words = (int64_t)P * K;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Yin) + (int64_t)n * img_words, 0, (int)(img_words * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yws = __builtin_amdgcn_make_buffer_rsrc(p.Yout + (int64_t)n * img_words, 0, (int)(img_words * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(p.Zm + (int64_t)n * img_words, 0, (int)(img_words * 4), 0x00020000);
    // this thread's outputs o = tid + 512 m = ((ch nrows + rr) W + v) of the band's residual rows:
    // padded row pr = r0 + rr + ph | (v + pw) << 12 | ch << 24,  -1 = none;  x at them (the same in every iteration)
    int oinfo[MC];
    float xr[MC];
    {
      const float inv_nr = 1.0f / (float)nrows;
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        const int o = tid + kCfThreads * m;
        // floor((o + 1/2) / d) in fp32 is exact for o < 2^14 (conv.hip, cgp_stage_field)
        const int rest = (int)(((float)o + 0.5f) * p.inv_w), v = o - rest * g.W;
        const int ch = (int)(((float)rest + 0.5f) * inv_nr), rr = rest - ch * nrows;
        const bool ok = m < mcount && o < outs;
        oinfo[m] = ok ? ((r0 + rr + g.ph) | ((v + g.pw) << 12) | (ch << 24)) : -1;
        xr[m] = 0.0f;
        if (ok) xr[m] = p.x[(((int64_t)n * g.C + ch) * g.H + r0 + rr) * g.W + v];
      }
    }
    if (p.bands > 1) {
      // rows of the band's LDS image that lie in the convolution's zero padding (first / last band): another band's
      // residual may be left there
      const int top = max(0, g.ph - c0), bot = r1 + g.ph - c0;                  // band rows [0, top) and [bot, RHB)
      for (int e = tid; e < g.C * p.RHB * p.RW; e += kCfThreads) {
        const int row = (e / p.RW) % p.RHB;
        if (row < top || row >= bot) rimg[e] = 0.0f;
      }
    }
   for (int it = 0; it < p.iters; ++it) {
    CF_STAMP(1);
    // ======================= phase A: residual rows of the band =======================
    float acc[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = 0.0f;
    float bf1[NT][4 * KQH];
    auto load_bf1 = [&](int h) {
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int t = 0; t < KQH; ++t) {
          const f32x4 v = *(const lds_f32x4*)(f1 + ((c * KQ + KQH * h + t) * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) bf1[c][4 * t + e] = v[e];
        }
    };
    if (KH == 1) load_bf1(0);
    // A operand of a 16-pixel row block: 16-byte pieces of the Ym rows straight into the lanes' MFMA slots
    auto load_a = [&](int i_c, int blk, f32x4 (&a)[KQ]) {
      const int npx = min(p.R, s1 - i_c) * Wz, f = 16 * blk + l15;
      const unsigned rowoff = (unsigned)((i_c * Wz + f) * K + 4 * q) * 4u;
#pragma unroll
      for (int t = 0; t < KQ; ++t) {
        const bool ok = f < npx && 16 * t + 4 * q < K;                       // (K % 4 == 0)
        unsigned o = ok ? rowoff + 64u * t : kCfOor;
        asm volatile("" : "+v"(o));         // (opaque: hipcc otherwise turns the select into a branch round a second load
                                            // of the same registers, with an s_waitcnt vmcnt(0) in front of it)
        a[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, o, 0, 0));
      }
    };
    f32x4 av[KQ];
    load_a(s0, wid, av);
    for (int i_c = s0; i_c < s1; i_c += p.R) {
      const int rows = min(p.R, s1 - i_c), npx = rows * Wz, nblk = (npx + 15) >> 4;
      // ---- COLS of the chunk's code rows: 16-pixel row blocks, wave w takes blocks w, w + 8, ... ----
      CF_STAMP(2 + 3 * ((i_c - s0) / p.R));
      for (int blk = wid; blk < nblk; blk += kCfWaves) {
        f32x4 an[KQ];
        load_a(i_c, blk + kCfWaves, an);                                       // (beyond the chunk: every piece out of range)
        __builtin_amdgcn_sched_barrier(0);                                     // ... issued BEFORE this block's MFMAs
        f32x4 cacc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) cacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < KH; ++h) {
          if (KH > 1) load_bf1(h);
#pragma unroll
          for (int t = 0; t < KQH; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < NT; ++c)
#if LASSO_CF_ABL & 1
                cacc[c][0] += av[KQH * h + t][e] * 1e-30f;
#else
                cacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[KQH * h + t][e], bf1[c][4 * t + e], cacc[c], 0, 0, 0);
#endif
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int fp = 16 * blk + 4 * q + rg;
          if (fp < npx) {
            const int ii = (int)(((float)fp + 0.5f) * p.inv_wz), v = fp - ii * Wz;
            lds_f32* const dst = cols + (ii * p.WP + v + g.kw - 1) * PITCH + l15;
#pragma unroll
            for (int c = 0; c < NT; ++c) dst[16 * c] = cacc[c][rg];
          }
        }
#pragma unroll
        for (int t = 0; t < KQ; ++t) av[t] = an[t];
      }
      load_a(i_c + p.R, wid, av);                                              // the next chunk's first block: in flight under the taps
      CF_LDS_BARRIER();
      CF_STAMP(3 + 3 * ((i_c - s0) / p.R));
      // ---- overlap-add: code rows ascending, taps b ascending (rr = kw - 1 - b descending) ----
      // KW = the kernel width when it is 3, 5 or 7 (no per-tap masks), else 0: widths up to 7 behind masks
      auto add_taps = [&](auto kw_tag) {
        constexpr int KW = decltype(kw_tag)::value;
        constexpr int NR = KW ? KW : kCfMaxKw;
#pragma unroll
        for (int m = 0; m < MC; ++m) {
          int oi = oinfo[m];
          asm volatile("" : "+v"(oi));      // (opaque per chunk: hipcc otherwise hoists every output's decoded fields and
                                            // tap addresses out of the chunk and iteration loops -- 150 registers at MC = 8)
          if (m >= mcount || oi < 0) continue;
          const int pr = oi & 0xfff, jb = (oi >> 12) & 0xfff, ch = oi >> 24;
          float s = acc[m];
          const int a0 = pr - i_c;                                             // tap row a = a0 - ii of chunk row ii
          // conv_synth_few_kernel walks the code pixels that reach into ITS band of image rows (old_rb of them; the
          // band of this output) in chunks of 128 from that band's first code row: a code row that holds a chunk
          // boundary strictly inside gives its first part's taps (pixels v < vs, the larger b) before the second part's
          const int ob = (int)(((float)(pr - g.ph) + 0.5f) * p.inv_old_rb), i_lo = max(0, ob * p.old_rb + g.ph - (g.kh - 1));
          for (int ii = 0; ii < rows; ++ii) {
            const int a = a0 - ii;
            if (a < 0 || a >= g.kh) continue;
            // tap (a, b) of this output comes from code pixel (i_c + ii, jb - b), stored at padded column jb - b + kw - 1:
            // b = kw - 1 - rr sits at lo + rr TS
            const lds_f32* const lo = cols + (ii * p.WP + jb) * PITCH + (ch * g.kh + a) * g.kw + g.kw - 1;
            float val[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) val[rr] = lo[rr * TS];             // (rr >= kw: inside the buffer, not used)
            const int f0 = (i_c + ii - i_lo) * Wz, fb = (f0 + Wz - 1) & ~127;
            const bool split = fb > f0;
            if (__builtin_amdgcn_ballot_w64(split) == 0) {
#pragma unroll
              for (int rr = NR - 1; rr >= 0; --rr)
                if (KW || rr < g.kw) s += val[rr];
            } else {
              const int r1s = split ? g.kw - 2 - jb + (fb - f0) : NR;          // first part: b >= jb - vs + 1  <=>  rr <= r1s
#pragma unroll
              for (int rr = NR - 1; rr >= 0; --rr)
                if (KW || rr < g.kw) s += rr <= r1s ? val[rr] : 0.0f;
#pragma unroll
              for (int rr = NR - 1; rr >= 0; --rr)
                if (KW || rr < g.kw) s += rr > r1s ? val[rr] : 0.0f;
            }
          }
          acc[m] = s;
        }
      };
#if LASSO_CF_ABL & 2
      if (p.lr < -1e30f)
#endif
      switch (g.kw) {
        case 3: add_taps(std::integral_constant<int, 3>{}); break;
        case 5: add_taps(std::integral_constant<int, 5>{}); break;
        case 7: add_taps(std::integral_constant<int, 7>{}); break;
        default: add_taps(std::integral_constant<int, 0>{}); break;
      }
      CF_LDS_BARRIER();
      CF_STAMP(4 + 3 * ((i_c - s0) / p.R));
    }
#pragma unroll
    for (int m = 0; m < MC; ++m) {
      int oi = oinfo[m];
      asm volatile("" : "+v"(oi));
      if (m >= mcount || oi < 0) continue;
      const int pr = oi & 0xfff, jb = (oi >> 12) & 0xfff, ch = oi >> 24;
      rimg[(ch * p.RHB + pr - c0) * p.RW + jb] = acc[m] - xr[m];
    }
    CF_LDS_BARRIER();
    CF_STAMP(26);

    // ======================= phase B: gradient, prox, momentum of the band's code rows =======================
    // A wave works on 16-pixel blocks x KQW 16-atom groups: all atoms of every eighth block, or (K > 32) a half / a
    // quarter of the atoms of every fourth / second block.  The product is issued TRANSPOSED (atoms x pixels: the W fragment is the A
    // operand), so lane (l15, q) ends up with pixel l15, atoms 16 nt + 4 q .. + 3 in the four registers of accumulator nt
    const int ag = wid / WPG, bfirst = wid % WPG;
    float bf2[S4][KQW];
    int to[S4];
#pragma unroll
    for (int s = 0; s < S4; ++s) {
#pragma unroll
      for (int h = 0; h < (KQW + 3) / 4; ++h) {
        const f32x4 v = *(const lds_f32x4*)(f2 + (s * 64 + lane) * NTP + (KQW * ag / 4 + h) * 4);
#pragma unroll
        for (int nt = 0; nt < KQW; ++nt)
          if (nt / 4 == h) bf2[s][nt] = KQW < 4 ? v[(KQW * ag) % 4 + nt] : v[nt % 4];
      }
      to[s] = tofl[4 * s + q];
    }
    const int bpx = (c1 - c0) * Wz, nb = (bpx + 15) >> 4;                      // the band's code pixels, 16-pixel blocks
    const float coef = p.coef[it];
    float dsum = 0.0f;
    auto fetch_zy = [&](int blk, unsigned (&off)[KQW], f32x4 (&yo)[KQW], f32x4 (&zo)[KQW]) {
      const int prow = 16 * blk + l15;
#pragma unroll
      for (int nt = 0; nt < KQW; ++nt) {
        const int col = 16 * (KQW * ag + nt) + 4 * q;
        off[nt] = (prow < bpx && col < K) ? (unsigned)((c0 * Wz + prow) * K + col) * 4u : kCfOor;
        asm volatile("" : "+v"(off[nt]));
#if LASSO_CF_ABL & 16
        yo[nt] = f32x4{(float)off[nt], 0.f, 1.f, 2.f}; zo[nt] = yo[nt];
#else
        yo[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, off[nt], 0, 0));
        zo[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, off[nt], 0, 0));
#endif
      }
    };
    unsigned off[KQW];
    f32x4 yo[KQW], zo[KQW];
    fetch_zy(bfirst, off, yo, zo);
    CF_STAMP(27);
    for (int blk = bfirst; blk < nb; blk += WPG) {
      CF_STAMP(32 + (blk - bfirst) / WPG);
      unsigned offn[KQW];
      f32x4 yn_[KQW], zn_[KQW];
      fetch_zy(blk + WPG, offn, yn_, zn_);                                     // the next block's pieces: in flight under the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      const int pl = min(16 * blk + l15, bpx - 1);
      const int pu = (int)(((float)pl + 0.5f) * p.inv_wz), pv = pl - pu * Wz;
      const int bp = pu * p.RW + pv;
      f32x4 acc2[KQW];
#pragma unroll
      for (int nt = 0; nt < KQW; ++nt) acc2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      float ar[S4];                                                            // the pixel operands of all steps in one batch of LDS reads
#pragma unroll
      for (int s = 0; s < S4; ++s) ar[s] = rimg[to[s] + bp];
      int ns = nsteps;                                                         // (opaque per block: sixteen loop-invariant
      asm volatile("" : "+s"(ns));                                             // step masks would live in spilled SGPR pairs)
#pragma unroll
      for (int s = 0; s < S4; ++s) {
        if (s < ns) {                                                          // (padded steps would multiply zeros; uniform)
#pragma unroll
#if LASSO_CF_ABL & 4
          for (int nt = 0; nt < KQW; ++nt) acc2[nt][s & 3] += ar[s];
#else
          for (int nt = 0; nt < KQW; ++nt) acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf2[s][nt], ar[s], acc2[nt], 0, 0, 0);
#endif
        }
      }
#pragma unroll
      for (int nt = 0; nt < KQW; ++nt) {
        const f32x4 gv = acc2[nt];
        f32x4 zn, yn;
        float ds = 0.0f;
#if LASSO_CF_ABL & 8
        zn = gv + yo[nt]; yn = zo[nt]; ds = gv[0];
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = __fsub_rn(yo[nt][e], __fmul_rn(p.lr, gv[e]));
          zn[e] = __fsub_rn(t, __builtin_amdgcn_fmed3f(t, -p.lam, p.lam));
          ds += __builtin_fabsf(__fsub_rn(zo[nt][e], zn[e]));
          yn[e] = __fadd_rn(zn[e], __fmul_rn(coef, __fsub_rn(zn[e], zo[nt][e])));
        }
#endif
        dsum += off[nt] != kCfOor ? ds : 0.0f;
#if LASSO_CF_ABL & 16
        dsum += zn[1] + yn[2];
#else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zn), zrs, off[nt], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yn), yws, off[nt], 0, 0);
#endif
      }
#pragma unroll
      for (int nt = 0; nt < KQW; ++nt) { off[nt] = offn[nt]; yo[nt] = yn_[nt]; zo[nt] = zn_[nt]; }
    }
    CF_STAMP(29);
    CF_STAMP_WAVE(48 + wid);
    // sum |z - z+| of this item and iteration: lanes by xor-shuffle, into the wave's own column (a fixed order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
    if (lane == 0) wred[it][wid] += dsum;
    // the next iteration's synthesis reads the y rows the other waves have just written: their stores complete
    // (vmcnt(0)) before the barrier, and the CU's vector L1 is shared by the workgroup's waves
    __syncthreads();
   }
  }
  { const int item = blockIdx.x, it = p.iters - 1; CF_STAMP(30); }
  if (tid < p.iters) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kCfWaves; ++w) s += wred[tid][w];
    p.dpart[tid * gridDim.x + blockIdx.x] = s;
  }
}

// delta[it] = sum of row it of the partial sums, in index order (deterministic); one workgroup per iteration
__global__ __launch_bounds__(256) void conv_fused_sums_kernel(const float* __restrict__ dpart, int count,
                                                              float* __restrict__ delta) {
  __shared__ float red[256];
  const float* row = dpart + (int64_t)blockIdx.x * count;
  float s = 0.0f;
  for (int e = threadIdx.x; e < count; e += 256) s += row[e];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) delta[blockIdx.x] = red[0];
}

struct FusedPlan { int NT, KQ, R, BR, bands, old_rb, outs_max; size_t lds; };

// The plan of ONE solve is computed ONCE: launch_conv_fused_pack() -- the first call of a solve -- plans
// afresh (environment switches included) and leaves the plan in this thread's slot; the iteration count, the y-buffer
// question and every launch of the same solve read it back instead of re-deriving it (a switch flipped in between can
// no longer make the pack and the launches disagree, and band mode no longer calls getenv once per iteration).
struct PlanSlot { bool valid; ConvGeom g; int cus; bool covered; FusedPlan pl; };
static thread_local PlanSlot tl_plan = {false, {}, 0, false, {}};
static bool same_geom(const ConvGeom& a, const ConvGeom& b) {
  return a.N == b.N && a.C == b.C && a.H == b.H && a.W == b.W && a.K == b.K && a.Hz == b.Hz && a.Wz == b.Wz &&
         a.kh == b.kh && a.kw == b.kw && a.sh == b.sh && a.sw == b.sw && a.ph == b.ph && a.pw == b.pw;
}
static bool fused_plan_fresh(const ConvGeom& g, int cus, FusedPlan* pl);
// false when the geometry is not covered
bool fused_plan(const ConvGeom& g, int cus, FusedPlan* pl, bool fresh = false) {
  if (!fresh && tl_plan.valid && tl_plan.cus == cus && same_geom(tl_plan.g, g)) {
    *pl = tl_plan.pl;
    return tl_plan.covered;
  }
  tl_plan.covered = fused_plan_fresh(g, cus, pl);
  tl_plan.valid = true; tl_plan.g = g; tl_plan.cus = cus; tl_plan.pl = *pl;
  return tl_plan.covered;
}

// LDS a workgroup may ask for on this device (the plan below budgets 150 KB of gfx950's 160 KB)
static size_t device_lds_limit() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
  return (size_t)std::max(v, 0);
}

static bool fused_plan_fresh(const ConvGeom& g, int cus, FusedPlan* pl) {
  if (const char* e = getenv("LASSO_CONV_FUSED"); e && e[0] == '0') return false;      // A/B and test switch: the two-kernel form
  const int ckk = g.C * g.kh * g.kw;
  if (g.sh != 1 || g.sw != 1 || g.C >= 8 || g.K < 4 || (g.K & 3) || g.K > 128 || ckk > 80 || g.kw > kCfMaxKw) return false;
  if (g.ph >= g.kh || g.pw >= g.kw || g.Wz > 128 || cus <= 0 || g.N <= 0 || (int64_t)g.C * g.W > kCfMaxOut * kCfThreads) return false;
  if ((int64_t)g.Hz * g.Wz * g.K * 4 >= ((int64_t)1 << 31) || (int64_t)g.Hz * g.Wz >= 16384 || g.H + g.ph >= 4096 || g.W + g.pw >= 4096) return false;
  pl->NT = (ckk + 15) / 16;
  pl->KQ = g.K <= 16 ? 1 : g.K <= 32 ? 2 : g.K <= 64 ? 4 : 8;
  // Whole images (a workgroup per image, up to 64 iterations per launch) or bands of code rows (more work items than
  // images; each band synthesises kh - 1 code rows of halo on either side again -- only while that is at most 60 % more
  // synthesis and K <= 64 --, one iteration per launch).  With an image for every CU: whole.  Below that the cheaper
  // of the two by rows of work per workgroup and launch round (synthesis + gradient rows; bands pay ~15 % for their
  // launch per iteration), whole images only from a third of the CUs on (measured, tools/ab_conv_small_batches.py:
  // N=255 ... 96 grey-scale 32 x 32 images 47 ... 38 us per iteration against 80 ... 58 of the two-kernel form, a tie
  // at N=64 where 3 x 32 x 32 images lose, 43 against 34; 3 x 32 x 32: N=200 whole 46 / bands 61, N=128 whole 44 / bands 37).
  pl->bands = 1;
  pl->BR = g.Hz;
  if (g.N > cus) {
    // more images than CUs: rounds of a workgroup per image; a thin last round on LARGE images costs more than the
    // two-kernel form's finer tiles (N=300 2x60x60 images, 32 3x3 atoms: 249 against 220 us per iteration; with 20x20
    // code grids 31 against 60)
    const int64_t rounds = (g.N + cus - 1) / cus;
    if ((double)g.N < 0.7 * (double)(rounds * cus) && (double)g.Hz * g.Wz * g.K * ckk > 1e6) return false;
  }
  if (g.N < cus) {
    const bool whole_ok = (int64_t)g.C * g.H * g.W <= kCfMaxOut * kCfThreads && 3 * (int64_t)g.N >= cus;
    const int want = (cus + g.N - 1) / g.N;
    const int br = (g.Hz + want - 1) / want, nb = (g.Hz + br - 1) / br;
    bool bands_ok = nb >= 2 && 10 * (br + 2 * (g.kh - 1)) <= 16 * br && pl->KQ <= 4 &&
                    (int64_t)g.C * std::min(g.H, br + g.kh - 1) * g.W <= kCfMaxOut * kCfThreads;
    // measured (profiles/r05_conv/ab_conv_fused.txt): N=64 3x64x64 images in four bands, 64 atoms 105 against 123 us per
    // iteration of the two-kernel form, 128 atoms 224 against 211 -- the synthesis in two halves does not carry the halo
    if (const char* e = getenv("LASSO_CONV_FUSED_BANDS"); e && e[0] == '0') bands_ok = false;
    const double cost_whole = whole_ok ? 2.0 * g.Hz : 1e300;
    const double cost_bands = bands_ok ? 1.15 * (2.0 * br + 2.0 * (g.kh - 1)) * (double)(((int64_t)g.N * nb + cus - 1) / cus) : 1e300;
    if (cost_whole >= 1e300 && cost_bands >= 1e300) return false;
    if (cost_bands < cost_whole) { pl->BR = br; pl->bands = nb; }
  }
  const int rhb = pl->BR + g.kh - 1;
  pl->outs_max = g.C * std::min(g.H, rhb) * g.W;
  if (pl->outs_max > kCfMaxOut * kCfThreads) return false;
  // conv_synth_few_kernel's bands of image rows for this problem (launch_conv_synth_few): the order of its overlap-add
  {
    int rb = std::min(g.H, kSynthFewOuts / (g.C * g.W));
    const int want = (cus + g.N - 1) / g.N;
    if (want > 1) rb = std::min(rb, std::max(std::min(g.kh, g.H), (g.H + want - 1) / want));
    pl->old_rb = std::max(rb, 1);
  }
  const int pitch = 16 * pl->NT + 1, wp = g.Wz + 2 * (g.kw - 1), ntp = pl->KQ > 4 ? 8 : 4;
  const size_t fixed = (size_t)(pl->NT * pl->KQ * 256 + 4 * pl->NT * 64 * ntp + 16 * pl->NT +
                                ((g.C * rhb * (g.W + 2 * g.pw) + 3) & ~3) + kCfMaxKw * pitch) * 4;
  const size_t budget = 150 * 1024;
  // a part with less LDS per workgroup, or with more CUs than the partial-sum buffer of a 64-iteration launch holds
  // (kConvDpart words: lasso_hip.hip's kConvDpart): the two-kernel form, not a failed launch
  if (device_lds_limit() < budget) return false;
  if ((int64_t)std::min<int64_t>((int64_t)g.N * pl->bands, cus) * (pl->bands > 1 ? 1 : kCfMaxIters) > kConvDpart) return false;
  if (fixed + (size_t)wp * pitch * 4 > budget) return false;
  const int srows = std::min(g.Hz, pl->BR + (pl->bands > 1 ? 2 * (g.kh - 1) : 0));      // code rows a band synthesises
  const int rmax = (int)std::min<size_t>((budget - fixed) / ((size_t)wp * pitch * 4), (size_t)srows);
  // rows per chunk: the fewest rounds of eight 16-pixel blocks over the band, then the fewest chunks
  int best = 0;
  int64_t best_rounds = INT64_MAX;
  for (int r = rmax; r >= 1; --r) {
    int64_t rounds = 0;
    for (int i = 0; i < srows; i += r) rounds += ((std::min(r, srows - i) * g.Wz + 15) / 16 + kCfWaves - 1) / kCfWaves;
    if (rounds < best_rounds) { best_rounds = rounds; best = r; }
  }
  pl->R = best;
  pl->lds = fixed + (size_t)best * wp * pitch * 4;
  return true;
}

template <int NT, int KQ, int MC>
hipError_t fused_launch(const ConvFused& p, int grid, size_t lds, hipStream_t stream) {
  const void* fn = reinterpret_cast<const void*>(&conv_fused_kernel<NT, KQ, MC>);
  if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_fused_kernel<NT, KQ, MC>), dim3(grid), dim3(kCfThreads), lds, stream, p);
  return hipGetLastError();
}

template <int NT, int KQ>
hipError_t fused_launch_mc(int outs, const ConvFused& p, int grid, size_t lds, hipStream_t stream) {
  if (outs <= 2 * kCfThreads) return fused_launch<NT, KQ, 2>(p, grid, lds, stream);
  if (outs <= 8 * kCfThreads) return fused_launch<NT, KQ, 8>(p, grid, lds, stream);
  return fused_launch<NT, KQ, kCfMaxOut>(p, grid, lds, stream);
}

template <int NT>
hipError_t fused_launch_kq(int kq, int outs, const ConvFused& p, int grid, size_t lds, hipStream_t stream) {
  if (kq == 1) return fused_launch_mc<NT, 1>(outs, p, grid, lds, stream);
  if (kq == 2) return fused_launch_mc<NT, 2>(outs, p, grid, lds, stream);
  if (kq == 4) return fused_launch_mc<NT, 4>(outs, p, grid, lds, stream);
  return fused_launch_mc<NT, 8>(outs, p, grid, lds, stream);
}

}  // namespace

size_t conv_fused_table_bytes() { return (size_t)(5 * 8 * 256 + 4 * 5 * 64 * 8 + 16 * 5) * 4; }

// The fragment tables of the fused kernel into `tables` (conv_fused_table_bytes()); *covered = false -> the geometry
// takes the two-kernel form and nothing is written.
hipError_t launch_conv_fused_pack(const float* w, void* tables, const ConvGeom& g, int cus, bool* covered,
                                  hipStream_t stream) {
  FusedPlan pl;
  *covered = fused_plan(g, cus, &pl, /*fresh=*/true);          // the solve's plan: every later call reads it back
  if (!*covered) return hipSuccess;
  const int ntp = pl.KQ > 4 ? 8 : 4;
  float* wf1 = (float*)tables;
  float* wf2 = wf1 + pl.NT * pl.KQ * 256;
  int* toff = (int*)(wf2 + 4 * pl.NT * 64 * ntp);
  const int total = pl.NT * pl.KQ * 256 + 4 * pl.NT * 64 * ntp + 16 * pl.NT;
  hipLaunchKernelGGL(conv_fused_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, wf1, wf2, toff, g,
                     pl.NT, pl.KQ, pl.BR + g.kh - 1, g.W + 2 * g.pw);
  return hipGetLastError();
}

// iterations one launch may take: 64 for whole images; 1 when images are cut into bands (a band's synthesis needs its
// neighbours' rows of the previous iteration: the launch boundary is the grid-wide barrier)
int conv_fused_max_iters(const ConvGeom& g, int cus) {
  FusedPlan pl;
  if (!fused_plan(g, cus, &pl)) return 0;
  return pl.bands > 1 ? 1 : kCfMaxIters;
}

// != 0: y is read from Yin and written to Yout, two DIFFERENT buffers (bands); 0: in place
int conv_fused_two_y_buffers(const ConvGeom& g, int cus) {
  FusedPlan pl;
  return fused_plan(g, cus, &pl) && pl.bands > 1;
}

// the instantiation launch_conv_fused would run for this geometry (the name rocprofv3 reports), or null
const char* conv_fused_kernel_name(const ConvGeom& g, int cus) {
  FusedPlan pl;
  if (!fused_plan(g, cus, &pl, /*fresh=*/true)) return nullptr;
  static thread_local char name[64];
  snprintf(name, sizeof(name), "lasso::conv_fused_kernel<%d, %d, %d>", pl.NT, pl.KQ, pl.outs_max <= 2 * kCfThreads ? 2 : pl.outs_max <= 8 * kCfThreads ? 8 : kCfMaxOut);
  return name;
}

// `iters` <= conv_fused_max_iters() iterations (ista.py:19-20,29,41-46) in one launch, iteration i with the momentum factor
// coefs[i]; delta_out (device, may be null) receives the sum |z - z+| of each.  dpart: iters x min(items, cus) words.
// Yin / Yout: see conv_fused_two_y_buffers.
hipError_t launch_conv_fused(const void* tables, const float* x, float* Zm, const float* Yin, float* Yout, float lr,
                             float lam, const float* coefs, int iters, float* dpart, int dpart_cap, float* delta_out,
                             const ConvGeom& g, int cus, hipStream_t stream) {
  FusedPlan pl;
  if (iters < 1 || iters > kCfMaxIters || !fused_plan(g, cus, &pl)) return hipErrorInvalidValue;
  if (pl.bands > 1 && (iters != 1 || Yin == Yout)) return hipErrorInvalidValue;
  const int ntp = pl.KQ > 4 ? 8 : 4;
  ConvFused p;
  p.Wf1 = (const float*)tables;
  p.Wf2 = p.Wf1 + pl.NT * pl.KQ * 256;
  p.toff = (const int*)(p.Wf2 + 4 * pl.NT * 64 * ntp);
  p.x = x; p.Zm = Zm; p.Yin = Yin; p.Yout = Yout; p.lr = lr; p.lam = lam; p.dpart = dpart; p.g = g;
  p.iters = iters;
  for (int i = 0; i < kCfMaxIters; ++i) p.coef[i] = i < iters ? coefs[i] : 0.0f;
  p.R = pl.R; p.WP = g.Wz + 2 * (g.kw - 1); p.RHB = pl.BR + g.kh - 1; p.RW = g.W + 2 * g.pw;
  p.BR = pl.BR; p.bands = pl.bands; p.old_rb = pl.old_rb;
  p.inv_wz = 1.0f / (float)g.Wz; p.inv_w = 1.0f / (float)g.W; p.inv_old_rb = 1.0f / (float)pl.old_rb;
  const int64_t items = (int64_t)g.N * pl.bands;
  const int grid = (int)std::min<int64_t>(items, cus);
  if (grid <= 0 || (int64_t)grid * iters > dpart_cap) return hipErrorInvalidValue;
  hipError_t e = hipErrorInvalidValue;
  switch (pl.NT) {
    case 1: e = fused_launch_kq<1>(pl.KQ, pl.outs_max, p, grid, pl.lds, stream); break;
    case 2: e = fused_launch_kq<2>(pl.KQ, pl.outs_max, p, grid, pl.lds, stream); break;
    case 3: e = fused_launch_kq<3>(pl.KQ, pl.outs_max, p, grid, pl.lds, stream); break;
    case 4: e = fused_launch_kq<4>(pl.KQ, pl.outs_max, p, grid, pl.lds, stream); break;
    case 5: e = fused_launch_kq<5>(pl.KQ, pl.outs_max, p, grid, pl.lds, stream); break;
  }
  if (e != hipSuccess || !delta_out) return e;
  hipLaunchKernelGGL(conv_fused_sums_kernel, dim3(iters), dim3(256), 0, stream, dpart, grid, delta_out);
  return hipGetLastError();
}

}  // namespace lasso
