// Synthesis side of the convolutional solver for images with FEW channels (C < 8: grey-scale and RGB patches,
// reference lasso/conv2d/ista.py:19, conv_transpose2d(z, W) - x), stride 1, as one kernel:
//   R[n][c][u][v] = sum_{a,b,k} Ym[(n, u+ph-a, v+pw-b)][k] W[k][c][a][b]  -  x[n][c][u][v]
// conv_synth.hip's view (columns = the channels, padded to one 16-wide MFMA block) wastes 13/16 of the matrix pipe
// at C = 3 and 15/16 at C = 1, and the explicit path writes the [C kh kw][M] matrix COLS^T to HBM and reads it back
// (34 MB per iteration at N=256, 1x32x32, 64 7x7 atoms; 79 MB at N=64, 3x64x64, 128 5x5 atoms).  Here:
//   GEMM view   rows = code pixels, columns = the C kh kw taps (NT blocks of 16), contraction over the K atoms --
//               COLS[pixel][tap] = sum_k Ym[pixel][k] W[k][tap]: 49 of 64 / 75 of 80 columns useful;
//   overlap-add the COLS block of 128 code pixels lives in LDS only; every thread owns up to 8 output pixels of the
//               workgroup's band of image rows, gathers their taps from the block (fixed order: code rows
//               ascending, then b ascending) and keeps the sums in registers until the band is complete.
// A workgroup (8 waves) owns (image n, band of RB image rows) and walks the code pixels that reach into the band
// -- rows u0+ph-kh+1 .. u0+RB-1+ph, a CONTIGUOUS range of Ym rows -- in chunks of 128: wave w takes 16 of them as
// one MFMA row block.  A operand: 16-byte global loads straight into the lanes' MFMA slots (atom order inside the
// contraction is permuted so that lane (row, q) holds atoms 16t + 4q .. +3 of its row; the B fragments use the same
// permutation), the next chunk's loads issued before the gather of the current one.  B operand: W for all taps in
// registers for the whole launch (NT K/4 values per lane; persistent over work items).
// Roofline: MFMA, 2 M' (16 NT) K flop with M' the code pixels including the bands' halo; HBM: Ym once (+ halo).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include "lasso_kernels.h"

// workgroup barrier that orders LDS traffic only (__syncthreads() also waits for every global operation in flight: the
// next chunk's A operands are requested before the barriers so that they arrive under the stores and the overlap-add)
#define SF_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

namespace lasso {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;

constexpr int kSfWaves = 8, kSfThreads = 64 * kSfWaves, kSfChunk = 16 * kSfWaves, kSfMaxOut = 8;

struct ConvSynthFew {
  const float* Ym;     // [N*Hz*Wz][K]
  const float* W;      // [K][C][kh][kw]
  const float* x;      // [N][C][H][W] or null
  float* R;            // [N][C][H][W]
  ConvGeom g;
  int rb, bands, items;   // image rows per band, bands per image, N * bands
};

// NT: 16-tap column blocks (C kh kw <= 16 NT); KQ: 16-atom groups (K <= 16 KQ); BLDS: the B fragments live in LDS
// (lane-linear: one conflict-free ds_read_b32 per MFMA) instead of NT * 4 KQ registers -- the combinations whose
// fragments do not fit beside the rest (16 NT KQ / 4 > ~100 of the 256 registers of two waves per SIMD)
template <int NT, int KQ, bool BLDS>
__global__ __launch_bounds__(kSfThreads, (NT * KQ <= 4 && !BLDS) ? 4 : 2) void conv_synth_few_kernel(const ConvSynthFew p) {
  constexpr int PITCH = 16 * NT + 1;           // odd: the gather's lanes walk consecutive pixels
  extern __shared__ __attribute__((aligned(16))) float sf_smem[];
  lds_f32* const cols = (lds_f32*)sf_smem;     // [kSfChunk][PITCH]
  lds_f32* const bl = cols + kSfChunk * PITCH; // BLDS: [4 KQ][NT][64]
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int ckk = g.C * g.kh * g.kw;

  // B fragments: bf[c][4 t + e] = W[k = 16 t + 4 q + e][tap = 16 c + l15]   (zero beyond K / the taps).  Buffer loads
  // whose offset is out of range where the fragment is zero (opaque offsets, no condition on the load or its use):
  // under a condition hipcc put every 4-byte load in a branch of its own with an s_waitcnt vmcnt(0) behind it -- 32 to
  // 64 L2 round trips in a row at the head of every launch (round 5: ~20 us of the 94 us launch at 3 x 5 x 5 taps, 128
  // atoms)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, g.K * ckk * 4, 0x00020000);
  auto w_at = [&](int k, int tap) {
    unsigned o = (tap < ckk && k < g.K) ? (unsigned)(k * ckk + tap) * 4u : 0xfffffff0u;
    asm volatile("" : "+v"(o));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, o, 0, 0));
  };
  float bf[BLDS ? 1 : NT][BLDS ? 1 : 4 * KQ];
  if constexpr (BLDS) {
    // the LDS copy [4 t + e][c][lane], one entry per thread and step
    constexpr int total = 4 * KQ * NT * 64, steps = (total + kSfThreads - 1) / kSfThreads;
    float tmp[steps];
#pragma unroll
    for (int j = 0; j < steps; ++j) {
      const int idx = min(tid + kSfThreads * j, total - 1);
      const int ln = idx & 63, rest = idx >> 6, c = rest % NT, te = rest / NT;
      tmp[j] = w_at(16 * (te >> 2) + 4 * (ln >> 4) + (te & 3), 16 * c + (ln & 15));
    }
#pragma unroll
    for (int j = 0; j < steps; ++j)
      if (tid + kSfThreads * j < total) bl[tid + kSfThreads * j] = tmp[j];
    __syncthreads();
  } else {
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int t = 0; t < KQ; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) bf[c][4 * t + e] = w_at(16 * t + 4 * q + e, 16 * c + l15);
  }

  const int outs = g.C * p.rb * g.W;           // outputs of a band (<= kSfMaxOut * kSfThreads)
  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int n = item / p.bands, u0 = (item - n * p.bands) * p.rb;
    const int i_lo = max(0, u0 + g.ph - (g.kh - 1)), i_hi = min(g.Hz - 1, u0 + p.rb - 1 + g.ph);
    const int npx = (i_hi - i_lo + 1) * g.Wz;                       // code pixels reaching into the band (may be <= 0)
    const int64_t pix0 = ((int64_t)n * g.Hz + i_lo) * g.Wz;        // ... a contiguous range of Ym rows
    // this thread's outputs: (v | row in band << 12 | channel << 24), -1 = none
    int oinfo[kSfMaxOut];
    float acc[kSfMaxOut];
#pragma unroll
    for (int m = 0; m < kSfMaxOut; ++m) {
      const int o = tid + kSfThreads * m;
      const int rest = o / g.W;
      oinfo[m] = o < outs ? ((o - rest * g.W) | ((rest % p.rb) << 12) | ((rest / p.rb) << 24)) : -1;
      acc[m] = 0.0f;
    }
    auto load_a = [&](int chunk, f32x4 (&av)[KQ]) {
      const int f = kSfChunk * chunk + 16 * wid + l15;
      const float* row = p.Ym + (pix0 + min(f, npx - 1)) * g.K;
#pragma unroll
      for (int t = 0; t < KQ; ++t) {
        const int k = 16 * t + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f < npx && k < g.K) v = *reinterpret_cast<const f32x4*>(row + k);     // (K % 4 == 0)
        av[t] = v;
      }
    };
    const int nchunks = npx > 0 ? (npx + kSfChunk - 1) / kSfChunk : 0;
    f32x4 av[KQ];
    if (nchunks > 0) load_a(0, av);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      // ---- COLS block of this chunk's 128 code pixels: wave w's 16 pixels x all taps ----
      f32x4 cacc[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) cacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KQ; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            float b;
            if constexpr (BLDS) b = bl[((4 * t + e) * NT + c) * 64 + lane];
            else b = bf[c][4 * t + e];
            cacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][e], b, cacc[c], 0, 0, 0);
          }
      if (chunk + 1 < nchunks) load_a(chunk + 1, av);              // in flight during the stores and the gather
      SF_LDS_BARRIER();                                             // the previous chunk's gather is done
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) cols[(16 * wid + 4 * q + rg) * PITCH + 16 * c + l15] = cacc[c][rg];
      SF_LDS_BARRIER();
      // ---- gather: code rows of this chunk, ascending; taps b ascending ----
      const int f_lo = kSfChunk * chunk, f_hi = min(f_lo + kSfChunk, npx) - 1;
      const int ia = i_lo + f_lo / g.Wz, ib = i_lo + f_hi / g.Wz, lim = f_hi - f_lo;
#pragma unroll
      for (int m = 0; m < kSfMaxOut; ++m) {
        if (oinfo[m] < 0) continue;
        const int v = oinfo[m] & 0xfff, u = u0 + ((oinfo[m] >> 12) & 0xfff), ch = oinfo[m] >> 24;
        const int jb = v + g.pw;                                   // code column j = jb - b
        float s = acc[m];
        for (int i = ia; i <= ib; ++i) {
          const int a = u + g.ph - i;
          if (a < 0 || a >= g.kh) continue;
          const int rowf = (i - i_lo) * g.Wz - f_lo;               // chunk-relative flat index of code pixel (i, 0)
          // taps with 0 <= j < Wz and the pixel inside this chunk: 0 <= rowf + j <= lim
          const int b_lo = max(0, max(jb - (g.Wz - 1), jb + rowf - lim));
          const int b_hi = min(g.kw - 1, min(jb, jb + rowf));
          const lds_f32* src = cols + (rowf + jb) * PITCH + (ch * g.kh + a) * g.kw;
          for (int b = b_lo; b <= b_hi; ++b) s += src[b - b * PITCH];
        }
        acc[m] = s;
      }
    }
    // ---- the band is complete: subtract x, store.  Through buffer descriptors of image n (an offset out of range where
    // the thread has no output: reads 0, store dropped), so that nothing sits under a branch: one load, its wait and
    // the store per output in a branch of its own were up to eight dependent HBM round trips per band, and hipcc
    // closed every such branch with an s_waitcnt vmcnt(0) that also waited for the previous store ----
    {
      const int chw = g.C * g.H * g.W;
      const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(p.R + (int64_t)n * chw, 0, chw * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x ? p.x : p.R) + (int64_t)n * chw, 0,
                                                                           p.x ? chw * 4 : 0, 0x00020000);
      // MCT = slots that can hold an output of this band (1, 2, 4 or 8): an out-of-range buffer operation still costs
      // its issue slot, and small images have one output per thread, not eight
      auto finish = [&](auto mc_tag) {
        constexpr int MCT = decltype(mc_tag)::value, B4 = MCT < 4 ? MCT : 4;
#pragma unroll
        for (int m0 = 0; m0 < MCT; m0 += B4) {        // (batches of four: eight live offsets and values more would cost the
          unsigned ooff[B4];                          // small instantiations their second workgroup per CU -- 128 registers)
          float xv[B4];
#pragma unroll
          for (int j = 0; j < B4; ++j) {
            const int oi = oinfo[m0 + j];
            const int v = oi & 0xfff, u = u0 + ((oi >> 12) & 0xfff), ch = oi >> 24;
            ooff[j] = (oi >= 0 && u < g.H) ? (unsigned)((ch * g.H + u) * g.W + v) * 4u : 0xfffffff0u;
            asm volatile("" : "+v"(ooff[j]));
            xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ooff[j], 0, 0));
          }
#pragma unroll
          for (int j = 0; j < B4; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[m0 + j] - xv[j]), rrs, ooff[j], 0, 0);
        }
      };
      const int mcount = (outs + kSfThreads - 1) / kSfThreads;
      if (mcount <= 1) finish(std::integral_constant<int, 1>{});
      else if (mcount <= 2) finish(std::integral_constant<int, 2>{});
      else if (mcount <= 4) finish(std::integral_constant<int, 4>{});
      else finish(std::integral_constant<int, kSfMaxOut>{});
    }
    SF_LDS_BARRIER();                                               // cols is rewritten by the next item
  }
}

template <int NT, int KQ>
hipError_t few_launch(const ConvSynthFew& p, int cus, hipStream_t stream) {
  constexpr bool BLDS = NT * KQ * 4 > 96;
  const size_t lds = (size_t)(kSfChunk * (16 * NT + 1) + (BLDS ? 4 * KQ * NT * 64 : 0)) * 4;
  const void* fn = reinterpret_cast<const void*>(&conv_synth_few_kernel<NT, KQ, BLDS>);
  if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  const int grid = std::min(p.items, (lds <= 80 * 1024 ? 2 : 1) * cus);
  hipLaunchKernelGGL((conv_synth_few_kernel<NT, KQ, BLDS>), dim3(grid), dim3(kSfThreads), lds, stream, p);
  return hipGetLastError();
}

template <int KQ>
hipError_t few_launch_nt(int nt, const ConvSynthFew& p, int cus, hipStream_t stream, bool* done) {
  *done = true;
  if (nt <= 1) return few_launch<1, KQ>(p, cus, stream);
  if (nt <= 2) return few_launch<2, KQ>(p, cus, stream);
  if (nt <= 4) return few_launch<4, KQ>(p, cus, stream);
  if (nt <= 5) return few_launch<5, KQ>(p, cus, stream);
  if (nt <= 8) return few_launch<8, KQ>(p, cus, stream);
  *done = false;
  return hipSuccess;
}

}  // namespace

// *done = false when the geometry is not covered (stride > 1, C >= 8: conv_synth.hip's range, K > 128 or not a
// multiple of 4, more than 128 taps, an image row too wide for eight outputs per thread): the caller goes on.
hipError_t launch_conv_synth_few(const float* Ym, const float* w, const float* x, float* r, const ConvGeom& g, int cus,
                                 bool* done, hipStream_t stream, int dry) {
  *done = false;
  const int ckk = g.C * g.kh * g.kw;
  if (g.sh != 1 || g.sw != 1 || g.C >= 8 || g.K < 4 || (g.K & 3) || g.K > 128 || ckk > 128 || (((uintptr_t)Ym) & 15)) return hipSuccess;
  if ((int64_t)g.C * g.W > kSfMaxOut * kSfThreads || g.W >= 4096 || cus <= 0) return hipSuccess;   // (12-bit fields)
  if ((int64_t)g.N * g.Hz * g.Wz * g.K >= INT32_MAX || (int64_t)g.C * g.H * g.W * 4 >= INT32_MAX ||
      (int64_t)g.K * g.C * g.kh * g.kw * 4 >= INT32_MAX) return hipSuccess;       // (32-bit buffer offsets inside an image / W)
  ConvSynthFew p;
  p.Ym = Ym; p.W = w; p.x = x; p.R = r; p.g = g;
  // band height: as tall as eight outputs per thread allow, but enough bands to fill the chip (a band re-reads
  // kh - 1 code rows of its neighbour: never below kh rows unless the image is that small)
  int rb = std::min(g.H, (kSfMaxOut * kSfThreads) / (g.C * g.W));
  const int want = (cus + g.N - 1) / g.N;                       // bands per image that give every CU a work item
  if (want > 1) rb = std::min(rb, std::max(std::min(g.kh, g.H), (g.H + want - 1) / want));
  p.rb = std::max(rb, 1);
  p.bands = (g.H + p.rb - 1) / p.rb;
  const int64_t items = (int64_t)g.N * p.bands;
  if (items <= 0 || items > INT32_MAX) return hipSuccess;
  p.items = (int)items;
  const int nt = (ckk + 15) / 16;
  if (dry) {                                       // (no launch: is the geometry covered?)
    *done = nt <= 8;
    return hipSuccess;
  }
  if (g.K <= 32) return few_launch_nt<2>(nt, p, cus, stream, done);
  if (g.K <= 64) return few_launch_nt<4>(nt, p, cus, stream, done);
  return few_launch_nt<8>(nt, p, cus, stream, done);
}

}  // namespace lasso
