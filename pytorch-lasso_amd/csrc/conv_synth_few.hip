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
#include "lasso_kernels.h"

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
__global__ __launch_bounds__(kSfThreads) void conv_synth_few_kernel(const ConvSynthFew p) {
  constexpr int PITCH = 16 * NT + 1;           // odd: the gather's lanes walk consecutive pixels
  extern __shared__ __attribute__((aligned(16))) float sf_smem[];
  lds_f32* const cols = (lds_f32*)sf_smem;     // [kSfChunk][PITCH]
  lds_f32* const bl = cols + kSfChunk * PITCH; // BLDS: [4 KQ][NT][64]
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int ckk = g.C * g.kh * g.kw;

  // B fragments: bf[c][4 t + e] = W[k = 16 t + 4 q + e][tap = 16 c + l15]   (zero beyond K / the taps)
  float bf[BLDS ? 1 : NT][BLDS ? 1 : 4 * KQ];
#pragma unroll
  for (int c0 = 0; c0 < NT; ++c0) {
    const int c = c0;
    if (BLDS && (c0 % kSfWaves) != wid) continue;      // (LDS copy: the column blocks are shared out over the waves)
    const int tap = 16 * c + l15;
#pragma unroll
    for (int t = 0; t < KQ; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 16 * t + 4 * q + e;
        const float v = (tap < ckk && k < g.K) ? p.W[(int64_t)k * ckk + tap] : 0.0f;
        if constexpr (BLDS) bl[((4 * t + e) * NT + c) * 64 + lane] = v;
        else bf[c][4 * t + e] = v;
      }
  }
  if constexpr (BLDS) __syncthreads();

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
      __syncthreads();                                             // the previous chunk's gather is done
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) cols[(16 * wid + 4 * q + rg) * PITCH + 16 * c + l15] = cacc[c][rg];
      __syncthreads();
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
    // ---- the band is complete: subtract x, store ----
#pragma unroll
    for (int m = 0; m < kSfMaxOut; ++m) {
      if (oinfo[m] < 0) continue;
      const int v = oinfo[m] & 0xfff, u = u0 + ((oinfo[m] >> 12) & 0xfff), ch = oinfo[m] >> 24;
      if (u >= g.H) continue;
      const int64_t idx = (((int64_t)n * g.C + ch) * g.H + u) * g.W + v;
      p.R[idx] = acc[m] - (p.x ? p.x[idx] : 0.0f);
    }
    __syncthreads();                                               // cols is rewritten by the next item
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
                                 bool* done, hipStream_t stream) {
  *done = false;
  const int ckk = g.C * g.kh * g.kw;
  if (g.sh != 1 || g.sw != 1 || g.C >= 8 || g.K < 4 || (g.K & 3) || g.K > 128 || ckk > 128 || (((uintptr_t)Ym) & 15)) return hipSuccess;
  if ((int64_t)g.C * g.W > kSfMaxOut * kSfThreads || g.W >= 4096 || cus <= 0) return hipSuccess;   // (12-bit fields)
  if ((int64_t)g.N * g.Hz * g.Wz * g.K >= INT32_MAX) return hipSuccess;
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
  if (g.K <= 32) return few_launch_nt<2>(nt, p, cus, stream, done);
  if (g.K <= 64) return few_launch_nt<4>(nt, p, cus, stream, done);
  return few_launch_nt<8>(nt, p, cus, stream, done);
}

}  // namespace lasso
