// Internal kernel-launch declarations shared by the .hip translation units and
// the C-ABI layer (lasso_hip.cpp).  Not part of the public boundary
// (include/lasso_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace lasso {

constexpr int kTileM = 16;          // batch rows per workgroup (one MFMA M-block)
constexpr int kFistaWaves = 8;      // waves per workgroup (2 per SIMD)
constexpr int kFistaThreads = kFistaWaves * 64;
constexpr int kFistaD = 256;        // padded feature dim handled by the fused kernel
constexpr int kFistaMaxK = 1024;    // largest padded dictionary size (LDS bound)

struct FistaTileParams {
  const float* X;    int64_t ldx;       // [n][d]
  const float* Wp;                       // [kFistaD][Kpad]   zero padded W
  const float* Wtp;                      // [Kpad][kFistaD]   zero padded W^T
  const float* z_in; int64_t ldz_in;     // nullable -> zeros
  const float* y_in; int64_t ldy_in;     // nullable -> y = z_in
  float* z_out;      int64_t ldz_out;
  float* y_out;      int64_t ldy_out;    // nullable
  const float* coef;                     // [iters] momentum coefficients (device)
  float* partials;                       // [iters][ntiles] sum|z-z_next| per tile, nullable
  int n, d, k;                           // real sizes
  int ntiles, iters;
  float lr, lam;                         // step size, alpha*lr
};

size_t fista_tile_lds_bytes(int kpad);
hipError_t launch_fista_tile(const FistaTileParams& p, int kpad, int grid, hipStream_t stream);

}  // namespace lasso
