// Instantiations of the fused FISTA tile kernel (fista_tile_sp_kernel.hpp) that leave out the all-padding feature chunks
// of GEMM-2's contraction (template parameter DS = ceil(d / 32) < D / 32): rows with fewer features than the tile's
// padded width.  Fixed-iteration launches only (the in-kernel stop rule keeps the full-width instantiations); a shape
// without an instantiation here runs the full-width kernel.  A separate translation unit so that it builds beside
// fista_tile_sp.hip.
#include "fista_tile_sp_kernel.hpp"

namespace lasso {

namespace {
template <int K, int M, int NW, int DS>
hipError_t go(const FistaTileParams& p, int grid, hipStream_t stream) {
  // GEMM-2 runs (K / D) passes x DS chunks = steps in pairs (two ring slots): an odd product has no instantiation
  constexpr int D = 512 * NW / M, S2 = (K / D) * DS;
  if constexpr (S2 % 2 == 0 && S2 >= 4) return sp::launch_ks<K, M, false, NW, DS>(p, grid, stream);
  else return hipErrorInvalidValue;
}
// K = every padded dictionary size of the tile geometry (M, NW); DS by switch
template <int M, int NW, int DS>
hipError_t by_k(const FistaTileParams& p, int kpad, int grid, hipStream_t stream) {
  constexpr int D = 512 * NW / M;
  switch (kpad) {
    case 256: return go<256, M, NW, DS>(p, grid, stream);
    case 384: if constexpr (D == 128) return go<384, M, NW, DS>(p, grid, stream); else return hipErrorInvalidValue;
    case 512: return go<512, M, NW, DS>(p, grid, stream);
    case 768: if constexpr (D == 256 || (D == 128 && NW == 4)) return go<768, M, NW, DS>(p, grid, stream); else return hipErrorInvalidValue;
    case 1024: if constexpr (D == 256 || (D == 128 && NW == 4)) return go<1024, M, NW, DS>(p, grid, stream); else return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

// dsteps = ceil(d / 32); hipErrorInvalidValue = no such instantiation (the caller runs the full-width kernel)
hipError_t launch_fista_tile_sp_ds(const FistaTileParams& p, int kpad, int dpad, int dsteps, int grid, hipStream_t stream,
                                   int waves) {
  if (p.stop_on || dsteps <= 0 || dsteps >= dpad / 32) return hipErrorInvalidValue;
  if (waves == 8 && dpad == 256) {                 // the flagship 16 x 256 tile: d in (128, 224]
    switch (dsteps) {
      case 5: return by_k<16, 8, 5>(p, kpad, grid, stream);
      case 6: return by_k<16, 8, 6>(p, kpad, grid, stream);
      case 7: return by_k<16, 8, 7>(p, kpad, grid, stream);
    }
  } else if (waves == 8 && dpad == 128) {          // 32 x 128 tiles (K <= 512): d <= 96
    if (kpad > 512) return hipErrorInvalidValue;
    switch (dsteps) {
      case 2: return by_k<32, 8, 2>(p, kpad, grid, stream);
      case 3: return by_k<32, 8, 3>(p, kpad, grid, stream);
    }
  } else if (waves == 4 && dpad == 128) {          // narrow 16 x 128 tiles (4 waves): d <= 96, any K
    switch (dsteps) {
      case 2: return by_k<16, 4, 2>(p, kpad, grid, stream);
      case 3: return by_k<16, 4, 3>(p, kpad, grid, stream);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace lasso
