// The fused persistent FISTA kernel: one 8-wave workgroup per 16-row tile, software-pipelined.
// Algorithm, LDS layouts, W streaming and MFMA operand convention: DESIGN.md section 3.1.
//
// Why: all waves of a workgroup run the same instruction stream in near lockstep (the
// two barriers per iteration re-align them), so any stretch in which a wave is NOT
// issuing MFMAs -- waiting for its ds_read fragments, issuing LDS-DMA, running the prox
// epilogue -- is a stretch in which EVERY wave on the SIMD is idle and the matrix pipe
// drains (ablations on MI355X: epilogue 7 %, DMA issue 5 %, fragment waits ~10 %).
// Here each wave hides those stretches behind its OWN MFMAs:
//   * B/A fragments of step g+1 are read into a second register set while the MFMAs of
//     step g run (the ring slot is released -- and refilled by LDS-DMA with step g+3 --
//     as soon as those reads have returned, a few MFMAs into step g);
//   * the prox/momentum epilogue of GEMM-2 pass p runs between the MFMAs of the first
//     step of pass p+1 (two alternating accumulator sets).
#include <stdlib.h>
#include "fista_tile_sp_kernel.hpp"

namespace lasso {

// waves = 8: tiles of 4096 / dpad rows; waves = 4 (dpad <= 128): half-height tiles, two workgroups per CU
hipError_t fista_tile_sp_occupancy(int kpad, int dpad, int* blocks_per_cu, int waves) {
  if (waves == 4) {
    if (dpad == 128) {
      switch (kpad) {
        case 256: return sp::occupancy_k<256, 16, 4>(blocks_per_cu);
        case 384: return sp::occupancy_k<384, 16, 4>(blocks_per_cu);
        case 512: return sp::occupancy_k<512, 16, 4>(blocks_per_cu);
        case 768: return sp::occupancy_k<768, 16, 4>(blocks_per_cu);
        case 1024: return sp::occupancy_k<1024, 16, 4>(blocks_per_cu);
      }
    } else if (dpad == 64) {
      if (kpad == 256) return sp::occupancy_k<256, 32, 4>(blocks_per_cu);
    }
    return hipErrorInvalidValue;
  }
  if (dpad == 256) {
    switch (kpad) {
      case 256: return sp::occupancy_k<256, 16>(blocks_per_cu);
      case 512: return sp::occupancy_k<512, 16>(blocks_per_cu);
      case 768: return sp::occupancy_k<768, 16>(blocks_per_cu);
      case 1024: return sp::occupancy_k<1024, 16>(blocks_per_cu);
    }
  } else if (dpad == 128) {
    switch (kpad) {
      case 256: return sp::occupancy_k<256, 32>(blocks_per_cu);
      case 384: return sp::occupancy_k<384, 32>(blocks_per_cu);
      case 512: return sp::occupancy_k<512, 32>(blocks_per_cu);
    }
  } else if (dpad == 64) {
    if (kpad == 256) return sp::occupancy_k<256, 64>(blocks_per_cu);
  }
  return hipErrorInvalidValue;
}

// rows per tile: 512 * waves / dpad  (8 waves: 256 -> 16, 128 -> 32, 64 -> 64; 4 waves: 128 -> 16, 64 -> 32)
hipError_t launch_fista_tile_sp(const FistaTileParams& p, int kpad, int dpad, int grid, hipStream_t stream, int waves) {
  // rows with fewer features than the tile's padded width: the instantiation that leaves the all-padding chunks of
  // GEMM-2's contraction out, when there is one (fista_tile_sp_ds.hip; LASSO_NO_DSTEPS=1 in the environment: A/B)
  static const bool ds_off = getenv("LASSO_NO_DSTEPS") != nullptr;
  if (!p.stop_on && !ds_off && (p.d + 31) / 32 < dpad / 32) {
    const hipError_t e = launch_fista_tile_sp_ds(p, kpad, dpad, (p.d + 31) / 32, grid, stream, waves);
    if (e != hipErrorInvalidValue) return e;
    (void)hipGetLastError();
  }
  if (waves == 4) {
    if (dpad == 128) {
      switch (kpad) {
        case 256: return sp::launch_k<256, 16, 4>(p, grid, stream);
        case 384: return sp::launch_k<384, 16, 4>(p, grid, stream);
        case 512: return sp::launch_k<512, 16, 4>(p, grid, stream);
        case 768: return sp::launch_k<768, 16, 4>(p, grid, stream);
        case 1024: return sp::launch_k<1024, 16, 4>(p, grid, stream);
      }
    } else if (dpad == 64) {
      if (kpad == 256) return sp::launch_k<256, 32, 4>(p, grid, stream);
    }
    return hipErrorInvalidValue;
  }
  if (dpad == 256) {
    switch (kpad) {
      case 256: return sp::launch_k<256, 16>(p, grid, stream);
      case 512: return sp::launch_k<512, 16>(p, grid, stream);
      case 768: return sp::launch_k<768, 16>(p, grid, stream);
      case 1024: return sp::launch_k<1024, 16>(p, grid, stream);
    }
  } else if (dpad == 128) {
    switch (kpad) {
      case 256: return sp::launch_k<256, 32>(p, grid, stream);
      case 384: return sp::launch_k<384, 32>(p, grid, stream);
      case 512: return sp::launch_k<512, 32>(p, grid, stream);
    }
  } else if (dpad == 64) {
    if (kpad == 256) return sp::launch_k<256, 64>(p, grid, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace lasso
