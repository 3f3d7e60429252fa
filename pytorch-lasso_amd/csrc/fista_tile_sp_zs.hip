// ZS instantiations of the fused FISTA tile kernel (fista_tile_sp_kernel.hpp): fixed-iteration launches from an all-zero
// code, whose first iteration has y = 0 and therefore r = -x without a GEMM (round 6; 5 % of a 10-iteration E-step).
// A separate translation unit so that it builds beside fista_tile_sp.hip.
#include <stdlib.h>
#include "fista_tile_sp_kernel.hpp"

namespace lasso {
namespace sp {

template <int K, int M, int NW>
hipError_t launch_zero_start(const FistaTileParams& p, int grid, hipStream_t stream) {
  return launch_ks<K, M, false, NW, 0, true>(p, grid, stream);
}

// every tile geometry launch_fista_tile_sp() dispatches to
#define LASSO_ZS(K, M, NW) template hipError_t launch_zero_start<K, M, NW>(const FistaTileParams&, int, hipStream_t);
LASSO_ZS(256, 16, 4) LASSO_ZS(384, 16, 4) LASSO_ZS(512, 16, 4) LASSO_ZS(768, 16, 4) LASSO_ZS(1024, 16, 4)
LASSO_ZS(256, 32, 4)
LASSO_ZS(256, 16, 8) LASSO_ZS(512, 16, 8) LASSO_ZS(768, 16, 8) LASSO_ZS(1024, 16, 8)
LASSO_ZS(256, 32, 8) LASSO_ZS(384, 32, 8) LASSO_ZS(512, 32, 8)
LASSO_ZS(256, 64, 8)
#undef LASSO_ZS

}  // namespace sp
}  // namespace lasso
