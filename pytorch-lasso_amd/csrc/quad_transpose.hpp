// DPP helper shared by the kernels that turn MFMA accumulators into 16-byte row pieces (bt_iter.hip, conv_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace lasso {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4 x 4 transpose inside every lane quad: on entry lane j of a quad holds v[0..3] = M[j][0..3], on exit M[0..3][j].
// Two butterfly stages (lane bit 0 with element bit 0, lane bit 1 with element bit 1), one DPP move per exchanged
// element.  Used to turn the MFMA C layout (a lane owns FOUR ROWS of one column) into 16-byte row pieces.
__device__ __forceinline__ void quad_transpose(f32x4& v, int j) {
  const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {            // pairs (0,1), (2,3): exchange with lane j ^ 1
    const float send = b0 ? v[2 * h] : v[2 * h + 1];
    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xf, 0xf, true));  // quad_perm:[1,0,3,2]
    if (b0) v[2 * h] = recv; else v[2 * h + 1] = recv;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {            // pairs (0,2), (1,3): exchange with lane j ^ 2
    const float send = b1 ? v[h] : v[h + 2];
    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xf, 0xf, true));  // quad_perm:[2,3,0,1]
    if (b1) v[h] = recv; else v[h + 2] = recv;
  }
}

}  // namespace lasso
