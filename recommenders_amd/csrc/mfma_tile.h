// mfma_tile.h -- 32x32 dot-product tiles on the f32 matrix cores, for the kernels whose
// results are compared with a float tolerance (in-batch softmax, its backward).
//
// v_mfma_f32_32x32x2_f32 takes two k values per step: lanes 0-31 supply k0, lanes 32-63
// supply k1.  These kernels use the "half split" k order -- lane half h owns features
// [h*DP/2, (h+1)*DP/2) -- so that a lane's operand fragment is DP/2 CONTIGUOUS floats of
// a row-major row and can be fetched with 16-byte loads straight from global/L2, no
// repacking.  (The top-K path needs the natural d order for bit-exactness and therefore
// uses the packed even/odd layout instead; see common.h.)
#pragma once

#include "common.h"

namespace tfrs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Fragment of row `row` (row-major, leading dim d) for lane half h: features h*DP/2 + s.
// Zero for invalid rows and for padded features >= d.
template <int DP>
__device__ __forceinline__ void load_row_frag(float (&v)[DP / 2], const float *base,
                                              int64_t row, bool valid, int d, int h,
                                              bool vec_ok) {
  constexpr int HALF = DP / 2;
  if (vec_ok) {  // d == DP and 16-byte aligned rows
    const f32x4 *p = reinterpret_cast<const f32x4 *>(base + row * d + h * HALF);
#pragma unroll
    for (int m = 0; m < HALF / 4; ++m) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid) x = p[m];
      v[4 * m + 0] = x[0];
      v[4 * m + 1] = x[1];
      v[4 * m + 2] = x[2];
      v[4 * m + 3] = x[3];
    }
  } else {
#pragma unroll
    for (int s = 0; s < HALF; ++s) {
      const int f = h * HALF + s;
      v[s] = (valid && f < d) ? base[row * d + f] : 0.0f;
    }
  }
}

// D[i][j] = sum_f A[i][f] * B[j][f]; lane l holds a = frag of A row (l&31), b = frag of B
// row (l&31).  Result: lane (j = l&31, h = l>>5), register r -> A row (r&3) + 8*(r>>2) + 4*h.
template <int DP>
__device__ __forceinline__ f32x16 tile_dot(const float (&a)[DP / 2], const float (&b)[DP / 2]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < DP / 2; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ int tile_row_of_reg(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Padded feature dims the register-resident kernels are instantiated for.
__host__ __device__ inline int softmax_padded_dim(int d) {
  if (d <= 8) return 8;
  if (d <= 16) return 16;
  if (d <= 32) return 32;
  if (d <= 64) return 64;
  return 128;
}

}  // namespace tfrs
