// topk_pack.hip -- row-major candidates[n, d]  <->  packed MFMA/LDS layout (common.h).
// HBM-bound copy kernels: one thread moves one 16-byte slot; a wave therefore writes
// 1 KiB of contiguous packed rows per instruction.
#include "common.h"

namespace tfrs {

__global__ void __launch_bounds__(256) pack_kernel(const float *__restrict__ cand,
                                                   int64_t n, int d, int dp,
                                                   char *__restrict__ packed,
                                                   int64_t dst_row, int64_t total_rows) {
  const int slots = dp / 4 + 1;  // incl. pad slot
  const int half = dp / 8;       // slots per plane
  const int64_t nslots = total_rows * slots;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nslots;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / slots;
    const int s = (int)(t - r * slots);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n && s < 2 * half) {
      const int h = s / half;  // 0: even features, 1: odd features
      const int m = s - h * half;
      const float *row = cand + r * d;
      const int k0 = 8 * m + h;  // features k0, k0+2, k0+4, k0+6
      v.x = (k0 < d) ? row[k0] : 0.f;
      v.y = (k0 + 2 < d) ? row[k0 + 2] : 0.f;
      v.z = (k0 + 4 < d) ? row[k0 + 4] : 0.f;
      v.w = (k0 + 6 < d) ? row[k0 + 6] : 0.f;
    }
    *reinterpret_cast<float4 *>(packed + (dst_row + r) * (int64_t)row_bytes(dp) +
                                (int64_t)s * 16) = v;
  }
}

__global__ void __launch_bounds__(256) unpack_kernel(const char *__restrict__ packed,
                                                     int64_t n, int d, int dp,
                                                     float *__restrict__ out) {
  const int half = dp / 8;
  const int64_t total = n * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / d;
    const int k = (int)(t - r * d);
    const int h = k & 1, s = k >> 1;  // plane, position inside the plane
    const float *row = reinterpret_cast<const float *>(packed + r * (int64_t)row_bytes(dp));
    out[t] = row[h * half * 4 + s];
  }
}

// Packs cand[n, d] into rows dst_row .. dst_row+n-1 and zero-fills rows up to
// zero_rows_to (exclusive, >= dst_row + n) so that whole kTileN stages can be read.
int launch_pack(const float *cand, int64_t n, int d, char *packed, int64_t dst_row,
                int64_t zero_rows_to, hipStream_t stream) {
  const int dp = padded_dim(d);
  const int64_t total_rows = zero_rows_to - dst_row;
  if (total_rows <= 0) return TFRS_OK;
  const int64_t nslots = total_rows * (dp / 4 + 1);
  int64_t blocks = (nslots + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, cand, n, d,
                     dp, packed, dst_row, total_rows);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_unpack(const char *packed, int64_t n, int d, float *out, hipStream_t stream) {
  if (n <= 0) return TFRS_OK;
  const int dp = padded_dim(d);
  int64_t blocks = (n * d + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, packed, n,
                     d, dp, out);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

}  // namespace tfrs
