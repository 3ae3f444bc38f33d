// logits_ce.hip -- Keras CategoricalCrossentropy(from_logits=True, reduction=SUM) on an EXPLICIT logits matrix
// (tasks/retrieval.py:86-87, :210), for the Retrieval paths that have to build the [B, C] matrix: multi-head
// (max-sim) queries :172-176, embedding dims above the fused kernels' envelope, batch metrics / hard-negative
// mining combined with a logit adjustment :205-208.  The default training path never comes here (it runs the
// fused in-batch softmax of softmax16.hip / softmax.hip, where the matrix does not exist).
//
//   loss_i = w_i * ( lse_i * sum_j y_ij  -  sum_j y_ij s_ij ),   lse_i = log sum_j exp(s_ij)   (max-subtracted)
//   dS_ij  = g * w_i * ( softmax(S_i)_j * sum_j y_ij  -  y_ij )
// One wave per row, two passes over the row (it is read from L2 the second time); HBM-bound:
// forward nq * nc * 8 bytes (logits + labels), backward nq * nc * 12.
#include "common.h"

namespace tfrs {

constexpr int kCeWaves = 4;

__device__ __forceinline__ float ce_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float ce_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ void __launch_bounds__(kCeWaves * 64) logits_ce_fwd_kernel(const float *__restrict__ logits,
                                                                      const float *__restrict__ labels,
                                                                      int64_t nq, int64_t nc,
                                                                      const float *__restrict__ weight,
                                                                      float *__restrict__ row_loss,
                                                                      float *__restrict__ lse_out,
                                                                      float *__restrict__ ysum_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kCeWaves + (threadIdx.x >> 6);
  if (row >= nq) return;
  const float *s = logits + row * nc, *y = labels + row * nc;
  float mx = -__builtin_inff();
  for (int64_t j = lane; j < nc; j += 64) mx = fmaxf(mx, s[j]);
  mx = ce_wave_max(mx);
  float se = 0.0f, ys = 0.0f, ysum = 0.0f;
  for (int64_t j = lane; j < nc; j += 64) {
    const float v = s[j], yy = y[j];
    se += __expf(v - mx);
    ys += yy * v;
    ysum += yy;
  }
  se = ce_wave_sum(se);
  ys = ce_wave_sum(ys);
  ysum = ce_wave_sum(ysum);
  if (lane == 0) {
    const float lse = mx + __logf(se);
    lse_out[row] = lse;
    ysum_out[row] = ysum;
    row_loss[row] = (weight ? weight[row] : 1.0f) * (lse * ysum - ys);
  }
}

__global__ void __launch_bounds__(kCeWaves * 64) logits_ce_bwd_kernel(const float *__restrict__ logits,
                                                                      const float *__restrict__ labels,
                                                                      int64_t nq, int64_t nc,
                                                                      const float *__restrict__ weight,
                                                                      const float *__restrict__ lse,
                                                                      const float *__restrict__ ysum,
                                                                      const float *__restrict__ gscale,
                                                                      float *__restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kCeWaves + (threadIdx.x >> 6);
  if (row >= nq) return;
  const float w = (weight ? weight[row] : 1.0f) * gscale[0];
  const float l = lse[row], ysr = ysum[row];
  const float *s = logits + row * nc, *y = labels + row * nc;
  float *d = dlogits + row * nc;
  for (int64_t j = lane; j < nc; j += 64) d[j] = w * (__expf(s[j] - l) * ysr - y[j]);
}

}  // namespace tfrs

using namespace tfrs;

extern "C" int tfrs_logits_ce_fwd(const float *logits, const float *labels, int64_t nq, int64_t nc,
                                  const float *sample_weight, float *row_loss, float *lse, float *ysum,
                                  void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && nc >= 1, "logits_ce_fwd: bad shape");
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(logits && labels && row_loss && lse && ysum, "logits_ce_fwd: NULL pointer");
  hipLaunchKernelGGL(logits_ce_fwd_kernel, dim3((unsigned)((nq + kCeWaves - 1) / kCeWaves)), dim3(kCeWaves * 64),
                     0, (hipStream_t)stream, logits, labels, nq, nc, sample_weight, row_loss, lse, ysum);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_logits_ce_bwd(const float *logits, const float *labels, int64_t nq, int64_t nc,
                                  const float *sample_weight, const float *lse, const float *ysum,
                                  const float *grad_scale, float *dlogits, void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && nc >= 1, "logits_ce_bwd: bad shape");
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(logits && labels && lse && ysum && grad_scale && dlogits, "logits_ce_bwd: NULL pointer");
  hipLaunchKernelGGL(logits_ce_bwd_kernel, dim3((unsigned)((nq + kCeWaves - 1) / kCeWaves)), dim3(kCeWaves * 64),
                     0, (hipStream_t)stream, logits, labels, nq, nc, sample_weight, lse, ysum, grad_scale, dlogits);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
