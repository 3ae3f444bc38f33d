// calibrate.hip -- what THIS box sustains, measured in the caller's process (bench.py's
// `roofline.measured_ceiling`): boxes of this pool differ by 3-7 % in the rate a kernel gets (the shader
// clock follows the power the operand data draws), so a fraction of the 2.5 PFLOP/s / 8 TB/s spec can only be
// compared between runs when the box's own ceiling stands beside it.
//
//   tfrs_calibrate_mfma_f16  a saturating v_mfma_f32_32x32x16_f16 loop: 2 waves per SIMD on every CU, four
//                            persistent accumulators per wave, operands in registers (uniform random fp16 in
//                            [-1, 1), the fill the guide quotes GEMM rates on), no memory traffic.  Also reports
//                            the shader clock it ran at: s_memtime cycles over the 100 MHz constant clock.
//   tfrs_calibrate_copy      a float4 copy (one 4-8 KiB chunk per workgroup, plain and nontemporal variants, the best
//                            one reported), half of the bytes read and half written: the ceiling of
//                            every HBM-bound kernel of the library (gather, segment-sum, DotInteraction, Adagrad).
//
// Both calls time with HIP events on `stream` and WAIT for them: they are measurement entry points, not part of
// any launch path.
#include "common.h"

namespace tfrs {

typedef float cal_f32x16 __attribute__((ext_vector_type(16)));
typedef float cal_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 cal_f16x8 __attribute__((ext_vector_type(8)));

constexpr int kCalAcc = 4;

__device__ __forceinline__ uint32_t cal_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void __launch_bounds__(256) cal_mfma_kernel(float *__restrict__ out, long long *__restrict__ clk,
                                                       int iters) {
  cal_f16x8 a[4], b[kCalAcc];
#pragma unroll
  for (int i = 0; i < 4 + kCalAcc; ++i) {
    cal_f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t h = cal_hash((uint32_t)(threadIdx.x * 97 + i * 8 + e) * 2654435761u + 12345u);
      v[e] = (_Float16)((float)(int)(h % 2001u) * 1e-3f - 1.0f);     // uniform in [-1, 1]
    }
    if (i < 4) a[i] = v; else b[i - 4] = v;
  }
  cal_f32x16 acc[kCalAcc];
#pragma unroll
  for (int n = 0; n < kCalAcc; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int n = 0; n < kCalAcc; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c & 3], b[n], acc[n], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float res = 0.f;
#pragma unroll
  for (int n = 0; n < kCalAcc; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) res += acc[n][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = res;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

__global__ void __launch_bounds__(256) cal_fill_kernel(cal_f32x4 *__restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const uint32_t h = cal_hash((uint32_t)i);
    p[i] = cal_f32x4{(float)(h & 1023u), (float)((h >> 10) & 1023u), (float)(h >> 20), 1.0f};
  }
}

// One workgroup per U KiB * 4 chunk, no loop: lane t moves pieces t, t + 256, ... of the chunk (every wave instruction is
// one contiguous 1 KiB line set); NT = nontemporal loads / stores.  The calibration reports the best variant.
template <int U, bool NT>
__global__ void __launch_bounds__(256) cal_copy_kernel(const cal_f32x4 *__restrict__ src, cal_f32x4 *__restrict__ dst,
                                                       size_t n4) {
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  cal_f32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n4) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n4) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u];
    }
  }
}

template <int U, bool NT>
static hipError_t cal_copy_time(const cal_f32x4 *src, cal_f32x4 *dst, size_t n4, int iters, hipStream_t s, float *ms) {
  const dim3 grid((unsigned)((n4 + 256 * U - 1) / (256 * U)));
  hipEvent_t e0, e1;
  hipError_t e;
  if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
  if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
  hipLaunchKernelGGL((cal_copy_kernel<U, NT>), grid, dim3(256), 0, s, src, dst, n4);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((cal_copy_kernel<U, NT>), grid, dim3(256), 0, s, src, dst, n4);
  (void)hipEventRecord(e1, s);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
  e = hipEventElapsedTime(ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return e;
}

}  // namespace tfrs

extern "C" size_t tfrs_calibrate_workspace_bytes(void) { return (size_t)512 * 256 * 4 + 256; }

extern "C" int tfrs_calibrate_mfma_f16(void *ws, size_t ws_bytes, int iters, double *tflops_h, double *shader_mhz_h,
                                       void *stream) {
  TFRS_CHECK_ARG(ws && ws_bytes >= tfrs_calibrate_workspace_bytes() && iters > 0 && iters <= (1 << 20),
                 "calibrate_mfma_f16: workspace of tfrs_calibrate_workspace_bytes() bytes and 0 < iters <= 2^20 needed");
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, cus = 0;
  TFRS_HIP(hipGetDevice(&dev));
  TFRS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int wgs = cus * 2;                        // one wave per SIMD per workgroup, two workgroups per CU
  TFRS_CHECK_ARG(wgs <= 512, "calibrate_mfma_f16: more than 256 compute units");
  float *out = (float *)ws;
  long long *clk = (long long *)((char *)ws + (size_t)512 * 256 * 4);
  hipEvent_t e0, e1;
  TFRS_HIP(hipEventCreate(&e0));
  TFRS_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(tfrs::cal_mfma_kernel, dim3(wgs), dim3(256), 0, s, out, clk, iters / 8 + 1);   // clocks settle
  TFRS_HIP(hipEventRecord(e0, s));
  hipLaunchKernelGGL(tfrs::cal_mfma_kernel, dim3(wgs), dim3(256), 0, s, out, clk, iters);
  TFRS_HIP(hipEventRecord(e1, s));
  TFRS_LAUNCH_CHECK();
  TFRS_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TFRS_HIP(hipEventElapsedTime(&ms, e0, e1));
  long long h[2] = {0, 0};
  TFRS_HIP(hipMemcpyAsync(h, clk, 16, hipMemcpyDeviceToHost, s));
  TFRS_HIP(hipStreamSynchronize(s));
  TFRS_HIP(hipEventDestroy(e0));
  TFRS_HIP(hipEventDestroy(e1));
  // every wave issues iters * 16 * kCalAcc MFMAs of 2 * 32 * 32 * 16 flop
  const double flop = (double)wgs * 4.0 * iters * 16.0 * tfrs::kCalAcc * 32768.0;
  if (tflops_h) *tflops_h = flop / ((double)ms * 1e-3) / 1e12;
  if (shader_mhz_h) *shader_mhz_h = h[1] > 0 ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;
  return TFRS_OK;
}

extern "C" int tfrs_calibrate_copy(void *ws, size_t ws_bytes, int iters, double *gbs_h, void *stream) {
  TFRS_CHECK_ARG(ws && ws_bytes >= (size_t)(64 << 20) && iters > 0 && ((uintptr_t)ws & 15) == 0,
                 "calibrate_copy: a 16-byte aligned workspace of at least 64 MiB and iters > 0 needed");
  hipStream_t s = (hipStream_t)stream;
  const size_t n4 = ws_bytes / 32;                // 16-byte elements per half
  tfrs::cal_f32x4 *src = (tfrs::cal_f32x4 *)ws, *dst = src + n4;
  int dev = 0, cus = 0;
  TFRS_HIP(hipGetDevice(&dev));
  TFRS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const dim3 grid((unsigned)(cus * 16));
  hipLaunchKernelGGL(tfrs::cal_fill_kernel, grid, dim3(256), 0, s, src, n4);
  TFRS_LAUNCH_CHECK();
  float ms = 1e30f, t = 0.f;
  TFRS_HIP((tfrs::cal_copy_time<4, false>(src, dst, n4, iters, s, &t))); ms = t < ms ? t : ms;
  TFRS_HIP((tfrs::cal_copy_time<8, false>(src, dst, n4, iters, s, &t))); ms = t < ms ? t : ms;
  TFRS_HIP((tfrs::cal_copy_time<4, true>(src, dst, n4, iters, s, &t))); ms = t < ms ? t : ms;
  TFRS_HIP((tfrs::cal_copy_time<8, true>(src, dst, n4, iters, s, &t))); ms = t < ms ? t : ms;
  if (gbs_h) *gbs_h = 2.0 * (double)n4 * 16.0 * iters / ((double)ms * 1e-3) / 1e9;
  return TFRS_OK;
}
