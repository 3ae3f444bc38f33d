// Saturating MFMA loop (round 5): what the fp16 matrix pipe of this chip sustains when
// NOTHING but v_mfma_f32_32x32x16_f16 is issued -- NACC independent accumulators per wave that persist over the
// whole kernel (no re-zeroing: one dependency chain of 16 * iters per accumulator), operands in registers, 1 / 2 / 4
// waves per SIMD -- on all-zero operands (must reach the guide's 2495 TFLOP/s before anything else measured with
// it is called a ceiling) and on random operands (the clock follows the power the operand data draws).  The
// shader clock is logged beside every rate: cycles of s_memtime (clock64) over the kernel against the 100 MHz
// wall clock, and the clock implied by the MFMA count (32 cycles per 32x32x16 instruction and SIMD).
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF16>
__global__ void __launch_bounds__(256) sat(const f16x8 *in, float *out, long long *clk, int iters) {
  f16x8 a[4], b[NACC];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 256 * i];
#pragma unroll
  for (int i = 0; i < NACC; ++i) b[i] = in[threadIdx.x + 256 * (4 + i)];
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if (BF16)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[c & 3]),
                                                           __builtin_bit_cast(bf16x8, b[n]), acc[n], 0, 0, 0);
        else
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c & 3], b[n], acc[n], 0, 0, 0);
      }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float res = 0.f;
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) res += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = res;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC, bool BF16>
void run(const char *name, const f16x8 *in, float *out, long long *clk, int waves_per_simd, int iters) {
  const int wgs = 256 * waves_per_simd;        // 256 CUs x (4 waves = 1 per SIMD) per workgroup
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  sat<NACC, BF16><<<wgs, 256>>>(in, out, clk, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  sat<NACC, BF16><<<wgs, 256>>>(in, out, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfma_per_simd = (double)waves_per_simd * iters * 16.0 * NACC;
  const double flop = mfma_per_simd * 1024.0 * 32768.0;
  printf("%-6s %d acc x chain %d, %d wave(s)/SIMD: %8.3f ms  %7.0f TFLOP/s  (%.3f of 2500)   clock: s_memtime %.0f MHz, "
         "implied by 32 cycles per MFMA %.0f MHz\n", name, NACC, 16 * iters, waves_per_simd, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 2500.0, (double)h[0] / ((double)h[1] / 100.0), mfma_per_simd * 32.0 / (ms * 1e3));
}

int main(int argc, char **argv) {
  f16x8 *in; float *out; long long *clk;
  hipMalloc(&in, 256 * 12 * sizeof(f16x8)); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 16);
  _Float16 *h = (_Float16 *)malloc(256 * 12 * 16);
  unsigned short *hb = (unsigned short *)malloc(256 * 12 * 16);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (const char *mode : {"zeros", "random", "random/8"}) {
    srand(1);
    for (int i = 0; i < 256 * 12 * 8; ++i) {
      float v = mode[0] == 'z' ? 0.f : (rand() % 2001 - 1000) / 1000.0f;
      if (strlen(mode) > 6) v *= 0.125f;               // the scan's operands: corpus rows scaled into [-1, 1] / sqrt(64)
      h[i] = (_Float16)v;
      unsigned u; memcpy(&u, &v, 4);
      hb[i] = (unsigned short)(u >> 16);
    }
    printf("---- operands: %s (fp16 / bf16 hold the same values)\n", mode);
    hipMemcpy(in, h, 256 * 12 * 16, hipMemcpyHostToDevice);
    for (int w : {1, 2}) run<4, false>("fp16", in, out, clk, w, iters / w);   // (132 registers: 3 waves fit, 4 do not)
    run<8, false>("fp16", in, out, clk, 1, iters / 2);
    run<2, false>("fp16", in, out, clk, 4, iters / 2);                           // 68 registers: 4 waves per SIMD
    hipMemcpy(in, hb, 256 * 12 * 16, hipMemcpyHostToDevice);
    for (int w : {1, 2}) run<4, true>("bf16", in, out, clk, w, iters / w);
  }
  return 0;
}
