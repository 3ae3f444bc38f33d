// Microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate for the accumulator-chain shapes the
// top-K scan uses (development tool).  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int CHAIN, bool EPI, bool INTERLEAVE = false>
__global__ void __launch_bounds__(512, 2) k(const f16x8 *in, float *out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 512 * i]; b[i] = in[threadIdx.x + 512 * (i + 4)]; }
  float res = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    if (INTERLEAVE) {
#pragma unroll
      for (int c = 0; c < CHAIN; ++c)
#pragma unroll
        for (int n = 0; n < NACC; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c & 3], b[n & 3], acc[n], 0, 0, 0);
    } else {
#pragma unroll
      for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int c = 0; c < CHAIN; ++c)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c & 3], b[n & 3], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
      if (EPI) {
        float m = acc[n][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = __builtin_fmaxf(m, acc[n][r]);
        res = __builtin_fmaxf(res, m);
      } else {
        res += acc[n][it & 15];
      }
    }
    asm volatile("" : "+v"(res));
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int NACC, int CHAIN, bool EPI, bool IL = false>
void run(const char *name, const f16x8 *in, float *out, int wgs) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, CHAIN, EPI, IL><<<wgs, 512>>>(in, out, 10);
  hipEventRecord(e0);
  k<NACC, CHAIN, EPI, IL><<<wgs, 512>>>(in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)wgs * 8 * iters * NACC * CHAIN * 32768.0;
  printf("%-28s wgs=%d  %.3f ms  %.0f TFLOP/s\n", name, wgs, ms, flop / ms / 1e9);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the same loop on the bf16 matrix instruction (same issue rate; 8-bit significands)
__global__ void __launch_bounds__(512, 2) kbf(const bf16x8 *in, float *out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 512 * i]; b[i] = in[threadIdx.x + 512 * (i + 4)]; }
  float res = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c & 3], b[n & 3], acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) res += acc[n][it & 15];
    asm volatile("" : "+v"(res));
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main(int argc, char **argv) {
  f16x8 *in; float *out;
  hipMalloc(&in, 512 * 8 * sizeof(f16x8)); hipMalloc(&out, 1024 * 512 * 4);
  _Float16 *h = (_Float16 *)malloc(512 * 8 * 16);
  // operand data decides the power draw and with it the sustained clock: random (default),
  // "zeros", "const" (all 0.5) or "mN" (random, mantissas truncated to N bits)
  const char *mode = argc > 1 ? argv[1] : "random";
  for (int i = 0; i < 512 * 8 * 8; ++i)
    h[i] = mode[0] == 'z' ? (_Float16)0.0f : mode[0] == 'c' ? (_Float16)0.5f
                                          : (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  if (mode[0] == 'm') {   // "mN": random operands truncated to N mantissa bits (N = 0 .. 10)
    const int keep = atoi(mode + 1);
    unsigned short *hu = (unsigned short *)h;
    for (int i = 0; i < 512 * 8 * 8; ++i) hu[i] &= (unsigned short)(0xFFFFu << (10 - keep));
  }
  hipMemcpy(in, h, 512 * 8 * 16, hipMemcpyHostToDevice);
  printf("operands: %s\n", mode);
  for (int wgs : {256, 512}) {
    run<2, 4, false>("2acc chain4 chain-major", in, out, wgs);
    run<2, 4, true, true>("2acc chain4 interleaved+epi", in, out, wgs);
    run<4, 4, false, true>("4acc chain4 interleaved", in, out, wgs);
  }
  {  // bf16 operands with the same values (converted on the host by truncation of the f32 bits)
    unsigned short *hb = (unsigned short *)malloc(512 * 8 * 16);
    for (int i = 0; i < 512 * 8 * 8; ++i) {
      float f = (float)h[i];
      unsigned u; memcpy(&u, &f, 4);
      hb[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(in, hb, 512 * 8 * 16, hipMemcpyHostToDevice);
    for (int wgs : {256, 512}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters = 4000;
      kbf<<<wgs, 512>>>((const bf16x8 *)in, out, 10);
      hipEventRecord(e0);
      kbf<<<wgs, 512>>>((const bf16x8 *)in, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)wgs * 8 * iters * 4 * 4 * 32768.0;
      printf("%-28s wgs=%d  %.3f ms  %.0f TFLOP/s\n", "bf16 4acc chain4 interleaved", wgs, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
