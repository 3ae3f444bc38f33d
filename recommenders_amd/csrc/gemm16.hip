// gemm16.hip -- f32-grade GEMM on the fp16 matrix cores (split operands), for the large dense
// products of the ranking models: Cross (layers/feature_interaction/dcn.py:151-186: x @ kernel
// with the cross formula fused in the epilogue), its gradients, and MLP / low-rank projections
// (layers/blocks.py:46-61).  At BASELINE configs[3] (B = 65536, d = 3456) one Cross layer is a
// 1.57 TFLOP product; the f32-input MFMA tops out at 157 TFLOP/s.
//
// C[M, N] = A[M, K] @ B[K, N] (row-major f32 in and out).  Every operand value x is split into two
// fp16 numbers, x * 2^a = hi + lo (a = power of two per A row / per B column that puts the largest
// magnitude of that row / column in [2^9, 2^10): no fp16 under- or overflow;
// |x * 2^a - hi - lo| <= 2^-22 |x * 2^a|), and the product is accumulated in f32 as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16: three MFMA products per f32 product, 16x the
// f32 MFMA rate each.  The scales are powers of two and are undone exactly in the epilogue.
//
//   prep_rows   A  -> Ah, Al [Mp, Kp] (row-major, K padded to 32), inv_a[Mp]
//   colmax +
//   prep_cols   B  -> Bh, Bl [Np, Kp] (one row per OUTPUT column: K contiguous), inv_b[Np]
//   gemm16      128 x 128 tile per workgroup (4 waves, 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles),
//               K step 32; both operand tiles are staged by direct-to-LDS copies (16 B per lane,
//               XOR-swizzled 16-byte slots instead of row padding, so an unpadded 64-byte tile
//               row is conflict-free for ds_read_b128) into a double buffer; completion tracked
//               by hand (s_waitcnt vmcnt + s_barrier), copies issued from inline assembly (see
//               softmax16.hip for why); epilogue: * inv_a[row] * inv_b[col] + bias [cross formula].
//
// Roofline: MFMA-bound; ALGORITHMIC flop 2 M N K, the pipe executes 3x that, priced against the
// dense fp16 peak.  Operand images cost M K + K N extra 4-byte writes per call (prep).
#include <stdlib.h>

#include <algorithm>

#include "mfma_tile.h"

namespace tfrs {

typedef _Float16 g16h8 __attribute__((ext_vector_type(8)));
typedef _Float16 g16h4 __attribute__((ext_vector_type(4)));

constexpr int kG16M = 128, kG16N = 128, kG16K = 32;
constexpr int kG16Img = 128 * 64;   // bytes of one 128-row x 32-half tile image in LDS

__device__ __forceinline__ uint32_t g16_f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float g16_u2f(uint32_t x) { return __builtin_bit_cast(float, x); }

// power-of-two scale that puts `maxabs` in [2^9, 2^10), and its inverse (1, 1 for ~zero rows)
__device__ __forceinline__ void g16_scale_of(float maxabs, float *s, float *inv) {
  const uint32_t e = (g16_f2u(maxabs) >> 23) & 0xffu;
  *s = 1.0f;
  *inv = 1.0f;
  if (e >= 27u && e < 255u) {
    *s = g16_u2f((263u - e) << 23);
    *inv = g16_u2f((e - 9u) << 23);
  }
}

// Image addressing.  An image holds `rows_p` rows of kp halves.  With kb == kp a row is contiguous
// (offset row * kp + k).  Long-K products (dW = x^T dz: K = batch) use K-BLOCKED images, kb < kp:
// block q = k / kb holds all rows' k in [q * kb, (q + 1) * kb) back to back, offset
// (q * rows_p + row) * kb + k % kb -- the 256-row operand tile of one K block is then ONE
// contiguous 256 * kb * 2-byte range instead of 256 pieces 2 * kp bytes apart (at K = 65536 every
// 32-byte piece of a K step sat in a different page: the product ran at 60 % of the rate of the
// K = 3456 ones).
__device__ __forceinline__ int64_t g16_img_off(int64_t row, int k, int kp, int kb, int64_t rows_p) {
  return kb == kp ? row * (int64_t)kp + k : ((int64_t)(k / kb) * rows_p + row) * kb + (k % kb);
}

// ---- prep: rows of A ------------------------------------------------------------------------
// one wave per row; rows in [m, mp) and columns in [k, kp) are zero
// `mul` (optional, same shape as x): the image is built from x * mul (dz = dy * x0 of the Cross
// backward never exists in HBM).
__global__ void __launch_bounds__(256) g16_prep_rows_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ mul, int64_t m,
                                                            int k, int kp, _Float16 *__restrict__ hi,
                                                            _Float16 *__restrict__ lo,
                                                            float *__restrict__ inv,
                                                            uint32_t *__restrict__ colmax, int np,
                                                            int kb, int64_t rows_p) {
  // first kernel of the chain: re-arms the column maxima the next kernel combines with atomicMax
  // (no hipMemsetAsync: memset nodes are not reliably ordered under HIP-graph replay)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np; i += (int64_t)gridDim.x * 256)
    colmax[i] = 0u;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool valid = row < m;
  const float *xr = x + row * (int64_t)k;
  const float *mr = mul ? mul + row * (int64_t)k : nullptr;
  const bool vec = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   (!mul || (reinterpret_cast<uintptr_t>(mul) & 15) == 0);
  float mx = 0.0f;
  if (valid) {
    if (vec) {
      for (int c = lane * 4; c < k; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(xr + c);
        if (mr) v = v * *reinterpret_cast<const f32x4 *>(mr + c);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    } else {
      for (int c = lane; c < k; c += 64) mx = fmaxf(mx, fabsf(mr ? xr[c] * mr[c] : xr[c]));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float s, iv;
  g16_scale_of(mx, &s, &iv);
  if (lane == 0) inv[row] = iv;
  for (int c = lane * 4; c < kp; c += 256) {
    const int64_t io = g16_img_off(row, c, kp, kb, rows_p);   // 4 consecutive k stay in one block
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      if (vec && c + 3 < k) {
        f32x4 t = *reinterpret_cast<const f32x4 *>(xr + c);
        if (mr) t = t * *reinterpret_cast<const f32x4 *>(mr + c);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c + u < k) v[u] = mr ? xr[c + u] * mr[c + u] : xr[c + u];
      }
    }
    g16h4 h4, l4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float sv = v[u] * s;
      h4[u] = (_Float16)sv;
      l4[u] = (_Float16)(sv - (float)h4[u]);
    }
    *reinterpret_cast<g16h4 *>(hi + io) = h4;
    *reinterpret_cast<g16h4 *>(lo + io) = l4;
  }
}

// ---- prep: rows of A into K-step-major images (kb = 16) -------------------------------------------
// One wave per 4 consecutive rows.  Lane l: K block (l >> 4) of the current group of 4 blocks, row
// (l & 15) >> 2, 4-half part l & 3: the wave READS four 256-byte row segments per step and WRITES,
// per K block, the 4 rows' 32-byte chunks as one contiguous 128-byte line (in the image block q
// holds [rows_p][16] halves).  The row-per-wave kernel above wrote 8-byte pieces 2 * rows_p * 16
// bytes apart into this layout: 0.62 ms per 65536 x 3456 activation instead of 0.39.
__global__ void __launch_bounds__(256) g16_prep_rows_ksm_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ mul, int64_t m,
                                                                int k, int kp, _Float16 *__restrict__ hi,
                                                                _Float16 *__restrict__ lo,
                                                                float *__restrict__ inv, int64_t rows_p) {
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  const int rr = (lane & 15) >> 2, part = lane & 3, qi = lane >> 4;
  const int64_t row = r0 + rr;
  const bool valid = row < m;
  const float *xr = x + row * (int64_t)k;
  const float *mr = mul ? mul + row * (int64_t)k : nullptr;
  const bool vec = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   (!mul || (reinterpret_cast<uintptr_t>(mul) & 15) == 0);
  auto load4 = [&](int c, float (&v)[4]) __attribute__((always_inline)) {
    v[0] = v[1] = v[2] = v[3] = 0.0f;
    if (!valid) return;
    if (vec && c + 3 < k) {
      f32x4 t = *reinterpret_cast<const f32x4 *>(xr + c);
      if (mr) t = t * *reinterpret_cast<const f32x4 *>(mr + c);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c + u < k) v[u] = mr ? xr[c + u] * mr[c + u] : xr[c + u];
    }
  };
  float mx = 0.0f;
  for (int q0 = 0; q0 * 16 < kp; q0 += 4) {
    float v[4];
    load4((q0 + qi) * 16 + part * 4, v);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  // lanes of one row: same (l >> 2) & 3 -> reduce over the part bits (1, 2) and the block bits (16, 32)
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float s, iv;
  g16_scale_of(mx, &s, &iv);
  if (part == 0 && qi == 0 && row < rows_p) inv[row] = iv;
  for (int q0 = 0; q0 * 16 < kp; q0 += 4) {
    const int q = q0 + qi;
    if (q * 16 >= kp) continue;
    float v[4];
    load4(q * 16 + part * 4, v);
    g16h4 h4, l4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float sv = v[u] * s;
      h4[u] = (_Float16)sv;
      l4[u] = (_Float16)(sv - (float)h4[u]);
    }
    const int64_t io = ((int64_t)q * rows_p + row) * 16 + part * 4;
    *reinterpret_cast<g16h4 *>(hi + io) = h4;
    *reinterpret_cast<g16h4 *>(lo + io) = l4;
  }
}

// Single-pass form of the kernel above for kp <= 8192: the four waves of a workgroup share FOUR rows, wave w
// holding K blocks [w nq4, (w + 1) nq4) of them in registers (lane roles as above: 4 blocks x 4 rows x 4
// parts per step, <= 16 steps of 8 floats), so a row is read ONCE -- the two-pass kernel reads it for the maximum and
// again for the conversion, and at 55 KB per wave the second read comes from HBM (0.51 ms for a
// 65536 x 3456 activation, 0.88 with the fused multiplier; both operands twice).  The row maxima meet in
// LDS.  Same scale, same conversion, same image as the two-pass kernel, bit for bit.
// V2 (round 6): rows that are only 8-byte aligned (k % 4 == 2: the DLRM top MLP reads a [batch, 5082] matrix -- 5050
// interaction pairs + 32 bottom-stack outputs) load 8 bytes per instruction instead of 16 and take the row's last, partial
// chunk from the clamped load; they used to fall back to the two-pass kernel: 2.08 ms for 131072 x 5082 (1.9 TB/s).
// NW = waves per workgroup (= per group of four rows): 8 for the long unaligned rows -- sixteen steps of 8-byte loads per
// lane on four waves ran at 2.4 TB/s (177 registers: two waves per SIMD), eight steps on eight waves fit twice the waves.
template <int STEPS, bool MUL, bool V2 = false, int NW = 4>
__global__ void __launch_bounds__(NW * 64) g16_prep_rows_ksm1_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ mul, int64_t m,
                                                                 int k, int kp, _Float16 *__restrict__ hi,
                                                                 _Float16 *__restrict__ lo,
                                                                 float *__restrict__ inv, int64_t rows_p) {
  __shared__ float s_mx[NW][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane roles per step: 8 K blocks x 4 rows x 2 halves of a block (8 floats in, one 16-byte store per image out)
  const int part = lane & 1, rr = (lane >> 1) & 3, qi = lane >> 3;
  const int64_t row = (int64_t)blockIdx.x * 4 + rr;
  const bool valid = row < m;
  const int64_t rowc = valid ? row : m - 1;     // (loads are unconditional: clamped, zeroed afterwards)
  const float *xr = x + rowc * (int64_t)k;
  const float *mr = MUL ? mul + rowc * (int64_t)k : nullptr;   // (launched only for k % 4 == 0 and 16-byte aligned operands)
  const int nq = kp / 16, nq4 = (nq + NW - 1) / NW;   // K blocks of a row, per wave
  const int q_base = wave * nq4;
  float v[STEPS][8];
  float mx = 0.0f;
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int q = q_base + 8 * st + qi;
    const bool in = 8 * st + qi < nq4 && q < nq && valid;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int c = q * 16 + part * 8 + hf * 4;
      const int cl = c + 3 < k ? c : k - 4;   // clamped: every load is issued, nothing waits before the last one
      f32x4 t;
      if (V2) {
        // (global loads need no natural alignment on gfx950: one 16-byte load from an 8-byte aligned address)
        typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
        t = *reinterpret_cast<const f32x4a8 *>(xr + cl);
        if (MUL) t = t * *reinterpret_cast<const f32x4a8 *>(mr + cl);
        // element c + u sits at t[u + 2] in the row's last chunk when k % 4 == 2 (its two valid elements are the upper
        // half of the clamped load), at t[u] everywhere else.  Plain selects: a nested ternary here compiled to a branch
        // per element with `s_waitcnt vmcnt(0)` behind every load -- one memory round trip per 16 bytes, 2.85 TB/s
        const bool part_chunk = c + 3 >= k;
        const float e0 = part_chunk ? t[2] : t[0], e1 = part_chunk ? t[3] : t[1];
        const float e2 = part_chunk ? 0.0f : t[2], e3 = part_chunk ? 0.0f : t[3];
        v[st][hf * 4 + 0] = (in && c + 0 < k) ? e0 : 0.0f;
        v[st][hf * 4 + 1] = (in && c + 1 < k) ? e1 : 0.0f;
        v[st][hf * 4 + 2] = (in && c + 2 < k) ? e2 : 0.0f;
        v[st][hf * 4 + 3] = (in && c + 3 < k) ? e3 : 0.0f;
      } else {
        t = *reinterpret_cast<const f32x4 *>(xr + cl);
        if (MUL) t = t * *reinterpret_cast<const f32x4 *>(mr + cl);
        const bool ok = in && c + 3 < k;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[st][hf * 4 + u] = ok ? t[u] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) mx = fmaxf(mx, fabsf(v[st][u]));
  }
  // lanes of one row: same rr (bits 1, 2) -> reduce over the half bit (1) and the block bits (8, 16, 32)
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 8));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (part == 0 && qi == 0) s_mx[wave][rr] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_mx[0][rr], s_mx[1][rr]), fmaxf(s_mx[2][rr], s_mx[3][rr]));
  if (NW == 8) mx = fmaxf(mx, fmaxf(fmaxf(s_mx[4][rr], s_mx[5][rr]), fmaxf(s_mx[6][rr], s_mx[7][rr])));
  float s, iv;
  g16_scale_of(mx, &s, &iv);
  if (wave == 0 && part == 0 && qi == 0 && row < rows_p) inv[row] = iv;
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int q = q_base + 8 * st + qi;
    if (8 * st + qi >= nq4 || q >= nq || row >= rows_p) continue;
    g16h8 h8, l8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float sv = v[st][u] * s;
      h8[u] = (_Float16)sv;
      l8[u] = (_Float16)(sv - (float)h8[u]);
    }
    const int64_t io = ((int64_t)q * rows_p + row) * 16 + part * 8;
    *reinterpret_cast<g16h8 *>(hi + io) = h8;
    *reinterpret_cast<g16h8 *>(lo + io) = l8;
  }
}

static void g16_launch_prep_rows_ksm(const float *x, const float *mul, int64_t m, int k, int kp, _Float16 *hi,
                                     _Float16 *lo, float *inv, int64_t rows_p, hipStream_t s) {
  const int nq4 = (kp / 16 + 3) / 4, steps = (nq4 + 7) / 8;
  const dim3 grid1((unsigned)(rows_p / 4));
  const bool vec = (k % 4 == 0) && k >= 4 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   (!mul || (reinterpret_cast<uintptr_t>(mul) & 15) == 0);
  // 8-byte aligned rows (k % 4 == 2): the same single-pass kernel with 8-byte loads
  const bool vec2 = !vec && (k % 4 == 2) && k >= 6 && ((reinterpret_cast<uintptr_t>(x) & 7) == 0) &&
                    (!mul || (reinterpret_cast<uintptr_t>(mul) & 7) == 0);
#define TFRS_KSM1V2(S, M) \
  hipLaunchKernelGGL((g16_prep_rows_ksm1_kernel<S, M, true>), grid1, dim3(256), 0, s, x, mul, m, k, kp, hi, lo, inv, rows_p)
  if (vec2 && steps <= 4) { if (mul) TFRS_KSM1V2(4, true); else TFRS_KSM1V2(4, false); return; }
  if (vec2 && steps <= 16) {    // (eight waves per four rows: eight steps per lane)
    if (mul)
      hipLaunchKernelGGL((g16_prep_rows_ksm1_kernel<8, true, true, 8>), grid1, dim3(512), 0, s, x, mul, m, k, kp, hi, lo, inv, rows_p);
    else
      hipLaunchKernelGGL((g16_prep_rows_ksm1_kernel<8, false, true, 8>), grid1, dim3(512), 0, s, x, mul, m, k, kp, hi, lo, inv, rows_p);
    return;
  }
#undef TFRS_KSM1V2
#define TFRS_KSM1(S, M) \
  hipLaunchKernelGGL((g16_prep_rows_ksm1_kernel<S, M>), grid1, dim3(256), 0, s, x, mul, m, k, kp, hi, lo, inv, rows_p)
  if (vec && steps <= 2) { if (mul) TFRS_KSM1(2, true); else TFRS_KSM1(2, false); }
  else if (vec && steps <= 4) { if (mul) TFRS_KSM1(4, true); else TFRS_KSM1(4, false); }
  else if (vec && steps <= 8) { if (mul) TFRS_KSM1(8, true); else TFRS_KSM1(8, false); }
  else if (vec && steps <= 16) { if (mul) TFRS_KSM1(16, true); else TFRS_KSM1(16, false); }   // kp <= 8192 (DLRM top MLP: 5082)
#undef TFRS_KSM1
  else
    hipLaunchKernelGGL(g16_prep_rows_ksm_kernel, dim3((unsigned)(rows_p / 16)), dim3(256), 0, s, x, mul, m, k, kp, hi,
                       lo, inv, rows_p);
}

// ---- prep: columns of B (transposed images) ------------------------------------------------
// B can be as large as A (dW = x^T dz: both operands are [batch, d] activations), so both passes
// are parallel over K as well: (1) column maxima of 64-column x 256-row slabs, combined with
// atomicMax on the bit patterns of the non-negative maxima (order-independent, hence
// reproducible; colmax is zeroed by the row-prep kernel that runs first), (2) 64 x 64 tiles
// scaled, split and transposed through LDS.
constexpr int kG16Slab = 256;   // (1024 left a 5082 x 1024 weight matrix with 80 workgroups of 32 serial rounds: 95 us)

// `psum` (optional): psum[slab, n] = column sums of the slab's rows (b * mul), combined
// afterwards in slab order by g16_colsum_kernel -- a deterministic db = sum_rows dz.
__global__ void __launch_bounds__(256) g16_colmax_kernel(const float *__restrict__ b,
                                                         const float *__restrict__ mul, int k, int n,
                                                         uint32_t *__restrict__ colmax,
                                                         float *__restrict__ psum) {
  __shared__ float red[4][64];
  __shared__ float reds[4][64];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * kG16Slab;
  const int c = tid & 63, rg = tid >> 6;
  float mx = 0.0f, sum = 0.0f;
  if (n0 + c < n) {
    const int k1 = k0 + kG16Slab < k ? k0 + kG16Slab : k;
    // eight rows per round, their loads issued together (a loop of one load + one add per iteration is one
    // memory round trip per iteration: 109 us for the 48 MB weight matrix, 290-380 us for a 906 MB activation);
    // rows past the slab are clamped for the load and skipped in the sum, whose order is unchanged
    for (int r0 = k0 + rg; r0 < k1; r0 += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u < k1 ? r0 + 4 * u : k1 - 1;
        const int64_t o = (int64_t)r * n + n0 + c;
        v[u] = mul ? b[o] * mul[o] : b[o];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + 4 * u < k1) {
          mx = fmaxf(mx, fabsf(v[u]));
          sum += v[u];
        }
      }
    }
  }
  red[rg][c] = mx;
  reds[rg][c] = sum;
  __syncthreads();
  if (tid < 64 && n0 + tid < n) {
    const float m4 = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));
    atomicMax(&colmax[n0 + tid], g16_f2u(m4));
    if (psum) psum[(int64_t)blockIdx.y * n + n0 + tid] = (reds[0][tid] + reds[1][tid]) + (reds[2][tid] + reds[3][tid]);
  }
}

__global__ void __launch_bounds__(256) g16_colsum_kernel(const float *__restrict__ psum, int nslab, int n,
                                                         float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  float s = 0.0f;
#pragma unroll 8
  for (int i = 0; i < nslab; ++i) s += psum[(int64_t)i * n + c];   // (eight loads in flight; the sum order is unchanged)
  out[c] = s;
}

// split-K: out[i] = sum over slices (fixed order: deterministic) of part[slice][i]
__global__ void __launch_bounds__(256) g16_splitk_reduce_kernel(const float *__restrict__ part, int nsplit,
                                                                int64_t count, float *__restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  if (i + 3 < count) {
    f32x4 acc = *reinterpret_cast<const f32x4 *>(part + i);
#pragma unroll 4
    for (int s = 1; s < nsplit; ++s) acc = acc + *reinterpret_cast<const f32x4 *>(part + (int64_t)s * count + i);
    *reinterpret_cast<f32x4 *>(out + i) = acc;
  } else {
    for (int64_t e = i; e < count; ++e) {
      float acc = part[e];
      for (int s = 1; s < nsplit; ++s) acc += part[(int64_t)s * count + e];
      out[e] = acc;
    }
  }
}

__global__ void __launch_bounds__(256) g16_zero_kernel(uint32_t *__restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

__global__ void __launch_bounds__(256) g16_prep_cols_kernel(const float *__restrict__ b,
                                                            const float *__restrict__ mul, int k, int n,
                                                            int kp, const uint32_t *__restrict__ colmax,
                                                            _Float16 *__restrict__ hi,
                                                            _Float16 *__restrict__ lo,
                                                            float *__restrict__ inv, int kb,
                                                            int64_t rows_p) {
  __shared__ float tile[64][65];
  __shared__ float s_scale[64];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  if (tid < 64) {
    float s, iv;
    g16_scale_of(g16_u2f(colmax[n0 + tid]), &s, &iv);
    s_scale[tid] = s;
    if (blockIdx.y == 0) inv[n0 + tid] = iv;
  }
  const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0) &&
                   (!mul || (reinterpret_cast<uintptr_t>(mul) & 15) == 0);
  if (vec) {   // 16 threads x 16 bytes per row, 16 rows per pass: the four passes' loads go out together
    const int cc = (tid & 15) * 4, rr = tid >> 4;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = k0 + rr + 16 * u < k ? k0 + rr + 16 * u : k - 1;          // clamped: the load is unconditional
      const int c4 = n0 + cc < n ? n0 + cc : n - 4;
      const int64_t o = (int64_t)r * n + c4;
      v[u] = *reinterpret_cast<const f32x4 *>(b + o);
      if (mul) v[u] = v[u] * *reinterpret_cast<const f32x4 *>(mul + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = k0 + rr + 16 * u < k && n0 + cc < n;
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[rr + 16 * u][cc + e] = ok ? v[u][e] : 0.0f;
    }
  } else {
    // rows that are not 16-byte aligned (n % 4 != 0: the [batch, 5082] input of the DLRM top MLP): one float per lane,
    // a wave instruction = 256 contiguous bytes of a row; ALL sixteen loads of a thread are issued (clamped coordinates)
    // before the first one is used -- with the load inside `if (in range)` in a runtime loop every one of them was a
    // memory round trip of its own: 1.45 ms for the 131072 x 5082 activation (2.75 TB/s)
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u;
      const int r = e >> 6, cc = e & 63;
      const int rc = k0 + r < k ? k0 + r : k - 1, ccl = n0 + cc < n ? n0 + cc : n - 1;
      const int64_t o = (int64_t)rc * n + ccl;
      v[u] = mul ? b[o] * mul[o] : b[o];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u;
      const int r = e >> 6, cc = e & 63;
      tile[r][cc] = (k0 + r < k && n0 + cc < n) ? v[u] : 0.0f;
    }
  }
  __syncthreads();
  const int col = tid >> 2, seg = (tid & 3) * 16;   // 16 consecutive k of one output column
  const float s = s_scale[col];
  const int64_t io = g16_img_off(n0 + col, k0 + seg, kp, kb, rows_p);   // kb is a multiple of 64
  _Float16 *hr = hi + io;
  _Float16 *lr = lo + io;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    g16h8 h8, l8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float sv = tile[seg + q * 8 + u][col] * s;
      h8[u] = (_Float16)sv;
      l8[u] = (_Float16)(sv - (float)h8[u]);
    }
    *reinterpret_cast<g16h8 *>(hr + q * 8) = h8;
    *reinterpret_cast<g16h8 *>(lr + q * 8) = l8;
  }
}

// ---- the GEMM -------------------------------------------------------------------------------
// epilogues (v = scaled accumulator + bias):
//   Bias     out = v
//   Cross    out = e0 * (v + diag * e1) + e1          e0 = x0, e1 = x   (Cross.call)
//   CrossDx0 out = e0 * (v + diag * e1)               e0 = dy, e1 = x   (dx0 = dy * z)
//   CrossDx  out = v + e0 + diag * e0 * e1            e0 = dy, e1 = x0  (dx = dz W^T + dy + diag dz)
enum { kG16EpiBias = 0, kG16EpiCross = 1, kG16EpiCrossDx0 = 2, kG16EpiCrossDx = 3 };

// aux (Cross only, optional): receives u = v + diag * x, the factor the backward multiplies dy by
// (dx0 = dy * u) -- saved by the training forward so that the backward needs two products, not three.
template <int EPI>
__device__ __forceinline__ float g16_epilogue(float v, const float *e0, const float *e1, float diag,
                                              int64_t o, float *aux = nullptr) {
  if (EPI == kG16EpiCross) {
    const float xv = e1[o];
    const float u = v + diag * xv;
    if (aux) aux[o] = u;
    return e0[o] * u + xv;
  }
  if (EPI == kG16EpiCrossDx0) return e0[o] * (v + diag * e1[o]);
  if (EPI == kG16EpiCrossDx) {
    const float dyv = e0[o];
    return v + dyv + diag * dyv * e1[o];
  }
  return v;
}

// The same epilogues on VALUES already in registers (e0v = e0[o], e1v = e1[o]); *u receives v + diag * x.
// The kernels' epilogues first issue ALL the loads of a 32 x 32 accumulator tile (unconditional, at
// clamped coordinates) and then compute and store: with `if (row >= m) continue; ... = e0[o] ...` inside
// the loop every load sat in a branch, the compiler could not count the loads in flight and waited for
// each one (vmcnt(0)) before the next -- 128 serial memory round trips per thread and tile, 1.3 ms of the
// 5.8 ms Cross forward product at configs[3] (the bias-only product of the same size: 4.45 ms).
template <int EPI>
__device__ __forceinline__ float g16_epilogue_v(float v, float e0v, float e1v, float diag, float *u) {
  if (EPI == kG16EpiCross) {
    *u = v + diag * e1v;
    return e0v * *u + e1v;
  }
  if (EPI == kG16EpiCrossDx0) return e0v * (v + diag * e1v);
  if (EPI == kG16EpiCrossDx) return v + e0v + diag * e0v * e1v;
  return v;
}

struct Gemm16Args {
  const _Float16 *ah, *al, *bh, *bl;   // [mp, kp], [np, kp]
  const float *inva, *invb;            // [mp], [np]
  int64_t m;
  int n, kp;
  const float *bias;                   // [n] or NULL
  const float *x0, *x;                 // epilogue operands e0, e1: [m, n] (see the enum below)
  float diag;
  float *out;                          // [m, n]
  float *aux;                          // [m, n] or NULL: see g16_epilogue
  int kb;                              // K block of the images (== kp: plain rows), see g16_img_off
  int64_t mp, np;                      // padded image rows (block stride of K-blocked images)
  int kt_per;                          // big kernel, split-K: K steps per blockIdx.y slice (0 = all);
                                       // slice y writes its partial product to out + y * m * n
  int raster;                          // big kernel: tile order of the workgroups (see gemm16_big_kernel)
  int act;                             // fused activation of v (common.h kAct*), before the epilogue formula
  float *pre;                          // [m, n] or NULL: receives v (the pre-activation) for the backward
  unsigned long long *clk;             // big kernel, measurement (TFRS_GEMM16_CLOCKS=1): += {shader cycles, 100 MHz ticks} per workgroup
};

__device__ __forceinline__ void g16_dma16(const char *gsrc_lane, const char *lds_wave_base) {
  const uint32_t m0v = (uint32_t)(uintptr_t)(
      __attribute__((address_space(3))) const char *)lds_wave_base;
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "v"(gsrc_lane), "s"(m0v)
               : "memory");
}
// the same copy with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the base advances with
// scalar adds, and a lane keeps ONE offset register per operand instead of a 64-bit address per image
__device__ __forceinline__ void g16_dma16_s(uint32_t voff, const char *sbase, const char *lds_wave_base) {
  const uint32_t m0v = (uint32_t)(uintptr_t)(
      __attribute__((address_space(3))) const char *)lds_wave_base;
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "v"(voff), "s"(sbase), "s"(m0v)
               : "memory");
}
__device__ __forceinline__ void g16_wait_dma() {
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8) | (0 << 14));
}

// LDS tile image: 128 rows x 4 slots of 16 bytes (32 halves), no padding; slot s of row r holds
// the k-slot s ^ ((r >> 2) & 3): the 16 lanes a ds_read_b128 serves together then hit 16
// different 16-byte bank groups.
__device__ __forceinline__ int g16_slot(int row, int kslot) { return kslot ^ ((row >> 2) & 3); }

// ACT: the instantiations that apply g.act / store g.pre (kept apart: the libm code of tanh / erf next to the
// accumulators costs the plain Cross epilogues of the 128 x 128 kernel 20 spilled registers)
template <int EPI, bool ACT = false>
__global__ void __launch_bounds__(256, 2) gemm16_kernel(const Gemm16Args g) {
  // [buffer][Ah | Al | Bh | Bl]
  __shared__ __attribute__((aligned(16))) char lds[2][4 * kG16Img];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane & 31, h = lane >> 5;

  // consecutive workgroups walk the N tiles of one M tile: the (large) A rows stay in L2
  const int nbn = (g.n + kG16N - 1) / kG16N;
  const int64_t bm = (int64_t)(blockIdx.x / nbn) * kG16M;
  const int bn = (int)(blockIdx.x % nbn) * kG16N;
  const int nk = g.kp / kG16K;

  // staging: wave w copies tile rows [32w, 32w + 32) of each of the 4 images, 16 rows per
  // instruction; lane i -> row 16u + (i >> 2), LDS slot i & 3 (the copy writes LDS linearly)
  const int sr = lane >> 2, ss = lane & 3;
  const char *src[4][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = wave * 32 + u * 16 + sr;
    const int ks = g16_slot(r, ss);
    src[0][u] = reinterpret_cast<const char *>(g.ah + (bm + r) * g.kp) + ks * 16;
    src[1][u] = reinterpret_cast<const char *>(g.al + (bm + r) * g.kp) + ks * 16;
    src[2][u] = reinterpret_cast<const char *>(g.bh + (int64_t)(bn + r) * g.kp) + ks * 16;
    src[3][u] = reinterpret_cast<const char *>(g.bl + (int64_t)(bn + r) * g.kp) + ks * 16;
  }
  auto stage = [&](int kt, char *buf) __attribute__((always_inline)) {
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        g16_dma16(src[im][u] + (int64_t)kt * (kG16K * 2), buf + im * kG16Img + (wave * 32 + u * 16) * 64);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;

  stage(0, lds[0]);
  g16_wait_dma();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const char *cur = lds[kt & 1];
    if (kt + 1 < nk) stage(kt + 1, lds[(kt + 1) & 1]);
    // fragments of this K step: k-slot q = 2 * kk + h
    g16h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + j, rb = wn * 64 + i * 32 + j;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int oa = ra * 64 + g16_slot(ra, 2 * kk + h) * 16;
        const int ob = rb * 64 + g16_slot(rb, 2 * kk + h) * 16;
        ah[i][kk] = *reinterpret_cast<const g16h8 *>(cur + oa);
        al[i][kk] = *reinterpret_cast<const g16h8 *>(cur + kG16Img + oa);
        bh[i][kk] = *reinterpret_cast<const g16h8 *>(cur + 2 * kG16Img + ob);
        bl[i][kk] = *reinterpret_cast<const g16h8 *>(cur + 3 * kG16Img + ob);
      }
    }
    // term-major: the products of one accumulator are 4 MFMAs apart, never back to back
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[i][kk] : ah[i][kk],
                                                                term == 1 ? bl[jn][kk] : bh[jn][kk],
                                                                acc[i][jn], 0, 0, 0);
    g16_wait_dma();     // this wave's share of the next K step has landed ...
    __syncthreads();    // ... and everybody else's; the current buffer is free again
  }

  // epilogue: acc[i][jn][r] = C'[row = bm + wm*64 + i*32 + tile_row_of_reg(r, h)][col = bn + wn*64 + jn*32 + j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int col = bn + wn * 64 + jn * 32 + j;
      const int colc = col < g.n ? col : g.n - 1;
      const float cs = g.invb[colc];
      const float bias = g.bias ? g.bias[colc] : 0.0f;
      float ra[16], e0v[16], e1v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = bm + wm * 64 + i * 32 + tile_row_of_reg(r, h);
        const int64_t rowc = row < g.m ? row : g.m - 1;
        const int64_t o = rowc * g.n + colc;
        ra[r] = g.inva[rowc];
        e0v[r] = EPI != kG16EpiBias ? g.x0[o] : 0.0f;
        e1v[r] = EPI != kG16EpiBias ? g.x[o] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = bm + wm * 64 + i * 32 + tile_row_of_reg(r, h);
        const float v0 = acc[i][jn][r] * (ra[r] * cs) + bias;
        const float v = ACT ? act_apply(g.act, v0) : v0;
        float u = 0.0f;
        const float res = g16_epilogue_v<EPI>(v, e0v[r], e1v[r], g.diag, &u);
        if (row < g.m && col < g.n) {
          const int64_t o = row * g.n + col;
          g.out[o] = res;
          if (EPI == kG16EpiCross && g.aux) g.aux[o] = u;
          if (ACT && g.pre) g.pre[o] = v0;
        }
      }
    }
}

// ---- the large-shape GEMM: 256 x 256 tiles, 4-deep ring ---------------------------------------
// The 128 x 128 kernel above loads 32 KB per 24 MFMAs per wave and keeps one K step in flight:
// measured 1.9 us per step against 0.64 us of MFMA work -- it waits for memory.  This variant
// quarters the bytes per flop (256 x 256 tile, 8 waves as 4 x 2, each 64 x 128 = 2 x 4 MFMA tiles)
// and keeps three K steps of 16 in flight in a 4-buffer ring (128 KB of LDS, one workgroup per
// CU, two waves per SIMD).  LDS rows are 32 bytes (16 halves): slot s of row r holds k-slot
// s ^ ((r >> 3) & 1), which again gives the 16 lanes of a ds_read_b128 16 distinct bank groups.
constexpr int kB16M = 256, kB16N = 256, kB16K = 16, kB16Ring = 4;
// measurement builds only (tools/ab_variants.sh; wrong results): 1 = no copies after the prologue, 2 = no waits / barriers
// in the K loop, 4 = fragments read once, 8 = K loop only (no epilogue loads / stores beyond one value per lane),
// 16 = every copy re-reads the tile's first K step (same cache lines: no traffic behind the L2)
#ifndef TFRS_G16_ABLATE
#define TFRS_G16_ABLATE 0
#endif
#if TFRS_G16_ABLATE
#ifndef TFRS_ALLOW_ABLATION
#error "TFRS_G16_ABLATE != 0 is a measurement build with wrong results: add -DTFRS_ALLOW_ABLATION to confirm"
#endif
extern "C" int tfrs_ablation_build_g16(void) { return TFRS_G16_ABLATE; }
#endif
constexpr int kB16Img = 256 * 32;          // bytes of one 256-row x 16-half image
constexpr int kB16Stage = 4 * kB16Img;     // Ah | Al | Bh | Bl

// s_waitcnt vmcnt(N) lgkmcnt(0): at most N copies of this wave in flight, and all of its own LDS
// reads returned (so that the barrier that follows really hands the oldest buffer over)
template <int N>
__device__ __forceinline__ void g16_wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

template <int EPI, bool ACT = false>
__global__ void __launch_bounds__(512) gemm16_big_kernel(const Gemm16Args g) {
  __shared__ __attribute__((aligned(16))) char lds[kB16Ring * kB16Stage];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;     // 4 x 2 waves; wave tile 64 (M) x 128 (N)
  const int j = lane & 31, h = lane >> 5;

  const long long clk_c0 = g.clk ? clock64() : 0, clk_w0 = g.clk ? wall_clock64() : 0;
  const int nbn = (g.n + kB16N - 1) / kB16N;
  // Tile of this workgroup.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB
  // L2), so in launch order an XCD's 32 resident workgroups were scattered over ~18 row panels of the output and
  // shared almost nothing in their L2 (raster 0).  raster >= 1: every XCD gets a CONTIGUOUS range of tile ids
  // (the bijective remap of the top-K scan); raster 2 additionally walks that range in groups of 4 row panels x
  // all their column panels column by column, so that 32 consecutive tiles are 4 row panels x 8 column panels:
  // 12 operand panels feed 32 workgroups instead of ~17 (row-major) or ~50 (round-robin).
  int tile = (int)blockIdx.x;
  if (g.raster >= 1 && gridDim.y == 1) {
    const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = tile & 7, pos = tile >> 3;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
  }
  int tm = tile / nbn, tn = tile % nbn;
  if (g.raster == 2 && gridDim.y == 1) {
    constexpr int kGroupM = 4;
    const int nbm = ((int)gridDim.x + nbn - 1) / nbn;
    const int per_group = kGroupM * nbn;
    const int grp = tile / per_group, first = grp * kGroupM;
    const int rows = nbm - first < kGroupM ? nbm - first : kGroupM;
    const int in_grp = tile - grp * per_group;
    tm = first + in_grp % rows;
    tn = in_grp / rows;
  }
  const int64_t bm = (int64_t)tm * kB16M;
  const int bn = tn * kB16N;
  const int nk_all = g.kp / kB16K;
  const int kbeg = g.kt_per ? (int)blockIdx.y * g.kt_per : 0;
  const int nk = g.kt_per ? (nk_all - kbeg < g.kt_per ? nk_all - kbeg : g.kt_per) : nk_all;
  float *const outp = g.out + (g.kt_per ? (int64_t)blockIdx.y * g.m * g.n : 0);

  // staging: one instruction = 32 rows x 2 slots; wave w copies rows [32w, 32w + 32) of each image.  The images are
  // K-step-major (kb == kB16K, checked by the launcher): the operand tile of K step t is ONE contiguous 256 x 32-byte
  // range, block t of the image, so a lane's source address advances by one block pitch per step -- four 64-bit adds
  // per stage (round 5 recomputed block and remainder with a scalar division per stage: 63 SALU instructions per step).
  const int sr = lane >> 1, ss = lane & 1;
  const int r = wave * 32 + sr;
  const int ks = ss ^ ((r >> 3) & 1);
  const int64_t blk_a = g.mp * (int64_t)(kB16K * 2), blk_b = g.np * (int64_t)(kB16K * 2);   // bytes per K step
  // wave-uniform bases (SGPRs) of the tile's rows in the four images at the tile's first K step; one per-lane offset
  const char *sb[4];
  sb[0] = reinterpret_cast<const char *>(g.ah + bm * kB16K) + kbeg * blk_a;
  sb[1] = reinterpret_cast<const char *>(g.al + bm * kB16K) + kbeg * blk_a;
  sb[2] = reinterpret_cast<const char *>(g.bh + (int64_t)bn * kB16K) + kbeg * blk_b;
  sb[3] = reinterpret_cast<const char *>(g.bl + (int64_t)bn * kB16K) + kbeg * blk_b;
  // (readfirstlane: the values ARE uniform -- kernel arguments and workgroup ids -- but the compiler's divergence
  // analysis does not always see it through the tile remap, and an "s" operand it believes divergent does not assemble)
#pragma unroll
  for (int im = 0; im < 4; ++im) {
    const uint64_t a = reinterpret_cast<uint64_t>(sb[im]);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    sb[im] = reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
  }
  const uint32_t voff = (uint32_t)(r * (kB16K * 2) + ks * 16);
  // stage(kt): the copies of step kt; calls are made in increasing kt, each advancing the four bases
  auto stage = [&](int kt) __attribute__((always_inline)) {
    char *buf = lds + (kt & (kB16Ring - 1)) * kB16Stage + wave * 32 * 32;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
      g16_dma16_s(voff, sb[im], buf + im * kB16Img);
      if (!(TFRS_G16_ABLATE & 16)) sb[im] += im < 2 ? blk_a : blk_b;
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.0f;

  // every wave issues exactly 4 copies per K step, so "at most 4 * s copies in flight" means
  // "all but the newest s steps have landed"
  auto wait_step = [&](int kt) __attribute__((always_inline)) {
    const int newer = nk - 1 - kt;             // steps issued after kt (at most 2 matter)
    if (newer >= 2) g16_wait_vm<8>(); else if (newer == 1) g16_wait_vm<4>(); else g16_wait_vm<0>();
    __builtin_amdgcn_s_barrier();              // step kt is complete for everybody, and nobody
                                               // reads the buffer of step kt - 1 any more
  };
  struct Frags {
    g16h8 ah[2], al[2], bh[4], bl[4];
  };
  auto read_frags = [&](int kt, Frags &f) __attribute__((always_inline)) {
    const char *cur = lds + (kt & (kB16Ring - 1)) * kB16Stage;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + j;
      const int oa = ra * 32 + (h ^ ((ra >> 3) & 1)) * 16;
      f.ah[i] = *reinterpret_cast<const g16h8 *>(cur + oa);
      f.al[i] = *reinterpret_cast<const g16h8 *>(cur + kB16Img + oa);
    }
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int rb = wn * 128 + jn * 32 + j;
      const int ob = rb * 32 + (h ^ ((rb >> 3) & 1)) * 16;
      f.bh[jn] = *reinterpret_cast<const g16h8 *>(cur + 2 * kB16Img + ob);
      f.bl[jn] = *reinterpret_cast<const g16h8 *>(cur + 3 * kB16Img + ob);
    }
  };
  // term-major: the three products of one accumulator are 8 MFMAs apart, never back to back
  auto mfmas = [&](const Frags &f) __attribute__((always_inline)) {
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? f.al[i] : f.ah[i],
                                                              term == 1 ? f.bl[jn] : f.bh[jn],
                                                              acc[i][jn], 0, 0, 0);
  };
  // Register double buffer: the fragments of step kt + 1 are read from LDS BEFORE the MFMAs of
  // step kt are issued, so the LDS latency and the barrier skew of a step hide under a full
  // step of matrix work (reading them at the top of their own step left the two waves of a
  // SIMD waiting together after every barrier: MFMA busy 47 %).
  // half(): any step (conditions evaluated at run time) -- the first and the last few steps of a tile.
  auto half = [&](int kt, Frags &cur, Frags &nxt) __attribute__((always_inline)) {
    if (kt + 1 < nk) {
      if (!(TFRS_G16_ABLATE & 2)) wait_step(kt + 1);
      if (kt + kB16Ring < nk && !(TFRS_G16_ABLATE & 1)) stage(kt + kB16Ring);   // into the buffer of step kt (already in registers)
      if (!(TFRS_G16_ABLATE & 4)) read_frags(kt + 1, nxt); else nxt = cur;
    }
    mfmas(cur);
  };
  // steady(): a step with kt + kB16Ring < nk -- NO branch between the LDS reads and the MFMAs.  Round 5 ran every
  // step through half(): the join behind its `if` made the compiler wait for lgkmcnt(0) in front of the MFMAs,
  // i.e. for the twelve reads of the NEXT step's fragments issued one line above -- the register double buffer was
  // defeated and both waves of a SIMD sat out the LDS latency together after every barrier.  Straight-line code
  // lets it count: the MFMAs of step kt need the reads of the previous step only (lgkmcnt(12)).
  // The schedule of a steady step is pinned by hand (sched_barrier(0) between the groups; left to itself the compiler
  // sinks the twelve fragment reads behind the MFMAs, right in front of the next step's lgkmcnt(0)): the barrier is
  // followed by MFMAs at once, the copies of step kt + 4 and the reads of step kt + 1's fragments are issued in the
  // shadow of the matrix work, and they have all returned long before the wait that ends the step.
  auto mfma_n = [&](const Frags &f, int n) __attribute__((always_inline)) {   // n = term * 8 + i * 4 + jn
    const int term = n >> 3, i = (n >> 2) & 1, jn = n & 3;
    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? f.al[i] : f.ah[i], term == 1 ? f.bl[jn] : f.bh[jn],
                                                        acc[i][jn], 0, 0, 0);
  };
  auto steady = [&](int kt, Frags &cur, Frags &nxt) __attribute__((always_inline)) {
    if (!(TFRS_G16_ABLATE & 2)) {
      g16_wait_vm<8>();
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 4; ++n) mfma_n(cur, n);
    __builtin_amdgcn_sched_barrier(0);
    if (!(TFRS_G16_ABLATE & 1)) stage(kt + kB16Ring);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 4; n < 8; ++n) mfma_n(cur, n);
    __builtin_amdgcn_sched_barrier(0);
    const char *nb = lds + ((kt + 1) & (kB16Ring - 1)) * kB16Stage;
    if (!(TFRS_G16_ABLATE & 4)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + j;
        const int oa = ra * 32 + (h ^ ((ra >> 3) & 1)) * 16;
        nxt.ah[i] = *reinterpret_cast<const g16h8 *>(nb + oa);
        nxt.al[i] = *reinterpret_cast<const g16h8 *>(nb + kB16Img + oa);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 8; n < 12; ++n) mfma_n(cur, n);
    __builtin_amdgcn_sched_barrier(0);
    if (!(TFRS_G16_ABLATE & 4)) {
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int rb = wn * 128 + jn * 32 + j;
        const int ob = rb * 32 + (h ^ ((rb >> 3) & 1)) * 16;
        nxt.bh[jn] = *reinterpret_cast<const g16h8 *>(nb + 2 * kB16Img + ob);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 12; n < 16; ++n) mfma_n(cur, n);
    __builtin_amdgcn_sched_barrier(0);
    if (!(TFRS_G16_ABLATE & 4)) {
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int rb = wn * 128 + jn * 32 + j;
        const int ob = rb * 32 + (h ^ ((rb >> 3) & 1)) * 16;
        nxt.bl[jn] = *reinterpret_cast<const g16h8 *>(nb + 3 * kB16Img + ob);
      }
    } else {
      nxt = cur;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 16; n < 24; ++n) mfma_n(cur, n);
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int p = 0; p < kB16Ring; ++p)
    if (p < nk) stage(p);
  Frags fa, fb;
  {
    const int newer = nk - 1;                  // steps 1..3 may still be in flight
    if (newer >= 3) g16_wait_vm<12>(); else if (newer == 2) g16_wait_vm<8>();
    else if (newer == 1) g16_wait_vm<4>(); else g16_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
  }
  read_frags(0, fa);
  int kt = 0;
  for (; kt + kB16Ring + 1 < nk; kt += 2) {    // both steps of the pair are steady ones
    steady(kt, fa, fb);
    steady(kt + 1, fb, fa);
  }
  for (; kt < nk; kt += 2) {                   // the last four or five steps
    half(kt, fa, fb);
    if (kt + 1 < nk) half(kt + 1, fb, fa);
  }
  if (TFRS_G16_ABLATE & 1) g16_wait_vm<0>();
  if (TFRS_G16_ABLATE & 8) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn)
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += acc[i][jn][q];
    if (sum == 123.456f) outp[tid] = sum;
    if (g.clk && tid == 0) {
      atomicAdd(g.clk, (unsigned long long)(clock64() - clk_c0));
      atomicAdd(g.clk + 1, (unsigned long long)(wall_clock64() - clk_w0));
    }
    return;
  }

  // Epilogue addressing: one 64-bit base per array for the tile (uniform) + 32-bit per-lane offsets (a 256-row
  // tile spans < 2^31 elements for n < 2^23, checked by the launcher): sixteen 64-bit addresses per batch next
  // to the 128 accumulators spilled.
  auto clocks_out = [&]() __attribute__((always_inline)) {
    if (g.clk && tid == 0) {
      atomicAdd(g.clk, (unsigned long long)(clock64() - clk_c0));
      atomicAdd(g.clk + 1, (unsigned long long)(wall_clock64() - clk_w0));
    }
  };
  const int64_t tile0 = bm * (int64_t)g.n + bn;
  const float *const x0b = EPI != kG16EpiBias ? g.x0 + tile0 : nullptr;
  const float *const xb = EPI != kG16EpiBias ? g.x + tile0 : nullptr;
  float *const outb = outp + tile0;
  float *const auxb = (EPI == kG16EpiCross && g.aux) ? g.aux + tile0 : nullptr;
  const int rows_here = (int)(g.m - bm < kB16M ? g.m - bm : kB16M);   // >= 1
  const int cols_here = g.n - bn < kB16N ? g.n - bn : kB16N;           // >= 1
  // (Round 6 measured a wide form of this epilogue -- accumulator tiles transposed through the free stage ring so that
  // every global access is a dwordx4, loads of the next half batch issued ahead -- bit-identical and NOT faster: Cross
  // forward 4.39 ms against 4.27 ms scalar, training pair 14.47 against 14.40 (profiles/r06_gemm16_ab.txt); its two
  // register sets next to the accumulators spilled 63-80 registers in every Cross instantiation.  Removed: the epilogue's
  // cost is its HBM traffic with the matrix pipe of the CU idle, not its instruction count.)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int lc = wn * 128 + jn * 32 + j;                           // column within the tile
      const int lcc = lc < cols_here ? lc : cols_here - 1;
      const float cs = g.invb[bn + lcc];
      const float bias = g.bias ? g.bias[bn + lcc] : 0.0f;
      // (two batches of eight rows; sixteen at once -- 48 values + 16 offsets next to the 128 accumulators -- spill
      // 9-11 registers in the Cross instantiations even with 32-bit offsets, 50 with 64-bit addresses)
#pragma unroll
      for (int q0 = 0; q0 < 16; q0 += 8) {
        float ra[8], e0v[8], e1v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int lr = wm * 64 + i * 32 + tile_row_of_reg(q0 + q, h);
          const int lrc = lr < rows_here ? lr : rows_here - 1;
          const uint32_t o = (uint32_t)lrc * (uint32_t)g.n + (uint32_t)lcc;
          ra[q] = g.inva[bm + lrc];
          e0v[q] = EPI != kG16EpiBias ? x0b[o] : 0.0f;
          e1v[q] = EPI != kG16EpiBias ? xb[o] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int lr = wm * 64 + i * 32 + tile_row_of_reg(q0 + q, h);
          const float v0 = acc[i][jn][q0 + q] * (ra[q] * cs) + bias;
          const float v = ACT ? act_apply(g.act, v0) : v0;
          float u = 0.0f;
          const float res = g16_epilogue_v<EPI>(v, e0v[q], e1v[q], g.diag, &u);
          if (lr < rows_here && lc < cols_here) {
            const uint32_t o = (uint32_t)lr * (uint32_t)g.n + (uint32_t)lc;
            outb[o] = res;
            if (EPI == kG16EpiCross && auxb) auxb[o] = u;
            if (ACT && g.pre) g.pre[(int64_t)tile0 + o] = v0;
          }
        }
      }
    }
  clocks_out();
}

// ---- host side ------------------------------------------------------------------------------
static inline size_t g16_al(size_t x) { return (x + 255) / 256 * 256; }
static inline int64_t g16_pad(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

struct G16Layout {
  int64_t mp, np;
  int kp;
  size_t ah, al, bh, bl, inva, invb, colmax, colmax_a, psum, part, clk, total;
  int nsplit;
};

// Split-K for long-K products whose 256 x 256 output tiles do not fill the chip (dW = x^T dz:
// 14 x 14 tiles at d = 3456, K = batch = 65536): the K range is cut into slices so that
// tiles x slices covers the 256 CUs a whole number of times (within 10 %), every slice writes its
// partial product, a second kernel sums the slices in fixed order.
static int g16_splits(int64_t m, int n, int k) {
  const int64_t tiles = ((m + kB16M - 1) / kB16M) * ((n + kB16N - 1) / kB16N);
  if (tiles >= 512 || k < 8192) return 1;
  const char *ev = option("TFRS_GEMM16_SPLITK");
  if (ev && *ev) return std::max(1, atoi(ev));
  // slices of at least 2048 K rows; the number that fills whole rounds of the chip's 256 workgroup slots best -- or, for
  // products of a handful of tiles (round 6: the 512 x 256 weight gradient of the bottom MLP at batch 131072 is TWO
  // tiles; unsplit it ran on 8 workgroups of the 128 x 128 kernel for 4.2 ms of a 31 ms DLRM step), as much of one
  // round as the K length allows
  int best = 1;
  double best_eff = 0.0;
  const int smax = tiles < 8 ? 128 : 16;
  for (int s = 2; s <= smax; ++s) {
    if (k / s < 2048) break;
    const double waves = (double)(tiles * s) / 256.0;
    const double eff = waves <= 1.0 ? waves : waves / (double)(int64_t)(waves + 0.999999);
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  return best;
}

static G16Layout g16_layout(int64_t m, int n, int k) {
  G16Layout L;
  L.mp = g16_pad(m, kB16M);   // both kernels' tiles divide 256
  L.np = g16_pad(n, kB16N);
  L.kp = (int)g16_pad(k, 64);   // 64: the column prep writes whole 64-k tiles
  size_t o = 0;
  const size_t ia = g16_al((size_t)L.mp * L.kp * 2), ib = g16_al((size_t)L.np * L.kp * 2);
  L.ah = o; o += ia; L.al = o; o += ia;
  L.bh = o; o += ib; L.bl = o; o += ib;
  L.inva = o; o += g16_al((size_t)L.mp * 4);
  L.invb = o; o += g16_al((size_t)L.np * 4);
  L.colmax = o; o += g16_al((size_t)L.np * 4);
  L.colmax_a = o; o += g16_al((size_t)L.mp * 4);
  L.psum = o; o += g16_al((size_t)((k + kG16Slab - 1) / kG16Slab) * L.np * 4);
  L.nsplit = g16_splits(m, n, k);
  L.part = o; o += L.nsplit > 1 ? g16_al((size_t)L.nsplit * m * n * 4) : 0;
  L.clk = o; o += 256;      // measurement: {cycles, ticks} sums of the last big-kernel launch (TFRS_GEMM16_CLOCKS)
  L.total = o;
  return L;
}

size_t gemm16_workspace_bytes(int64_t m, int n, int k) { return g16_layout(m, n, k).total; }

// One operand of the product C[M, N] = A[M, K] @ B[K, N].  `t` = the array holds the TRANSPOSE
// (A given as [K, M], B given as [N, K]); `mul` = optional elementwise multiplier of the same
// shape/layout.  Either way the images end up K-contiguous per output row / column, so no
// transposed copy of an activation is ever materialised (dW = x^T dz reads x and dz as they are).
struct G16Operand {
  const float *p;
  const float *mul;
  bool t;
};

template <int EPI, bool ACT = false>
static void g16_launch(const Gemm16Args &g, bool big, hipStream_t s) {
  if (big) {
    const dim3 grid((unsigned)(((g.m + kB16M - 1) / kB16M) * ((g.n + kB16N - 1) / kB16N)));
    hipLaunchKernelGGL((gemm16_big_kernel<EPI, ACT>), grid, dim3(512), 0, s, g);
  } else {
    // tiles that hold at least one real row and column (the kernel derives its tile
    // coordinates from ceil(n / 128), not from the 256-padded image sizes)
    const dim3 grid((unsigned)(((g.m + kG16M - 1) / kG16M) * ((g.n + kG16N - 1) / kG16N)));
    hipLaunchKernelGGL((gemm16_kernel<EPI, ACT>), grid, dim3(256), 0, s, g);
  }
}

// C = A @ B (+ bias) with epilogue `epi` on (e0, e1, diag); colsum (optional, B not transposed):
// colsum[n] = sum_k (B * mul)[k, n].  ws from gemm16_workspace_bytes(m, n, k).
int gemm16_run_ex(const G16Operand &a, const G16Operand &b, int64_t m, int n, int k,
                  const float *bias, int epi, const float *e0, const float *e1, float diag,
                  float *out, float *colsum, void *ws, hipStream_t s, float *aux = nullptr, int act = 0,
                  float *pre = nullptr) {
  const G16Layout L = g16_layout(m, n, k);
  char *w = static_cast<char *>(ws);
  _Float16 *ah = reinterpret_cast<_Float16 *>(w + L.ah), *al = reinterpret_cast<_Float16 *>(w + L.al);
  _Float16 *bh = reinterpret_cast<_Float16 *>(w + L.bh), *bl = reinterpret_cast<_Float16 *>(w + L.bl);
  float *inva = reinterpret_cast<float *>(w + L.inva), *invb = reinterpret_cast<float *>(w + L.invb);
  uint32_t *colmax = reinterpret_cast<uint32_t *>(w + L.colmax);
  uint32_t *colmax_a = reinterpret_cast<uint32_t *>(w + L.colmax_a);
  float *psum = reinterpret_cast<float *>(w + L.psum);
  const unsigned nslab = (unsigned)((k + kG16Slab - 1) / kG16Slab);
  // kernel choice first: it decides the image layout
  const char *tv = option("TFRS_GEMM16_TILE");
  const int forced = (tv && *tv) ? atoi(tv) : 0;
  const int64_t big_tiles = (L.mp / kB16M) * (L.np / kB16N);
  const bool splitk = L.nsplit > 1 && epi == kG16EpiBias && !bias && !act && !pre && forced != 128;
  const bool big = (forced == 256 || (forced != 128 && big_tiles >= 512) || splitk) && n < (1 << 23);   // (32-bit tile offsets in its epilogue)
  // the 256 x 256 kernel reads K-step-major images (kb = 16 halves): the operand tile of one K
  // step is 256 rows x 32 bytes CONTIGUOUS, so every direct-to-LDS copy instruction moves eight
  // full 128-byte lines instead of 32 quarter lines 2 * kp bytes apart
  const int kb = big ? kB16K : L.kp;
  // column maxima are combined with atomicMax: re-armed by a kernel, not a memset node
  hipLaunchKernelGGL(g16_zero_kernel, dim3((unsigned)((L.np + L.mp + 255) / 256)), dim3(256), 0, s,
                     colmax, (int)(L.np + L.mp));   // colmax and colmax_a are adjacent
  if (!a.t && kb == kB16K) {
    g16_launch_prep_rows_ksm(a.p, a.mul, m, k, L.kp, ah, al, inva, L.mp, s);
  } else if (!a.t) {
    hipLaunchKernelGGL(g16_prep_rows_kernel, dim3((unsigned)(L.mp / 4)), dim3(256), 0, s, a.p, a.mul, m, k,
                       L.kp, ah, al, inva, colmax, 0, kb, L.mp);
  } else {   // a.p is [K, M]: image row i = column i of the array
    hipLaunchKernelGGL(g16_colmax_kernel, dim3((unsigned)(L.mp / 64), nslab), dim3(256), 0, s, a.p, a.mul,
                       k, (int)m, colmax_a, (float *)nullptr);
    hipLaunchKernelGGL(g16_prep_cols_kernel, dim3((unsigned)(L.mp / 64), (unsigned)(L.kp / 64)), dim3(256),
                       0, s, a.p, a.mul, k, (int)m, L.kp, colmax_a, ah, al, inva, kb, L.mp);
  }
  if (!b.t) {
    hipLaunchKernelGGL(g16_colmax_kernel, dim3((unsigned)(L.np / 64), nslab), dim3(256), 0, s, b.p, b.mul,
                       k, n, colmax, colsum ? psum : (float *)nullptr);
    hipLaunchKernelGGL(g16_prep_cols_kernel, dim3((unsigned)(L.np / 64), (unsigned)(L.kp / 64)), dim3(256),
                       0, s, b.p, b.mul, k, n, L.kp, colmax, bh, bl, invb, kb, L.np);
    if (colsum)
      hipLaunchKernelGGL(g16_colsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, psum,
                         (int)nslab, n, colsum);
  } else if (kb == kB16K) {   // b.p is [N, K]: already one row per output column
    g16_launch_prep_rows_ksm(b.p, b.mul, (int64_t)n, k, L.kp, bh, bl, invb, L.np, s);
  } else {
    hipLaunchKernelGGL(g16_prep_rows_kernel, dim3((unsigned)(L.np / 4)), dim3(256), 0, s, b.p, b.mul,
                       (int64_t)n, k, L.kp, bh, bl, invb, colmax, 0, kb, L.np);
  }
  TFRS_LAUNCH_CHECK();
  Gemm16Args g = {};
  g.ah = ah; g.al = al; g.bh = bh; g.bl = bl; g.inva = inva; g.invb = invb;
  g.m = m; g.n = n; g.kp = L.kp;
  g.bias = bias; g.x0 = e0; g.x = e1; g.diag = diag; g.out = out; g.aux = aux;
  g.kb = kb; g.mp = L.mp; g.np = L.np;
  g.act = act; g.pre = pre;
  {
    const char *cv = option("TFRS_GEMM16_CLOCKS");
    g.clk = nullptr;
    if (cv && cv[0] == '1' && big) {
      g.clk = reinterpret_cast<unsigned long long *>(w + L.clk);
      hipLaunchKernelGGL(g16_zero_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<uint32_t *>(g.clk), 4);
    }
  }
  {
    const char *rv = option("TFRS_GEMM16_RASTER");
    g.raster = (rv && *rv) ? atoi(rv) : 1;   // (measured: 0 / 1 / 2 within 0.5 % on the Cross products, 1 and 2
                                             // 2.4 % ahead on the DLRM top MLP -- profiles/r05_gemm_raster.jsonl)
  }
  // large shapes: 256 x 256 tiles with the 4-deep ring; otherwise (few tiles: fill the chip)
  // the 128 x 128 kernel.  TFRS_GEMM16_TILE = 128 | 256 forces one.
  if (splitk) {
    const int nk_all = L.kp / kB16K;
    g.kt_per = (nk_all + L.nsplit - 1) / L.nsplit;
    const int slices = (nk_all + g.kt_per - 1) / g.kt_per;
    g.out = reinterpret_cast<float *>(w + L.part);
    const dim3 grid((unsigned)(((m + kB16M - 1) / kB16M) * ((n + kB16N - 1) / kB16N)), (unsigned)slices);
    hipLaunchKernelGGL((gemm16_big_kernel<kG16EpiBias>), grid, dim3(512), 0, s, g);
    const int64_t count = m * (int64_t)n;
    hipLaunchKernelGGL(g16_splitk_reduce_kernel, dim3((unsigned)((count / 4 + 256) / 256)), dim3(256), 0, s,
                       g.out, slices, count, out);
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  if (act || pre) {     // (only the forward products take an activation: Dense and Cross)
    if (epi == kG16EpiCross) g16_launch<kG16EpiCross, true>(g, big, s);
    else g16_launch<kG16EpiBias, true>(g, big, s);
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  switch (epi) {
    case kG16EpiCross: g16_launch<kG16EpiCross>(g, big, s); break;
    case kG16EpiCrossDx0: g16_launch<kG16EpiCrossDx0>(g, big, s); break;
    case kG16EpiCrossDx: g16_launch<kG16EpiCrossDx>(g, big, s); break;
    default: g16_launch<kG16EpiBias>(g, big, s); break;
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// C = A @ B (+ bias) [cross epilogue when x0 != NULL]; ws from gemm16_workspace_bytes
int gemm16_run(const float *a, const float *b, int64_t m, int n, int k, const float *bias,
               const float *x0, const float *x, float diag, float *out, void *ws, hipStream_t s,
               float *aux) {
  return gemm16_run_ex({a, nullptr, false}, {b, nullptr, false}, m, n, k, bias,
                       x0 ? kG16EpiCross : kG16EpiBias, x0, x, diag, out, nullptr, ws, s, x0 ? aux : nullptr);
}

// out = epilogue(act(a @ b + bias)), optionally storing the pre-activation: Dense with a fused activation
// (x0 == NULL) or Cross with a preactivation / low-rank input (x0, x given): see tfrs_dense_fwd_act / tfrs_cross_fwd_act
int gemm16_run_act(const float *a, const float *b, int64_t m, int n, int k, const float *bias, int act,
                   const float *x0, const float *x, float diag, float *out, float *pre, void *ws,
                   hipStream_t s) {
  return gemm16_run_ex({a, nullptr, false}, {b, nullptr, false}, m, n, k, bias,
                       x0 ? kG16EpiCross : kG16EpiBias, x0, x, diag, out, nullptr, ws, s, nullptr, act, pre);
}

// Cross backward (layers/feature_interaction/dcn.py:151-186 under models/base.py:77), full rank,
// linear preactivation.  With z = x W + b + diag x and dz = dy * x0:
//   dx0 = dy * z                     product 1: x @ W,     epilogue CrossDx0 (z never stored)
//   dx  = dz W^T + dy + diag * dz    product 2: dz @ W^T,  B = W read as [N, K] (no transpose copy),
//                                    A image built from dy * x0 on the fly, epilogue CrossDx
//   dW  = x^T dz                     product 3: both operands read K-major as they lie in HBM
//   db  = column sums of dz          by-product of product 3's column-maximum pass
// One workspace serves the three products in turn.
size_t gemm16_cross_bwd_workspace_bytes(int64_t batch, int d) {
  return std::max(g16_layout(batch, d, d).total, g16_layout(d, d, (int)batch).total);
}

// out = a * b (+ c) (+ e): the addends are optional.  c may alias out (accumulation in place: same index read and written
// by the same thread).
__global__ void __launch_bounds__(256) g16_mul_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      int64_t count, float *out, const float *c, const float *e) {
  const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) |
                     reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(e)) & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    for (; i * 4 + 3 < count; i += stride) {
      const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
      float4 r = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
      if (c) {
        const float4 z = reinterpret_cast<const float4 *>(c)[i];
        r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w;
      }
      if (e) {
        const float4 z = reinterpret_cast<const float4 *>(e)[i];
        r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w;
      }
      reinterpret_cast<float4 *>(out)[i] = r;
    }
    for (int64_t t = (count & ~(int64_t)3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride)
      out[t] = a[t] * b[t] + (c ? c[t] : 0.0f) + (e ? e[t] : 0.0f);
  } else {
    for (; i < count; i += stride) out[i] = a[i] * b[i] + (c ? c[i] : 0.0f) + (e ? e[i] : 0.0f);
  }
}

// `u` (optional) = x W + b + diag x saved by the training forward (tfrs_cross_fwd_f16_train): the
// first product is then replaced by the elementwise dx0 = dy * u.
// dx0_add (with u only; may alias dx0): dx0 = dy * u + dx0_add -- a STACK of Cross layers on one x0 (dcn.py:47-56)
// accumulates x0's gradient across its layers in place instead of leaving 906 MB additions to the autograd engine;
// add_dx != 0 (first layer of such a stack, where x IS x0): dx0 = dy * u + dx0_add + dx, formed after the dx product.
int gemm16_cross_bwd(const float *x0, const float *x, const float *kernel, const float *bias,
                     float diag, const float *dy, int64_t batch, int d, float *dx0, float *dx,
                     float *dkernel, float *dbias, void *ws, hipStream_t s, const float *u, const float *dx0_add,
                     int add_dx) {
  int rc = TFRS_OK;
  const int64_t count = batch * (int64_t)d;
  const dim3 mgrid((unsigned)std::min<int64_t>((count / 4 + 255) / 256 + 1, 256 * 32));
  if (u) {
    if (!add_dx) {
      hipLaunchKernelGGL(g16_mul_kernel, mgrid, dim3(256), 0, s, dy, u, count, dx0, dx0_add, (const float *)nullptr);
      TFRS_LAUNCH_CHECK();
    }
  } else {
    rc = gemm16_run_ex({x, nullptr, false}, {kernel, nullptr, false}, batch, d, d, bias,
                       kG16EpiCrossDx0, dy, x, diag, dx0, nullptr, ws, s);
  }
  if (rc != TFRS_OK) return rc;
  rc = gemm16_run_ex({dy, x0, false}, {kernel, nullptr, true}, batch, d, d, nullptr, kG16EpiCrossDx, dy,
                     x0, diag, dx, nullptr, ws, s);
  if (rc != TFRS_OK) return rc;
  if (u && add_dx) {
    hipLaunchKernelGGL(g16_mul_kernel, mgrid, dim3(256), 0, s, dy, u, count, dx0, dx0_add, (const float *)dx);
    TFRS_LAUNCH_CHECK();
  }
  return gemm16_run_ex({x, nullptr, true}, {dy, x0, false}, d, d, (int)batch, nullptr, kG16EpiBias, nullptr,
                       nullptr, 0.0f, dkernel, dbias, ws, s);
}


// scores[nq, nc] = q @ c^T, c read as the transposed operand
int gemm16_scores(const float *q, const float *c, int64_t nq, int nc, int d, float *out, void *ws,
                  hipStream_t s) {
  return gemm16_run_ex({q, nullptr, false}, {c, nullptr, true}, nq, nc, d, nullptr, kG16EpiBias, nullptr,
                       nullptr, 0.0f, out, nullptr, ws, s);
}

size_t gemm16_dense_bwd_workspace_bytes(int64_t batch, int din, int dout) {
  return std::max(g16_layout(batch, din, dout).total, g16_layout(din, dout, (int)batch).total);
}

// dx = dy W^T (W read as [N = din, K = dout]), dW = x^T dy, db = column sums of dy
// (addend != NULL: dx = dy W^T + addend, through the CrossDx epilogue with diag = 0)
int gemm16_dense_bwd(const float *x, const float *kernel, const float *dy, int64_t batch, int din,
                     int dout, float *dx, float *dkernel, float *dbias, void *ws, hipStream_t s,
                     const float *addend) {
  int rc = TFRS_OK;
  if (dx)
    rc = gemm16_run_ex({dy, nullptr, false}, {kernel, nullptr, true}, batch, din, dout, nullptr,
                       addend ? kG16EpiCrossDx : kG16EpiBias, addend, addend, 0.0f, dx, nullptr, ws, s);
  if (rc != TFRS_OK) return rc;
  if (dkernel)
    return gemm16_run_ex({x, nullptr, true}, {dy, nullptr, false}, din, dout, (int)batch, nullptr,
                         kG16EpiBias, nullptr, nullptr, 0.0f, dkernel, dbias, ws, s);
  if (dbias) {   // bias gradient alone: the column pass of the B operand, no product
    const G16Layout L = g16_layout(din, dout, (int)batch);
    char *w = static_cast<char *>(ws);
    uint32_t *colmax = reinterpret_cast<uint32_t *>(w + L.colmax);
    float *psum = reinterpret_cast<float *>(w + L.psum);
    const unsigned nslab = (unsigned)((batch + kG16Slab - 1) / kG16Slab);
    hipLaunchKernelGGL(g16_zero_kernel, dim3((unsigned)((L.np + 255) / 256)), dim3(256), 0, s, colmax, (int)L.np);
    hipLaunchKernelGGL(g16_colmax_kernel, dim3((unsigned)(L.np / 64), nslab), dim3(256), 0, s, dy,
                       (const float *)nullptr, (int)batch, dout, colmax, psum);
    hipLaunchKernelGGL(g16_colsum_kernel, dim3((unsigned)((dout + 255) / 256)), dim3(256), 0, s, psum,
                       (int)nslab, dout, dbias);
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}

}  // namespace tfrs
