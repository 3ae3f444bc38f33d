// topk_pack.hip -- row-major candidates[n, d]  <->  packed MFMA/LDS layout (common.h).
// HBM-bound copy kernels: one thread moves one 16-byte slot; a wave therefore writes
// 1 KiB of contiguous packed rows per instruction.
#include <algorithm>

#include "common.h"

namespace tfrs {

__global__ void __launch_bounds__(256) pack_kernel(const float *__restrict__ cand,
                                                   int64_t n, int d, int dp,
                                                   char *__restrict__ packed,
                                                   int64_t dst_row, int64_t total_rows,
                                                   int64_t mul, int64_t add,
                                                   int32_t *__restrict__ rowmap,
                                                   uint32_t *__restrict__ flags) {
  const int slots = dp / 4;
  const int half = dp / 8;       // slots per plane
  const int64_t nslots = total_rows * slots;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nslots;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / slots;
    const int s = (int)(t - r * slots);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t src = (rowmap && r < n) ? (mul * r + add) % n : r;   // shuffled block
    if (rowmap && r < n && s == 0) rowmap[dst_row + r] = (int32_t)(dst_row + src);
    if (r < n && s < 2 * half) {
      const int h = s / half;  // 0: even features, 1: odd features
      const int m = s - h * half;
      const float *row = cand + src * d;
      const int k0 = 8 * m + h;  // features k0, k0+2, k0+4, k0+6
      v.x = (k0 < d) ? row[k0] : 0.f;
      v.y = (k0 + 2 < d) ? row[k0 + 2] : 0.f;
      v.z = (k0 + 4 < d) ? row[k0 + 4] : 0.f;
      v.w = (k0 + 6 < d) ? row[k0 + 6] : 0.f;
      // non-finite candidates (NaN / Inf: exponent bits all ones) void the filter's error bound for their whole stage:
      // flagged once per offending slot into the index's host-visible flag word (bit 0), BruteForce.index raises
      if (flags && (nonfinite_bits(v.x) | nonfinite_bits(v.y) | nonfinite_bits(v.z) | nonfinite_bits(v.w)))
        atomicOr(flags, kNonfiniteCandidates);
    }
    *reinterpret_cast<float4 *>(packed + (dst_row + r) * (int64_t)row_bytes(dp) +
                                (int64_t)s * 16) = v;
  }
}

__global__ void __launch_bounds__(256) unpack_kernel(const char *__restrict__ packed,
                                                     int64_t n, int d, int dp,
                                                     const int32_t *__restrict__ rowmap,
                                                     float *__restrict__ out) {
  const int half = dp / 8;
  const int64_t total = n * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / d;
    const int k = (int)(t - r * d);
    const int h = k & 1, s = k >> 1;  // plane, position inside the plane
    const float *row = reinterpret_cast<const float *>(packed + r * (int64_t)row_bytes(dp));
    out[(rowmap ? (int64_t)rowmap[r] : r) * d + k] = row[h * half * 4 + s];
  }
}

// Packs cand[n, d] into rows dst_row .. dst_row+n-1 and zero-fills rows up to
// zero_rows_to (exclusive, >= dst_row + n) so that whole kTileN stages can be read.
static int64_t gcd64(int64_t a, int64_t b) {
  while (b) { const int64_t t = a % b; a = b; b = t; }
  return a;
}

int launch_pack(const float *cand, int64_t n, int d, char *packed, int64_t dst_row,
                int64_t zero_rows_to, int32_t *rowmap, hipStream_t stream, uint32_t *flags) {
  const int dp = padded_dim(d);
  const int64_t total_rows = zero_rows_to - dst_row;
  if (total_rows <= 0) return TFRS_OK;
  const int64_t nslots = total_rows * (dp / 4);
  int64_t blocks = (nslots + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  int64_t mul = 1, add = 0;
  if (rowmap && n > 1) {   // multiplicative shuffle: stride ~ n / golden ratio, coprime to n
    mul = (int64_t)((double)n * 0.6180339887498949) | 1;
    while (gcd64(mul, n) != 1) mul += 2;
    add = n / 3;
  }
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, cand, n, d,
                     dp, packed, dst_row, total_rows, mul, add, rowmap, flags);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_unpack(const char *packed, int64_t n, int d, const int32_t *rowmap, float *out,
                  hipStream_t stream) {
  if (n <= 0) return TFRS_OK;
  const int dp = padded_dim(d);
  int64_t blocks = (n * d + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, packed, n,
                     d, dp, rowmap, out);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// ---- fp16 prefilter image (common.h) ------------------------------------------------
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  union {
    f16x2_t h;
    uint32_t u;
  } v;
  v.h[0] = (_Float16)lo;  // v_cvt_pk_f16_f32, round to nearest even
  v.h[1] = (_Float16)hi;
  return v.u;
}

// One workgroup (256 threads) per stage of kTileN rows, reading the f32 image:
//   pass 1: threads 0..127 own one row each -> row norm, row max |x|; block max of both;
//   pass 2: every thread converts 16-byte slots (8 halves, natural feature order) of x / scale.
// Rows >= the valid row count are zero in the f32 image and stay zero here.
__global__ void __launch_bounds__(256) pack16_stage_kernel(const char *__restrict__ packed, int dp,
                                                           int dp16, int64_t stage0,
                                                           char *__restrict__ packed16,
                                                           StageMeta *__restrict__ meta,
                                                           float *__restrict__ norm_max) {
  __shared__ float s_norm[kTileN], s_amax[kTileN];
  __shared__ float s_scale;
  const int64_t stage = stage0 + blockIdx.x;
  const int64_t row0 = stage * kTileN;
  const int tid = threadIdx.x;
  {
    // two threads per row (one per parity plane), 16-byte loads
    const int r = tid >> 1, pl = tid & 1;
    const float4 *plane = reinterpret_cast<const float4 *>(packed + (row0 + r) * (int64_t)row_bytes(dp)) +
                          pl * (dp / 8);
    float ssq = 0.0f, amax = 0.0f;
    for (int m = 0; m < dp / 8; ++m) {
      const float4 v = plane[m];
      ssq = __builtin_fmaf(v.x, v.x, ssq);
      ssq = __builtin_fmaf(v.y, v.y, ssq);
      ssq = __builtin_fmaf(v.z, v.z, ssq);
      ssq = __builtin_fmaf(v.w, v.w, ssq);
      amax = fmaxf(fmaxf(amax, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))),
                   fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
    }
    ssq += __shfl_xor(ssq, 1);
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    if (pl == 0) {
      s_norm[r] = __builtin_sqrtf(ssq) * kNormSlack;
      s_amax[r] = amax;
    }
  }
  __syncthreads();
  if (tid < 64) {
    float nm = fmaxf(s_norm[tid], s_norm[tid + 64]);
    float am = fmaxf(s_amax[tid], s_amax[tid + 64]);
    for (int off = 32; off > 0; off >>= 1) {
      nm = fmaxf(nm, __shfl_xor(nm, off));
      am = fmaxf(am, __shfl_xor(am, off));
    }
    if (tid == 0) {
      const float sc = pow2_ceil(am);
      s_scale = sc;
      meta[stage].norm = nm;
      meta[stage].scale = sc;
      meta[stage].inv_scale = 1.0f / sc;
      meta[stage].pad_ = 0.0f;
      atomicMax(reinterpret_cast<uint32_t *>(norm_max), __float_as_uint(nm));  // nm >= 0
    }
  }
  __syncthreads();
  const float inv = 1.0f / s_scale;  // exact: power of two
  const int slots = dp16 / 8 + 1;
  for (int t = tid; t < kTileN * slots; t += 256) {
    const int r = t / slots;
    const int s = t - r * slots;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (s < slots - 1 && 8 * s < dp) {
      // features 8s .. 8s+7 = float4 #s of the even plane interleaved with float4 #s of the odd one
      const float4 *ev = reinterpret_cast<const float4 *>(packed + (row0 + r) * (int64_t)row_bytes(dp));
      const float4 e = ev[s], o = ev[dp / 8 + s];
      w[0] = pack_f16x2(e.x * inv, o.x * inv);
      w[1] = pack_f16x2(e.y * inv, o.y * inv);
      w[2] = pack_f16x2(e.z * inv, o.z * inv);
      w[3] = pack_f16x2(e.w * inv, o.w * inv);
    }
    *reinterpret_cast<uint4 *>(packed16 + (row0 + r) * (int64_t)row_bytes16(dp16) + (int64_t)s * 16) =
        make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// The same stage for dp >= 16 with every byte of the f32 image read ONCE: thread (row r = tid / 2, parity plane
// pl = tid & 1) holds its plane in registers (dp / 8 16-byte loads in flight; the kernel above walked them
// with a runtime loop, one memory round trip per piece, and read the stage a second time for the conversion:
// 2.3 TB/s, 4.4 ms per 12.5 M x 128), converts it after the workgroup has agreed on the scale, swaps half of
// its packed halves with the row's other thread (feature 2i sits in the even plane, 2i + 1 in the odd one; word i
// of the image is the pair) and stores dp bytes of the row.
template <int DP>
__global__ void __launch_bounds__(256) pack16_stage_regs_kernel(const char *__restrict__ packed, int64_t stage0,
                                                                char *__restrict__ packed16,
                                                                StageMeta *__restrict__ meta,
                                                                float *__restrict__ norm_max) {
  static_assert(DP >= 16, "dp = 8 keeps pack16_stage_kernel (its image row is padded to 16 halves)");
  constexpr int N4 = DP / 8;    // float4 per plane
  constexpr int NW = DP / 4;    // packed fp16 pairs per plane = image words this thread stores
  __shared__ float s_norm[kTileN], s_amax[kTileN];
  __shared__ float s_scale;
  const int64_t stage = stage0 + blockIdx.x;
  const int64_t row0 = stage * kTileN;
  const int tid = threadIdx.x;
  const int r = tid >> 1, pl = tid & 1;
  const float4 *plane = reinterpret_cast<const float4 *>(packed + (row0 + r) * (int64_t)row_bytes(DP)) + pl * N4;
  float4 v[N4];
#pragma unroll
  for (int m = 0; m < N4; ++m) v[m] = plane[m];
  float ssq = 0.0f, amax = 0.0f;
#pragma unroll
  for (int m = 0; m < N4; ++m) {
    ssq = __builtin_fmaf(v[m].x, v[m].x, ssq);
    ssq = __builtin_fmaf(v[m].y, v[m].y, ssq);
    ssq = __builtin_fmaf(v[m].z, v[m].z, ssq);
    ssq = __builtin_fmaf(v[m].w, v[m].w, ssq);
    amax = fmaxf(fmaxf(amax, fmaxf(__builtin_fabsf(v[m].x), __builtin_fabsf(v[m].y))),
                 fmaxf(__builtin_fabsf(v[m].z), __builtin_fabsf(v[m].w)));
  }
  ssq += __shfl_xor(ssq, 1);
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  if (pl == 0) {
    s_norm[r] = __builtin_sqrtf(ssq) * kNormSlack;
    s_amax[r] = amax;
  }
  __syncthreads();
  if (tid < 64) {
    float nm = fmaxf(s_norm[tid], s_norm[tid + 64]);
    float am = fmaxf(s_amax[tid], s_amax[tid + 64]);
    for (int off = 32; off > 0; off >>= 1) {
      nm = fmaxf(nm, __shfl_xor(nm, off));
      am = fmaxf(am, __shfl_xor(am, off));
    }
    if (tid == 0) {
      const float sc = pow2_ceil(am);
      s_scale = sc;
      meta[stage].norm = nm;
      meta[stage].scale = sc;
      meta[stage].inv_scale = 1.0f / sc;
      meta[stage].pad_ = 0.0f;
      atomicMax(reinterpret_cast<uint32_t *>(norm_max), __float_as_uint(nm));  // nm >= 0
    }
  }
  __syncthreads();
  const float inv = 1.0f / s_scale;  // exact: power of two
  // own plane -> packed pairs: pk[j] = (value 2j, value 2j + 1) of the plane
  uint32_t pk[NW];
#pragma unroll
  for (int m = 0; m < N4; ++m) {
    pk[2 * m] = pack_f16x2(v[m].x * inv, v[m].y * inv);
    pk[2 * m + 1] = pack_f16x2(v[m].z * inv, v[m].w * inv);
  }
  // the even thread stores image words [0, NW) and needs the odd plane's values [0, NW) = its pairs [0, NW / 2);
  // the odd thread stores words [NW, 2 NW) and needs the even plane's pairs [NW / 2, NW)
  uint32_t other[NW / 2];
#pragma unroll
  for (int j = 0; j < NW / 2; ++j) other[j] = __shfl_xor(pl == 0 ? pk[NW / 2 + j] : pk[j], 1);
  uint32_t w[NW];
#pragma unroll
  for (int j = 0; j < NW / 2; ++j) {
    const uint32_t e = pl == 0 ? pk[j] : other[j];            // (feature 2i, 2i + 2) of the even plane
    const uint32_t o = pl == 0 ? other[j] : pk[NW / 2 + j];   // (2i + 1, 2i + 3) of the odd plane
    w[2 * j] = (e & 0xFFFFu) | (o << 16);
    w[2 * j + 1] = (e >> 16) | (o & 0xFFFF0000u);
  }
  char *dst = packed16 + (row0 + r) * (int64_t)row_bytes16(DP) + pl * (NW * 4);
#pragma unroll
  for (int q4 = 0; q4 < NW / 4; ++q4)
    *reinterpret_cast<uint4 *>(dst + 16 * q4) = make_uint4(w[4 * q4], w[4 * q4 + 1], w[4 * q4 + 2], w[4 * q4 + 3]);
  if (pl == 1) *reinterpret_cast<uint4 *>(dst + NW * 4) = make_uint4(0u, 0u, 0u, 0u);   // the row's pad slot
}

// 16 lanes per query: each lane reads 16-byte pieces of the row (coalesced: a wave covers four
// consecutive rows per pass), a 4-step xor tree combines the 16 partial sums / maxima.  (The
// one-thread-per-query version walked its row with a runtime-bound loop -- one memory round trip
// per feature, 10 us for 8192 x 64.)
__global__ void __launch_bounds__(256) query_kappa_kernel(const float *__restrict__ q, int64_t nq,
                                                          int d, float *__restrict__ qk,
                                                          float *__restrict__ qscale,
                                                          uint32_t *__restrict__ zero_u32,
                                                          uint32_t *__restrict__ flags) {
  const int sub = threadIdx.x & 15;
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool ok = r < nq;
  const float *row = q + (ok ? r : 0) * d;
  float ssq = 0.0f, amax = 0.0f;
  if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
    float4 v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // d <= 128: at most two pieces per lane, both in flight
      const int c = sub + 16 * i;
      v[i] = (ok && 4 * c < d) ? *reinterpret_cast<const float4 *>(row + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ssq = __builtin_fmaf(v[i].x, v[i].x, ssq); ssq = __builtin_fmaf(v[i].y, v[i].y, ssq);
      ssq = __builtin_fmaf(v[i].z, v[i].z, ssq); ssq = __builtin_fmaf(v[i].w, v[i].w, ssq);
      amax = fmaxf(fmaxf(amax, fmaxf(__builtin_fabsf(v[i].x), __builtin_fabsf(v[i].y))),
                   fmaxf(__builtin_fabsf(v[i].z), __builtin_fabsf(v[i].w)));
    }
  } else {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {      // d <= 128 = 16 lanes x 8 features
      const int k = sub + 16 * i;
      x[i] = (ok && k < d) ? row[k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ssq = __builtin_fmaf(x[i], x[i], ssq);
      amax = fmaxf(amax, __builtin_fabsf(x[i]));
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    ssq += __shfl_xor(ssq, off);
    amax = fmaxf(amax, __shfl_xor(amax, off));
  }
  if (ok && sub == 0) {
    // a NaN / Inf element (or a norm beyond the f32 range) makes the sum of squares non-finite: bit 1 of the flag word
    if (flags && nonfinite_bits(ssq)) atomicOr(flags, kNonfiniteQueries);
    if (zero_u32) zero_u32[r] = 0u;
    qk[r] = __builtin_sqrtf(ssq) * kNormSlack * kF16Kappa;
    qscale[r] = pow2_ceil(amax);
  }
}

// flags |= bit when any of x[0, count) -- or of the optional second array y[0, county) -- is NaN / Inf (the search
// paths that do not run query_kappa_kernel; Streaming over blocks read in place: queries + carried state, ONE launch)
__global__ void __launch_bounds__(256) nonfinite_flag_kernel(const float *__restrict__ x, int64_t count,
                                                             const float *__restrict__ y, int64_t county,
                                                             uint32_t *__restrict__ flags, uint32_t bit) {
  uint32_t bad = 0u;
  const int64_t total = count + county;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    bad |= nonfinite_bits(i < count ? x[i] : y[i - count]);
  if (__ballot(bad != 0u) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags, bit);
}

int launch_nonfinite_flag(const float *x, int64_t count, uint32_t *flags, uint32_t bit, hipStream_t stream,
                          const float *y, int64_t county) {
  if (!y) county = 0;
  if (count + county <= 0 || !flags) return TFRS_OK;
  const int64_t blocks = std::min<int64_t>((count + county + 1023) / 1024, 1024);
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, count, y, county, flags, bit);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_pack16(const char *packed, int d, int64_t row_begin, int64_t row_end, char *packed16,
                  StageMeta *meta, float *norm_max, hipStream_t stream) {
  if (row_end <= row_begin) return TFRS_OK;
  const int64_t s0 = row_begin / kTileN;
  const int64_t s1 = (row_end + kTileN - 1) / kTileN;
  const dim3 grid((unsigned)(s1 - s0)), block(256);
  switch (padded_dim(d)) {
    case 16: hipLaunchKernelGGL(pack16_stage_regs_kernel<16>, grid, block, 0, stream, packed, s0, packed16, meta, norm_max); break;
    case 32: hipLaunchKernelGGL(pack16_stage_regs_kernel<32>, grid, block, 0, stream, packed, s0, packed16, meta, norm_max); break;
    case 64: hipLaunchKernelGGL(pack16_stage_regs_kernel<64>, grid, block, 0, stream, packed, s0, packed16, meta, norm_max); break;
    case 128: hipLaunchKernelGGL(pack16_stage_regs_kernel<128>, grid, block, 0, stream, packed, s0, packed16, meta, norm_max); break;
    default:
      hipLaunchKernelGGL(pack16_stage_kernel, grid, block, 0, stream, packed, padded_dim(d), padded_dim16(d), s0,
                         packed16, meta, norm_max);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_query_kappa(const float *q, int64_t nq, int d, float *qk, float *qscale,
                       uint32_t *zero_u32, hipStream_t stream, uint32_t *flags) {
  if (nq <= 0) return TFRS_OK;
  hipLaunchKernelGGL(query_kappa_kernel, dim3((unsigned)((nq * 16 + 255) / 256)), dim3(256), 0,
                     stream, q, nq, d, qk, qscale, zero_u32, flags);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

}  // namespace tfrs
