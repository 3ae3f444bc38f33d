// dedup.hip -- support kernels of the de-duplicated index (robustness of BruteForce.call,
// layers/factorized_top_k.py:586-607, on corpora that hold many EXACT copies of the same row:
// catalogues with default / cold-start embeddings, popularity-weighted duplicates).
//
// tf.math.top_k breaks ties by the lower index, so hundreds of copies of a top-K row all belong to
// the candidates of the K-th place and no score threshold can separate them: survivor lists of the
// filtered scans overflow and the queries fall to the exact-recompute path (bench.py
// `robustness.zipf_duplicates`: 68x the i.i.d. step).  The host therefore indexes the DISTINCT rows
// (found with the 64-bit row hash below + an exact comparison of hash neighbours) and keeps, per
// distinct row, the ascending list of the original rows it stands for; a search returns the best
// distinct rows and `tfrs_topk_expand_duplicates` turns them back into the exact top-K of the
// original corpus: every original row is a candidate with its distinct row's score, order
// (score descending, original row ascending) -- exactly tf.math.top_k on the full corpus.
#include "common.h"

namespace tfrs {

// 64-bit hash of a row's bit pattern (equal rows hash equal; collisions are resolved by the exact comparison of
// hash neighbours on the host side).  Index-time work, read once -- and therefore read coalesced: LPR lanes
// share a row, lane l takes the 16-byte pieces l, l + LPR, ...; a piece is mixed together with its position,
// the pieces' values are added up (wrapping) across the row's lanes and finalised.  (The first version walked
// each row with one thread, 64 cache lines per load instruction: 27 ms for 12.5 M x 128, 0.24 TB/s.)
__device__ __forceinline__ uint64_t hash_mix64(uint64_t h) {
  h ^= h >> 33;
  h *= 0xFF51AFD7ED558CCDull;
  h ^= h >> 33;
  h *= 0xC4CEB9FE1A85EC53ull;
  h ^= h >> 33;
  return h;
}
template <int LPR>
__global__ void __launch_bounds__(256) row_hash64_pieces_kernel(const uint4 *__restrict__ x, int64_t n, int pieces,
                                                                int d, uint64_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = t / LPR;
  const int l = (int)(t - r * LPR);
  uint64_t h = 0;
  if (r < n) {
    const uint4 *p = x + r * pieces;
    for (int c = l; c < pieces; c += LPR) {
      const uint4 w = p[c];
      const uint64_t k = (uint64_t)(c + 1) * 0x9E3779B97F4A7C15ull;
      const uint64_t a = ((uint64_t)w.y << 32) | w.x, b = ((uint64_t)w.w << 32) | w.z;
      h += hash_mix64(a ^ k) + hash_mix64(b + (k << 1 | 1ull));
    }
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) h += __shfl_xor(h, off);   // (all lanes of the wave take part)
  if (r < n && l == 0) out[r] = hash_mix64(h ^ (uint64_t)d) & 0x7FFFFFFFFFFFFFFFull;   // (non-negative as int64: torch sorts signed)
}

// rows that are not a whole number of aligned 16-byte pieces: one thread per row
__global__ void __launch_bounds__(256) row_hash64_kernel(const uint32_t *__restrict__ x, int64_t n, int d,
                                                         uint64_t *__restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t *p = x + r * d;
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)d;
  for (int k = 0; k < d; ++k) {
    h ^= (uint64_t)p[k] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
  }
  out[r] = h & 0x7FFFFFFFFFFFFFFFull;   // (non-negative as int64: torch sorts signed)
}

template <int LPR>
static void launch_row_hash_pieces(const float *rows, int64_t n, int d, uint64_t *out, hipStream_t s) {
  const int64_t threads = n * LPR;
  hipLaunchKernelGGL(row_hash64_pieces_kernel<LPR>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const uint4 *>(rows), n, d / 4, d, out);
}

}  // namespace tfrs

extern "C" int tfrs_row_hash64(const float *rows, int64_t n, int d, uint64_t *out, void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(n >= 0 && d >= 1, "row_hash64: bad shape");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(rows && out, "row_hash64: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (d % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0 && n * 64 < (1ll << 39)) {
    const int pieces = d / 4;   // lanes per row: the largest power of two <= pieces, at most 32
    if (pieces >= 32) launch_row_hash_pieces<32>(rows, n, d, out, s);
    else if (pieces >= 16) launch_row_hash_pieces<16>(rows, n, d, out, s);
    else if (pieces >= 8) launch_row_hash_pieces<8>(rows, n, d, out, s);
    else if (pieces >= 4) launch_row_hash_pieces<4>(rows, n, d, out, s);
    else if (pieces >= 2) launch_row_hash_pieces<2>(rows, n, d, out, s);
    else launch_row_hash_pieces<1>(rows, n, d, out, s);
  } else {
    hipLaunchKernelGGL(row_hash64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t *>(rows), n, d, out);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_topk_expand_duplicates(const float *scores, const int32_t *distinct_rows, int64_t nq,
                                           int k_in, const int64_t *dup_start, const int32_t *dup_rows,
                                           int k_out, float *out_scores, int32_t *out_rows, void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(nq >= 0 && k_in >= 1, "topk_expand_duplicates: bad shape");
  TFRS_CHECK_ARG(k_out >= 1 && k_out <= TFRS_MAX_K, "topk_expand_duplicates: k_out=%d outside [1, %d]", k_out,
                 TFRS_MAX_K);
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(scores && distinct_rows && dup_start && dup_rows && out_scores && out_rows,
                 "topk_expand_duplicates: NULL pointer");
  SelectArgs se = {};
  se.nq = nq;
  se.k = k_out;
  se.state_len = 0;
  se.source = kSrcExpand;
  se.part_scores = scores;
  se.part_idx = distinct_rows;
  se.nparts = 1;
  se.k_in = k_in;
  se.dup_start = dup_start;
  se.dup_rows = dup_rows;
  se.d = 8;
  se.out_scores = out_scores;
  se.out_idx = out_rows;
  return launch_select(se, (hipStream_t)stream);
}
