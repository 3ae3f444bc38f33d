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

// 64-bit hash of a row's bit pattern (xor-multiply-rotate over its 32-bit words; equal rows hash
// equal, collisions are resolved by the exact comparison on the host side).  One thread per row:
// index-time work, read once.
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

}  // namespace tfrs

extern "C" int tfrs_row_hash64(const float *rows, int64_t n, int d, uint64_t *out, void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(n >= 0 && d >= 1, "row_hash64: bad shape");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(rows && out, "row_hash64: NULL pointer");
  hipLaunchKernelGGL(row_hash64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint32_t *>(rows), n, d, out);
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
