// common.h -- shared host/device helpers for libtfrs_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tfrs_hip.h"

namespace tfrs {

// ---- error reporting (thread-local message behind tfrs_last_error) ----------
void set_error(const char *fmt, ...);

#define TFRS_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      ::tfrs::set_error(__VA_ARGS__);    \
      return TFRS_EINVAL;                \
    }                                    \
  } while (0)

#define TFRS_HIP(call)                                                        \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess) {                                                   \
      ::tfrs::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                  \
      return TFRS_EHIP;                                                       \
    }                                                                         \
  } while (0)

#define TFRS_LAUNCH_CHECK() TFRS_HIP(hipGetLastError())

// ---- packed candidate layout -------------------------------------------------
// Row r of the packed corpus occupies row_bytes(dp) bytes:
//   slots 0 .. dp/8-1      : even features  (d = 0, 2, 4, ...)  4 floats per 16-B slot
//   slots dp/8 .. dp/4-1   : odd features   (d = 1, 3, 5, ...)
//   slot  dp/4             : zero pad (makes the row stride an ODD number of 16-B
//                            slots, so ds_read_b128 of 16 consecutive rows at one
//                            slot index hits 16 distinct 16-B bank groups)
// v_mfma_f32_32x32x2_f32 consumes features (2s, 2s+1) at step s: lanes 0-31 supply
// k = 2s from the even plane, lanes 32-63 supply k = 2s+1 from the odd plane, so the
// accumulation order is d = 0, 1, 2, ... exactly.
constexpr int kTileN = 128;  // candidate rows per LDS stage; packed buffers are padded to it

__host__ __device__ inline int padded_dim(int d) {
  // supported register-resident dims: 8, 16, 32, 64, 128
  if (d <= 8) return 8;
  if (d <= 16) return 16;
  if (d <= 32) return 32;
  if (d <= 64) return 64;
  return 128;
}
__host__ __device__ inline int row_bytes(int dp) { return dp * 4 + 16; }
__host__ __device__ inline int64_t padded_rows(int64_t n) {
  return (n + kTileN - 1) / kTileN * kTileN;
}

// ---- ordered keys ------------------------------------------------------------
// 64-bit key whose unsigned order is (score descending-first, index ascending-first)
// when sorted in DESCENDING key order: high word = monotone map of the f32 score,
// low word = ~index.  Key 0 is reserved for "empty" (no finite/inf score maps to a
// zero high word).  -0.0f is canonicalised to +0.0f so that it ties with +0.0f as
// tf.math.top_k's float comparison does.
__device__ inline uint32_t f32_orderable(float s) {
  s = s + 0.0f;  // -0 -> +0
  uint32_t b = __float_as_uint(s);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float f32_from_orderable(uint32_t u) {
  uint32_t b = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
  return __uint_as_float(b);
}
__device__ inline uint64_t make_key(float s, int32_t idx) {
  return ((uint64_t)f32_orderable(s) << 32) | (uint32_t)(~(uint32_t)idx);
}
__device__ inline float key_score(uint64_t k) { return f32_from_orderable((uint32_t)(k >> 32)); }
__device__ inline int32_t key_index(uint64_t k) { return (int32_t)(~(uint32_t)k); }

// ---- top-K internals shared between translation units -------------------------
struct ScanArgs {
  const float *q;        // [nq, d] row-major
  int64_t nq;
  int d;                 // true feature dim (<= DP)
  const char *packed;    // packed corpus base (row 0)
  int64_t c_begin;       // first candidate row of this round (multiple of kTileN)
  int64_t c_end;         // one past the last valid row
  int64_t split_len;     // rows per split (multiple of kTileN)
  int n_qtiles;          // ceil(nq / 256)
  int n_splits;
  // FILTER mode: every lane (query j, lane half h) of the wave that owns a query appends the
  // scores above thr[query] to its PRIVATE segment seg = 2*split + h of that query's list:
  //   buf[(query * nseg + seg) * cap_l + e],  cnt[query * nseg + seg] = number appended
  // (counts may exceed cap_l: the excess was dropped and the query is recomputed exactly).
  const float *thr;      // [nq] current K-th best score per query
  uint32_t *cnt;         // [nq, nseg]
  uint2 *buf;            // [nq, nseg, cap_l] (score bits, row index)
  uint32_t cap_l;        // entries per segment
  int nseg;              // 2 * n_splits
  // MATERIALIZE mode
  float *dense;          // [nq, ld_dense] scores for rows c_begin..c_end
  int64_t ld_dense;
};

int launch_scan(const ScanArgs &a, bool materialize, hipStream_t stream);

enum SelectSource { kSrcDense = 0, kSrcList = 1, kSrcParts = 2 };

struct SelectArgs {
  int64_t nq;
  int k;
  // prior state (may be empty: state_len == 0)
  const float *state_scores;  // [nq, k]
  const int32_t *state_idx;   // [nq, k]
  int state_len;
  int source;
  // kSrcDense: scores[nq, ld] for candidate rows idx_base .. idx_base + n_dense - 1
  const float *dense;
  int64_t ld_dense;
  int64_t n_dense;
  // kSrcList: segmented survivor lists written by scan_kernel (see ScanArgs); a query with
  // any cnt > cap_l lost entries and is recomputed exactly from the packed corpus
  const uint2 *buf;
  const uint32_t *cnt;
  uint32_t cap_l;
  int nseg;
  // recompute fallback inputs (rows [rc_begin, rc_end) of the packed corpus)
  const float *q;
  int d;
  const char *packed;
  int64_t rc_begin, rc_end;
  // kSrcParts: scores/idx[nparts, nq, k_in]
  const float *part_scores;
  const int32_t *part_idx;
  int nparts, k_in;
  // all sources: value added to source-local row numbers
  int64_t idx_base;
  // outputs
  float *out_scores;   // [nq, k]
  int32_t *out_idx;    // [nq, k]
  float *out_thr;      // [nq] K-th best score, or -inf while fewer than k entries (may be NULL)
};

int launch_select(const SelectArgs &a, hipStream_t stream);
int launch_pack(const float *cand, int64_t n, int d, char *packed, int64_t dst_row,
                int64_t zero_rows_to, hipStream_t stream);
int launch_unpack(const char *packed, int64_t n, int d, float *out, hipStream_t stream);

}  // namespace tfrs
