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

// hipFuncSetAttribute(kernel, MaxDynamicSharedMemorySize, bytes), remembered per (kernel, DEVICE):
// the attribute lives in the per-device function object, so a process that drives several GPUs (or
// several host threads on distinct streams -- the header promises re-entrancy there) must set it on
// each device it launches on.  A process-wide `static bool` (round 2) left the second device at the
// 64 KB default and its first large-LDS launch failed.  Thread-safe; the hit path is one
// hipGetDevice plus a thread-local compare.
hipError_t ensure_dynamic_lds(const void *kernel, int bytes);

// ---- the library's ONE configuration plane ----------------------------------------------------
// Every measurement / test switch (the TFRS_* names listed in INTEGRATION.md) is read through
// option(): a value set with tfrs_set_option() wins, else the process environment is consulted, so
// the switches are addressable through the C ABI (and enumerable: tfrs_get_option) instead of being
// scattered getenv calls.  Returns nullptr when the option is unset.  The returned pointer stays
// valid until the same option is set again from this thread's point of view (values are interned).
const char *option(const char *name);

// ---- packed candidate layout -------------------------------------------------
// Row r of the packed corpus occupies row_bytes(dp) = 4 dp bytes (round 5: no pad slot in memory, so a dim-64 row
// is two whole 128-byte lines for the kernels that re-score single rows -- with the pad it straddled three):
//   slots 0 .. dp/8-1      : even features  (d = 0, 2, 4, ...)  4 floats per 16-B slot
//   slots dp/8 .. dp/4-1   : odd features   (d = 1, 3, 5, ...)
// In LDS (scan_kernel's stages) the row stride is lds_row_bytes(dp) = 4 dp + 16, an ODD number of 16-B slots,
// so ds_read_b128 of 16 consecutive rows at one slot index hits 16 distinct 16-B bank groups; the stage copy
// re-pitches (every lane of the copy computes its own source address anyway).
// v_mfma_f32_32x32x2_f32 consumes features (2s, 2s+1) at step s: lanes 0-31 supply
// k = 2s from the even plane, lanes 32-63 supply k = 2s+1 from the odd plane, so the
// accumulation order is d = 0, 1, 2, ... exactly.
// ---- activations fused into the GEMM epilogues (Keras names: layers/blocks.py:46-52 Dense(activation=...),
// dcn.py:176-181 Cross(preactivation=...)); codes of the C ABI (include/tfrs_hip.h TFRS_ACT_*) -----------------
enum { kActNone = 0, kActRelu = 1, kActSigmoid = 2, kActTanh = 3, kActSilu = 4, kActGelu = 5 };
__device__ __forceinline__ float act_apply(int act, float v) {
  switch (act) {
    case kActRelu: return v > 0.0f ? v : 0.0f;
    case kActSigmoid: return 1.0f / (1.0f + __expf(-v));
    case kActTanh: return tanhf(v);
    case kActSilu: return v / (1.0f + __expf(-v));
    case kActGelu: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));    // Keras gelu(approximate=False)
    default: return v;
  }
}
// d act(p) / d p at the PRE-activation p
__device__ __forceinline__ float act_grad(int act, float p) {
  switch (act) {
    case kActRelu: return p > 0.0f ? 1.0f : 0.0f;
    case kActSigmoid: { const float s = 1.0f / (1.0f + __expf(-p)); return s * (1.0f - s); }
    case kActTanh: { const float t = tanhf(p); return 1.0f - t * t; }
    case kActSilu: { const float s = 1.0f / (1.0f + __expf(-p)); return s * (1.0f + p * (1.0f - s)); }
    case kActGelu: return 0.5f * (1.0f + erff(p * 0.70710678118654752f)) +
                          p * 0.3989422804014327f * __expf(-0.5f * p * p);
    default: return 1.0f;
  }
}
// ... and from the OUTPUT y = act(p) where that determines it (relu, sigmoid, tanh)
__device__ __forceinline__ float act_grad_from_output(int act, float y) {
  switch (act) {
    case kActRelu: return y > 0.0f ? 1.0f : 0.0f;
    case kActSigmoid: return y * (1.0f - y);
    case kActTanh: return 1.0f - y * y;
    default: return 1.0f;
  }
}

constexpr int kTileN = 128;  // candidate rows per LDS stage; packed buffers are padded to it

__host__ __device__ inline int padded_dim(int d) {
  // supported register-resident dims: 8, 16, 32, 64, 128
  if (d <= 8) return 8;
  if (d <= 16) return 16;
  if (d <= 32) return 32;
  if (d <= 64) return 64;
  return 128;
}
__host__ __device__ inline int row_bytes(int dp) { return dp * 4; }
__host__ __device__ inline int lds_row_bytes(int dp) { return dp * 4 + 16; }
__host__ __device__ inline int64_t padded_rows(int64_t n) {
  return (n + kTileN - 1) / kTileN * kTileN;
}

// ---- fp16 prefilter image -------------------------------------------------------
// A second, half-size image of the corpus feeds the 16-bit matrix cores (16x the f32 MFMA
// rate).  Its scores only FILTER: every prefilter score carries a rigorous error bound
//   |s_16 - s_f32| <= ||q|| * ||c|| * kF16Kappa + kF16Tiny,
// candidates whose prefilter score is within the bound of the running K-th best are kept and
// re-scored with the exact f32 fma chain before anything is returned.
//
// fp16 (11-bit significand) rather than bf16 (8-bit) because the bound is 8x tighter at the
// same MFMA rate; its narrow exponent range is handled by power-of-two scales that are
// folded into the filter threshold (no per-score work): every stage of kTileN rows stores
// x / s_stage with s_stage = 2^ceil(log2(max |x| in the stage)), every query is scaled by its
// own 2^ceil(log2(max |q_d|)).  With |x/s| <= 1:
//   element error <= 2^-11 |x| (normal range) or 2^-25 s (fp16 subnormal range)
//   => sum over d: 2^-10 sum|q_d c_d| + 2^-25 (s_c ||q||_1 + s_q ||c||_1)
//                  <= ||q|| * N * (2^-10 + 2^-23 sqrt(D)),  N = max row norm of the stage
//   + D * 2^-21 ||q|| ||c|| for the f32 accumulation orders (matrix core vs. fma chain)
// so kappa = 2^-10 + 2^-19.5 + 2^-14 (D = 128) = 1.04e-3; kF16Kappa adds 5 %.
// Row r of the image occupies row_bytes16(dp16) bytes: dp16 halves in natural feature order
// + one 16-byte zero pad slot (odd number of 16-B slots per row -> conflict-free
// ds_read_b128, same argument as the f32 image).  Per stage: StageMeta {norm, scale}.
constexpr float kF16Kappa = 0.0011f;
constexpr float kF16Tiny = 1.0e-30f;    // absolute slack (f32 subnormal flushes)
constexpr float kNormSlack = 1.0002f;   // covers the rounding of the f32 norm computation

struct StageMeta {
  float norm;       // max row 2-norm of the stage's rows (rounded up)
  float scale;      // power of two >= max |x| of the stage (1 for an all-zero stage)
  float inv_scale;  // 1 / scale (exact)
  float pad_;
};

__host__ __device__ inline int padded_dim16(int d) {
  if (d <= 16) return 16;
  if (d <= 32) return 32;
  if (d <= 64) return 64;
  return 128;
}
__host__ __device__ inline int row_bytes16(int dp16) { return dp16 * 2 + 16; }

// smallest power of two >= x for finite x > 0 (1 for x == 0 or non-finite), clamped so
// that both the scale and its reciprocal are normal floats
__device__ inline float pow2_ceil(float x) {
  if (!(x > 0.0f) || !(x < __builtin_inff())) return 1.0f;
  const uint32_t b = __float_as_uint(x);
  int e = (int)(b >> 23) - 127;           // floor(log2 x) for normal x, -127 for subnormal
  if ((b & 0x7FFFFFu) != 0u) e += 1;
  if (e < -120) e = -120;
  if (e > 120) e = 120;
  return __uint_as_float((uint32_t)(e + 127) << 23);
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

// ---- row-major candidate blocks read in place (Streaming over a dataset) ------------------------
// Streaming.call (layers/factorized_top_k.py:404-509) re-reads its candidate dataset on every call
// (:384-390: "index_from_dataset" only keeps the dataset).  The blocks of ONE group of consecutive
// dataset elements -- device pointers to row-major f32 [rows, d] -- are described by a RawTable
// in device memory; image-local row r of the group is row r - row_start[b] of block b.  The kernels
// that need a candidate's f32 features (the raw scan, the fp16 packer, the exact re-scoring of
// survivors and the exact-recompute fallback) fetch them through it, so no packed f32 image of
// the stream is ever built.  Requirements of this path: d in {8, 16, 32, 64, 128} (a row is a whole
// number of 16-byte pieces, d == padded_dim(d)) and 16-byte aligned block pointers.
constexpr int kRawMaxBlocks = 192;   // descriptor table by value: 16 + 193 * 8 + 192 * 8 = 3096 bytes < 4 KiB of kernel arguments
struct RawTable {
  int32_t n_blocks;
  int32_t uniform_rows;   // > 0: every block but the last holds exactly this many rows (block = row / uniform_rows)
  int64_t total_rows;
  int64_t row_start[kRawMaxBlocks + 1];
  const float *ptr[kRawMaxBlocks];
};
__device__ inline int raw_find_block(const RawTable *t, int64_t row) {
  const int nb = t->n_blocks;
  if (t->uniform_rows > 0) {
    const int64_t b = row / t->uniform_rows;
    return b < nb ? (int)b : nb - 1;
  }
  int lo = 0, hi = nb - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t->row_start[mid] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__device__ inline const float *raw_row_ptr(const RawTable *t, int64_t row, int d) {
  const int b = raw_find_block(t, row);
  return t->ptr[b] + (row - t->row_start[b]) * d;
}
// Exact score of one candidate read from its block: the d-ordered fma chain of the MFMA f32 path
// (`qs`: the query in LDS, read as broadcast float4; d is a multiple of 8).
__device__ __forceinline__ float raw_score(const RawTable *t, int64_t row, int d, const float *qs) {
  const float4 *c4 = reinterpret_cast<const float4 *>(raw_row_ptr(t, row, d));
  const float4 *q4 = reinterpret_cast<const float4 *>(qs);
  float acc = 0.0f;
#pragma unroll 8
  for (int m = 0; m < d / 4; ++m) {
    const float4 c = c4[m], q = q4[m];
    acc = __builtin_fmaf(c.x, q.x, acc);
    acc = __builtin_fmaf(c.y, q.y, acc);
    acc = __builtin_fmaf(c.z, q.z, acc);
    acc = __builtin_fmaf(c.w, q.w, acc);
  }
  return acc;
}
// Copies a table passed BY VALUE (kernel arguments) to device memory: no host-to-device copy, no
// synchronisation, capturable in a HIP graph.
int launch_raw_table_write(const RawTable &table_h, RawTable *table_dev, hipStream_t stream);

// Raw scan (topk_raw.hip): exact f32 scores (v_mfma_f32_32x32x2_f32, the d-ordered fma chain) of the
// group's rows [c_begin, c_end) read straight from the row-major blocks, for small query batches
// (HBM-bound: every candidate byte is read once from HBM and never written back).
//   grid      = n_splits x n_qtiles workgroups of 256 threads; a query tile = qg groups of 32 queries
//               (qg = 1 or 2), resident in ALL four waves as MFMA B operands; wave w scores 32 of a
//               stage's 128 rows against them.
//   FILTER      scores > thr[q] are appended to the (query, split) segment of the survivor list,
//               slot taken from a workgroup-shared LDS counter:
//               buf[(q * cap_l + e) * nseg + split], cnt[q * nseg + split]  (nseg == n_splits;
//               counts beyond cap_l: entries were dropped, the query is recomputed exactly)
//   MATERIALIZE dense[q * ld_dense + (row - c_begin)]
struct RawScanArgs {
  const float *q;
  int64_t nq;
  int d;
  const RawTable *table;
  int64_t c_begin, c_end;   // c_begin: a multiple of kTileN
  int64_t split_len;        // rows per split (a multiple of kTileN)
  int n_splits;
  int n_qtiles;             // ceil(nq / (32 * qg))
  int qg;                   // 1 or 2
  const float *thr;
  uint32_t *cnt;
  uint2 *buf;
  uint32_t cap_l;
  int nseg;
  float *dense;
  int64_t ld_dense;
  // rawscan16_kernel (fp16 prefilter straight from the blocks): thr[q] is a proven lower bound of the
  // query's final K-th score; qk / qscale as launch_query_kappa writes them; norm_max: atomicMax (float
  // bits) of the largest row norm met; zero_word / zero_aux: re-armed like Scan16Args' (topk_scan16.hip)
  const float *qk;
  const float *qscale;
  int64_t thr_stride;    // rawscan16: query q's bound is thr[q * thr_stride] (0 -> 1); lets the carried state's K-th column serve
  float *norm_max;
  uint32_t *zero_word;
  uint32_t *zero_aux;
};
int launch_rawscan(const RawScanArgs &a, bool materialize, hipStream_t stream);
// qg: 1, 2, 4 or 8 groups of 32 queries per workgroup (n_qtiles = ceil(nq / (32 * qg))); survivors carry
// PREFILTER scores (re-scored by launch_list_topk16 with the table), layout [nq, cap_l, nseg = n_splits]
int launch_rawscan16(const RawScanArgs &a, hipStream_t stream);
// dim 128, one query tile of 65 .. 128 queries: workgroups of two waves on 64-row stages, two per CU (the caller plans twice the splits)
bool rawscan16_half(int d, int qg, int n_qtiles);
// fp16 prefilter image (+ StageMeta, global max row norm) of the group's rows [0, table.total_rows),
// zero rows up to the next stage boundary, straight from the row-major blocks
int launch_pack16_raw(const RawTable *table, int64_t n_rows, int d, char *packed16, StageMeta *meta,
                      float *norm_max, hipStream_t stream);

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
  // scores above thr[query] to its PRIVATE segment seg = 2*split + h of that query's list
  // (entry-major, so that the select kernel reads entry e of 64 segments as one 512-byte row):
  //   buf[(query * cap_l + e) * nseg + seg],  cnt[query * nseg + seg] = number appended
  // (counts may exceed cap_l: the excess was dropped and the query is recomputed exactly).
  const float *thr;      // [nq] current K-th best score per query
  int tie_ge;            // keep scores >= thr (shuffled index: an equal score later in the image
                         // may belong to an EARLIER original row and win the tie)
  uint32_t *cnt;         // [nq, nseg]
  uint2 *buf;            // [nq, cap_l, nseg] (score bits, row index)
  uint32_t cap_l;        // entries per segment
  int nseg;              // 2 * n_splits
  // MATERIALIZE mode
  float *dense;          // [nq, ld_dense] scores for rows c_begin..c_end
  int64_t ld_dense;
  // FILTER mode, paged search (K beyond TFRS_MAX_K: tfrs_bruteforce_topk_below): scores above
  // ceil_score[query] belong to rows returned by an earlier page and are not appended (NULL: no
  // ceiling); the exact (score, row) ceiling is applied when keys are formed (SelectArgs::ceil_key)
  const float *ceil_score;
};

int launch_scan(const ScanArgs &a, bool materialize, hipStream_t stream);

// fp16 prefilter scan (topk_scan16.hip).  A launch covers the stage list
// stage0 + i * stage_stride, i < n_stages, cut into n_splits slices of stages_per_split;
// a workgroup owns 512 queries (8 waves x 2 groups of 32) x one slice.
//   FILTER (default): a score survives when s_16 > lower[q] - qk[q] * meta[stage].norm - tiny
//     (lower[q]: proven lower bound of the query's final exact K-th score; qk = ||q|| * kappa)
//     and is appended to the query's survivor list (layout: see SelectArgs).
//   BINMAX (binmax != NULL): binmax[q * ld_binmax + 2 * (i / bin_stages) + h] = max prefilter
//     score of lane half h over bin_stages consecutive list stages (64 * bin_stages candidates).
//   MATERIALIZE (dense != NULL): dense[q * ld_dense + i * 128 + r] = prefilter score of row r
//     of stage i.
struct Scan16Args {
  const float *q;        // [nq, d] row-major f32 (scaled + converted to fp16 in registers)
  int64_t nq;
  int d;
  const char *packed16;  // fp16 image base (row 0)
  const StageMeta *meta; // [rows / kTileN]
  int64_t stage0;
  int n_stages;
  int stage_stride;
  int stages_per_split;
  int64_t row_limit;     // rows >= row_limit (zero padding of the last stage) never survive
  int n_qtiles;          // ceil(nq / 512)
  int n_splits;
  const float *lower;    // [nq]
  const float *qk;       // [nq]
  const float *qscale;   // [nq] power of two >= max |q_d|
  uint32_t *cnt;         // [nq, nseg]
  uint2 *buf;            // [nq, cap_l, nseg] (prefilter score bits, row index)
  uint32_t cap_l;
  int nseg;              // 2 * n_splits
  float *dense;
  int64_t ld_dense;
  float *binmax;
  int64_t ld_binmax;
  int bin_stages;        // stages per bin (stages_per_split must be a multiple of it)
  // FILTER (second-generation kernel): per-query overflow lists for survivors whose segment is
  // full (corpora ordered by cluster put a query's survivors into one or two segments):
  // ovf_cnt[q] entries in ovf_buf[q * ovf_cap ...]; NULL = none (a full segment then flags the
  // query for the exact redo, as the first-generation kernel does)
  uint32_t *ovf_cnt;
  uint2 *ovf_buf;
  uint32_t ovf_cap;
  int drain_every;       // FILTER: all waves drain at every drain_every-th stage end (see drain_min)
  int drain_min;         // FILTER (second-generation kernel): queue entries that trigger a
                         // drain at a stage end (0 -> 1)
  uint32_t *zero_aux;    // FILTER: four more words re-armed the same way (redo reason counters) or NULL
  uint32_t *zero_word;   // FILTER: word re-armed (= 0) for the kernel that follows (the
                         // flagged-query counter); NULL otherwise.  In-kernel instead of a
                         // hipMemsetAsync because memset nodes are not reliably ordered against
                         // kernel nodes when a captured call is replayed from a HIP graph.
};
int launch_scan16(const Scan16Args &a, hipStream_t stream);
constexpr int kScan16QueriesPerWg = 512;

enum SelectSource { kSrcDense = 0, kSrcList = 1, kSrcParts = 2, kSrcRecompute = 3, kSrcExpand = 4 };

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
  // kSrcParts: scores/idx[nparts, nq, k_in]; idx < 0 marks an empty slot
  const float *part_scores;
  const int32_t *part_idx;
  int nparts, k_in;
  int64_t part_stride;  // elements between consecutive parts (0 = dense: nq * k_in)
  // all sources: value added to source-local row numbers
  int64_t idx_base;
  // shuffled index: image row -> original row (NULL: identity); applied when keys are formed, so
  // every tie is decided by ORIGINAL row numbers
  const int32_t *rowmap;
  // outputs
  float *out_scores;   // [nq, k]
  int32_t *out_idx;    // [nq, k]
  float *out_thr;      // [nq] K-th best score, or -inf while fewer than k entries (may be NULL)
  // kSrcExpand (de-duplicated index): part_scores / part_idx[nq, k_in] are the best DISTINCT rows
  // (idx = distinct-row number, < 0 empty); distinct row u stands for the original rows
  // dup_rows[dup_start[u] .. dup_start[u + 1]) (ascending).  Every original row is a candidate with
  // its distinct row's score; at most k rows per distinct row can matter.
  const int64_t *dup_start;
  const int32_t *dup_rows;
  // paged search: only keys strictly BELOW ceil_key[query] -- i.e. after it in (score descending,
  // row ascending) order -- are candidates (NULL: no ceiling)
  const uint64_t *ceil_key;
  // launch_recompute: optional buffer of nq * kRecomputeChunks partial key lists of k keys; when
  // set, each flagged query's rows are spread over recompute_chunks(nq, flagged) workgroups --
  // the fewer queries are flagged, the more workgroups share one query -- and merged afterwards
  uint64_t *part_keys;
  // kSrcRecompute: exact keys of rows [rc_begin, rc_end) (slow, always correct).
  // launch_recompute only: only_flagged != NULL restricts the work to the listed queries,
  // only_flagged[0] = count, only_flagged[1 + s] = query index of slot s.
  const uint32_t *only_flagged;
  // raw != NULL: exact scores are computed from the row-major blocks of the table (rc_begin / rc_end
  // and the list's row numbers are group-local rows) instead of the packed f32 image
  const RawTable *raw;
};

int launch_select(const SelectArgs &a, hipStream_t stream);
// exact top-K of rows [rc_begin, rc_end) for the queries flagged in a.only_flagged (all queries
// when NULL), one workgroup per query; writes out_scores / out_idx of those queries only
int launch_recompute(const SelectArgs &a, hipStream_t stream);
constexpr int kRecomputeChunks = 32;      // partial lists per query the workspace holds (for ALL nq queries)
constexpr int kRecomputeMaxChunks = 256;  // grid.y of the recompute launch: chunks one flagged query can use
constexpr int kRecomputeSlotsX = 8;       // grid.x: flagged slots in flight (workgroups stride over the slots)
// chunks per flagged query, decided on the device from the flagged count: the whole list budget
// (nq * kRecomputeChunks) is shared by the flagged queries, capped by the grid.  One flagged query
// of 8192 is scanned by 256 workgroups (~0.1 ms on a 1M x 64 corpus) instead of 32 (1.3 ms).
__host__ __device__ inline int recompute_chunks(int64_t nq, int64_t flagged) {
  const int64_t budget = nq * kRecomputeChunks / (flagged > 0 ? flagged : 1);
  return (int)(budget < kRecomputeMaxChunks ? (budget < 1 ? 1 : budget) : kRecomputeMaxChunks);
}
// rowmap != NULL: the block is stored SHUFFLED -- image row dst_row + r holds block row
// (mul * r + add) mod n (mul coprime to n, near n / golden ratio: consecutive block rows land far
// apart) and rowmap[dst_row + r] = dst_row + that row.
// bits of an index handle's host-visible flag word (tfrs_index_nonfinite)
constexpr uint32_t kNonfiniteCandidates = 1u, kNonfiniteQueries = 2u;
__device__ __forceinline__ uint32_t nonfinite_bits(float x) {
  return ((__builtin_bit_cast(uint32_t, x) & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
}
int launch_nonfinite_flag(const float *x, int64_t count, uint32_t *flags, uint32_t bit, hipStream_t stream,
                          const float *y = nullptr, int64_t county = 0);
int launch_pack(const float *cand, int64_t n, int d, char *packed, int64_t dst_row,
                int64_t zero_rows_to, int32_t *rowmap, hipStream_t stream, uint32_t *flags = nullptr);
int launch_unpack(const char *packed, int64_t n, int d, const int32_t *rowmap, float *out,
                  hipStream_t stream);
// (Re)builds the fp16 image, StageMeta and the global max row norm for the stages that hold
// rows [row_begin, row_end) from the f32 image (stage-local and idempotent: a partially
// filled tail stage is simply rebuilt by the next append).  norm_max: atomicMax on float bits.
int launch_pack16(const char *packed, int d, int64_t row_begin, int64_t row_end, char *packed16,
                  StageMeta *meta, float *norm_max, hipStream_t stream);
// qk[q] = ||q||_2 * kNormSlack * kF16Kappa;  qscale[q] = 2^ceil(log2 max|q_d|)
// (also re-arms zero_u32[q] = 0 when given: the per-query overflow counters of the filter pass)
// (flags: |= kNonfiniteQueries when a query row holds NaN / Inf)
int launch_query_kappa(const float *q, int64_t nq, int d, float *qk, float *qscale,
                       uint32_t *zero_u32, hipStream_t stream, uint32_t *flags = nullptr);
constexpr uint32_t kOvfCap = 1024;   // overflow entries per query (= the list kernel's capacity)
constexpr uint32_t kOvfPerSeg = 256;  // ... of which at most this many from one segment (a segment
                                      // that overflows by more flags the query for the exact redo)
// topk_select16.hip: lower[q] = (K-th largest bin maximum) - eps[q]
int launch_bin_threshold(const float *binmax, int64_t ld, int n_bins, int64_t nq, int k,
                         const float *qk, const float *norm_max, float *lower, float *raw,
                         hipStream_t stream);
// topk_select16.hip: survivor list of prefilter scores -> exact top-K; queries that need the
// exact recompute path (list overflow / retained set too large) are appended to the list
// redo[1 + slot] with redo[0] = their count (must be zero on entry)
int launch_list_topk16(const float *q, int64_t nq, int d, const char *packed, const uint2 *buf,
                       const uint32_t *cnt, uint32_t cap_l, int nseg, int k, const float *qk,
                       const float *norm_max, float *out_scores, int32_t *out_idx, uint32_t *redo,
                       int64_t idx_base, const uint32_t *ovf_cnt, const uint2 *ovf_buf,
                       uint32_t ovf_cap, const int32_t *rowmap, const float *verify_raw,
                       uint32_t *redo_reason, hipStream_t stream, const RawTable *raw = nullptr);

}  // namespace tfrs
