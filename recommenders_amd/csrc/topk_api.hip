// topk_api.hip -- C-ABI entry points of the top-K retrieval path and the round driver.
//
// A query batch is answered in "rounds" over growing candidate ranges:
//   round 0  : rows [0, n0)   scores materialised ([nq, n0], n0 ~ 4096) -> select top-K
//              (establishes a per-query threshold = current K-th best score)
//   round i>0: rows [lo, hi) with hi - lo ~ (rho - 1) * lo: fused MFMA scan that keeps
//              only scores above the threshold (expected K*(rho-1) per query) -> merge
// Every round is exact (a threshold taken from a subset is a lower bound of the final
// K-th score), the [nq, n] matrix never exists, nothing synchronises with the host, and
// per-query lists that overflow (adversarial candidate order) are recomputed exactly
// inside the merge kernel.  Streaming.call blocks reuse the same driver with the
// carried-in state as round "-1".
#include <stdlib.h>

#include <algorithm>
#include <new>

#include "common.h"

namespace tfrs {

static int64_t env_i64(const char *name, int64_t dflt) {
  const char *v = option(name);
  if (!v || !*v) return dflt;
  return atoll(v);
}

struct TopkTuning {
  int64_t prefix;  // rows scored densely before filtering starts
  int64_t rho;     // geometric growth of the filtered ranges
  int64_t target_wgs;
  int64_t thr_wgs;       // workgroups of the threshold pass (TFRS_TOPK_WGS_THR; default: target_wgs)
  bool f16_filter;  // TFRS_TOPK_FILTER=f16 (default) | f32
  int64_t sample;   // fp16 path: the threshold pass scans every `sample`-th stage
  int64_t min_bins; // fp16 path: sampled bins required per query, in units of K
  int64_t drain_min; // fp16 filter kernel: queue entries that trigger a drain at a stage end
  int64_t drain_every; // ... and the stage period at which every wave drains whatever it holds
  // Shuffled indexes (tfrs_index::rowmap): the threshold pass may sample more sparsely and take
  // a statistically chosen rank of the bin maxima instead of the K-th (plan_sample); the exact
  // top-K never depends on that choice -- a query whose bound turns out too high is redone.
  bool stat;             // TFRS_TOPK_STAT (default 1)
  int64_t sample_stat;   // sampling stride of the statistical plan (TFRS_TOPK_SAMPLE_STAT, 16)
  double p_fail;         // accepted probability that a query needs the redo (TFRS_TOPK_STAT_PFAIL)
};

static TopkTuning tuning() {
  TopkTuning t;
  t.prefix = std::max<int64_t>(kTileN, env_i64("TFRS_TOPK_PREFIX", 4096));
  t.prefix = padded_rows(t.prefix);
  t.rho = std::max<int64_t>(2, env_i64("TFRS_TOPK_RHO", 8));
  t.target_wgs = std::max<int64_t>(1, env_i64("TFRS_TOPK_WGS", 512));
  t.thr_wgs = std::max<int64_t>(1, env_i64("TFRS_TOPK_WGS_THR", t.target_wgs));
  const char *f = option("TFRS_TOPK_FILTER");
  t.f16_filter = !(f && (f[0] == 'f' || f[0] == 'F') && f[1] == '3');
  t.sample = std::max<int64_t>(1, env_i64("TFRS_TOPK_SAMPLE", 4));
  t.min_bins = std::max<int64_t>(1, env_i64("TFRS_TOPK_MINBINS", 8));  // in 64-candidate bins
  t.drain_min = std::max<int64_t>(1, env_i64("TFRS_SCAN16_DRAIN", 24));
  t.drain_every = std::max<int64_t>(1, env_i64("TFRS_SCAN16_DRAIN_EVERY", 4));
  t.stat = env_i64("TFRS_TOPK_STAT", 1) != 0;
  t.sample_stat = std::max<int64_t>(1, env_i64("TFRS_TOPK_SAMPLE_STAT", 16));
  const char *pf = option("TFRS_TOPK_STAT_PFAIL");
  t.p_fail = pf ? std::min(1.0, std::max(1e-12, atof(pf))) : 1e-7;
  return t;
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- optional per-launch timing of the scan kernel (tfrs_profile_*) ---------------------
// HIP events recorded on the launch stream around every scan launch while enabled; read
// back (after the caller synchronised) as total milliseconds / launches / algorithmic flop.
struct ScanProfile {
  bool enabled = false;
  static constexpr int kMax = 4096;
  hipEvent_t start[kMax], stop[kMax];
  double flop[kMax];
  int kind[kMax];  // 0 = f32 scan, 1 = fp16 filter pass, 2 = fp16 threshold (bin-max) pass
  int created = 0;
  int used = 0;
};
static ScanProfile g_prof;

static bool prof_begin(hipStream_t stream, double flop, int kind, int *slot) {
  if (!g_prof.enabled || g_prof.used >= ScanProfile::kMax) return false;
  const int i = g_prof.used;
  if (i >= g_prof.created) {
    if (hipEventCreate(&g_prof.start[i]) != hipSuccess) return false;
    if (hipEventCreate(&g_prof.stop[i]) != hipSuccess) return false;
    g_prof.created = i + 1;
  }
  g_prof.flop[i] = flop;
  g_prof.kind[i] = kind;
  (void)hipEventRecord(g_prof.start[i], stream);
  *slot = i;
  g_prof.used = i + 1;
  return true;
}
static void prof_end(hipStream_t stream, int slot) { (void)hipEventRecord(g_prof.stop[slot], stream); }

static int timed_scan(const ScanArgs &sa, bool materialize, hipStream_t stream) {
  int slot = 0;
  const bool on = prof_begin(stream, 2.0 * (double)sa.nq * (double)(sa.c_end - sa.c_begin) * sa.d, 0, &slot);
  const int rc = launch_scan(sa, materialize, stream);
  if (on) prof_end(stream, slot);
  return rc;
}

static int timed_scan16(const Scan16Args &sa, double flop, hipStream_t stream) {
  int slot = 0;
  const bool on = prof_begin(stream, flop, sa.binmax ? 2 : 1, &slot);
  const int rc = launch_scan16(sa, stream);
  if (on) prof_end(stream, slot);
  return rc;
}

// fp16 path: stages scanned by the threshold pass (only full stages are sampled) and the
// stride actually used: the requested one, reduced until min_bins * rank bins are available.
struct SamplePlan {
  int64_t stride;
  int64_t n_stages;    // 0: the fp16 path is not applicable
  int64_t bin_stages;  // stages per bin: 4 when that still leaves 8 * K bins, else 1
  int rank;            // the bound is the rank-th largest bin maximum (K, or less: statistical plan)
  bool stat;           // rank < K is possible: the list kernel verifies the bound
  int64_t n_bins() const { return 2 * ((n_stages + bin_stages - 1) / bin_stages); }
};

// Statistical rank.  The image rows are a pseudo-random permutation of the corpus and the
// threshold pass scores every stride-th stage, i.e. a fraction f = 1 / stride of the rows.  Order
// the rows by score: the bound taken from the m-th best SAMPLED row has at least K rows above
// it unless m or more of the K - 1 best rows were sampled, P = P(Binomial(K - 1, f) >= m).  The
// smallest m with P <= p_fail leaves about m / f survivors per query instead of K / f.  (Bins
// that hold two of the best sampled rows and the low-discrepancy shuffle both err towards a
// lower bound, i.e. towards more survivors.)  Nothing is assumed for correctness: the list
// kernel checks that K survivors score above the bound and flags the query for the exact redo
// otherwise.
static int stat_rank(int k, int64_t stride, double p_fail) {
  if (stride <= 1 || k <= 1) return k;
  const double f = 1.0 / (double)stride;
  const int n = k - 1;
  // tail[m] = P(X >= m), summed from the top
  double pmf = __builtin_pow(f, n);   // P(X = n)
  double tail = 0.0;
  int m = n + 1;                       // P(X >= n + 1) = 0
  for (int x = n; x >= 1; --x) {
    tail += pmf;
    if (tail > p_fail) break;
    m = x;
    // P(X = x - 1) = P(X = x) * x / (n - x + 1) * (1 - f) / f
    pmf = pmf * (double)x / (double)(n - x + 1) * (1.0 - f) / f;
    if (pmf == 0.0) {                  // underflow far in the tail: restart from the exact term
      pmf = __builtin_exp(__builtin_lgamma(n + 1.0) - __builtin_lgamma((double)x) -
                          __builtin_lgamma(n - x + 2.0) + (x - 1) * __builtin_log(f) +
                          (n - x + 1) * __builtin_log1p(-f));
    }
  }
  return std::max(1, std::min(m, k));
}

// P(Binomial(n, f) <= x)
static double binom_cdf(int n, double f, int x) {
  if (x < 0) return 0.0;
  if (x >= n) return 1.0;
  double pmf = __builtin_exp((double)n * __builtin_log1p(-f));   // P(X = 0); n <= 1024: no underflow
  double cdf = pmf;
  for (int i = 1; i <= x; ++i) {
    pmf *= (double)(n - i + 1) / (double)i * f / (1.0 - f);
    cdf += pmf;
  }
  return cdf;
}

static SamplePlan plan_sample(int64_t n, int k, const TopkTuning &t, bool stat) {
  const int64_t full = n / kTileN;
  SamplePlan p = {stat ? t.sample_stat : t.sample, 0, 1, k, false};
  for (;; --p.stride) {
    p.rank = stat ? stat_rank(k, p.stride, t.p_fail) : k;
    if (p.stride <= 1) break;
    // The survivor list must fit the list kernel's 1024 slots or the query is redone exactly
    // (correct, but ~1 ms).  Its length is the position T of the rank-th sampled row among all
    // rows ordered by score (negative binomial, mean rank * stride), times ~1.3 for the eps band:
    // P(1.35 T > 1024) = P(Binomial(758, 1 / stride) < rank) must be negligible.
    if (binom_cdf(758, 1.0 / (double)p.stride, p.rank - 1) > 1e-8) continue;
    if (2 * (full / p.stride) >= t.min_bins * (int64_t)p.rank) break;
  }
  p.stat = stat && p.rank < k;
  const int64_t ns = full / p.stride;
  if (2 * ns >= std::max<int64_t>(t.min_bins * (int64_t)p.rank, k)) p.n_stages = ns;
  // 4 stages per bin when that still leaves 8 * rank bins; beyond 4096 bins per query the bins
  // are widened further (the threshold kernel merges them down to <= 1024 values anyway), so
  // the bin-maxima buffer stays at nq * 4096 floats however large the corpus is
  if (2 * (ns / 4) >= 8 * (int64_t)p.rank) p.bin_stages = std::max<int64_t>(4, (2 * ns + 4095) / 4096);
  return p;
}
// (the statistical plan applies to shuffled indexes only)
static bool use_stat(const TopkTuning &t, const int32_t *rowmap) { return t.stat && rowmap != nullptr; }
static int64_t dense_rows(int64_t n, int k, const TopkTuning &t) {
  const int64_t want = std::max<int64_t>(t.prefix, padded_rows(k));
  int64_t cols = std::min<int64_t>(padded_rows(n), want);
  if (t.f16_filter && k <= 512) {
    for (int stat = 0; stat < 2; ++stat) {   // the workspace serves either plan
      const SamplePlan sp = plan_sample(n, k, t, stat != 0);
      if (sp.n_stages > 0) cols = std::max<int64_t>(cols, padded_rows(sp.n_bins()));
    }
  }
  return cols;
}

// Survivor lists.  A filtered round keeps, per query, about K * (rho - 1) scores (the rows of
// the round that beat the K-th best of the rows before it) plus, on the fp16 path, the band
// within eps of the bound: list_mean() budgets 1.5x that.  The list is cut into nseg private
// segments (2 per candidate split); a segment's load is ~Poisson(mean / nseg), so its capacity
// is mean + 8 sigma + 8 -- overflow probability < 1e-12 per segment on exchangeable data, and
// an overflow only costs time (the query's range is recomputed exactly), never correctness.
static int64_t list_mean(int k, const TopkTuning &t) {
  // f32 rounds: K * (rho - 1); fp16 path: K * sample (threshold from 1/sample of the rows)
  const int64_t stat = t.stat ? (int64_t)stat_rank(k, t.sample_stat, t.p_fail) * t.sample_stat : 0;
  return std::max<int64_t>(256, (3 * std::max<int64_t>((int64_t)k * std::max<int64_t>(t.rho - 1, t.sample), stat)) / 2);
}
static uint32_t segment_cap(int k, int nseg, const TopkTuning &t) {
  const double m = (double)list_mean(k, t) / nseg;
  return (uint32_t)(m + 8.0 * __builtin_sqrt(m) + 8.0);
}
// (sized for the fp16 prefilter geometry, 512 queries per workgroup: it has the fewer query
// tiles and therefore the more splits)
static int max_splits(int64_t nq, const TopkTuning &t) {
  const int64_t n_qtiles = (nq + kScan16QueriesPerWg - 1) / kScan16QueriesPerWg;
  return (int)std::max<int64_t>(1, (t.target_wgs + n_qtiles - 1) / n_qtiles);
}
constexpr int kMaxKF16 = 512;
// Survivor-list entries reserved per query: nseg * segment_cap(nseg) grows with nseg, so the
// largest segment count bounds every round.
static int64_t list_entries_per_query(int64_t nq, int k, const TopkTuning &t) {
  const int nseg = 2 * max_splits(nq, t);
  // (+ 1 per segment: segment_cap truncates, so nseg * segment_cap(nseg) is monotone in nseg only
  // up to that rounding -- 254 segments x 57 needed more than 256 x 56)
  return (int64_t)nseg * ((int64_t)segment_cap(k, nseg, t) + 1);
}

struct RoundWs {
  float *thr;
  float *thr_raw;   // [nq] fp16 path: the bin maximum the bound was taken from (thr + eps)
  uint32_t *cnt;
  float *dense;
  int64_t ld_dense;
  uint2 *buf;
  int64_t entries;  // per query
  float *qk;        // [nq]
  float *qscale;    // [nq]
  uint32_t *redo;   // [1 + nq + 4] count, flagged query list, reason counters (topk_select16.hip)
  uint64_t *part_keys;  // [nq, kRecomputeChunks, k] partial lists of the exact-recompute fallback
  uint32_t *ovf_cnt;    // [nq] fp16 path: survivors beyond their segment's capacity ...
  uint2 *ovf_buf;       // [nq, kOvfCap] ... go here
  char *end;
};

static size_t round_ws_bytes(int64_t nq, int64_t n, int k, const TopkTuning &t, int64_t ld_dense = 0) {
  size_t b = 0;
  b += 2 * align_up((size_t)nq * 4);                                    // thr, thr_raw
  b += align_up((size_t)nq * 2 * max_splits(nq, t) * 4);                 // cnt[nq, nseg]
  b += align_up((size_t)nq * (ld_dense > 0 ? ld_dense : dense_rows(n, k, t)) * 4);   // dense
  b += align_up((size_t)nq * list_entries_per_query(nq, k, t) * 8);      // buf
  b += 2 * align_up((size_t)nq * 4) + align_up((size_t)(nq + 5) * 4);    // qk, qscale, redo
  b += align_up((size_t)nq * kRecomputeChunks * k * 8);                  // part_keys
  b += align_up((size_t)nq * 4) + align_up((size_t)nq * kOvfCap * 8);    // ovf_cnt, ovf_buf
  return b;
}

static RoundWs carve_round_ws(char *p, int64_t nq, int64_t n, int k, const TopkTuning &t,
                              int64_t ld_dense = 0) {
  RoundWs w;
  w.thr = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * 4);
  w.thr_raw = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * 4);
  w.cnt = reinterpret_cast<uint32_t *>(p);
  p += align_up((size_t)nq * 2 * max_splits(nq, t) * 4);
  w.ld_dense = ld_dense > 0 ? ld_dense : dense_rows(n, k, t);
  w.dense = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * w.ld_dense * 4);
  w.entries = list_entries_per_query(nq, k, t);
  w.buf = reinterpret_cast<uint2 *>(p);
  p += align_up((size_t)nq * w.entries * 8);
  w.qk = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * 4);
  w.qscale = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * 4);
  w.redo = reinterpret_cast<uint32_t *>(p);
  p += align_up((size_t)(nq + 5) * 4);
  w.part_keys = reinterpret_cast<uint64_t *>(p);
  p += align_up((size_t)nq * kRecomputeChunks * k * 8);
  w.ovf_cnt = reinterpret_cast<uint32_t *>(p);
  p += align_up((size_t)nq * 4);
  w.ovf_buf = reinterpret_cast<uint2 *>(p);
  p += align_up((size_t)nq * kOvfCap * 8);
  w.end = p;
  return w;
}

__global__ void thr_from_state_kernel(const float *state_scores, int64_t nq, int k,
                                      float *thr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nq) thr[r] = state_scores[r * k + (k - 1)];
}

static void plan_splits(int64_t rows, int n_qtiles, const TopkTuning &t, int64_t *split_len,
                        int *n_splits) {
  const int64_t stages = (rows + kTileN - 1) / kTileN;
  int64_t want = (t.target_wgs + n_qtiles - 1) / n_qtiles;
  want = std::max<int64_t>(1, std::min<int64_t>(want, (stages + 3) / 4));
  const int64_t per = (stages + want - 1) / want;
  *split_len = per * kTileN;
  *n_splits = (int)((stages + per - 1) / per);
}

// Answers one block of `n` packed candidate rows (global row numbers idx_base + row).
// state_*[nq, k] holds `state_len` sorted entries drawn from `seen` earlier candidates.
static int run_rounds(const float *q, int64_t nq, int d, const char *packed, int64_t n,
                      int64_t idx_base, int64_t seen, int k, float *state_scores,
                      int32_t *state_idx, int state_len, const RoundWs &w,
                      const TopkTuning &t, hipStream_t stream, int *new_len,
                      const int32_t *rowmap = nullptr, const float *ceil_score = nullptr,
                      const uint64_t *ceil_key = nullptr) {
  int len = state_len;
  if (n <= 0 || nq <= 0) {
    *new_len = len;
    return TFRS_OK;
  }
  const int n_qtiles = (int)((nq + 255) / 256);
  int64_t lo = 0;
  int rc;

  ScanArgs sa = {};
  sa.q = q;
  sa.nq = nq;
  sa.d = d;
  sa.packed = packed;
  sa.n_qtiles = n_qtiles;
  sa.thr = w.thr;
  sa.cnt = w.cnt;
  sa.buf = w.buf;
  sa.dense = w.dense;
  sa.ld_dense = w.ld_dense;
  sa.tie_ge = rowmap != nullptr;
  sa.ceil_score = ceil_score;

  SelectArgs se = {};
  se.ceil_key = ceil_key;
  se.rowmap = rowmap;
  se.nq = nq;
  se.k = k;
  se.state_scores = state_scores;
  se.state_idx = state_idx;
  se.q = q;
  se.d = d;
  se.packed = packed;
  se.idx_base = idx_base;
  se.out_scores = state_scores;
  se.out_idx = state_idx;
  se.out_thr = w.thr;

  if (len < k) {
    // dense round: every score of rows [0, n0) is a candidate
    const int64_t n0 = std::min<int64_t>(n, w.ld_dense);
    sa.c_begin = 0;
    sa.c_end = n0;
    plan_splits(n0, n_qtiles, t, &sa.split_len, &sa.n_splits);
    if ((rc = timed_scan(sa, /*materialize=*/true, stream)) != TFRS_OK) return rc;
    se.state_len = len;
    se.source = kSrcDense;
    se.dense = w.dense;
    se.ld_dense = w.ld_dense;
    se.n_dense = n0;
    se.idx_base = idx_base;  // dense column e is packed row e
    if ((rc = launch_select(se, stream)) != TFRS_OK) return rc;
    len = (int)std::min<int64_t>(k, (int64_t)len + n0);
    seen += n0;
    lo = n0;
  } else if (lo < n) {
    hipLaunchKernelGGL(thr_from_state_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256),
                       0, stream, state_scores, nq, k, w.thr);
    TFRS_LAUNCH_CHECK();
  }

  while (lo < n) {  // state is full here (len == k), w.thr is current
    int64_t span = std::max<int64_t>((t.rho - 1) * std::max<int64_t>(seen, 1), kTileN);
    span = padded_rows(span);
    const int64_t hi = (n - lo <= span) ? n : lo + span;
    sa.c_begin = lo;
    sa.c_end = hi;
    plan_splits(hi - lo, n_qtiles, t, &sa.split_len, &sa.n_splits);
    sa.nseg = 2 * sa.n_splits;
    sa.cap_l = segment_cap(k, sa.nseg, t);
    if ((int64_t)sa.nseg * sa.cap_l > w.entries) {
      set_error("topk: survivor workspace too small (%d segments x %u)", sa.nseg, sa.cap_l);
      return TFRS_ENOMEM;
    }
    if ((rc = timed_scan(sa, /*materialize=*/false, stream)) != TFRS_OK) return rc;
    se.state_len = len;
    se.source = kSrcList;
    se.buf = w.buf;
    se.cnt = w.cnt;
    se.cap_l = sa.cap_l;
    se.nseg = sa.nseg;
    se.rc_begin = lo;
    se.rc_end = hi;
    if ((rc = launch_select(se, stream)) != TFRS_OK) return rc;
    seen += hi - lo;
    lo = hi;
  }
  *new_len = len;
  return TFRS_OK;
}

// ---- fp16-prefiltered path (BruteForce.call) ---------------------------------------------
// Four launches, no rounds:
//   1. query_kappa: ||q|| * kappa and the power-of-two query scales;
//   2. threshold pass: BINMAX scan of every `stride`-th stage of the fp16 image -> the largest
//      prefilter score per 64-candidate bin; bin maxima belong to distinct candidates, so
//      (K-th largest bin maximum) - eps is a proven lower bound of the final K-th score
//      (select kernel, kSrcDense + thr_eps);
//   3. filter pass: FILTER scan of ALL rows against that bound (about K * stride survivors
//      per query instead of a [nq, n] score matrix);
//   4. select: prefilter top-K with its 2*eps retention band, exact f32 re-scoring of the
//      retained entries, exact top-K (bit-identical to the f32 path).
struct F16Image {
  const char *packed16;
  const StageMeta *meta;
  const float *norm_max;
};

static void plan_stage_splits(int64_t n_stages, int n_qtiles, const TopkTuning &t, int *per,
                              int *n_splits) {
  int64_t want = (t.target_wgs + n_qtiles - 1) / n_qtiles;
  want = std::max<int64_t>(1, std::min<int64_t>(want, (n_stages + 3) / 4));
  *per = (int)((n_stages + want - 1) / want);
  *n_splits = (int)((n_stages + *per - 1) / *per);
}

// One range [row_lo, row_lo + n) of a group of blocks through the block-fed fp16 filter (rawscan16_kernel,
// topk_raw.hip: the filter pass reads the f32 blocks themselves -- no image, hence no threshold pass over
// one): bound[q * bound_stride] is a proven lower bound of query q's final K-th score (the carried state's
// K-th column); w.qk / w.qscale hold launch_query_kappa's output; the range's exact top-K goes to out_*.
static int run_raw16_range(const float *q, int64_t nq, int d, const RawTable *raw, float *norm_max,
                           int64_t row_lo, int64_t n, int64_t idx_base, int k, const float *bound,
                           int64_t bound_stride, int qg, float *out_scores, int32_t *out_idx, const RoundWs &w,
                           const TopkTuning &t, hipStream_t stream) {
  int rc;
  RawScanArgs sa = {};
  sa.q = q;
  sa.nq = nq;
  sa.d = d;
  sa.table = raw;
  sa.c_begin = row_lo;
  sa.c_end = row_lo + n;
  sa.qg = qg;
  sa.n_qtiles = (int)((nq + 32 * qg - 1) / (32 * qg));
  {
    // TFRS_TOPK_WGS (512) counts two resident workgroups per CU, which is what the image-fed filters and the block-fed
    // ones up to dim 64 hold.  At dim 128 two raw stages fill the LDS: ONE workgroup per CU, and 512 splits ran as two
    // rounds -- 12.5 M x 128, 1 / 64 / 128 / 512 queries 1.20 / 1.34 / 1.47 / 2.60 ms against 1.18 / 1.305 / 1.42 / 2.565 with
    // one round (384 splits, a round and a half: 1.41 / 1.59 / 1.71 / 3.06).  The two-wave form (rawscan16_half) holds two.
    TopkTuning t2 = t;
    // (eight query groups per wave -- 129 .. 256 queries up to dim 64 -- need more than 128 registers: one workgroup per CU
    // as well; 25 M x 64, 256 queries 2.41 -> 2.35 ms)
    if (d >= 128 || qg == 8) t2.target_wgs = std::max<int64_t>(1, t.target_wgs / 2);
    if (rawscan16_half(d, qg, sa.n_qtiles))
      t2.target_wgs = std::min<int64_t>(2 * t2.target_wgs, 2 * (int64_t)max_splits(nq, t));   // (nseg <= 2 * max_splits: the list budget)
    plan_splits(n, sa.n_qtiles, t2, &sa.split_len, &sa.n_splits);
  }
  sa.nseg = sa.n_splits;
  sa.cap_l = segment_cap(k, sa.nseg, t);
  if ((int64_t)sa.nseg * sa.cap_l > w.entries || sa.nseg > 2 * max_splits(nq, t)) {
    set_error("topk: survivor workspace too small (%d segments x %u)", sa.nseg, sa.cap_l);
    return TFRS_ENOMEM;
  }
  sa.thr = bound;
  sa.thr_stride = bound_stride;
  sa.cnt = w.cnt;
  sa.buf = w.buf;
  sa.qk = w.qk;
  sa.qscale = w.qscale;
  sa.norm_max = norm_max;
  sa.zero_word = reinterpret_cast<uint32_t *>(w.redo);
  sa.zero_aux = w.redo + 1 + nq;
  int slot = -1;
  const bool timed = prof_begin(stream, 2.0 * (double)nq * (double)n * d, 1, &slot);
  rc = launch_rawscan16(sa, stream);
  if (timed) prof_end(stream, slot);
  if (rc != TFRS_OK) return rc;
  // prefilter top-K + exact re-scoring from the blocks; flagged queries (list overflow, retained set too
  // large) are answered by the exact recompute path of the generic select kernel
  if ((rc = launch_list_topk16(q, nq, d, /*packed=*/nullptr, w.buf, w.cnt, sa.cap_l, sa.nseg, k, w.qk, norm_max,
                               out_scores, out_idx, w.redo, idx_base, /*ovf_cnt=*/nullptr, w.ovf_buf, kOvfCap,
                               /*rowmap=*/nullptr, /*thr_raw=*/nullptr, w.redo + 1 + nq, stream, raw)) != TFRS_OK)
    return rc;
  SelectArgs se = {};
  se.nq = nq;
  se.k = k;
  se.q = q;
  se.d = d;
  se.source = kSrcRecompute;
  se.only_flagged = w.redo;
  se.part_keys = w.part_keys;
  se.idx_base = idx_base;
  se.rc_begin = row_lo;
  se.rc_end = row_lo + n;
  se.raw = raw;
  se.out_scores = out_scores;
  se.out_idx = out_idx;
  return launch_recompute(se, stream);
}

// lower_preset: w.thr already holds a proven lower bound per query (Streaming: the carried
// state's exact K-th score) -> the threshold pass is skipped.
// row_lo (a multiple of kTileN) / n: the rows [row_lo, row_lo + n) of the image are searched (survivor
// row numbers stay image-absolute; sp must have been planned for n rows).  raw != NULL: exact scores come
// from the row-major blocks of the table (Streaming groups, topk_raw.hip) and `packed` is unused.
static int run_f16(const float *q, int64_t nq, int d, const char *packed, const F16Image &img,
                   int64_t n, int64_t idx_base, int k, const SamplePlan &sp, bool lower_preset,
                   float *out_scores, int32_t *out_idx, const RoundWs &w, const TopkTuning &t,
                   hipStream_t stream, const int32_t *rowmap = nullptr, int64_t row_lo = 0,
                   const RawTable *raw = nullptr, uint32_t *flags = nullptr) {
  const int n_qtiles = (int)((nq + kScan16QueriesPerWg - 1) / kScan16QueriesPerWg);
  const int64_t stage_lo = row_lo / kTileN;
  int rc;
  // (the per-query overflow counters are re-armed here whether or not the filter pass uses them; flags: the index
  // handle's host-visible word, |= kNonfiniteQueries when a query row holds NaN / Inf)
  if ((rc = launch_query_kappa(q, nq, d, w.qk, w.qscale, w.ovf_cnt, stream, flags)) != TFRS_OK) return rc;

  Scan16Args s16 = {};
  s16.q = q;
  s16.nq = nq;
  s16.d = d;
  s16.packed16 = img.packed16;
  s16.meta = img.meta;
  s16.n_qtiles = n_qtiles;
  s16.qk = w.qk;
  s16.qscale = w.qscale;
  s16.row_limit = row_lo + n;
  s16.stage0 = stage_lo;

  // threshold pass
  if (!lower_preset) {
  s16.n_stages = (int)sp.n_stages;
  s16.stage_stride = (int)sp.stride;
  {
    TopkTuning tt = t;
    tt.target_wgs = t.thr_wgs;
    plan_stage_splits(sp.n_stages, n_qtiles, tt, &s16.stages_per_split, &s16.n_splits);
  }
  s16.bin_stages = (int)sp.bin_stages;
  s16.stages_per_split = (int)((s16.stages_per_split + sp.bin_stages - 1) / sp.bin_stages * sp.bin_stages);
  s16.n_splits = (int)((sp.n_stages + s16.stages_per_split - 1) / s16.stages_per_split);
  s16.binmax = w.dense;
  s16.ld_binmax = w.ld_dense;
  if ((rc = timed_scan16(s16, 2.0 * (double)nq * (double)sp.n_stages * kTileN * d, stream)) != TFRS_OK)
    return rc;
  if ((rc = launch_bin_threshold(w.dense, w.ld_dense, (int)sp.n_bins(), nq, sp.rank, w.qk,
                                 img.norm_max, w.thr, w.thr_raw, stream)) != TFRS_OK)
    return rc;
  }

  // filter pass over all rows
  const int64_t all_stages = (n + kTileN - 1) / kTileN;
  s16.binmax = nullptr;
  s16.n_stages = (int)all_stages;
  s16.stage_stride = 1;
  {
    // dim 128: two 34.8 KB stages + the survivor queues leave room for ONE filter workgroup per CU (scan16f_kernel<128, 8, 2>:
    // 111 KB of LDS), so TFRS_TOPK_WGS = 512 -- two per CU -- ran as two rounds: one round measures - 3.8 / - 2.4 / - 1.8 % at
    // 1 / 64 / 512 queries over 12.5 M x 128 and - 5 ... 6 % up to 1024 queries over 2 M x 128 (8192 queries: - 0.3 ... 0.5 %)
    TopkTuning tf = t;
    if (padded_dim16(d) >= 128) tf.target_wgs = std::max<int64_t>(1, t.target_wgs / 2);
    plan_stage_splits(all_stages, n_qtiles, tf, &s16.stages_per_split, &s16.n_splits);
  }
  s16.lower = w.thr;
  s16.cnt = w.cnt;
  s16.buf = w.buf;
  s16.nseg = 2 * s16.n_splits;
  s16.cap_l = segment_cap(k, s16.nseg, t);
  if ((int64_t)s16.nseg * s16.cap_l > w.entries) {
    set_error("topk: survivor workspace too small (%d segments x %u)", s16.nseg, s16.cap_l);
    return TFRS_ENOMEM;
  }
  s16.drain_min = (int)t.drain_min;
  s16.drain_every = (int)t.drain_every;
  // overflow lists exist only in the second-generation filter kernel, which launch_scan16 selects
  // under exactly this condition (32-bit survivor offsets, TFRS_SCAN16_V != 1)
  const char *gen = option("TFRS_SCAN16_V");
  const bool use_ovf = !(gen && gen[0] == '1') &&
                       (uint64_t)nq * s16.cap_l * (uint64_t)s16.nseg < (1ull << 32);
  s16.ovf_cnt = use_ovf ? w.ovf_cnt : nullptr;
  s16.ovf_buf = w.ovf_buf;
  s16.ovf_cap = kOvfCap;
  s16.zero_word = reinterpret_cast<uint32_t *>(w.redo);   // the flagged-query counter, re-armed
  s16.zero_aux = w.redo + 1 + nq;
  if ((rc = timed_scan16(s16, 2.0 * (double)nq * (double)n * d, stream)) != TFRS_OK) return rc;

  // prefilter top-K + exact re-scoring; flagged queries (list overflow, retained set too
  // large) are answered by the exact recompute path of the generic select kernel
  if ((rc = launch_list_topk16(q, nq, d, packed, w.buf, w.cnt, s16.cap_l, s16.nseg, k, w.qk,
                               img.norm_max, out_scores, out_idx, w.redo, idx_base, s16.ovf_cnt, w.ovf_buf,
                               kOvfCap, rowmap, (sp.stat && !lower_preset) ? w.thr_raw : nullptr,
                               w.redo + 1 + nq, stream, raw)) != TFRS_OK)
    return rc;
  SelectArgs se = {};
  se.nq = nq;
  se.k = k;
  se.q = q;
  se.d = d;
  se.packed = packed;
  se.source = kSrcRecompute;
  se.only_flagged = w.redo;
  se.part_keys = w.part_keys;
  se.rowmap = rowmap;
  se.idx_base = idx_base;
  se.rc_begin = row_lo;
  se.rc_end = row_lo + n;
  se.raw = raw;
  se.out_scores = out_scores;
  se.out_idx = out_idx;
  return launch_recompute(se, stream);
}

}  // namespace tfrs

using namespace tfrs;

extern "C" int tfrs_profile_enable(int on) {
  g_prof.enabled = on != 0;
  g_prof.used = 0;
  return TFRS_OK;
}

static int profile_sum(int kind, double *scan_ms_h, int *launches_h, double *flop_h) {
  double ms = 0.0, flop = 0.0;
  int cnt = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    if (kind >= 0 && g_prof.kind[i] != kind) continue;
    float t = 0.f;
    TFRS_HIP(hipEventSynchronize(g_prof.stop[i]));
    TFRS_HIP(hipEventElapsedTime(&t, g_prof.start[i], g_prof.stop[i]));
    ms += t;
    flop += g_prof.flop[i];
    ++cnt;
  }
  if (scan_ms_h) *scan_ms_h = ms;
  if (launches_h) *launches_h = cnt;
  if (flop_h) *flop_h = flop;
  return TFRS_OK;
}

extern "C" int tfrs_profile_read_kind(int kind, double *scan_ms_h, int *launches_h,
                                      double *flop_h) {
  TFRS_CHECK_ARG(kind >= 0 && kind <= 2, "profile_read_kind: kind must be 0, 1 or 2");
  return profile_sum(kind, scan_ms_h, launches_h, flop_h);
}

extern "C" int tfrs_profile_read(double *scan_ms_h, int *launches_h, double *flop_h) {
  const int rc = profile_sum(-1, scan_ms_h, launches_h, flop_h);
  g_prof.used = 0;
  return rc;
}

// ----------------------------------------------------------------------------------------
// index handle
// ----------------------------------------------------------------------------------------
struct tfrs_index {
  char *packed = nullptr;    // f32 image (exact scores)
  char *packed16 = nullptr;  // fp16 image (prefilter), meta[capacity / kTileN], norm_max
  StageMeta *meta = nullptr;
  float *norm_max = nullptr;
  int64_t n = 0;         // valid rows
  int64_t capacity = 0;  // rows allocated (multiple of kTileN)
  int d = 0;
  // Shuffled storage (default for indexes of >= 65536 rows; TFRS_INDEX_SHUFFLE=0 disables): every
  // appended block is stored in a pseudo-random row order and rowmap[image row] = original row.
  // The fp16-prefiltered search takes its threshold from bin maxima of sampled stages, which
  // presumes that a query's best candidates are spread over the image; a corpus stored cluster by
  // cluster (any locality in the row order) breaks that -- the shuffle restores it for every
  // input order.  Keys are formed from ORIGINAL row numbers, so results and tie order are unchanged.
  int32_t *rowmap = nullptr;
  // Host-visible flag word (pinned, mapped into the device address space; allocated with the first reserve and
  // kept for the life of the handle): kNonfiniteCandidates is set by the packer when an indexed row holds NaN / Inf,
  // kNonfiniteQueries by the query pass of a search.  Kernels write it with atomicOr ONLY in the error case, the
  // host reads it without touching the stream (tfrs_index_nonfinite).
  uint32_t *flags_h = nullptr, *flags_d = nullptr;
};

static void index_free(tfrs_index *index) {
  if (index->packed) (void)hipFree(index->packed);
  if (index->packed16) (void)hipFree(index->packed16);
  if (index->meta) (void)hipFree(index->meta);  // norm_max lives in the same block
  if (index->rowmap) (void)hipFree(index->rowmap);
  index->rowmap = nullptr;
  index->packed = index->packed16 = nullptr;
  index->meta = nullptr;
  index->norm_max = nullptr;
}

// the handle's host-visible flag word, allocated on first use
static int index_flag_word(tfrs_index *index) {
  if (index->flags_h) return TFRS_OK;
  hipError_t fe = hipHostMalloc(reinterpret_cast<void **>(&index->flags_h), 64, hipHostMallocMapped);
  if (fe == hipSuccess) fe = hipHostGetDevicePointer(reinterpret_cast<void **>(&index->flags_d), index->flags_h, 0);
  if (fe != hipSuccess) {
    if (index->flags_h) (void)hipHostFree(index->flags_h);
    index->flags_h = index->flags_d = nullptr;
    set_error("index: hipHostMalloc of the flag word failed: %s", hipGetErrorString(fe));
    return TFRS_ENOMEM;
  }
  *index->flags_h = 0u;
  return TFRS_OK;
}

extern "C" int tfrs_index_create(tfrs_index_t **out_h) {
  TFRS_CHECK_ARG(out_h != nullptr, "index_create: NULL output");
  *out_h = new (std::nothrow) tfrs_index();
  if (!*out_h) {
    set_error("index_create: out of host memory");
    return TFRS_ENOMEM;
  }
  return TFRS_OK;
}

extern "C" int tfrs_index_destroy(tfrs_index_t *index) {
  if (!index) return TFRS_OK;
  index_free(index);
  if (index->flags_h) (void)hipHostFree(index->flags_h);
  delete index;
  return TFRS_OK;
}

extern "C" int tfrs_index_reserve(tfrs_index_t *index, int64_t capacity, int d, void *stream) {
  TFRS_CHECK_ARG(index != nullptr, "index_reserve: NULL index");
  TFRS_CHECK_ARG(capacity >= 0 && d >= 1, "index_reserve: bad shape [%lld, %d]",
                 (long long)capacity, d);
  if (d > TFRS_MAX_DIM) {
    set_error("index: embedding dim %d > %d is not implemented", d, TFRS_MAX_DIM);
    return TFRS_ENOTIMPL;
  }
  index_free(index);
  int frc = index_flag_word(index);
  if (frc != TFRS_OK) return frc;
  *index->flags_h = 0u;          // (re-index: a new corpus, a clean record)
  index->n = 0;
  index->d = d;
  index->capacity = padded_rows(std::max<int64_t>(capacity, 1));
  const size_t bytes = (size_t)index->capacity * row_bytes(padded_dim(d));
  const size_t bytes16 = (size_t)index->capacity * row_bytes16(padded_dim16(d));
  const size_t nstages = (size_t)(index->capacity / kTileN);
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&index->packed), bytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&index->packed16), bytes16);
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void **>(&index->meta), (nstages + 1) * sizeof(StageMeta));
  const char *sh = option("TFRS_INDEX_SHUFFLE");
  const bool shuffle = capacity >= 65536 && !(sh && sh[0] == '0');
  if (e == hipSuccess && shuffle)
    e = hipMalloc(reinterpret_cast<void **>(&index->rowmap), (size_t)index->capacity * sizeof(int32_t));
  if (e != hipSuccess) {
    index_free(index);
    index->capacity = 0;
    set_error("index: hipMalloc(%zu + %zu) failed: %s", bytes, bytes16, hipGetErrorString(e));
    return TFRS_ENOMEM;
  }
  index->norm_max = reinterpret_cast<float *>(index->meta + nstages);
  TFRS_HIP(hipMemsetAsync(index->meta, 0, (nstages + 1) * sizeof(StageMeta),
                          (hipStream_t)stream));
  return TFRS_OK;
}

extern "C" int tfrs_index_append(tfrs_index_t *index, const float *block, int64_t nb,
                                 void *stream) {
  TFRS_CHECK_ARG(index && index->packed, "index_append: reserve first");
  TFRS_CHECK_ARG(nb >= 0 && (nb == 0 || block), "index_append: bad block");
  TFRS_CHECK_ARG(index->n + nb <= index->capacity, "index_append: %lld + %lld rows exceed capacity %lld",
                 (long long)index->n, (long long)nb, (long long)index->capacity);
  // zero-fill up to the next stage boundary so whole stages can always be read
  const int64_t zero_to = padded_rows(index->n + nb);
  int rc = launch_pack(block, nb, index->d, index->packed, index->n, zero_to, index->rowmap,
                       (hipStream_t)stream, index->flags_d);
  if (rc != TFRS_OK) return rc;
  // (re)build the fp16 image of every stage this block touched, from the f32 image
  rc = launch_pack16(index->packed, index->d, index->n, zero_to, index->packed16, index->meta,
                     index->norm_max, (hipStream_t)stream);
  if (rc != TFRS_OK) return rc;
  index->n += nb;
  return TFRS_OK;
}

extern "C" int tfrs_index_set(tfrs_index_t *index, const float *candidates, int64_t n, int d,
                              void *stream) {
  TFRS_CHECK_ARG(candidates != nullptr || n == 0, "index_set: NULL candidates");
  int rc = tfrs_index_reserve(index, n, d, stream);
  if (rc != TFRS_OK) return rc;
  return tfrs_index_append(index, candidates, n, stream);
}

extern "C" int tfrs_index_nonfinite(const tfrs_index_t *index, int reset_mask, int32_t *flags_h) {
  TFRS_CHECK_ARG(index && flags_h, "index_nonfinite: NULL argument");
  *flags_h = 0;
  if (!index->flags_h) return TFRS_OK;           // nothing indexed yet
  // plain read of pinned host memory: reflects every kernel that has COMPLETED (synchronise first for a verdict on
  // work still in flight); the device only ever ORs bits in, the host only clears them here
  const uint32_t v = __atomic_load_n(index->flags_h, __ATOMIC_ACQUIRE);
  *flags_h = (int32_t)v;
  if (reset_mask) __atomic_fetch_and(index->flags_h, ~(uint32_t)reset_mask, __ATOMIC_ACQ_REL);
  return TFRS_OK;
}

extern "C" int tfrs_index_note_nonfinite(tfrs_index_t *index, const float *x, int64_t count, const float *y,
                                         int64_t county, int bits, void *stream) {
  TFRS_CHECK_ARG(index && count >= 0 && county >= 0 && bits > 0, "index_note_nonfinite: bad argument");
  TFRS_CHECK_ARG((x || count == 0) && (y || county == 0), "index_note_nonfinite: NULL array");
  const int rc = index_flag_word(index);
  if (rc != TFRS_OK) return rc;
  if (!x) return launch_nonfinite_flag(y, county, index->flags_d, (uint32_t)bits, (hipStream_t)stream);
  return launch_nonfinite_flag(x, count, index->flags_d, (uint32_t)bits, (hipStream_t)stream, y, county);
}

extern "C" int64_t tfrs_index_size(const tfrs_index_t *index) { return index ? index->n : -1; }
extern "C" int tfrs_index_dim(const tfrs_index_t *index) { return index ? index->d : -1; }

extern "C" int tfrs_index_unpack(const tfrs_index_t *index, float *out, void *stream) {
  TFRS_CHECK_ARG(index && index->packed, "index_unpack: not indexed");
  return launch_unpack(index->packed, index->n, index->d, index->rowmap, out, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// BruteForce.call
// ----------------------------------------------------------------------------------------
extern "C" size_t tfrs_bruteforce_topk_workspace_bytes(int64_t nq, int64_t n, int d, int k) {
  (void)d;
  if (nq <= 0 || n <= 0 || k <= 0) return 256;
  return round_ws_bytes(nq, n, k, tuning());
}

extern "C" int tfrs_bruteforce_topk(const tfrs_index_t *index, const float *queries,
                                    int64_t nq, int k, float *out_scores, int32_t *out_idx,
                                    void *workspace, size_t workspace_bytes, void *stream) {
  if (!index || !index->packed) {
    set_error("bruteforce_topk: the index has not been built");
    return TFRS_ESTATE;
  }
  TFRS_CHECK_ARG(nq >= 0, "bruteforce_topk: nq < 0");
  TFRS_CHECK_ARG(k >= 1 && k <= TFRS_MAX_K, "bruteforce_topk: k=%d outside [1, %d]", k,
                 TFRS_MAX_K);
  TFRS_CHECK_ARG((int64_t)k <= index->n,
                 "input must have at least k columns (k=%d, candidates=%lld)", k,
                 (long long)index->n);
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(queries && out_scores && out_idx && workspace, "bruteforce_topk: NULL pointer");
  const TopkTuning t = tuning();
  const size_t need = round_ws_bytes(nq, index->n, k, t);
  if (workspace_bytes < need) {
    set_error("bruteforce_topk: workspace %zu < required %zu", workspace_bytes, need);
    return TFRS_ENOMEM;
  }
  const RoundWs w = carve_round_ws(static_cast<char *>(workspace), nq, index->n, k, t);
  if (t.f16_filter && k <= kMaxKF16) {
    const SamplePlan sp = plan_sample(index->n, k, t, use_stat(t, index->rowmap));
    if (sp.n_stages > 0) {
      const F16Image img = {index->packed16, index->meta, index->norm_max};
      return run_f16(queries, nq, index->d, index->packed, img, index->n, /*idx_base=*/0, k, sp,
                     /*lower_preset=*/false, out_scores, out_idx, w, t, (hipStream_t)stream,
                     index->rowmap, /*row_lo=*/0, /*raw=*/nullptr, index->flags_d);
    }
  }
  // (all-f32 rounds: IEEE comparisons, but the same contract -- a NaN / Inf query row is recorded in the flag word)
  int frc = launch_nonfinite_flag(queries, nq * (int64_t)index->d, index->flags_d, kNonfiniteQueries, (hipStream_t)stream);
  if (frc != TFRS_OK) return frc;
  int new_len = 0;
  return run_rounds(queries, nq, index->d, index->packed, index->n, /*idx_base=*/0,
                    /*seen=*/0, k, out_scores, out_idx, /*state_len=*/0, w, t,
                    (hipStream_t)stream, &new_len, index->rowmap);
}

// ---- paged search: K beyond TFRS_MAX_K --------------------------------------------------------
// tf.math.top_k has no limit on k (layers/factorized_top_k.py:605); the selection kernels hold K
// slots per wave.  A caller that wants more asks page by page: page p returns the best
// k <= TFRS_MAX_K rows among those that come strictly AFTER the last row of page p - 1 in the
// result order (score descending, row ascending), so the concatenated pages are exactly the sorted
// top-(sum of k).  Always the all-f32 rounds (the ceiling is applied where scores are compared and
// where keys are formed).
namespace tfrs {
__global__ void ceil_keys_kernel(const float *scores, const int32_t *rows, int64_t nq, int64_t ld,
                                 float *ceil_score, uint64_t *ceil_key) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nq) return;
  const float s = scores[r * ld];
  ceil_score[r] = s;
  ceil_key[r] = make_key(s, rows[r * ld]);
}
}  // namespace tfrs

extern "C" size_t tfrs_bruteforce_topk_below_workspace_bytes(int64_t nq, int64_t n, int d, int k) {
  if (nq <= 0 || n <= 0 || k <= 0 || d <= 0) return 256;
  return round_ws_bytes(nq, n, k, tuning()) + align_up((size_t)nq * 4) + align_up((size_t)nq * 8);
}

extern "C" int tfrs_bruteforce_topk_below(const tfrs_index_t *index, const float *queries, int64_t nq, int k,
                                          const float *last_scores, const int32_t *last_rows, int64_t last_ld,
                                          float *out_scores, int32_t *out_idx, void *workspace,
                                          size_t workspace_bytes, void *stream) {
  if (!index || !index->packed) {
    set_error("bruteforce_topk_below: the index has not been built");
    return TFRS_ESTATE;
  }
  TFRS_CHECK_ARG(nq >= 0 && k >= 1 && k <= TFRS_MAX_K, "bruteforce_topk_below: k=%d outside [1, %d]", k, TFRS_MAX_K);
  TFRS_CHECK_ARG((int64_t)k <= index->n, "input must have at least k columns (k=%d, candidates=%lld)", k,
                 (long long)index->n);
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(queries && out_scores && out_idx && workspace, "bruteforce_topk_below: NULL pointer");
  TFRS_CHECK_ARG((last_scores == nullptr) == (last_rows == nullptr) && (last_scores == nullptr || last_ld >= 1),
                 "bruteforce_topk_below: last_scores / last_rows must both be given (with their row stride) or both be NULL");
  const TopkTuning t = tuning();
  const size_t need = tfrs_bruteforce_topk_below_workspace_bytes(nq, index->n, index->d, k);
  if (workspace_bytes < need) {
    set_error("bruteforce_topk_below: workspace %zu < required %zu", workspace_bytes, need);
    return TFRS_ENOMEM;
  }
  const RoundWs w = carve_round_ws(static_cast<char *>(workspace), nq, index->n, k, t);
  float *ceil_score = nullptr;
  uint64_t *ceil_key = nullptr;
  if (last_scores) {
    ceil_key = reinterpret_cast<uint64_t *>(w.end);
    ceil_score = reinterpret_cast<float *>(w.end + align_up((size_t)nq * 8));
    hipLaunchKernelGGL(tfrs::ceil_keys_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, last_scores, last_rows, nq, last_ld, ceil_score, ceil_key);
    TFRS_LAUNCH_CHECK();
  }
  int frc = launch_nonfinite_flag(queries, nq * (int64_t)index->d, index->flags_d, kNonfiniteQueries, (hipStream_t)stream);
  if (frc != TFRS_OK) return frc;
  int new_len = 0;
  return run_rounds(queries, nq, index->d, index->packed, index->n, /*idx_base=*/0, /*seen=*/0, k, out_scores,
                    out_idx, /*state_len=*/0, w, t, (hipStream_t)stream, &new_len, index->rowmap, ceil_score,
                    ceil_key);
}

extern "C" int tfrs_bruteforce_topk_redo_count(const void *workspace, int64_t nq, int64_t n, int k,
                                               int32_t *redo_count_h, void *stream) {
  TFRS_CHECK_ARG(workspace && redo_count_h && nq > 0 && n > 0 && k > 0,
                 "bruteforce_topk_redo_count: bad argument");
  const TopkTuning t = tuning();
  *redo_count_h = 0;
  if (!(t.f16_filter && k <= kMaxKF16) || plan_sample(n, k, t, false).n_stages <= 0) return TFRS_OK;
  const RoundWs w = carve_round_ws(static_cast<char *>(const_cast<void *>(workspace)), nq, n, k, t);
  uint32_t v = 0;
  TFRS_HIP(hipMemcpyAsync(&v, w.redo, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  TFRS_HIP(hipStreamSynchronize((hipStream_t)stream));
  *redo_count_h = (int32_t)v;
  return TFRS_OK;
}

extern "C" int tfrs_bruteforce_topk_redo_reasons(const void *workspace, int64_t nq, int64_t n, int k,
                                                 int32_t *reasons_h, void *stream) {
  TFRS_CHECK_ARG(workspace && reasons_h && nq > 0 && n > 0 && k > 0,
                 "bruteforce_topk_redo_reasons: bad argument");
  const TopkTuning t = tuning();
  for (int i = 0; i < 4; ++i) reasons_h[i] = 0;
  if (!(t.f16_filter && k <= kMaxKF16) || plan_sample(n, k, t, false).n_stages <= 0) return TFRS_OK;
  const RoundWs w = carve_round_ws(static_cast<char *>(const_cast<void *>(workspace)), nq, n, k, t);
  uint32_t v[4] = {0u, 0u, 0u, 0u};
  TFRS_HIP(hipMemcpyAsync(v, w.redo + 1 + nq, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  TFRS_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < 4; ++i) reasons_h[i] = (int32_t)v[i];
  return TFRS_OK;
}

extern "C" int tfrs_debug_topk_plan(int64_t n, int k, int shuffled, int64_t *plan_h) {
  TFRS_CHECK_ARG(n > 0 && k >= 1 && k <= TFRS_MAX_K && plan_h, "debug_topk_plan: bad argument");
  const TopkTuning t = tuning();
  SamplePlan sp = {1, 0, 1, k, false};
  if (t.f16_filter && k <= kMaxKF16) sp = plan_sample(n, k, t, shuffled != 0 && t.stat);
  plan_h[0] = sp.stride;
  plan_h[1] = sp.n_stages;
  plan_h[2] = sp.bin_stages;
  plan_h[3] = sp.rank;
  plan_h[4] = sp.stat ? 1 : 0;
  return TFRS_OK;
}

// Test hook: the raw fp16 prefilter scores of rows [row_begin, row_end) (multiples of 128
// except at the corpus end), so that tests can check the error bound the filter relies on.
extern "C" int tfrs_debug_fp16_scores(const tfrs_index_t *index, const float *queries,
                                      int64_t nq, int64_t row_begin, int64_t row_end,
                                      float *out, float *scratch, void *stream) {
  if (!index || !index->packed16) {
    set_error("debug_fp16_scores: the index has not been built");
    return TFRS_ESTATE;
  }
  TFRS_CHECK_ARG(queries && out && scratch && nq >= 0, "debug_fp16_scores: NULL pointer");
  TFRS_CHECK_ARG(row_begin >= 0 && row_begin % kTileN == 0 && row_end >= row_begin &&
                     row_end <= index->n, "debug_fp16_scores: bad row range");
  const TopkTuning t = tuning();
  Scan16Args a = {};
  a.q = queries;
  a.nq = nq;
  a.d = index->d;
  a.packed16 = index->packed16;
  a.meta = index->meta;
  int rc = launch_query_kappa(queries, nq, index->d, scratch, scratch + nq, nullptr, (hipStream_t)stream);
  if (rc != TFRS_OK) return rc;
  a.qk = scratch;
  a.qscale = scratch + nq;
  a.n_qtiles = (int)((nq + kScan16QueriesPerWg - 1) / kScan16QueriesPerWg);
  a.stage0 = row_begin / kTileN;
  a.n_stages = (int)((row_end - row_begin + kTileN - 1) / kTileN);
  a.stage_stride = 1;
  a.row_limit = row_end;
  plan_stage_splits(a.n_stages, a.n_qtiles, t, &a.stages_per_split, &a.n_splits);
  a.dense = out;
  a.ld_dense = (int64_t)a.n_stages * kTileN;
  return launch_scan16(a, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// Streaming.call, one candidate block
// ----------------------------------------------------------------------------------------
// Streaming blocks go through the fp16 prefilter when they are large enough: at least 16
// stages, and -- for a block that cannot take its bound from the carried state -- enough
// bins for the threshold pass.
static bool stream_block_f16(int64_t nb, int k, const TopkTuning &t) {
  return t.f16_filter && k <= kMaxKF16 && nb >= 16 * kTileN;
}
static size_t stream_f16_extra_bytes(int64_t nq, int64_t nb, int d, int k) {
  return align_up((size_t)padded_rows(nb) * row_bytes16(padded_dim16(d))) +              // fp16 image
         align_up((size_t)(padded_rows(nb) / kTileN + 1) * sizeof(StageMeta)) +           // meta + norm_max
         2 * align_up((size_t)nq * k * 4);                                                // block top-K
}

extern "C" size_t tfrs_streaming_topk_workspace_bytes(int64_t nq, int64_t nb, int d, int k) {
  if (nq <= 0 || nb <= 0 || k <= 0 || d <= 0 || d > TFRS_MAX_DIM) return 256;
  const TopkTuning t = tuning();
  size_t b = round_ws_bytes(nq, nb, k, t) + align_up((size_t)padded_rows(nb) * row_bytes(padded_dim(d)));
  if (stream_block_f16(nb, k, t)) b += stream_f16_extra_bytes(nq, nb, d, k);
  return b;
}

extern "C" int tfrs_streaming_topk_update(const float *queries, int64_t nq, int d,
                                          const float *cand_block, int64_t nb,
                                          int64_t base_row, int k, float *state_scores,
                                          int32_t *state_idx, int32_t state_len,
                                          int32_t *new_len_h, void *workspace,
                                          size_t workspace_bytes, void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && nb >= 0 && d >= 1, "streaming_topk_update: bad shape");
  if (d > TFRS_MAX_DIM) {
    set_error("streaming_topk_update: embedding dim %d > %d is not implemented", d,
              TFRS_MAX_DIM);
    return TFRS_ENOTIMPL;
  }
  TFRS_CHECK_ARG(k >= 1 && k <= TFRS_MAX_K, "streaming_topk_update: k=%d outside [1, %d]", k,
                 TFRS_MAX_K);
  TFRS_CHECK_ARG(state_len >= 0 && state_len <= k, "streaming_topk_update: bad state_len");
  TFRS_CHECK_ARG(base_row >= 0 && base_row + nb <= 0x7FFFFFFFll,
                 "streaming_topk_update: row numbers exceed int32");
  if (new_len_h) *new_len_h = state_len;
  if (nq == 0 || nb == 0) return TFRS_OK;
  TFRS_CHECK_ARG(queries && cand_block && state_scores && state_idx && workspace,
                 "streaming_topk_update: NULL pointer");
  const TopkTuning t = tuning();
  const size_t need = tfrs_streaming_topk_workspace_bytes(nq, nb, d, k);
  if (workspace_bytes < need) {
    set_error("streaming_topk_update: workspace %zu < required %zu", workspace_bytes, need);
    return TFRS_ENOMEM;
  }
  const RoundWs w = carve_round_ws(static_cast<char *>(workspace), nq, nb, k, t);
  char *packed = w.end;
  int rc = launch_pack(cand_block, nb, d, packed, 0, padded_rows(nb), nullptr, (hipStream_t)stream);
  if (rc != TFRS_OK) return rc;
  int new_len = state_len;
  if (stream_block_f16(nb, k, t)) {
    // fp16-prefiltered block: bound from the carried state when it is full (no threshold
    // pass at all), else from the block's own bin maxima; then merge the block's exact top-K
    // into the state.
    const bool preset = (state_len == k);
    const SamplePlan sp = preset ? SamplePlan{1, 0, 1, k, false} : plan_sample(nb, k, t, false);
    if (preset || sp.n_stages > 0) {
      hipStream_t st = (hipStream_t)stream;
      char *p = packed + align_up((size_t)padded_rows(nb) * row_bytes(padded_dim(d)));
      char *packed16 = p;
      p += align_up((size_t)padded_rows(nb) * row_bytes16(padded_dim16(d)));
      StageMeta *meta = reinterpret_cast<StageMeta *>(p);
      const size_t nstages = (size_t)(padded_rows(nb) / kTileN);
      float *norm_max = reinterpret_cast<float *>(meta + nstages);
      p += align_up((nstages + 1) * sizeof(StageMeta));
      float *blk_scores = reinterpret_cast<float *>(p);
      p += align_up((size_t)nq * k * 4);
      int32_t *blk_idx = reinterpret_cast<int32_t *>(p);
      TFRS_HIP(hipMemsetAsync(norm_max, 0, sizeof(float), st));
      if ((rc = launch_pack16(packed, d, 0, padded_rows(nb), packed16, meta, norm_max, st)) != TFRS_OK)
        return rc;
      if (preset) {
        hipLaunchKernelGGL(thr_from_state_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0,
                           st, state_scores, nq, k, w.thr);
        TFRS_LAUNCH_CHECK();
      }
      const F16Image img = {packed16, meta, norm_max};
      if ((rc = run_f16(queries, nq, d, packed, img, nb, base_row, k, sp, preset, blk_scores,
                        blk_idx, w, t, st)) != TFRS_OK)
        return rc;
      SelectArgs se = {};
      se.nq = nq;
      se.k = k;
      se.state_scores = state_scores;
      se.state_idx = state_idx;
      se.state_len = state_len;
      se.source = kSrcParts;
      se.part_scores = blk_scores;
      se.part_idx = blk_idx;
      se.nparts = 1;
      se.k_in = k;
      se.d = 8;
      se.out_scores = state_scores;
      se.out_idx = state_idx;
      if ((rc = launch_select(se, st)) != TFRS_OK) return rc;
      new_len = (int)std::min<int64_t>(k, (int64_t)state_len + nb);
      if (new_len_h) *new_len_h = new_len;
      return TFRS_OK;
    }
  }
  rc = run_rounds(queries, nq, d, packed, nb, /*idx_base=*/base_row, /*seen=*/base_row, k,
                  state_scores, state_idx, state_len, w, t, (hipStream_t)stream, &new_len);
  if (new_len_h) *new_len_h = new_len;
  return rc;
}

// ----------------------------------------------------------------------------------------
// Streaming.call, a GROUP of candidate blocks read in place (topk_raw.hip)
// ----------------------------------------------------------------------------------------
// The per-block entry point above costs a pack + pack16 + ~6 search launches per block and writes
// every candidate byte back to HBM twice (f32 image, fp16 image) before reading it again: over a
// dataset of 191 blocks of 65536 x 128 that is launch-bound at small query batches (18 ms at
// B = 1: 0.045 of the HBM rate) and 2x BruteForce at B = 8192.  Here the blocks of a group are
// described once (RawTable) and searched where they lie:
//   * nq <= raw_max_nq (TFRS_STREAM_RAW_MAX_NQ, default 64): all-f32 rounds with the raw scan
//     kernel -- dense round over the first rows while the state is not full, then geometric
//     filtered rounds (rho = 8) with the threshold of the running state; every candidate byte is
//     read from HBM once per 64 queries;
//   * larger batches: ONE fp16 image of the group built straight from the blocks, then
//     geometric fp16-prefiltered rounds (rho = TFRS_STREAM_RHO16, default 4) whose bound is the
//     running state's exact K-th score (the first range takes it from its own bin maxima);
//     survivors are re-scored exactly from the blocks.
// Identical results to the per-block path and to the oracle's Streaming fold: every round is exact.
namespace tfrs {

static int64_t stream_raw_max_nq() { return std::max<int64_t>(0, env_i64("TFRS_STREAM_RAW_MAX_NQ", 64)); }
static int64_t stream_rho16() { return std::max<int64_t>(2, env_i64("TFRS_STREAM_RHO16", 4)); }
constexpr int64_t kStreamFirstRange = 262144;   // rows of the first fp16 range (threshold pass of its own)

// Query batches up to this size take the block-fed fp16 filter (rawscan16_kernel: one pass over the f32
// blocks, <= 256 queries per workgroup); 0 sends them to the exact f32-MFMA scan (<= TFRS_STREAM_RAW_MAX_NQ
// queries) or the fp16 image (above)
static int64_t stream_raw16_max_nq() { return std::max<int64_t>(0, env_i64("TFRS_STREAM_RAW16_MAX_NQ", 1024)); }
// (round 5: beyond one wave's registers -- 128 queries at dim 128, 256 below -- the waves of an 8-wave workgroup split
// the queries, 512 per workgroup, and the stage is converted once per workgroup through LDS: rawscan16w_kernel.  Dims
// below 32 keep the 256-query limit.  Measured on 12.5 M x 128 (profiles/r05_streaming_wide.txt): 257 queries 2.62 ms
// against 3.30 through the group's fp16 image, 512: 2.9-3.1 against 3.7; from ~800 queries on the image wins again
// (1024: 5.2-5.6 against 5.0) -- the kernel's scoring phase runs at a third of the matrix pipe's rate.
// Round 6: rawscan16pc_kernel (producer / consumer waves) takes 15-20 % off these -- 512 queries 2.46-2.51 ms -- and wins
// up to two full query tiles: 768 / 1024 queries 4.50 / 4.46 ms against 4.74 / 4.97 through the image, 1536: 8.6 against 7.8;
// the limit moved from 640 to 1024.)
static int64_t raw16_max_nq(int d) { return d >= 32 ? stream_raw16_max_nq() : std::min<int64_t>(stream_raw16_max_nq(), 256); }
// ... and, up to dim 64, from this size on: with one group of 32 queries the exact scan is copy-bound as well
// there and needs half the launches per range (12.5 M x 64, one query: 0.89 against 1.00 ms); at dim 128 its
// 64 matrix-core steps per row cost as much as the copy (one query 1.71 against 1.64 ms, 32: 1.88 against 1.70)
static int64_t stream_raw16_min_nq() { return std::max<int64_t>(1, env_i64("TFRS_STREAM_RAW16_MIN_NQ", 33)); }
static bool group_uses_raw16(int64_t nq, int d, int k, const TopkTuning &t) {
  return t.f16_filter && k <= kMaxKF16 && nq >= (d <= 64 ? stream_raw16_min_nq() : 1) &&
         nq <= raw16_max_nq(d);
}
// query groups of 32 per workgroup: eight up to dim 64, four at dim 128 (eight resident groups of dim 128 do
// not fit the register file).  More queries than one workgroup holds make several query tiles per split of
// the rows; the tiles of a split are neighbours in the XCD-aware workgroup order, so the second one reads the
// rows from the XCD's L2
static int raw16_qg(int64_t nq, int d) {
  // the 512-query workgroup (rawscan16pc_kernel) from this many queries on (TFRS_STREAM_RAW16_WIDE_FROM): beyond what
  // ONE resident query tile holds -- 128 queries at dim 128 (four groups per wave), 256 below.  Two tiles of rawscan16_kernel
  // read every row twice (the second time from L2): 12.5 M x 128, 129 / 192 / 256 queries 2.64 / 2.59 / 2.61 ms against
  // 1.94 / 1.97 / 2.03 here; at dim 64 (eight groups per wave) 129 .. 256 queries measure the same either way
  const int64_t wide_from = env_i64("TFRS_STREAM_RAW16_WIDE_FROM", d >= 128 ? 129 : 257);
  if (nq >= std::max<int64_t>(wide_from, 1) && d >= 32) return 16;     // rawscan16w_kernel: 512 queries per workgroup
  const int cap = d <= 64 ? 8 : 4;
  const int want = nq <= 32 ? 1 : nq <= 64 ? 2 : nq <= 128 ? 4 : 8;
  return std::min(want, cap);
}
static bool group_uses_f16(int64_t nq, int64_t n, int d, int k, const TopkTuning &t) {
  if (group_uses_raw16(nq, d, k, t)) return false;
  if (!(t.f16_filter && k <= kMaxKF16) || nq <= stream_raw_max_nq()) return false;
  // the first range must be able to take its bound from bin maxima
  return plan_sample(std::min<int64_t>(n, kStreamFirstRange), k, t, false).n_stages > 0;
}
static int raw_qg(int64_t nq) { return nq <= 32 ? 1 : 2; }
static int64_t group_ld_dense(int64_t n, int k, const TopkTuning &t) {
  return std::max(dense_rows(n, k, t), dense_rows(std::min<int64_t>(n, kStreamFirstRange), k, t));
}
// survivor-list entries per query of a raw round: nseg = n_splits segments of segment_cap each
static int raw_max_splits(int64_t nq, const TopkTuning &t) {
  const int64_t n_qtiles = (nq + 32 * raw_qg(nq) - 1) / (32 * raw_qg(nq));
  return (int)std::max<int64_t>(1, (t.target_wgs + n_qtiles - 1) / n_qtiles);
}
static int64_t raw_list_entries(int64_t nq, int k, const TopkTuning &t) {
  const int nseg = raw_max_splits(nq, t);
  int64_t worst = 0;
  for (int sg = 1; sg <= nseg; sg *= 2) worst = std::max<int64_t>(worst, (int64_t)sg * ((int64_t)segment_cap(k, sg, t) + 1));
  return std::max<int64_t>(worst, (int64_t)nseg * ((int64_t)segment_cap(k, nseg, t) + 1));
}

struct GroupWs {
  RawTable *table;
  RoundWs w;
  char *packed16;        // fp16 path only
  StageMeta *meta;
  float *norm_max;
  float *blk_scores;
  int32_t *blk_idx;
};

static size_t group_ws_bytes(int64_t nq, int64_t n, int d, int k, const TopkTuning &t) {
  size_t b = align_up(sizeof(RawTable));
  b += round_ws_bytes(nq, n, k, t, group_ld_dense(n, k, t));
  b += align_up((size_t)nq * raw_list_entries(nq, k, t) * 8) + align_up((size_t)nq * raw_max_splits(nq, t) * 4);
  if (group_uses_f16(nq, n, d, k, t)) {
    b += align_up((size_t)padded_rows(n) * row_bytes16(padded_dim16(d)));
    b += align_up((size_t)(padded_rows(n) / kTileN + 1) * sizeof(StageMeta));
    b += 2 * align_up((size_t)nq * k * 4);
  }
  if (group_uses_raw16(nq, d, k, t)) b += align_up(sizeof(float)) + 2 * align_up((size_t)nq * k * 4);   // norm_max, one round's top-K
  return b;
}

// All-f32 rounds over the group's rows [0, n) through the raw scan kernel (same round structure as
// run_rounds).  raw_buf / raw_cnt: the survivor lists of the raw geometry (one segment per split).
static int run_rounds_raw(const float *q, int64_t nq, int d, const RawTable *table, int64_t n,
                          int64_t idx_base, int64_t seen, int k, float *state_scores, int32_t *state_idx,
                          int state_len, const RoundWs &w, uint2 *raw_buf, uint32_t *raw_cnt,
                          int64_t raw_entries, const TopkTuning &t, hipStream_t stream, int *new_len) {
  int len = state_len;
  if (n <= 0 || nq <= 0) {
    *new_len = len;
    return TFRS_OK;
  }
  const int qg = raw_qg(nq);
  const int n_qtiles = (int)((nq + 32 * qg - 1) / (32 * qg));
  int rc;
  RawScanArgs sa = {};
  sa.q = q;
  sa.nq = nq;
  sa.d = d;
  sa.table = table;
  sa.n_qtiles = n_qtiles;
  sa.qg = qg;
  sa.thr = w.thr;
  sa.cnt = raw_cnt;
  sa.buf = raw_buf;
  sa.dense = w.dense;
  sa.ld_dense = w.ld_dense;

  SelectArgs se = {};
  se.nq = nq;
  se.k = k;
  se.state_scores = state_scores;
  se.state_idx = state_idx;
  se.q = q;
  se.d = d;
  se.raw = table;
  se.idx_base = idx_base;
  se.out_scores = state_scores;
  se.out_idx = state_idx;
  se.out_thr = w.thr;

  int64_t lo = 0;
  if (len < k) {
    const int64_t n0 = std::min<int64_t>(n, w.ld_dense);
    sa.c_begin = 0;
    sa.c_end = n0;
    plan_splits(n0, n_qtiles, t, &sa.split_len, &sa.n_splits);
    if ((rc = launch_rawscan(sa, /*materialize=*/true, stream)) != TFRS_OK) return rc;
    se.state_len = len;
    se.source = kSrcDense;
    se.dense = w.dense;
    se.ld_dense = w.ld_dense;
    se.n_dense = n0;
    if ((rc = launch_select(se, stream)) != TFRS_OK) return rc;
    len = (int)std::min<int64_t>(k, (int64_t)len + n0);
    seen += n0;
    lo = n0;
  } else {
    hipLaunchKernelGGL(thr_from_state_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream,
                       state_scores, nq, k, w.thr);
    TFRS_LAUNCH_CHECK();
  }
  while (lo < n) {   // the state is full here (len == k), w.thr is current
    int64_t span = std::max<int64_t>((t.rho - 1) * std::max<int64_t>(seen, 1), kTileN);
    span = padded_rows(span);
    const int64_t hi = (n - lo <= span) ? n : lo + span;
    sa.c_begin = lo;
    sa.c_end = hi;
    plan_splits(hi - lo, n_qtiles, t, &sa.split_len, &sa.n_splits);
    sa.nseg = sa.n_splits;
    sa.cap_l = segment_cap(k, sa.nseg, t);
    if ((int64_t)sa.nseg * sa.cap_l > raw_entries) {
      set_error("streaming blocks: survivor workspace too small (%d segments x %u)", sa.nseg, sa.cap_l);
      return TFRS_ENOMEM;
    }
    if ((rc = launch_rawscan(sa, /*materialize=*/false, stream)) != TFRS_OK) return rc;
    se.state_len = len;
    se.source = kSrcList;
    se.buf = raw_buf;
    se.cnt = raw_cnt;
    se.cap_l = sa.cap_l;
    se.nseg = sa.nseg;
    se.rc_begin = lo;
    se.rc_end = hi;
    if ((rc = launch_select(se, stream)) != TFRS_OK) return rc;
    seen += hi - lo;
    lo = hi;
  }
  *new_len = len;
  return TFRS_OK;
}

}  // namespace tfrs

extern "C" size_t tfrs_streaming_topk_blocks_workspace_bytes(int64_t nq, int64_t total_rows, int d, int k) {
  if (nq <= 0 || total_rows <= 0 || k <= 0 || d <= 0 || d > TFRS_MAX_DIM) return 256;
  return group_ws_bytes(nq, total_rows, d, k, tuning());
}

extern "C" int tfrs_streaming_topk_update_blocks(const float *queries, int64_t nq, int d,
                                                 const float *const *blocks_h, const int64_t *block_rows_h,
                                                 int nblocks, int64_t base_row, int64_t seen_rows, int k,
                                                 float *state_scores, int32_t *state_idx, int32_t state_len,
                                                 int32_t *new_len_h, void *workspace, size_t workspace_bytes,
                                                 void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && d >= 1 && nblocks >= 0, "streaming_topk_update_blocks: bad shape");
  TFRS_CHECK_ARG(d == padded_dim(d) && d <= TFRS_MAX_DIM,
                 "streaming_topk_update_blocks: dim %d is not one of 8, 16, 32, 64, 128 (use the per-block entry point)", d);
  TFRS_CHECK_ARG(nblocks <= kRawMaxBlocks, "streaming_topk_update_blocks: %d blocks > %d per call", nblocks,
                 kRawMaxBlocks);
  TFRS_CHECK_ARG(k >= 1 && k <= TFRS_MAX_K, "streaming_topk_update_blocks: k=%d outside [1, %d]", k, TFRS_MAX_K);
  TFRS_CHECK_ARG(state_len >= 0 && state_len <= k, "streaming_topk_update_blocks: bad state_len");
  TFRS_CHECK_ARG(seen_rows >= 0 && base_row >= 0, "streaming_topk_update_blocks: negative row counters");
  if (new_len_h) *new_len_h = state_len;
  if (nblocks == 0) return TFRS_OK;
  TFRS_CHECK_ARG(blocks_h && block_rows_h, "streaming_topk_update_blocks: NULL block list");
  RawTable table;
  table.n_blocks = 0;
  table.uniform_rows = 0;
  int64_t total = 0;
  bool uniform = true;
  int64_t first_rows = -1;
  for (int b = 0; b < nblocks; ++b) {
    const int64_t nb = block_rows_h[b];
    TFRS_CHECK_ARG(nb >= 0, "streaming_topk_update_blocks: block %d has %lld rows", b, (long long)nb);
    if (nb == 0) continue;   // empty dataset elements contribute nothing
    TFRS_CHECK_ARG(blocks_h[b] != nullptr && (reinterpret_cast<uintptr_t>(blocks_h[b]) & 15) == 0,
                   "streaming_topk_update_blocks: block %d is NULL or not 16-byte aligned", b);
    const int i = table.n_blocks++;
    table.ptr[i] = blocks_h[b];
    table.row_start[i] = total;
    // uniform: every block BEFORE the last one has the rows of the first
    if (first_rows < 0) first_rows = nb;
    else if (table.row_start[i] != (int64_t)i * first_rows) uniform = false;
    total += nb;
  }
  table.row_start[table.n_blocks] = total;
  for (int i = table.n_blocks + 1; i <= kRawMaxBlocks; ++i) table.row_start[i] = total;
  for (int i = table.n_blocks; i < kRawMaxBlocks; ++i) table.ptr[i] = nullptr;
  if (uniform && first_rows > 0 && first_rows <= 0x7FFFFFFF) {
    // (the last block may be shorter, never longer: a longer one would hold rows of "block n")
    const int64_t last = total - table.row_start[table.n_blocks > 0 ? table.n_blocks - 1 : 0];
    if (table.n_blocks <= 1 || last <= first_rows) table.uniform_rows = (int32_t)first_rows;
  }
  table.total_rows = total;
  TFRS_CHECK_ARG(base_row + total <= 0x7FFFFFFFll, "streaming_topk_update_blocks: row numbers exceed int32");
  if (nq == 0 || total == 0) return TFRS_OK;
  TFRS_CHECK_ARG(queries && state_scores && state_idx && workspace, "streaming_topk_update_blocks: NULL pointer");
  const TopkTuning t = tuning();
  const size_t need = group_ws_bytes(nq, total, d, k, t);
  if (workspace_bytes < need) {
    set_error("streaming_topk_update_blocks: workspace %zu < required %zu", workspace_bytes, need);
    return TFRS_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream;
  char *p = static_cast<char *>(workspace);
  RawTable *table_dev = reinterpret_cast<RawTable *>(p);
  p += align_up(sizeof(RawTable));
  const RoundWs w = carve_round_ws(p, nq, total, k, t, group_ld_dense(total, k, t));
  p = w.end;
  uint2 *raw_buf = reinterpret_cast<uint2 *>(p);
  const int64_t raw_entries = raw_list_entries(nq, k, t);
  p += align_up((size_t)nq * raw_entries * 8);
  uint32_t *raw_cnt = reinterpret_cast<uint32_t *>(p);
  p += align_up((size_t)nq * raw_max_splits(nq, t) * 4);
  int rc = launch_raw_table_write(table, table_dev, st);
  if (rc != TFRS_OK) return rc;
  int new_len = state_len;

  if (group_uses_raw16(nq, d, k, t)) {
    // ---- block-fed fp16 filter: an exact dense round fills the state (it is the bound of everything
    // after it), then geometrically growing ranges, each ONE pass over its f32 rows --------------------
    float *norm_max = reinterpret_cast<float *>(p);
    p += align_up(sizeof(float));
    float *blk_scores = reinterpret_cast<float *>(p);
    p += align_up((size_t)nq * k * 4);
    int32_t *blk_idx = reinterpret_cast<int32_t *>(p);
    TFRS_HIP(hipMemsetAsync(norm_max, 0, sizeof(float), st));
    int64_t lo = 0, seen = seen_rows;
    if (new_len < k) {
      const int64_t n0 = std::min<int64_t>(total, w.ld_dense);
      if ((rc = run_rounds_raw(queries, nq, d, table_dev, n0, base_row, seen, k, state_scores, state_idx, new_len,
                               w, raw_buf, raw_cnt, raw_entries, t, st, &new_len)) != TFRS_OK)
        return rc;
      lo = n0;
      seen += n0;
    }
    if (lo < total && (rc = launch_query_kappa(queries, nq, d, w.qk, w.qscale, nullptr, st)) != TFRS_OK) return rc;
    while (lo < total) {   // the state is full here: lo > 0 only after a dense round of >= k rows
      int64_t span = std::max<int64_t>((t.rho - 1) * std::max<int64_t>(seen, 1), 16 * kTileN);
      span = padded_rows(span);
      int64_t hi = (total - lo <= span) ? total : lo + span;
      if (total - hi < 16 * kTileN) hi = total;
      // the bound of the range: the state's exact K-th score, read where it lies (column k - 1)
      if ((rc = run_raw16_range(queries, nq, d, table_dev, norm_max, lo, hi - lo, base_row, k, state_scores + (k - 1),
                                k, raw16_qg(nq, d), blk_scores, blk_idx, w, t, st)) != TFRS_OK)
        return rc;
      SelectArgs se = {};
      se.nq = nq;
      se.k = k;
      se.state_scores = state_scores;
      se.state_idx = state_idx;
      se.state_len = new_len;
      se.source = kSrcParts;
      se.part_scores = blk_scores;
      se.part_idx = blk_idx;
      se.nparts = 1;
      se.k_in = k;
      se.d = 8;
      se.out_scores = state_scores;
      se.out_idx = state_idx;
      if ((rc = launch_select(se, st)) != TFRS_OK) return rc;
      seen += hi - lo;
      lo = hi;
    }
    if (new_len_h) *new_len_h = new_len;
    return TFRS_OK;
  }
  if (!group_uses_f16(nq, total, d, k, t)) {
    rc = run_rounds_raw(queries, nq, d, table_dev, total, base_row, seen_rows, k, state_scores, state_idx,
                        state_len, w, raw_buf, raw_cnt, raw_entries, t, st, &new_len);
    if (new_len_h) *new_len_h = new_len;
    return rc;
  }

  // ---- fp16-prefiltered rounds over ONE image of the group ---------------------------------------
  char *packed16 = p;
  p += align_up((size_t)padded_rows(total) * row_bytes16(padded_dim16(d)));
  StageMeta *meta = reinterpret_cast<StageMeta *>(p);
  const size_t nstages = (size_t)(padded_rows(total) / kTileN);
  float *norm_max = reinterpret_cast<float *>(meta + nstages);
  p += align_up((nstages + 1) * sizeof(StageMeta));
  float *blk_scores = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * k * 4);
  int32_t *blk_idx = reinterpret_cast<int32_t *>(p);
  TFRS_HIP(hipMemsetAsync(norm_max, 0, sizeof(float), st));
  if ((rc = launch_pack16_raw(table_dev, total, d, packed16, meta, norm_max, st)) != TFRS_OK) return rc;
  const F16Image img = {packed16, meta, norm_max};
  const int64_t rho = stream_rho16();
  int64_t lo = 0, seen = seen_rows;
  while (lo < total) {
    const bool preset = (new_len == k);
    int64_t hi;
    SamplePlan sp = {1, 0, 1, k, false};
    if (preset) {
      int64_t span = std::max<int64_t>((rho - 1) * std::max<int64_t>(seen, 1), 16 * kTileN);
      span = padded_rows(span);
      hi = (total - lo <= span) ? total : lo + span;
      // a short tail (fewer than 16 stages) is not worth a round of its own
      if (total - hi < 16 * kTileN) hi = total;
      hipLaunchKernelGGL(thr_from_state_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st,
                         state_scores, nq, k, w.thr);
      TFRS_LAUNCH_CHECK();
    } else {
      // the state is not full: this range's own bin maxima give the bound (guaranteed plan).  Only the
      // FIRST range of a stream gets here (it fills the state: >= 8 * K bins means >= K rows).
      hi = std::min<int64_t>(total, lo + kStreamFirstRange);
      if (total - hi < 16 * kTileN) hi = total;
      sp = plan_sample(hi - lo, k, t, false);
      if (sp.n_stages <= 0) {
        set_error("streaming_topk_update_blocks: internal: no threshold plan for %lld rows", (long long)(hi - lo));
        return TFRS_ESTATE;
      }
    }
    if ((rc = run_f16(queries, nq, d, /*packed=*/nullptr, img, hi - lo, base_row, k, sp, preset, blk_scores,
                      blk_idx, w, t, st, /*rowmap=*/nullptr, lo, table_dev)) != TFRS_OK)
      return rc;
    SelectArgs se = {};
    se.nq = nq;
    se.k = k;
    se.state_scores = state_scores;
    se.state_idx = state_idx;
    se.state_len = new_len;
    se.source = kSrcParts;
    se.part_scores = blk_scores;
    se.part_idx = blk_idx;
    se.nparts = 1;
    se.k_in = k;
    se.d = 8;
    se.out_scores = state_scores;
    se.out_idx = state_idx;
    if ((rc = launch_select(se, st)) != TFRS_OK) return rc;
    new_len = (int)std::min<int64_t>(k, (int64_t)new_len + (hi - lo));
    seen += hi - lo;
    lo = hi;
  }
  if (new_len_h) *new_len_h = new_len;
  return TFRS_OK;
}

// ----------------------------------------------------------------------------------------
// top-K update from a materialised score block (embedding dims above TFRS_MAX_DIM)
// ----------------------------------------------------------------------------------------
extern "C" int tfrs_topk_update_from_scores(const float *scores, int64_t nq, int64_t nb, int64_t ld,
                                            int64_t base_row, int k, float *state_scores,
                                            int32_t *state_idx, int32_t state_len,
                                            int32_t *new_len_h, void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && nb >= 0 && ld >= nb, "topk_update_from_scores: bad shape");
  TFRS_CHECK_ARG(k >= 1 && k <= TFRS_MAX_K, "topk_update_from_scores: k=%d outside [1, %d]", k, TFRS_MAX_K);
  TFRS_CHECK_ARG(state_len >= 0 && state_len <= k, "topk_update_from_scores: bad state_len");
  TFRS_CHECK_ARG(base_row >= 0 && base_row + nb <= 0x7FFFFFFFll,
                 "topk_update_from_scores: row numbers exceed int32");
  if (new_len_h) *new_len_h = state_len;
  if (nq == 0 || nb == 0) return TFRS_OK;
  TFRS_CHECK_ARG(scores && state_scores && state_idx, "topk_update_from_scores: NULL pointer");
  SelectArgs se = {};
  se.nq = nq;
  se.k = k;
  se.state_scores = state_scores;
  se.state_idx = state_idx;
  se.state_len = state_len;
  se.source = kSrcDense;
  se.dense = scores;
  se.ld_dense = ld;
  se.n_dense = nb;
  se.idx_base = base_row;
  se.d = 8;
  se.out_scores = state_scores;
  se.out_idx = state_idx;
  const int rc = launch_select(se, (hipStream_t)stream);
  if (rc == TFRS_OK && new_len_h) *new_len_h = (int32_t)std::min<int64_t>(k, (int64_t)state_len + nb);
  return rc;
}

// ----------------------------------------------------------------------------------------
// merge of partial lists
// ----------------------------------------------------------------------------------------
extern "C" size_t tfrs_topk_merge_workspace_bytes(int64_t, int, int, int) { return 256; }

static int topk_merge_impl(const float *scores_parts, const int32_t *idx_parts, int nparts,
                           int64_t part_stride, int64_t nq, int k_in, int k_out, float *out_scores,
                           int32_t *out_idx, void *stream) {
  TFRS_CHECK_ARG(nparts >= 1 && k_in >= 1 && nq >= 0, "topk_merge: bad shape");
  TFRS_CHECK_ARG(k_out >= 1 && k_out <= TFRS_MAX_K && (int64_t)k_out <= (int64_t)nparts * k_in,
                 "topk_merge: k_out=%d must be in [1, min(%d, nparts*k_in=%lld)]", k_out,
                 TFRS_MAX_K, (long long)nparts * k_in);
  TFRS_CHECK_ARG(part_stride == 0 || part_stride >= nq * (int64_t)k_in,
                 "topk_merge: part_stride smaller than one part");
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(scores_parts && idx_parts && out_scores && out_idx, "topk_merge: NULL pointer");
  SelectArgs se = {};
  se.nq = nq;
  se.k = k_out;
  se.state_len = 0;
  se.source = kSrcParts;
  se.part_scores = scores_parts;
  se.part_idx = idx_parts;
  se.nparts = nparts;
  se.k_in = k_in;
  se.part_stride = part_stride;
  se.d = 8;
  se.out_scores = out_scores;
  se.out_idx = out_idx;
  se.out_thr = nullptr;
  return launch_select(se, (hipStream_t)stream);
}

extern "C" int tfrs_topk_merge(const float *scores_parts, const int32_t *idx_parts, int nparts,
                               int64_t nq, int k_in, int k_out, float *out_scores,
                               int32_t *out_idx, void *, size_t, void *stream) {
  return topk_merge_impl(scores_parts, idx_parts, nparts, 0, nq, k_in, k_out, out_scores, out_idx,
                         stream);
}

extern "C" int tfrs_topk_merge_strided(const float *scores_parts, const int32_t *idx_parts,
                                       int nparts, int64_t part_stride, int64_t nq, int k_in,
                                       int k_out, float *out_scores, int32_t *out_idx,
                                       void *stream) {
  return topk_merge_impl(scores_parts, idx_parts, nparts, part_stride, nq, k_in, k_out, out_scores,
                         out_idx, stream);
}

// ----------------------------------------------------------------------------------------
// _exclude, rank-of-positive, id match: small per-query kernels
// ----------------------------------------------------------------------------------------
namespace tfrs {

// One wave per query: adjusted = score - 1e5 * isin (float32, like the reference),
// then an exact top-kout by (adjusted desc, column asc) via rank counting (kin <= 1029).
__global__ void __launch_bounds__(256) exclude_kernel(const float *scores, const int32_t *ids,
                                                      int64_t nq, int kin,
                                                      const int32_t *exclude, int ne, int kout,
                                                      float *out_scores, int32_t *out_ids) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= nq) return;
  float *adj = reinterpret_cast<float *>(smem) + (size_t)wave * kin;
  for (int c = lane; c < kin; c += 64) {
    const int32_t id = ids[row * kin + c];
    bool isin = false;
    for (int e = 0; e < ne; ++e) isin = isin || (exclude[row * ne + e] == id);
    adj[c] = scores[row * kin + c] - (isin ? 1.0e5f : 0.0f);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < kin; c += 64) {
    const float v = adj[c];
    int rank = 0;
    for (int o = 0; o < kin; ++o) {
      const float u = adj[o];
      rank += (u > v) || (u == v && o < c);
    }
    if (rank < kout) {
      out_scores[row * kout + rank] = scores[row * kin + c];
      out_ids[row * kout + rank] = ids[row * kin + c];
    }
  }
}

struct KsArg {
  int32_t v[16];
};

__global__ void __launch_bounds__(256) rank_of_positive_kernel(
    const float *q, const float *c, int64_t nq, int d, const float *topk, int kmax,
    const KsArg ks, int nks, float *out_hits) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nq) return;
  // positive score: the same d-ordered fma chain as the scoring kernels, so the positive
  // ties exactly with its own copy in the corpus.
  float pos = 0.0f;
  for (int k = 0; k < d; ++k) pos = __builtin_fmaf(q[row * d + k], c[row * d + k], pos);
  // sorted descending: #greater within the first kk columns == #greater overall, capped
  int greater = 0;
  for (int j = lane; j < kmax; j += 64) greater += (topk[row * kmax + j] > pos) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) greater += __shfl_xor(greater, off);
  const bool finite = __builtin_isfinite(pos);
  if (lane < nks) out_hits[(int64_t)lane * nq + row] = (finite && greater < ks.v[lane & 15]) ? 1.0f : 0.0f;
}

__global__ void __launch_bounds__(256) id_match_kernel(const int32_t *ids, const int32_t *true_ids,
                                                       int64_t nq, int kmax, const KsArg ks,
                                                       int nks, float *out_hits) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nq) return;
  const int32_t t = true_ids[row];
  int first = kmax;  // first column that matches
  for (int j = lane; j < kmax; j += 64)
    if (ids[row * kmax + j] == t) first = min(first, j);
  for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
  if (lane < nks) out_hits[(int64_t)lane * nq + row] = (first < ks.v[lane & 15]) ? 1.0f : 0.0f;
}

}  // namespace tfrs

extern "C" int tfrs_topk_exclude(const float *scores, const int32_t *ids, int64_t nq, int kin,
                                 const int32_t *exclude, int ne, int k, float *out_scores,
                                 int32_t *out_ids, void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && kin >= 1 && ne >= 0 && k >= 1, "topk_exclude: bad shape");
  TFRS_CHECK_ARG(kin <= 4096, "topk_exclude: kin=%d too large", kin);
  if (nq == 0) return TFRS_OK;
  const int kout = k < kin ? k : kin;
  hipLaunchKernelGGL(exclude_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256),
                     (size_t)4 * kin * sizeof(float), (hipStream_t)stream, scores, ids, nq, kin,
                     exclude, ne, kout, out_scores, out_ids);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

static int make_ks(const int32_t *ks_h, int nks, tfrs::KsArg *out) {
  TFRS_CHECK_ARG(ks_h && nks >= 1 && nks <= 16, "need between 1 and 16 values of k");
  for (int i = 0; i < 16; ++i) out->v[i] = (i < nks) ? ks_h[i] : 0;
  return TFRS_OK;
}

extern "C" int tfrs_rank_of_positive(const float *queries, const float *true_candidates,
                                     int64_t nq, int d, const float *topk_scores, int kmax,
                                     const int32_t *ks_h, int nks, float *out_hits,
                                     void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && d >= 1 && kmax >= 1, "rank_of_positive: bad shape");
  tfrs::KsArg ks_d;
  int rc = make_ks(ks_h, nks, &ks_d);
  if (rc != TFRS_OK) return rc;
  if (nq == 0) return TFRS_OK;
  hipLaunchKernelGGL(rank_of_positive_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, queries, true_candidates, nq, d, topk_scores, kmax,
                     ks_d, nks, out_hits);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_id_match_topk(const int32_t *retrieved_ids, const int32_t *true_ids,
                                  int64_t nq, int kmax, const int32_t *ks_h, int nks,
                                  float *out_hits, void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && kmax >= 1, "id_match_topk: bad shape");
  tfrs::KsArg ks_d;
  int rc = make_ks(ks_h, nks, &ks_d);
  if (rc != TFRS_OK) return rc;
  if (nq == 0) return TFRS_OK;
  hipLaunchKernelGGL(id_match_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, retrieved_ids, true_ids, nq, kmax, ks_d, nks, out_hits);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
