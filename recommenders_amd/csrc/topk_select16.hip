// topk_select16.hip -- the two per-query kernels of the fp16-prefiltered BruteForce path
// (topk_api.hip run_f16): lean replacements of the generic sort-merge select kernel for the
// two shapes that path produces.  One wave per query, data in registers, selection by
// MSB-first radix counting (ballot + popcount) instead of bitonic sort-merge in LDS.
//
//   bin_threshold_kernel : K-th largest of the threshold pass's bin maxima -> lower[q]
//   list_topk16_kernel   : survivor list (prefilter scores) -> K-th prefilter score ->
//                          retain everything within 2*eps -> exact f32 re-scoring (same
//                          d-ordered fma chain as the MFMA f32 path) -> sorted exact top-K
//                          = tf.math.top_k of BruteForce.call (layers/factorized_top_k.py:605).
// Queries whose survivor list overflowed, or whose retained set does not fit, are flagged in
// a redo list and answered by the exact recompute kernels (topk_select.hip) afterwards.
#include "common.h"

namespace tfrs {

constexpr int kSel16Waves = 4;
constexpr int kSlots = 16;     // values per lane held in registers (64 * 16 = 1024 per query)
// Timing ablations of list_topk16_kernel (tools/ab_variants.sh; results are WRONG with any of them): 1 = no row
// loads in the exact re-scoring, 2 = no final sort, 4 = stop behind the gather, 16 = twice the LDS per workgroup
// (half the resident waves).
#ifndef TFRS_LIST16_ABLATE
#define TFRS_LIST16_ABLATE 0
#endif
#if TFRS_LIST16_ABLATE
// An ablation build computes WRONG results by design: it needs -DTFRS_ALLOW_ABLATION next to -DTFRS_LIST16_ABLATE=..., and the
// marker symbol below makes recommenders_amd/_lib.py refuse the library unless TFRS_ALLOW_ABLATION=1 is set.
#ifndef TFRS_ALLOW_ABLATION
#error "TFRS_LIST16_ABLATE != 0 is a measurement build with wrong results: add -DTFRS_ALLOW_ABLATION to confirm"
#endif
extern "C" int tfrs_ablation_build_list16(void) { return TFRS_LIST16_ABLATE; }
#endif
constexpr int kRadixBits = 24; // the K-th key is resolved to its top 24 bits (rounded DOWN)

__device__ __forceinline__ uint32_t sel16_mbcnt(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ void sel16_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Top kRadixBits bits of the k-th largest of the wave's 64 * kSlots orderable keys (0 = empty
// slot; returns 0 when fewer than k non-empty keys exist).  The result is <= the true k-th
// largest key, i.e. a lower bound of the k-th largest score.
// Slots >= nslots (wave-uniform) hold no keys and are skipped.
//
// The search is one bit per pass; the bits every non-empty key shares (prefilter scores of one query's survivors:
// the same sign and exponent and a few mantissa bits, typically 9-11 of the 24) are not searched: with at least k
// keys the bit-by-bit search would set exactly those bits of the prefix and leave k alone, with fewer it returns 0.
__device__ __forceinline__ uint32_t radix_kth(const uint32_t (&key)[kSlots], int k, int nslots = kSlots) {
  uint32_t kmax = 0u, kmin = 0xFFFFFFFFu;
  int n = 0;
#pragma unroll
  for (int s = 0; s < kSlots; ++s)
    if (s < nslots) {
      kmax = max(kmax, key[s]);
      kmin = min(kmin, key[s] != 0u ? key[s] : 0xFFFFFFFFu);
      n += (int)__popcll(__ballot(key[s] != 0u));
    }
  if (n < k) return 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
  }
  kmax = __builtin_amdgcn_readfirstlane(kmax);   // (the same in every lane: SGPRs, a uniform loop below)
  kmin = __builtin_amdgcn_readfirstlane(kmin);   // (n >= k >= 1: kmin is a key)
  constexpr uint32_t kLowBits = (1u << (32 - kRadixBits)) - 1u;
  const uint32_t diff = kmax ^ kmin;
  if ((diff & ~kLowBits) == 0u) return kmax & ~kLowBits;   // the keys agree on all searched bits
  const int top = 31 - __builtin_clz(diff);                  // highest bit two keys differ in (>= 32 - kRadixBits)
  uint32_t prefix = top == 31 ? 0u : (kmax & ~((2u << top) - 1u));
#pragma unroll 1
  for (int bit = top; bit >= 32 - kRadixBits; --bit) {
    const uint32_t test = prefix | (1u << bit);
    const uint32_t himask = ~((1u << bit) - 1u);
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
      if (s < nslots) cnt += (int)__popcll(__ballot((key[s] & himask) == test));
    if (cnt >= k) prefix = test; else k -= cnt;
  }
  return prefix;
}

// ---- threshold from the bin maxima ------------------------------------------------------
// binmax[q, n_bins] (prefilter scores of DISTINCT candidates, one per 64-candidate bin).
// Adjacent bins are first merged in groups of `group` (the maximum of a group is still the
// score of one candidate), giving <= 1024 values per query; lower[q] = (k-th largest) - eps,
// raw[q] = that k-th largest itself.  k is K, or the smaller statistical rank of topk_api.hip's
// plan_sample -- then the list kernel checks the bound against raw[q].
__global__ void __launch_bounds__(kSel16Waves * 64) bin_threshold_kernel(
    const float *__restrict__ binmax, int64_t ld, int n_bins, int group, int64_t nq, int k,
    const float *__restrict__ qk, const float *__restrict__ norm_max, float *__restrict__ lower,
    float *__restrict__ raw) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kSel16Waves + (threadIdx.x >> 6);
  if (row >= nq) return;
  const float *src = binmax + row * ld;
  uint32_t key[kSlots];
  if (group == 1) {
    // one bin per lane and slot (up to 1024 bins: configs[1] has 355): the slots' loads are issued together,
    // unconditionally at a clamped index -- guarded by `if (b < n_bins)` each was awaited on its own
    // (vmcnt(0)): 16 serial round trips, 24 us for a kernel that reads 11 MB
    float v[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int b = s * 64 + lane;
      v[s] = src[b < n_bins ? b : n_bins - 1];
    }
#pragma unroll
    for (int s = 0; s < kSlots; ++s) key[s] = (s * 64 + lane < n_bins) ? f32_orderable(v[s]) : 0u;
  } else
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    const int b0 = (s * 64 + lane) * group;
    float m = -__builtin_inff();
    if (group % 4 == 0) {
      for (int g = 0; g < group; g += 4) {
        if (b0 + g + 3 < n_bins) {
          const float4 v = *reinterpret_cast<const float4 *>(src + b0 + g);
          m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        } else {
          for (int e = 0; e < 4; ++e)
            if (b0 + g + e < n_bins) m = fmaxf(m, src[b0 + g + e]);
        }
      }
    } else {
      for (int g = 0; g < group; ++g)
        if (b0 + g < n_bins) m = fmaxf(m, src[b0 + g]);
    }
    key[s] = (b0 < n_bins) ? f32_orderable(m) : 0u;
  }
  // (slots past the last bin hold no keys: at configs[1] 355 bins occupy 6 of the 16 slots)
  const int nslots = min(kSlots, (n_bins + group * 64 - 1) / (group * 64));
  const uint32_t kth = radix_kth(key, k, nslots);
  if (lane == 0) {
    const float eps = qk[row] * norm_max[0] + kF16Tiny;
    // no K-th value (fewer than K bins, or -inf scores): no bound
    const bool have = kth > f32_orderable(-__builtin_inff());
    lower[row] = have ? f32_from_orderable(kth) - eps : -__builtin_inff();
    raw[row] = have ? f32_from_orderable(kth) : -__builtin_inff();
  }
}

int launch_bin_threshold(const float *binmax, int64_t ld, int n_bins, int64_t nq, int k,
                         const float *qk, const float *norm_max, float *lower, float *raw,
                         hipStream_t stream) {
  if (nq <= 0) return TFRS_OK;
  int group = 1;
  if (n_bins > 64 * kSlots) group = ((n_bins + 64 * kSlots - 1) / (64 * kSlots) + 3) / 4 * 4;
  hipLaunchKernelGGL(bin_threshold_kernel, dim3((unsigned)((nq + kSel16Waves - 1) / kSel16Waves)),
                     dim3(kSel16Waves * 64), 0, stream, binmax, ld, n_bins, group, nq, k, qk,
                     norm_max, lower, raw);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// ---- survivor list -> exact top-K ----------------------------------------------------------
// Exact score of one candidate from the packed f32 corpus (d-ordered fma chain; `qs` is the
// zero-padded query in LDS, read as broadcast float4).
__device__ __forceinline__ float packed_score16(const char *packed, int64_t row, int dp,
                                                const float *qs) {
  const float4 *ev = reinterpret_cast<const float4 *>(packed + row * (int64_t)row_bytes(dp));
  const float4 *od = ev + dp / 8;
  const float4 *q4 = reinterpret_cast<const float4 *>(qs);
  float acc = 0.0f;
#pragma unroll 4
  for (int m = 0; m < dp / 8; ++m) {  // (unrolled: 8 independent 16-byte loads in flight)
    const float4 e = ev[m], o = od[m];
    const float4 qa = q4[2 * m], qb = q4[2 * m + 1];
    acc = __builtin_fmaf(e.x, qa.x, acc);
    acc = __builtin_fmaf(o.x, qa.y, acc);
    acc = __builtin_fmaf(e.y, qa.z, acc);
    acc = __builtin_fmaf(o.y, qa.w, acc);
    acc = __builtin_fmaf(e.z, qb.x, acc);
    acc = __builtin_fmaf(o.z, qb.y, acc);
    acc = __builtin_fmaf(e.w, qb.z, acc);
    acc = __builtin_fmaf(o.w, qb.w, acc);
  }
  return acc;
}

// Sorts x[0..P) descending (bitonic network in LDS, one wave).
template <int P>
__device__ __forceinline__ void sort_desc16(uint64_t *x, int lane) {
#pragma unroll 1
  for (int size = 2; size <= P; size <<= 1) {
#pragma unroll 1
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < P / 2; t += 64) {
        const int i = 2 * t - (t & (stride - 1));
        const int jx = i + stride;
        const bool desc = ((i & size) == 0);
        const uint64_t va = x[i], vb = x[jx];
        if ((va < vb) == desc) {
          x[i] = vb;
          x[jx] = va;
        }
      }
      sel16_lds_sync();
    }
  }
}

struct List16Args {
  int64_t nq;
  int k;
  const float *q;
  int d;
  const char *packed;
  const uint2 *buf;      // [nq, cap_l, nseg]
  const uint32_t *cnt;   // [nq, nseg]
  uint32_t cap_l;
  int nseg;
  const float *qk;
  const float *norm_max;
  float *out_scores;     // [nq, k]
  int32_t *out_idx;      // [nq, k] (row + idx_base; -1 marks an empty slot: fewer than k survivors)
  uint32_t *redo;        // [1 + nq]: count, then the queries to answer with the exact recompute path
  int64_t idx_base;      // added to the image-local row numbers
  const uint32_t *ovf_cnt;   // [nq] or NULL: survivors that did not fit their segment ...
  const uint2 *ovf_buf;      // [nq, ovf_cap]
  uint32_t ovf_cap;
  const int32_t *rowmap;     // shuffled index: image row -> original row (NULL: identity)
  // Statistical bound (topk_api.hip plan_sample): raw[q] is the bin maximum the bound came from.
  // Rows outside the list have prefilter scores <= raw - eps - eps_stage, i.e. exact scores
  // <= raw - eps; when K list entries have prefilter scores >= raw, their exact scores are
  // >= raw - eps and the exact top-K is inside the list.  Otherwise the query is redone.  (With
  // the K-th bin maximum as the bound this holds by construction and verify_raw is NULL.)
  const float *verify_raw;
  // [4] or NULL: how many queries were flagged because {0: a segment or the list overflowed,
  // 1: the statistical bound did not hold, 2: the retained set did not fit}; [3]: the longest
  // survivor list of the call if one exceeded 3/4 of the capacity 64 * kSlots (else 0)
  uint32_t *redo_reason;
  // raw != NULL: survivors are re-scored from the row-major blocks of the table (their row numbers are
  // group-local rows) instead of the packed f32 image
  const RawTable *raw;
};

// KP: slots for the retained set (>= K + band); the list itself may hold up to 64 * kSlots.
template <int KP>
__global__ void __launch_bounds__(kSel16Waves * 64) list_topk16_kernel(const List16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * kSel16Waves + wave;
  if (row >= a.nq) return;
  constexpr int kCap = 64 * kSlots;
  uint2 *ent = reinterpret_cast<uint2 *>(smem) + (size_t)wave * kCap;   // compacted list; later
  uint64_t *ex = reinterpret_cast<uint64_t *>(ent);                      // reused for exact keys
  float *qs = reinterpret_cast<float *>(smem + (size_t)kSel16Waves * kCap * sizeof(uint2)) +
              (size_t)wave * TFRS_MAX_DIM;
  const int K = a.k;
  const int dp = padded_dim(a.d);
  for (int i = lane; i < dp; i += 64) qs[i] = (i < a.d) ? a.q[row * a.d + i] : 0.0f;

  // ---- gather the segmented list into LDS (lanes <-> segments, entry-major rows) ----------
  // Small batches split the corpus over many workgroups, i.e. many short segments per query
  // (1024 at B = 1): the counts of 16 x 64 segments are fetched as one batch of independent
  // loads, then the first two entries of all of them as a second batch, so the gather costs
  // two memory latencies instead of two per 64 segments.
  const uint2 *qbuf = a.buf + (row * (int64_t)a.cap_l) * a.nseg;
  int total = 0;       // wave-uniform
  bool bad = false;    // overflowed segment or list longer than kCap
  auto push = [&](bool p, uint2 v) __attribute__((always_inline)) {
    const uint64_t mask = __ballot(p);
    const int pos = total + (int)sel16_mbcnt(mask);
    if (p && pos < kCap) ent[pos] = v;
    total += (int)__popcll(mask);
  };
  constexpr int kCntBatch = 16;
  if (a.nseg <= 64) {
    // Large batches: one segment per lane, ~7 entries each.  Eight entries per round as one batch
    // of independent loads (the first round is issued straight behind the count), so the gather
    // costs 1 + ceil(longest segment / 8) memory latencies.
    // (every load of this gather is UNCONDITIONAL at a clamped segment / entry and selected afterwards: under
    // `if (e < c)` each one was awaited on its own -- the listing had 134 vmcnt(0) waits, at most 6 loads in flight)
    const int lseg = lane < a.nseg ? lane : a.nseg - 1;
    const uint32_t ecap = a.cap_l - 1u;
    uint32_t c = a.cnt[row * a.nseg + lseg];
    if (lane >= a.nseg) c = 0u;
    if (a.ovf_cnt) {   // the excess (up to kOvfPerSeg per segment) went to the overflow list
      bad = c > a.cap_l + kOvfPerSeg;
      c = min(c, a.cap_l);
    } else {
      bad = c > a.cap_l;
    }
    if (__ballot(bad) == 0ull) {
      for (uint32_t e0 = 0; __ballot(e0 < c) != 0ull; e0 += 8) {
        uint2 w[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) w[x] = qbuf[(int64_t)min(e0 + x, ecap) * a.nseg + lseg];
#pragma unroll
        for (int x = 0; x < 8; ++x) push(e0 + x < c, w[x]);
      }
    }
  } else
  for (int sb0 = 0; sb0 < a.nseg && !bad; sb0 += 64 * kCntBatch) {
    uint32_t cnts[kCntBatch];
#pragma unroll
    for (int b = 0; b < kCntBatch; ++b) {
      const int sg = sb0 + b * 64 + lane;
      cnts[b] = a.cnt[row * a.nseg + (sg < a.nseg ? sg : a.nseg - 1)];   // unconditional, clamped (see above)
    }
#pragma unroll
    for (int b = 0; b < kCntBatch; ++b)
      if (sb0 + b * 64 + lane >= a.nseg) cnts[b] = 0u;
#pragma unroll
    for (int b = 0; b < kCntBatch; ++b) {
      if (a.ovf_cnt) {   // the excess (up to kOvfPerSeg per segment) went to the overflow list
        bad = bad || (cnts[b] > a.cap_l + kOvfPerSeg);
        cnts[b] = min(cnts[b], a.cap_l);
      } else {
        bad = bad || (cnts[b] > a.cap_l);
      }
    }
    if (__ballot(bad) != 0ull) {   // (uniform) the query is redone anyway
      bad = true;
      break;
    }
    // first two entries of every segment of the batch: one wave of independent loads
    uint2 v[kCntBatch][2];
#pragma unroll
    for (int b = 0; b < kCntBatch; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int sg = sb0 + b * 64 + lane;
        v[b][e] = qbuf[(int64_t)min((uint32_t)e, a.cap_l - 1u) * a.nseg + (sg < a.nseg ? sg : a.nseg - 1)];
      }
#pragma unroll
    for (int b = 0; b < kCntBatch; ++b) {
      if (sb0 + b * 64 >= a.nseg) break;   // uniform
#pragma unroll
      for (int e = 0; e < 2; ++e) push((uint32_t)e < cnts[b], v[b][e]);
      // longer segments (large batches: few splits, ~7 entries each): 4 entries per round (the loop condition is a
      // ballot: the wave-wide maximum it replaces was six dependent cross-lane shuffles per batch of 64 segments,
      // 96 in a row for the 1024 segments of a single query)
      for (uint32_t e0 = 2; __ballot(e0 < cnts[b]) != 0ull; e0 += 4) {
        uint2 w[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int sg = sb0 + b * 64 + lane;
          w[x] = qbuf[(int64_t)min(e0 + x, a.cap_l - 1u) * a.nseg + (sg < a.nseg ? sg : a.nseg - 1)];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) push(e0 + x < cnts[b], w[x]);
      }
    }
  }
  if (a.ovf_cnt && __ballot(bad) == 0ull) {   // the query's overflow list (usually empty)
    const uint32_t novf = a.ovf_cnt[row];
    if (novf > a.ovf_cap) bad = true;
    else
      for (uint32_t e0 = 0; e0 < novf; e0 += 64) {
        const bool p = e0 + lane < novf;
        push(p, p ? a.ovf_buf[row * (int64_t)a.ovf_cap + e0 + lane] : make_uint2(0u, 0u));
      }
  }
  // (only lists beyond 3/4 of the capacity report: one atomic per QUERY on one word costs 60 us per 8192 queries)
  const bool any_bad = __ballot(bad) != 0ull;
  if (lane == 0 && a.redo_reason && total > 3 * kCap / 4 && !any_bad)
    atomicMax(&a.redo_reason[3], (uint32_t)total);
  if (any_bad || total > kCap) {
    if (lane == 0) a.redo[1 + atomicAdd(a.redo, 1u)] = (uint32_t)row;
    if (lane == 0 && a.redo_reason) atomicAdd(&a.redo_reason[0], 1u);
    return;
  }
  sel16_lds_sync();
  if (TFRS_LIST16_ABLATE & 4) { if (lane == 0) a.out_idx[row * K] = total; return; }

  // ---- list -> registers; K-th largest prefilter score ---------------------------------------
  uint32_t key[kSlots], rowid[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    const int e = s * 64 + lane;
    key[s] = 0u;
    rowid[s] = 0u;
    if (e < total) {
      const uint2 v = ent[e];
      key[s] = f32_orderable(__uint_as_float(v.x));
      rowid[s] = v.y;
    }
  }
  const float eps = a.qk[row] * a.norm_max[0] + kF16Tiny;
  const uint32_t kth = radix_kth(key, K, (total + 63) / 64);   // (a lower bound of the K-th largest; 0: fewer than K)
  if (a.verify_raw && (kth == 0u || kth < f32_orderable(a.verify_raw[row]))) {
    if (lane == 0) a.redo[1 + atomicAdd(a.redo, 1u)] = (uint32_t)row;
    if (lane == 0 && a.redo_reason) atomicAdd(&a.redo_reason[1], 1u);
    return;
  }
  // everything whose prefilter score is within 2*eps of the K-th one may belong to the exact
  // top-K (common.h); kth == 0: fewer than K entries -> keep all of them
  uint32_t lo_key = 0u;
  if (kth != 0u && eps < __builtin_inff()) lo_key = f32_orderable(f32_from_orderable(kth) - 2.0f * eps);
  sel16_lds_sync();  // all lanes have read `ent`: it can be overwritten with exact keys

  // ---- retained entries: compact their row numbers into LDS --------------------------------
  uint32_t *keep = reinterpret_cast<uint32_t *>(ent);
  int m = 0;  // wave-uniform
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    const bool kp = key[s] != 0u && key[s] >= lo_key;
    const uint64_t mask = __ballot(kp);
    const int pos = m + (int)sel16_mbcnt(mask);
    if (kp && pos < kCap) keep[pos] = rowid[s];   // (m <= total <= kCap: the region holds them all)
    m += (int)__popcll(mask);
  }
  if (m > KP) {
    // The retained set is larger than the usual K + band (near-duplicate clusters: hundreds of rows
    // within 2 eps of the K-th score).  Round 2 sent such a query to the corpus-wide exact recompute
    // (0.13 ms each); every retained row is already in LDS, so they are all re-scored here -- up to
    // the list capacity, 64 * kSlots, whose 8 bytes per entry are exactly the room their exact keys
    // need -- and sorted once.  (Reason 2 of redo_reason counts these queries; nothing is redone.)
    if (lane == 0 && a.redo_reason) atomicAdd(&a.redo_reason[2], 1u);
    sel16_lds_sync();
    uint32_t kall[kSlots];
#pragma unroll
    for (int u = 0; u < kSlots; ++u) kall[u] = (u * 64 + lane < m) ? keep[u * 64 + lane] : 0u;
    sel16_lds_sync();  // `keep` is dead: the region becomes `ex`
#pragma unroll 1
    for (int u = 0; u < kSlots; ++u) {
      uint64_t kk = 0ull;
      if (u * 64 + lane < m)
        kk = make_key(a.raw ? raw_score(a.raw, (int64_t)kall[u], a.d, qs) : packed_score16(a.packed, (int64_t)kall[u], dp, qs),
                      (int32_t)((a.rowmap ? (int64_t)a.rowmap[kall[u]] : (int64_t)kall[u]) + a.idx_base));
      ex[u * 64 + lane] = kk;
    }
    sel16_lds_sync();
    sort_desc16<kCap>(ex, lane);
    for (int i = lane; i < K; i += 64) {
      const uint64_t kk = ex[i];
      a.out_scores[row * K + i] = kk ? key_score(kk) : -__builtin_inff();
      a.out_idx[row * K + i] = kk ? key_index(kk) : -1;
    }
    return;
  }
  sel16_lds_sync();
  // ---- exact re-scoring, all lanes busy: entry u*64 + lane -> ex[u*64 + lane] -----------------
  uint32_t kid[KP / 64];
#pragma unroll
  for (int u = 0; u < KP / 64; ++u) kid[u] = (u * 64 + lane < m) ? keep[u * 64 + lane] : 0u;
  sel16_lds_sync();  // `keep` is dead: the region becomes `ex`
#pragma unroll
  for (int u = 0; u < KP / 64; ++u) {
    if (u * 64 < m) {  // wave-uniform
      uint64_t kk = 0ull;
      if (u * 64 + lane < m)
        kk = make_key((TFRS_LIST16_ABLATE & 1) ? (float)kid[u] * 1e-9f :
                      a.raw ? raw_score(a.raw, (int64_t)kid[u], a.d, qs) : packed_score16(a.packed, (int64_t)kid[u], dp, qs),
                      (int32_t)((a.rowmap ? (int64_t)a.rowmap[kid[u]] : (int64_t)kid[u]) + a.idx_base));
      ex[u * 64 + lane] = kk;
    }
  }
  for (int i = m + lane; i < KP; i += 64) ex[i] = 0ull;
  sel16_lds_sync();
  if (!(TFRS_LIST16_ABLATE & 2)) {
    if (KP > 128 && m <= 128) sort_desc16<128>(ex, lane); else sort_desc16<KP>(ex, lane);
  }

  for (int i = lane; i < K; i += 64) {
    const uint64_t kk = ex[i];
    a.out_scores[row * K + i] = kk ? key_score(kk) : -__builtin_inff();
    a.out_idx[row * K + i] = kk ? key_index(kk) : -1;
  }
}

template <int KP>
static int launch_list16_kp(const List16Args &a, hipStream_t stream) {
  const size_t lds = (size_t)kSel16Waves * (64 * kSlots * sizeof(uint2) + TFRS_MAX_DIM * sizeof(float)) *
                     ((TFRS_LIST16_ABLATE & 16) ? 2 : 1);
  const dim3 grid((unsigned)((a.nq + kSel16Waves - 1) / kSel16Waves));
  hipLaunchKernelGGL((list_topk16_kernel<KP>), grid, dim3(kSel16Waves * 64), lds, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_list_topk16(const float *q, int64_t nq, int d, const char *packed, const uint2 *buf,
                       const uint32_t *cnt, uint32_t cap_l, int nseg, int k, const float *qk,
                       const float *norm_max, float *out_scores, int32_t *out_idx, uint32_t *redo,
                       int64_t idx_base, const uint32_t *ovf_cnt, const uint2 *ovf_buf,
                       uint32_t ovf_cap, const int32_t *rowmap, const float *verify_raw,
                       uint32_t *redo_reason, hipStream_t stream, const RawTable *raw) {
  if (nq <= 0) return TFRS_OK;
  List16Args a;
  a.raw = raw;
  a.idx_base = idx_base;
  a.nq = nq;
  a.k = k;
  a.q = q;
  a.d = d;
  a.packed = packed;
  a.buf = buf;
  a.cnt = cnt;
  a.cap_l = cap_l;
  a.nseg = nseg;
  a.qk = qk;
  a.norm_max = norm_max;
  a.out_scores = out_scores;
  a.out_idx = out_idx;
  a.redo = redo;
  a.ovf_cnt = ovf_cnt;
  a.ovf_buf = ovf_buf;
  a.ovf_cap = ovf_cap;
  a.rowmap = rowmap;
  a.verify_raw = verify_raw;
  a.redo_reason = redo_reason;
  const int need = 2 * k;  // K + room for the 2*eps band
  if (need <= 128) return launch_list16_kp<128>(a, stream);
  if (need <= 256) return launch_list16_kp<256>(a, stream);
  if (need <= 512) return launch_list16_kp<512>(a, stream);
  return launch_list16_kp<1024>(a, stream);
}

}  // namespace tfrs
