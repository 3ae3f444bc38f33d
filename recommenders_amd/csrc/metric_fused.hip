// metric_fused.hip -- FactorizedTopK.update_state, score mode, without the top-K.
//
// Reference: metrics/factorized_top_k.py:133-137 (positive score, top-max(ks) retrieval) and
// :181-192 (`in_top_k(targets = 0, predictions = concat([positive, top_k_scores]), k)`).
// tf.math.in_top_k is true iff fewer than k predictions are STRICTLY greater than the target's
// (and the target's prediction is finite), and the retrieved list holds the max(ks) best scores of
// the corpus, so for every k <= max(ks)
//
//     hit_k[b]  <=>  #{ candidates j : score(q_b, cand_j) > pos_b } < k .
//
// The hit tests therefore need one integer per query -- how many corpus rows beat the positive --
// and never the sorted list: `rank_count_kernel` scores a block of candidates on the f32 matrix
// cores (the same d-ordered fma chain as every scoring kernel of this library, so the positive
// ties exactly with its own copy in the corpus) and counts, `hits_update_kernel` turns the counts
// into the weighted means of the metric.  Candidate rows may be read through an id indirection
// (row j = table[ids[j]]): the README quickstart's `movies.batch(128).map(item_model)` with an
// Embedding tower is then ONE launch -- gather, 4096 x 1682 scores and the rank of the positive --
// instead of 14 gathers + 14 streaming top-K updates (README.md:69-80).
//
// Roofline: MFMA f32 (157.3 TFLOP/s): 2 * D flop per (query, candidate); at the quickstart shapes
// (0.88 GFLOP) the launch is latency bound (one L2 round trip + <= 4 tiles of 32 x D/2 MFMAs per
// workgroup).
#include <algorithm>

#include <type_traits>

#include "common.h"

namespace tfrs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RankCountArgs {
  const float *q;        // [nq, d]
  const float *true_c;   // [nq, d]
  int64_t nq;
  int d;
  const float *cand;     // [vocab, d] (ids != nullptr) or [nc, d]
  const void *ids;       // nullptr: candidate j is row j
  int ids_i64;
  int64_t nc;            // candidates scored by this launch
  int64_t vocab;         // rows of `cand` (ids outside [0, vocab) score as zero rows)
  uint32_t *counts;      // [nq], += ; bit 31 = the positive's score is not finite
  int tiles_per_split;   // 32-row tiles per workgroup
  int n_qtiles;          // ceil(nq / 128)
  int mark_nonfinite;    // 1: this launch also flags non-finite positives (first block of a stream)
};

constexpr uint32_t kNonFiniteBit = 0x80000000u;

template <int DP>
struct RankGeom {
  static constexpr int kMaxTiles = DP <= 64 ? 7 : 3;     // <= 61 KB of LDS
  static constexpr int kPitch = DP + 4;            // floats per LDS row: even plane | odd plane | pad
  static constexpr int kLdsFloats = kMaxTiles * 32 * kPitch;
};

struct HitsArgs {
  uint32_t *counts;      // [nq]; re-armed (zeroed) for the next update
  int64_t nq;
  int32_t ks[16];
  int nks;
  const float *weight;   // [nq] or nullptr (all ones)
  float *state;          // [2 * nks]: weighted hit totals, then weight totals (tf.keras.metrics.Mean)
  float *results;        // [nks]: total / count after this update (0 when count == 0)
  float *hits;           // [nks, nq] per-example hit indicators, or nullptr
};

// counts -> weighted hit totals of every k, by ONE workgroup of NT threads (NT / 64 <= 16 waves).
// The reduction order is fixed (lane -> xor tree -> wave partials in order), so the metric state is
// bit-reproducible from run to run.  (Folding this into the last workgroup of the counting launch was
// measured: 30 us against 24 us for the two launches -- every workgroup then pays a vmcnt drain, two
// barriers and a ticket round trip, and the fold runs on 256 threads behind the slowest one.)
template <int NT>
__device__ __forceinline__ void hits_fold(const HitsArgs &a, float (*part_s)[17]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float tot[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) tot[i] = 0.0f;
  float wsum = 0.0f;
  // (the running totals travel with the first batch of counts: read behind the fold they were a round trip of their own)
  const int st = tid < a.nks ? tid : 0;
  const float state_t = a.state[st], state_c = a.state[a.nks + st];
  constexpr int kB = 16;                                  // loads per thread in flight: one memory round
  for (int64_t q0 = 0; q0 < a.nq; q0 += kB * NT) {       // trip covers 4096 queries at NT = 256
    uint32_t cv[kB];
    float wv[kB];
    auto fetch = [&](auto has_w) __attribute__((always_inline)) {   // unconditional loads (clamped), see rank_count_kernel
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        const int64_t q = q0 + u * NT + tid;
        const int64_t qc = q < a.nq ? q : a.nq - 1;
        cv[u] = a.counts[qc];
        wv[u] = decltype(has_w)::value ? a.weight[qc] : 1.0f;
      }
    };
    if (a.weight) fetch(std::true_type{}); else fetch(std::false_type{});
#pragma unroll
    for (int u = 0; u < kB; ++u)
      if (!(q0 + u * NT + tid < a.nq)) {
        cv[u] = 0u;
        wv[u] = 0.0f;
      }
#pragma unroll
    for (int u = 0; u < kB; ++u) {
      const int64_t q = q0 + u * NT + tid;
      if (q < a.nq) {
        // re-arm for the next sweep (write-through: the next launch's atomics act on memory)
        __hip_atomic_store(a.counts + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wsum += wv[u];
        const bool finite = (cv[u] & kNonFiniteBit) == 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < a.nks) {
            const float hit = (finite && cv[u] < (uint32_t)a.ks[i]) ? 1.0f : 0.0f;
            tot[i] += wv[u] * hit;
            if (a.hits) a.hits[(int64_t)i * a.nq + q] = hit;
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < a.nks)                                        // (uniform: unused slots cost nothing)
      for (int off = 32; off > 0; off >>= 1) tot[i] += __shfl_xor(tot[i], off);
  for (int off = 32; off > 0; off >>= 1) wsum += __shfl_xor(wsum, off);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) part_s[wave][i] = tot[i];
    part_s[wave][16] = wsum;
  }
  __syncthreads();
  if (tid < a.nks) {
    float t = 0.0f, w = 0.0f;
    for (int v = 0; v < NT / 64; ++v) {
      t += part_s[v][tid];
      w += part_s[v][16];
    }
    const float total = state_t + t;
    const float count = state_c + w;
    a.state[tid] = total;
    a.state[a.nks + tid] = count;
    a.results[tid] = count > 0.0f ? total / count : 0.0f;
  }
}

__global__ void __launch_bounds__(1024) hits_update_kernel(const HitsArgs a) {
  __shared__ float part_s[16][17];
  hits_fold<1024>(a, part_s);
}

// One workgroup = 4 waves = 128 queries x one split of <= kMaxTiles candidate tiles.  The split's
// rows are staged ONCE into LDS in the packed layout of common.h (even features | odd features |
// pad, odd number of 16-byte slots per row: conflict-free ds_read_b128) and shared by the waves;
// a wave keeps its 32 queries as the MFMA B operand, operands swapped (A = candidates) so that a
// lane's 16 accumulators belong to ONE query and the compare with pos[query] is lane-local.
template <int DP>
__global__ void __launch_bounds__(256, 1) rank_count_kernel(const RankCountArgs a) {
  using G = RankGeom<DP>;
  __shared__ __attribute__((aligned(16))) float tile_s[G::kLdsFloats];
  __shared__ float pos_s[128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const int qt = blockIdx.x % a.n_qtiles;
  const int split = blockIdx.x / a.n_qtiles;
  const int64_t ntiles = (a.nc + 31) / 32;
  const int64_t t0 = (int64_t)split * a.tiles_per_split;
  const int nt = (int)min<int64_t>(a.tiles_per_split, ntiles - t0);
  const int64_t c0 = t0 * 32;
  const int rows = (int)min<int64_t>((int64_t)nt * 32, a.nc - c0);
  const int d = a.d;

  // ---- stage the split's candidate rows (through the id indirection) into LDS ----------------
  // de-interleaved: feature k of row r -> tile_s[r * kPitch + (k & 1) * DP/2 + (k >> 1)].
  // Fully unrolled over the compile-time maximum (predicated): all id loads are issued, then all
  // row loads, then the LDS writes -- two memory round trips per workgroup, not two per iteration
  // (a runtime-bound loop serialised them: 23 us at the quickstart shapes instead of 5).
  if ((d & 3) == 0) {
    constexpr int kIter = (G::kMaxTiles * 32 * (DP / 4) + 255) / 256;   // (7 tiles: not a multiple of 256 at DP <= 16)
    const int cpr = d >> 2;                             // float4 chunks per row
    // Every load is UNCONDITIONAL (dead slots re-read row c0 / table row 0 and are zeroed by selects):
    // a load inside `if (r < rows ...)` makes the number of loads in flight unknown to the compiler,
    // which then waits for each one (vmcnt(0)) before the next -- the "two round trips" above were 37.
    int64_t src[kIter];
    auto load_ids = [&](auto mode) __attribute__((always_inline)) {   // 0: no indirection, 1: int64 ids, 2: int32 ids
#pragma unroll
      for (int i = 0; i < kIter; ++i) {
        const int e = tid + i * 256;
        const int r = e / (DP / 4), c = e - r * (DP / 4);
        const bool live = r < rows && c < cpr;
        const int64_t row = c0 + (live ? r : 0);
        int64_t id = row;
        if (decltype(mode)::value == 1) id = ((const int64_t *)a.ids)[row];
        if (decltype(mode)::value == 2) id = (int64_t)((const int32_t *)a.ids)[row];
        src[i] = (live && id >= 0 && id < a.vocab) ? id : -1;
      }
    };
    if (!a.ids) load_ids(std::integral_constant<int, 0>{});
    else if (a.ids_i64) load_ids(std::integral_constant<int, 1>{});
    else load_ids(std::integral_constant<int, 2>{});
    f32x4 v[kIter];
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
      const int e = tid + i * 256;
      const int c = e % (DP / 4);
      const bool ok = src[i] >= 0;
      const f32x4 t = *reinterpret_cast<const f32x4 *>(a.cand + (ok ? src[i] : 0) * d + 4 * (ok ? c : 0));
      v[i] = ok ? t : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
      const int e = tid + i * 256;
      const int r = e / (DP / 4), c = e - r * (DP / 4);
      if (r < nt * 32) {
        float *row = tile_s + r * G::kPitch;
        *reinterpret_cast<float2 *>(row + 2 * c) = make_float2(v[i][0], v[i][2]);             // features 4c, 4c+2
        *reinterpret_cast<float2 *>(row + DP / 2 + 2 * c) = make_float2(v[i][1], v[i][3]);    // features 4c+1, 4c+3
      }
    }
  } else {
    constexpr int kIter = (G::kMaxTiles * 32 * DP + 255) / 256;
    constexpr int kBatch = kIter < 16 ? kIter : 16;      // 16 loads in flight per thread
#pragma unroll 1
    for (int i0 = 0; i0 < kIter; i0 += kBatch) {
      float v[kBatch];
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        const int e = tid + (i0 + i) * 256;
        const int r = e / DP, k = e - r * DP;
        v[i] = 0.0f;
        if (r < rows && k < d) {
          int64_t src = c0 + r;
          if (a.ids) src = a.ids_i64 ? ((const int64_t *)a.ids)[src] : (int64_t)((const int32_t *)a.ids)[src];
          if (src >= 0 && src < a.vocab) v[i] = a.cand[src * d + k];
        }
      }
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        const int e = tid + (i0 + i) * 256;
        const int r = e / DP, k = e - r * DP;
        if (r < nt * 32) tile_s[r * G::kPitch + (k & 1) * (DP / 2) + (k >> 1)] = v[i];
      }
    }
  }

  // (no barrier between the phases: the compiler hoists every phase's loads to the top, so the
  // workgroup pays ~two memory round trips in total -- ids -> rows, with the query / positive rows
  // alongside -- instead of one or two per phase; the price is registers, which are free at one
  // workgroup per CU)
  // ---- positives ----------------------------------------------------------------------------------
  {
    // the d-ordered fma chain from +0 of the scoring kernels (oracle/c/oracle_core.c), so that the
    // positive ties exactly with its own copy among the candidates.  All loads are issued before
    // the chain and unconditionally (threads 128-255 repeat the work of 0-127: a branch around the
    // loads would cost the batching); padded features add +0 * 0.
    const int64_t r = (int64_t)qt * 128 + (tid & 127);
    const int64_t rc = r < a.nq ? r : a.nq - 1;
    float p = 0.0f;
    {
      const float *qr = a.q + rc * d, *cr = a.true_c + rc * d;
      if ((d & 3) == 0) {
        f32x4 qv[DP / 4], cv[DP / 4];
#pragma unroll
        for (int c = 0; c < DP / 4; ++c) {
          const int off = 4 * c < d ? 4 * c : d - 4;
          qv[c] = *reinterpret_cast<const f32x4 *>(qr + off);
          cv[c] = *reinterpret_cast<const f32x4 *>(cr + off);
        }
#pragma unroll
        for (int c = 0; c < DP / 4; ++c) {
          if (4 * c < d) {
            p = __builtin_fmaf(qv[c][0], cv[c][0], p);
            p = __builtin_fmaf(qv[c][1], cv[c][1], p);
            p = __builtin_fmaf(qv[c][2], cv[c][2], p);
            p = __builtin_fmaf(qv[c][3], cv[c][3], p);
          }
        }
      } else {
        // (odd dims: 16 features per round trip keeps the register footprint of this branch small)
        constexpr int kChunk = DP < 16 ? DP : 16;
#pragma unroll 1
        for (int k0 = 0; k0 < d; k0 += kChunk) {
          float qv[kChunk], cv[kChunk];
#pragma unroll
          for (int k = 0; k < kChunk; ++k) {
            const int kc = k0 + k < d ? k0 + k : d - 1;
            qv[k] = qr[kc];
            cv[k] = cr[kc];
          }
#pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (k0 + k < d) p = __builtin_fmaf(qv[k], cv[k], p);
        }
      }
    }
    if (tid < 128) pos_s[tid] = p;
  }

  // ---- this wave's 32 queries -> MFMA B operand ---------------------------------------
  const int64_t qrow = (int64_t)qt * 128 + wave * 32 + j;
  const bool qvalid = qrow < a.nq;
  float bq[DP / 2];
  if ((d & 3) == 0) {
#pragma unroll
    for (int c = 0; c < DP / 4; ++c) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(a.q + (qvalid ? qrow : a.nq - 1) * d + (4 * c < d ? 4 * c : d - 4));
      if (!(qvalid && 4 * c < d)) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      bq[2 * c] = h ? v[1] : v[0];
      bq[2 * c + 1] = h ? v[3] : v[2];
    }
  } else {
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) {
      const int k = 2 * s + h;
      bq[s] = (qvalid && k < d) ? a.q[qrow * d + k] : 0.0f;
    }
  }
  __syncthreads();
  const float pos = pos_s[wave * 32 + j];

  uint32_t cnt = 0;
  for (int t = 0; t < nt; ++t) {
    const float *ap = tile_s + (t * 32 + j) * G::kPitch + h * (DP / 2);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int m = 0; m < DP / 8; ++m) {
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(ap + 4 * m);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], bq[4 * m + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], bq[4 * m + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], bq[4 * m + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], bq[4 * m + 3], acc, 0, 0, 0);
    }
    // acc[r] = score(query j, candidate c0 + 32 t + (r & 3) + 8 (r >> 2) + 4 h)
    const int base = t * 32 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = base + (r & 3) + 8 * (r >> 2);
      cnt += (off < rows && acc[r] > pos) ? 1u : 0u;
    }
  }
  cnt += __shfl_xor(cnt, 32);
  if (h == 0 && qvalid) {
    if (a.mark_nonfinite && split == 0 && !__builtin_isfinite(pos)) cnt |= kNonFiniteBit;
    if (cnt != 0u) atomicAdd(a.counts + qrow, cnt);
  }
}

template <int DP>
static void launch_rank_count(const RankCountArgs &a, int nsplits, hipStream_t s) {
  hipLaunchKernelGGL((rank_count_kernel<DP>), dim3((unsigned)(nsplits * a.n_qtiles)), dim3(256), 0, s, a);
}

}  // namespace tfrs

extern "C" int tfrs_rank_count_accumulate(const float *queries, const float *true_candidates, int64_t nq,
                                          int d, const float *candidates, const void *cand_ids,
                                          int ids_i64, int64_t nc, int64_t vocab, uint32_t *counts,
                                          int first_block, void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(nq >= 0 && nc >= 0 && d >= 1 && vocab >= 0, "rank_count: bad shape");
  if (d > 128) {
    set_error("rank_count: embedding dim %d is above the fused kernels' 128 (use the top-K path)", d);
    return TFRS_ENOTIMPL;
  }
  if (nq == 0 || nc == 0) return TFRS_OK;
  TFRS_CHECK_ARG(queries && true_candidates && candidates && counts, "rank_count: NULL pointer");
  TFRS_CHECK_ARG(cand_ids || nc <= vocab, "rank_count: more candidates than rows");
  RankCountArgs a;
  a.q = queries; a.true_c = true_candidates; a.nq = nq; a.d = d; a.cand = candidates; a.ids = cand_ids;
  a.ids_i64 = ids_i64; a.nc = nc; a.vocab = vocab; a.counts = counts; a.mark_nonfinite = first_block ? 1 : 0;
  a.n_qtiles = (int)((nq + 127) / 128);
  const int dp = padded_dim(d);
  const int max_tiles = dp <= 64 ? 7 : 3;
  const int64_t ntiles = (nc + 31) / 32;
  // One workgroup per CU where the problem allows it: the launch is bound by the f32 matrix pipe
  // (64 cycles per 32x32x2 step), so what matters is that every SIMD gets the same number of
  // wave-tiles -- 256 workgroups x 4 waves put exactly one wave on each of the 1024 SIMDs (at the
  // quickstart shapes 7 tiles each against 6.6 ideal) -- and that each workgroup pays its load
  // latencies once.  At most max_tiles tiles per workgroup (LDS).
  int64_t want_splits = std::max<int64_t>(1, 256 / a.n_qtiles);
  int64_t tps = std::min<int64_t>(max_tiles, std::max<int64_t>(1, (ntiles + want_splits - 1) / want_splits));
  const int64_t nsplits = (ntiles + tps - 1) / tps;
  TFRS_CHECK_ARG(nsplits * a.n_qtiles <= 0x7FFFFFFF, "rank_count: grid too large (%lld x %d workgroups)",
                 (long long)nsplits, a.n_qtiles);
  a.tiles_per_split = (int)tps;
  hipStream_t s = (hipStream_t)stream;
  switch (dp) {
    case 8: launch_rank_count<8>(a, (int)nsplits, s); break;
    case 16: launch_rank_count<16>(a, (int)nsplits, s); break;
    case 32: launch_rank_count<32>(a, (int)nsplits, s); break;
    case 64: launch_rank_count<64>(a, (int)nsplits, s); break;
    default: launch_rank_count<128>(a, (int)nsplits, s); break;
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_topk_hits_update(uint32_t *counts, int64_t nq, const int32_t *ks_h, int nks,
                                     const float *sample_weight, float *state, float *results,
                                     float *hits, void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(nq >= 0, "topk_hits_update: bad shape");
  TFRS_CHECK_ARG(ks_h && nks >= 1 && nks <= 16, "need between 1 and 16 values of k");
  TFRS_CHECK_ARG(counts && state && results, "topk_hits_update: NULL pointer");
  HitsArgs a;
  a.counts = counts; a.nq = nq; a.nks = nks; a.weight = sample_weight; a.state = state; a.results = results;
  a.hits = hits;
  for (int i = 0; i < 16; ++i) a.ks[i] = i < nks ? ks_h[i] : 0;
  hipLaunchKernelGGL(hits_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
