// topk_select.hip -- exact per-query top-K selection / merge.
//
// Restates tf.math.top_k (values descending, ties -> lower index) for:
//   * the Streaming reduce step  concat(state, new) -> top_k  (layers/factorized_top_k.py:459-472)
//   * the final top_k of BruteForce.call (:605) over the survivors of the scan filter
//   * the multi-GPU merge of per-shard top-K lists.
//
// One wave per query.  Candidates are turned into 64-bit keys (common.h) whose
// descending order is (score desc, index asc).  The wave keeps `best[KP]` sorted in LDS,
// streams the source items 64 at a time, drops everything that cannot beat the current
// K-th key (ballot/mbcnt compaction into `chunk`), and only when `chunk` fills does it
// pay for a bitonic sort + bitonic merge-prune.  LDS ops of one wave execute in order,
// so no workgroup barrier is needed; rows are independent.
//
// Integer/compare work on L2-resident lists: not a roofline-relevant kernel (a few % of
// the scan time); it is kept simple and exact.
#include <algorithm>

#include "common.h"

namespace tfrs {

constexpr int kSelWaves = 4;

__device__ __forceinline__ uint32_t sel_mbcnt(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Sorts x[0..KP) descending (bitonic network, one wave).
template <int KP>
__device__ __forceinline__ void bitonic_sort_desc(uint64_t *x, int lane) {
#pragma unroll 1
  for (int size = 2; size <= KP; size <<= 1) {
#pragma unroll 1
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < KP / 2; t += 64) {
        const int i = 2 * t - (t & (stride - 1));
        const int jx = i + stride;
        const bool desc = ((i & size) == 0);
        const uint64_t va = x[i], vb = x[jx];
        if ((va < vb) == desc) {
          x[i] = vb;
          x[jx] = va;
        }
      }
      wave_lds_sync();
    }
  }
}

// x[0..KP) is bitonic -> sorted descending.
template <int KP>
__device__ __forceinline__ void bitonic_merge_desc(uint64_t *x, int lane) {
#pragma unroll 1
  for (int stride = KP >> 1; stride > 0; stride >>= 1) {
    for (int t = lane; t < KP / 2; t += 64) {
      const int i = 2 * t - (t & (stride - 1));
      const int jx = i + stride;
      const uint64_t va = x[i], vb = x[jx];
      if (va < vb) {
        x[i] = vb;
        x[jx] = va;
      }
    }
    wave_lds_sync();
  }
}

// best (sorted desc) <- top KP of best U chunk[0..fill); chunk is consumed.
template <int KP>
__device__ __forceinline__ void absorb_chunk(uint64_t *best, uint64_t *chunk, int fill,
                                             int lane) {
  for (int i = fill + lane; i < KP; i += 64) chunk[i] = 0ull;
  wave_lds_sync();
  bitonic_sort_desc<KP>(chunk, lane);
  // element-wise max of a descending and an ascending (reversed) sequence is bitonic and
  // holds the KP largest of the union
  for (int i = lane; i < KP; i += 64) {
    const uint64_t o = chunk[KP - 1 - i];
    if (o > best[i]) best[i] = o;
  }
  wave_lds_sync();
  bitonic_merge_desc<KP>(best, lane);
}

// Exact score of one candidate from the packed corpus: the same d-ordered fma chain as the
// MFMA path (re-scoring of prefilter survivors; queries whose scan list overflowed).  `qs`
// is the query, zero-padded to dp floats, in LDS (all lanes read the same address:
// broadcast); the candidate row is fetched with 16-byte loads, 4 even + 4 odd features each.
__device__ __forceinline__ float packed_score(const char *packed, int64_t row, int dp,
                                              const float *qs) {
  const float4 *ev = reinterpret_cast<const float4 *>(packed + row * (int64_t)row_bytes(dp));
  const float4 *od = ev + dp / 8;
  const float4 *q4 = reinterpret_cast<const float4 *>(qs);
  float acc = 0.0f;
#pragma unroll 4
  for (int m = 0; m < dp / 8; ++m) {  // (unrolled: 8 independent 16-byte loads in flight)
    const float4 e = ev[m], o = od[m];
    const float4 qa = q4[2 * m], qb = q4[2 * m + 1];  // features 8m .. 8m+7
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

template <int KP>
__global__ void __launch_bounds__(kSelWaves * 64) select_kernel(const SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * kSelWaves + wave;
  if (row >= a.nq) return;  // whole wave

  uint64_t *best = reinterpret_cast<uint64_t *>(smem) + (size_t)wave * 2 * KP;
  uint64_t *chunk = best + KP;
  float *qs = reinterpret_cast<float *>(smem + (size_t)kSelWaves * 2 * KP * sizeof(uint64_t)) +
              (size_t)wave * TFRS_MAX_DIM;
  const int K = a.k;
  const int source = a.source;
  const int dp = padded_dim(a.d);

  // ---- seed with the prior state (already sorted by construction) -----------------
  if (a.state_len > 0) {   // uniform; both loads of every slot unconditional (clamped) and in flight together
    int32_t si[KP / 64];
    float ss[KP / 64];
#pragma unroll
    for (int u = 0; u < KP / 64; ++u) {
      const int i = u * 64 + lane;
      const int ic = i < a.state_len ? i : 0;
      si[u] = a.state_idx[row * K + ic];
      ss[u] = a.state_scores[row * K + ic];
    }
#pragma unroll
    for (int u = 0; u < KP / 64; ++u) {
      const int i = u * 64 + lane;
      // (a paged search can leave a query with fewer than state_len candidates below its ceiling:
      // empty slots carry row -1 and must not come back as a real (0.0, row 0) entry)
      best[i] = (i < a.state_len && si[u] >= 0) ? make_key(ss[u], si[u]) : 0ull;
    }
  } else {
    for (int i = lane; i < KP; i += 64) best[i] = 0ull;
  }
  if (source == kSrcList || source == kSrcRecompute) {  // the query, zero-padded, for exact scoring
    for (int i = lane; i < dp; i += 64) qs[i] = (i < a.d) ? a.q[row * a.d + i] : 0.0f;
  }
  wave_lds_sync();

  int fill = 0;                           // wave-uniform
  uint64_t kth = best[K - 1];             // 0 while fewer than K entries: everything passes
  const uint64_t ceil = a.ceil_key ? a.ceil_key[row] : ~0ull;   // paged search: candidates lie below it

  // Offers one key per lane: keys that cannot beat the current bound are dropped, the rest
  // are compacted into `chunk`, which is sorted and merged into `best` when it fills.
  auto consume = [&](uint64_t key) {
    const bool p = key > kth && key < ceil;  // key 0 (empty) never passes
    const uint64_t mask = __ballot(p);
    if (mask == 0ull) return;
    if (fill + 64 > KP) {
      absorb_chunk<KP>(best, chunk, fill, lane);
      fill = 0;
      kth = best[K - 1];
    }
    if (p) chunk[fill + sel_mbcnt(mask)] = key;
    fill += (int)__popcll(mask);
  };

  // exact keys of rows [rc_begin, rc_end): the slow but always-correct path
  auto recompute_range = [&]() {
    const int64_t m = a.rc_end - a.rc_begin;
    for (int64_t base = 0; base < m; base += 64) {
      const int64_t e = base + lane;
      uint64_t key = 0ull;
      if (e < m) {
        const int64_t crow = a.rc_begin + e;
        key = make_key(a.raw ? raw_score(a.raw, crow, a.d, qs) : packed_score(a.packed, crow, dp, qs),
                       (int32_t)((a.rowmap ? (int64_t)a.rowmap[crow] : crow) + a.idx_base));
      }
      consume(key);
    }
  };

  if (source == kSrcRecompute) {
    recompute_range();
  } else if (source == kSrcList) {
    // a segment whose count exceeds its capacity lost entries: recompute the query exactly.
    // Counts are fetched kSegBatch x 64 at a time as one batch of independent loads (the raw scan of
    // small query batches leaves up to 512 short segments per query: walked 64 at a time with a
    // dependent count -> entry chain this kernel took 44 us per round at B = 1).
    constexpr int kSegBatch = 8;
    bool ovf = false;
    for (int sb0 = 0; sb0 < a.nseg; sb0 += 64 * kSegBatch) {
      uint32_t c[kSegBatch];
#pragma unroll
      for (int b = 0; b < kSegBatch; ++b) {
        const int sg = sb0 + b * 64 + lane;
        c[b] = a.cnt[row * a.nseg + (sg < a.nseg ? sg : a.nseg - 1)];
      }
#pragma unroll
      for (int b = 0; b < kSegBatch; ++b) ovf = ovf || (sb0 + b * 64 + lane < a.nseg && c[b] > a.cap_l);
    }
    if (__ballot(ovf) != 0ull) {
      recompute_range();
    } else if (a.nseg <= 64) {
      // large query batches: one segment per lane, ~K (rho - 1) / nseg entries each, 4 per round trip
      const uint2 *qbuf = a.buf + (row * (int64_t)a.cap_l) * a.nseg;
      const uint32_t c = (lane < a.nseg) ? a.cnt[row * a.nseg + lane] : 0u;
      uint32_t cmax = c;
      for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, off));
      for (uint32_t e0 = 0; e0 < cmax; e0 += 4) {
        uint2 ent[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ent[u] = make_uint2(0u, 0u);
          if (e0 + u < c) ent[u] = qbuf[(int64_t)(e0 + u) * a.nseg + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uint64_t key = 0ull;
          if (e0 + u < c)
            key = make_key(__uint_as_float(ent[u].x),
                           (int32_t)((a.rowmap ? (int64_t)a.rowmap[ent[u].y] : (int64_t)ent[u].y) + a.idx_base));
          consume(key);
        }
      }
    } else {
      // lanes <-> segments; entry e of 64 consecutive segments is one 512-B row
      const uint2 *qbuf = a.buf + (row * (int64_t)a.cap_l) * a.nseg;
      for (int sb0 = 0; sb0 < a.nseg; sb0 += 64 * kSegBatch) {
        uint32_t c[kSegBatch];
#pragma unroll
        for (int b = 0; b < kSegBatch; ++b) {
          const int sg = sb0 + b * 64 + lane;
          c[b] = a.cnt[row * a.nseg + (sg < a.nseg ? sg : a.nseg - 1)];
        }
        uint32_t cmax = 0u;
#pragma unroll
        for (int b = 0; b < kSegBatch; ++b) {
          if (sb0 + b * 64 + lane >= a.nseg) c[b] = 0u;
          cmax = max(cmax, c[b]);
        }
        for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, off));
        for (uint32_t e0 = 0; e0 < cmax; e0 += 2) {
          uint2 ent[kSegBatch][2];
#pragma unroll
          for (int b = 0; b < kSegBatch; ++b)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              // (clamped, unconditional loads: a load under `if` is awaited on its own)
              const int sg = min(sb0 + b * 64 + lane, a.nseg - 1);
              const uint32_t e = min(e0 + u, a.cap_l - 1);
              ent[b][u] = qbuf[(int64_t)e * a.nseg + sg];
            }
#pragma unroll
          for (int b = 0; b < kSegBatch; ++b) {
            if (sb0 + b * 64 >= a.nseg) break;   // (uniform)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              uint64_t key = 0ull;
              if (e0 + u < c[b])
                key = make_key(__uint_as_float(ent[b][u].x),
                               (int32_t)((a.rowmap ? (int64_t)a.rowmap[ent[b][u].y] : (int64_t)ent[b][u].y) + a.idx_base));
              consume(key);
            }
          }
        }
      }
    }
  } else if (source == kSrcDense) {
    if (!a.rowmap && (reinterpret_cast<uintptr_t>(a.dense + row * a.ld_dense) & 15) == 0) {
      // One wave walks the row, so the walk is bound by how many bytes it keeps in flight: eight 16-byte loads per lane and
      // batch (8 KB per wave), unconditional at a clamped column, the NEXT batch issued before this one is offered.  With
      // one conditional 4-byte load per 64 scores the dense round of a single streamed query -- 65536 scores -- was 1024
      // serial round trips (55 us of a 1.3 ms call); 4-byte loads in batches of 16 still 64 of them (43 us).  The keys
      // carry their column, so the order in which they are offered does not matter.
      constexpr int kU = 8;
      typedef float f4a __attribute__((ext_vector_type(4)));
      const float *drow = a.dense + row * a.ld_dense;
      const int64_t n4 = a.n_dense & ~(int64_t)3;          // whole 16-byte pieces; the last n % 4 scores: below
      f4a v[kU], nx[kU];
      auto fetch = [&](f4a (&dst)[kU], int64_t base) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int64_t e = base + (int64_t)(u * 64 + lane) * 4;
          dst[u] = *reinterpret_cast<const f4a *>(drow + (e + 4 <= n4 ? e : (n4 >= 4 ? n4 - 4 : 0)));
        }
      };
      if (n4 >= 4) {
        fetch(v, 0);
        for (int64_t base = 0; base < n4; base += 256 * kU) {
          fetch(nx, base + 256 * kU);   // (unconditional: behind the row's end every lane re-reads the last piece and drops it)
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const int64_t e = base + (int64_t)(u * 64 + lane) * 4;
            if (base + u * 256 < n4) {   // uniform
#pragma unroll
              for (int c = 0; c < 4; ++c)
                consume(e + 4 <= n4 ? make_key(v[u][c], (int32_t)(a.idx_base + e + c)) : 0ull);
            }
          }
#pragma unroll
          for (int u = 0; u < kU; ++u) v[u] = nx[u];
        }
      }
      {
        const int64_t e = n4 + lane;
        if (n4 < a.n_dense) consume(e < a.n_dense ? make_key(drow[e], (int32_t)(a.idx_base + e)) : 0ull);
      }
    } else {
      for (int64_t base = 0; base < a.n_dense; base += 64) {
        const int64_t e = base + lane;
        consume(e < a.n_dense ? make_key(a.dense[row * a.ld_dense + e],
                                         (int32_t)(a.idx_base + (a.rowmap ? (int64_t)a.rowmap[e] : e)))
                              : 0ull);
      }
    }
  } else if (source == kSrcExpand) {
    // 64 distinct results per pass: every lane fetches its result's score, distinct row and
    // duplicate range (three dependent loads for 64 results at once).  Results that stand for ONE
    // original row -- the common case -- are offered in a single consume; the others are walked
    // one by one, <= K rows each, and skipped outright once K entries with a better score are held
    // (results arrive in descending score order, so on a corpus of popular duplicates the walk
    // ends after the first few).
    for (int j0 = 0; j0 < a.k_in; j0 += 64) {
      const int jj = j0 + lane;
      int32_t u = -1;
      float sc = 0.0f;
      if (jj < a.k_in) {
        u = a.part_idx[row * a.k_in + jj];
        sc = a.part_scores[row * a.k_in + jj];
      }
      int64_t lo = 0;
      int cnt = 0;
      if (u >= 0) {
        lo = a.dup_start[u];
        const int64_t c = a.dup_start[u + 1] - lo;
        cnt = (int)(c < (int64_t)K ? c : (int64_t)K);
      }
      consume(cnt == 1 ? make_key(sc, (int32_t)(a.dup_rows[lo] + a.idx_base)) : 0ull);
      uint64_t multi = __ballot(cnt > 1);
      while (multi != 0ull) {
        const int src = (int)__builtin_ctzll(multi);
        multi &= multi - 1ull;
        const float s1 = __shfl(sc, src);
        const int c1 = __shfl(cnt, src);
        const int64_t l1 = ((int64_t)__shfl((int)(lo >> 32), src) << 32) | (uint32_t)__shfl((int)lo, src);
        if (kth != 0ull && fill == 0 && key_score(kth) > s1) continue;   // K better entries are held
        for (int base = 0; base < c1; base += 64) {
          const int e = base + lane;
          consume(e < c1 ? make_key(s1, (int32_t)(a.dup_rows[l1 + e] + a.idx_base)) : 0ull);
        }
        if (fill > 0) {   // settle, so that the early exit above sees the true K-th key
          absorb_chunk<KP>(best, chunk, fill, lane);
          fill = 0;
          kth = best[K - 1];
        }
      }
    }
  } else {
    // four entries per lane and round trip, indices and scores loaded unconditionally at a clamped entry (the merge
    // of a range's result into the carried state -- 2 x K entries -- was four pairs of dependent conditional loads)
    const int64_t m = (int64_t)a.nparts * a.k_in;
    const int64_t pstride = a.part_stride ? a.part_stride : a.nq * (int64_t)a.k_in;
    constexpr int kU = 4;
    for (int64_t base = 0; base < m; base += 64 * kU) {
      int32_t pi[kU];
      float ps[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t e = base + u * 64 + lane;
        const int64_t ec = e < m ? e : m - 1;
        const int64_t part = ec / a.k_in, jj = ec - part * a.k_in;
        const int64_t off = part * pstride + row * a.k_in + jj;
        pi[u] = a.part_idx[off];
        ps[u] = a.part_scores[off];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t e = base + u * 64 + lane;
        if (base + u * 64 < m)   // uniform
          consume((e < m && pi[u] >= 0) ? make_key(ps[u], pi[u]) : 0ull);
      }
    }
  }
  if (fill > 0) absorb_chunk<KP>(best, chunk, fill, lane);

  // ---- write the new state --------------------------------------------------------------
  if (a.out_scores) {
    for (int i = lane; i < K; i += 64) {
      const uint64_t key = best[i];
      a.out_scores[row * K + i] = key ? key_score(key) : 0.0f;
      a.out_idx[row * K + i] = key ? key_index(key) : -1;   // -1 marks an empty slot
    }
  }
  if (a.out_thr && lane == 0) {
    const uint64_t key = best[K - 1];
    a.out_thr[row] = key ? key_score(key) : -__builtin_inff();
  }
}

// ---- exact recompute of flagged queries, one WORKGROUP per query ---------------------------
// The always-correct fallback of the fp16-prefiltered path (list overflow / retained set too
// large): NW waves split rows [rc_begin, rc_end) between them, each keeps its own top-K with
// the same consume/absorb machinery, wave 0 merges the NW partial lists.  Unflagged queries'
// workgroups exit at once.  (A single wave per query took 16 ms for one flagged query on a
// 1M-row corpus; 16 waves bring the worst case to ~1 ms.)
template <int KP, int NW>
__global__ void __launch_bounds__(NW * 64) recompute_kernel(const SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // flagged queries: a.only_flagged[0] = count, a.only_flagged[1 + s] = query of slot s
  // (NULL: every query, slot = query); workgroups stride over the slots
  const int64_t nslots = a.only_flagged ? (int64_t)a.only_flagged[0] : a.nq;
  // the row range is cut into chunks: gridDim.y of them without a partial-list buffer, else as many
  // as the list budget allows for this many flagged queries (surplus workgroups exit)
  const int nchunks = a.part_keys ? recompute_chunks(a.nq, nslots) : (int)gridDim.y;
  const int chunk_id = blockIdx.y;
  if (chunk_id >= nchunks) return;
  for (int64_t slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
  const int64_t row = a.only_flagged ? (int64_t)a.only_flagged[1 + slot] : slot;
  __syncthreads();  // LDS reuse across slots
  uint64_t *best = reinterpret_cast<uint64_t *>(smem) + (size_t)wave * 2 * KP;
  uint64_t *chunk = best + KP;
  float *qs = reinterpret_cast<float *>(smem + (size_t)NW * 2 * KP * sizeof(uint64_t));
  const int K = a.k;
  const int dp = padded_dim(a.d);
  for (int i = threadIdx.x; i < dp; i += NW * 64) qs[i] = (i < a.d) ? a.q[row * a.d + i] : 0.0f;
  for (int i = lane; i < KP; i += 64) best[i] = 0ull;
  __syncthreads();

  int fill = 0;
  uint64_t kth = 0ull;
  const uint64_t ceil = a.ceil_key ? a.ceil_key[row] : ~0ull;
  auto consume = [&](uint64_t key) {
    const bool p = key > kth && key < ceil;
    const uint64_t mask = __ballot(p);
    if (mask == 0ull) return;
    if (fill + 64 > KP) {
      absorb_chunk<KP>(best, chunk, fill, lane);
      fill = 0;
      kth = best[K - 1];
    }
    if (p) chunk[fill + sel_mbcnt(mask)] = key;
    fill += (int)__popcll(mask);
  };
  const int64_t m = a.rc_end - a.rc_begin;
  const int64_t per = ((m + (int64_t)NW * nchunks - 1) / ((int64_t)NW * nchunks) + 63) / 64 * 64;
  const int64_t lo = ((int64_t)chunk_id * NW + wave) * per;
  const int64_t hi = (lo + per < m) ? lo + per : m;
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t e = base + lane;
    uint64_t key = 0ull;
    if (e < hi) {
      const int64_t crow = a.rc_begin + e;
      key = make_key(a.raw ? raw_score(a.raw, crow, a.d, qs) : packed_score(a.packed, crow, dp, qs),
                       (int32_t)((a.rowmap ? (int64_t)a.rowmap[crow] : crow) + a.idx_base));
    }
    consume(key);
  }
  if (fill > 0) absorb_chunk<KP>(best, chunk, fill, lane);
  fill = 0;
  __syncthreads();
  if (wave != 0) continue;
  // wave 0: fold the other waves' sorted lists (their first K keys) into its own
  kth = best[K - 1];
  for (int w = 1; w < NW; ++w) {
    const uint64_t *other = reinterpret_cast<const uint64_t *>(smem) + (size_t)w * 2 * KP;
    for (int base = 0; base < K; base += 64) {
      const int e = base + lane;
      consume(e < K ? other[e] : 0ull);
    }
  }
  if (fill > 0) absorb_chunk<KP>(best, chunk, fill, lane);
  if (a.part_keys) {  // partial list of this chunk: merged by recompute_merge_kernel
    uint64_t *dst = a.part_keys + ((size_t)slot * nchunks + chunk_id) * K;
    for (int i = lane; i < K; i += 64) dst[i] = best[i];
    continue;
  }
  for (int i = lane; i < K; i += 64) {
    const uint64_t key = best[i];
    a.out_scores[row * K + i] = key ? key_score(key) : 0.0f;
    a.out_idx[row * K + i] = key ? key_index(key) : -1;   // -1 marks an empty slot
  }
  }  // slot loop
}

// One wave per flagged query: top-K of the nchunks sorted partial key lists.
template <int KP>
__global__ void __launch_bounds__(64) recompute_merge_kernel(const SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  uint64_t *best = reinterpret_cast<uint64_t *>(smem);
  uint64_t *chunk = best + KP;
  const int K = a.k;
  const int64_t nslots = a.only_flagged ? (int64_t)a.only_flagged[0] : a.nq;
  const int nchunks = recompute_chunks(a.nq, nslots);
  for (int64_t slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
  const int64_t row = a.only_flagged ? (int64_t)a.only_flagged[1 + slot] : slot;
  wave_lds_sync();
  for (int i = lane; i < KP; i += 64) best[i] = 0ull;
  wave_lds_sync();
  int fill = 0;
  uint64_t kth = 0ull;
  const uint64_t *src = a.part_keys + (size_t)slot * nchunks * K;
  const int total = nchunks * K;
  for (int base0 = 0; base0 < total; base0 += 4 * 64) {
    uint64_t keys[4];      // four independent loads per round trip (up to 25600 keys per query)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = base0 + u * 64 + lane;
      keys[u] = e < total ? src[e] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t key = keys[u];
      const bool p = key > kth;
      const uint64_t mask = __ballot(p);
      if (mask == 0ull) continue;
      if (fill + 64 > KP) {
        absorb_chunk<KP>(best, chunk, fill, lane);
        fill = 0;
        kth = best[K - 1];
      }
      if (p) chunk[fill + sel_mbcnt(mask)] = key;
      fill += (int)__popcll(mask);
    }
  }
  if (fill > 0) absorb_chunk<KP>(best, chunk, fill, lane);
  for (int i = lane; i < K; i += 64) {
    const uint64_t key = best[i];
    a.out_scores[row * K + i] = key ? key_score(key) : 0.0f;
    a.out_idx[row * K + i] = key ? key_index(key) : -1;   // -1 marks an empty slot
  }
  }  // slot loop
}

template <int KP, int NW>
static int launch_recompute_kp(const SelectArgs &a, hipStream_t stream) {
  const size_t lds = (size_t)NW * 2 * KP * sizeof(uint64_t) + TFRS_MAX_DIM * sizeof(float);
  if (lds > 64 * 1024) TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&recompute_kernel<KP, NW>), (int)lds));
  // with a partial-list buffer the rows of a flagged query are spread over up to
  // kRecomputeMaxChunks workgroups (a single CU streams only ~30 GB/s of candidate rows) -- how
  // many is decided on the device from the flagged count (recompute_chunks) -- then merged
  const int gy = a.part_keys ? kRecomputeMaxChunks : 1;
  // small fixed grid striding over the flagged slots (usually none: the launch costs ~3 us)
  const unsigned gx = a.part_keys ? (unsigned)std::min<int64_t>(a.nq, kRecomputeSlotsX)
                                  : (a.only_flagged ? (unsigned)std::min<int64_t>(a.nq, 64) : (unsigned)a.nq);
  hipLaunchKernelGGL((recompute_kernel<KP, NW>), dim3(gx, (unsigned)gy), dim3(NW * 64), lds, stream, a);
  TFRS_LAUNCH_CHECK();
  if (a.part_keys) {
    hipLaunchKernelGGL((recompute_merge_kernel<KP>), dim3((unsigned)std::min<int64_t>(a.nq, 64)), dim3(64),
                       (size_t)2 * KP * sizeof(uint64_t), stream, a);
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}

int launch_recompute(const SelectArgs &a, hipStream_t stream) {
  if (a.nq <= 0) return TFRS_OK;
  TFRS_CHECK_ARG(a.k >= 1 && a.k <= TFRS_MAX_K, "recompute: k=%d outside [1, %d]", a.k, TFRS_MAX_K);
  if (a.k <= 64) return launch_recompute_kp<64, 4>(a, stream);
  if (a.k <= 128) return launch_recompute_kp<128, 4>(a, stream);
  if (a.k <= 256) return launch_recompute_kp<256, 4>(a, stream);
  if (a.k <= 512) return launch_recompute_kp<512, 4>(a, stream);
  return launch_recompute_kp<1024, 4>(a, stream);
}

template <int KP>
static int launch_select_kp(const SelectArgs &a, hipStream_t stream) {
  const size_t lds = (size_t)kSelWaves * (2 * KP * sizeof(uint64_t) + TFRS_MAX_DIM * sizeof(float));
  if (lds > 64 * 1024) TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&select_kernel<KP>), (int)lds));
  const dim3 grid((unsigned)((a.nq + kSelWaves - 1) / kSelWaves));
  hipLaunchKernelGGL((select_kernel<KP>), grid, dim3(kSelWaves * 64), lds, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_select(const SelectArgs &a, hipStream_t stream) {
  if (a.nq <= 0) return TFRS_OK;
  TFRS_CHECK_ARG(a.k >= 1 && a.k <= TFRS_MAX_K, "select: k=%d outside [1, %d]", a.k,
                 TFRS_MAX_K);
  TFRS_CHECK_ARG(a.state_len >= 0 && a.state_len <= a.k, "select: bad state_len %d",
                 a.state_len);
  const int need = a.k;
  if (need <= 64) return launch_select_kp<64>(a, stream);
  if (need <= 128) return launch_select_kp<128>(a, stream);
  if (need <= 256) return launch_select_kp<256>(a, stream);
  if (need <= 512) return launch_select_kp<512>(a, stream);
  return launch_select_kp<1024>(a, stream);
}

}  // namespace tfrs
