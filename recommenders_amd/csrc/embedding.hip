// embedding.hip -- embedding row gather, segment (combiner) reduce, and the
// deterministic scatter-add backward with optional fused row-wise Adagrad.
//
// Replaces tf.gather behind tf.keras.layers.Embedding (README.md:62-66,77-78), the
// combiner lookup of the TPUEmbedding CPU branch
// (layers/embedding/tpu_embedding_layer.py:913-919) and the IndexedSlices gradient +
// optimizer.apply_gradients of tfrs.Model.train_step (models/base.py:77-78).
//
// All three are HBM-bound byte movers.  Layout rule: a table row is read/written as
// 16-byte pieces by D/4 consecutive lanes, so a wave touches 64/(D/4) whole rows per
// instruction (full 128..512-byte lines), with several independent row loads in flight
// per lane to cover the ~2 us random-HBM latency.  Algorithmic bytes per gathered row:
// D*4 read + D*4 written + the id.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace tfrs {

// Denominator of the fused Adagrad update.  adagrad == 1: sqrt(acc + eps), tf.keras.optimizers.Adagrad of TF >= 2.11 /
// tf-keras (`variable.assign_sub(lr * grad / sqrt(accumulator + epsilon))`); adagrad == 2: sqrt(acc) + eps, the
// optimizer_v2 / ResourceApplyAdagradV2 form of TF <= 2.10 (the reference's release script pins TF 2.9.0,
// tools/build_scripts/release.sh:6) -- also torch.optim.Adagrad's, which the tests cross-check it against.
__device__ __forceinline__ float adagrad_denom(float acc, float eps, int adagrad) {
  return adagrad == 2 ? sqrtf(acc) + eps : sqrtf(acc + eps);
}


template <typename IdT>
__device__ __forceinline__ int64_t load_id(const void *ids, int64_t i) {
  return (int64_t) reinterpret_cast<const IdT *>(ids)[i];
}

// ---- dense gather ---------------------------------------------------------------------
// VEC = 4: d % 4 == 0 (16-byte pieces); VEC = 1: any d.
template <typename IdT, int VEC, int UNROLL = 4, bool NT = false>
__global__ void __launch_bounds__(256) gather_kernel(const float *__restrict__ table,
                                                     int64_t vocab, int d,
                                                     const void *__restrict__ ids, int64_t n,
                                                     float *__restrict__ out,
                                                     int32_t *err_flag) {
  const int per_row = d / VEC;
  const int64_t total = n * per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t + (UNROLL - 1) * stride < total; t += UNROLL * stride) {
    int64_t src[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t e = t + u * stride;
      const int64_t row = e / per_row;
      const int c = (int)(e - row * per_row);
      const int64_t id = load_id<IdT>(ids, row);
      ok[u] = (id >= 0 && id < vocab);
      src[u] = (ok[u] ? id : 0) * per_row + c;
    }
    if (VEC == 4) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const float4 *sp = reinterpret_cast<const float4 *>(table) + src[u];
        if (NT) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          const f4 x = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(sp));
          v[u] = make_float4(x[0], x[1], x[2], x[3]);
        } else {
          v[u] = *sp;
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (!ok[u]) {
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (err_flag) *err_flag = 1;
        }
        float4 *dp = reinterpret_cast<float4 *>(out) + (t + u * stride);
        if (NT) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 x = {v[u].x, v[u].y, v[u].z, v[u].w};
          __builtin_nontemporal_store(x, reinterpret_cast<f4 *>(dp));
        } else {
          *dp = v[u];
        }
      }
    } else {
      float v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = table[src[u]];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (!ok[u]) {
          v[u] = 0.f;
          if (err_flag) *err_flag = 1;
        }
        out[t + u * stride] = v[u];
      }
    }
  }
  for (; t < total; t += stride) {
    const int64_t row = t / per_row;
    const int c = (int)(t - row * per_row);
    const int64_t id = load_id<IdT>(ids, row);
    const bool ok = (id >= 0 && id < vocab);
    if (!ok && err_flag) *err_flag = 1;
    if (VEC == 4) {
      float4 v = ok ? reinterpret_cast<const float4 *>(table)[id * per_row + c]
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4 *>(out)[t] = v;
    } else {
      out[t] = ok ? table[id * per_row + c] : 0.f;
    }
  }
}

// ---- segment reduce (sum / mean / sqrtn combiner) -------------------------------------
// One group of `per_row` lanes per output row; the group walks the row's id segment in
// order (so the float32 sum order is the id order, like the oracle).
template <typename IdT, int VEC>
__global__ void __launch_bounds__(256) segment_reduce_kernel(
    const float *__restrict__ table, int64_t vocab, int d, const void *__restrict__ ids,
    const void *__restrict__ row_splits, const float *__restrict__ weights, int64_t nrows,
    int combiner, float *__restrict__ out, int32_t *err_flag) {
  const int per_row = d / VEC;
  const int64_t total = nrows * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / per_row;
    const int c = (int)(t - row * per_row);
    const int64_t lo = load_id<IdT>(row_splits, row), hi = load_id<IdT>(row_splits, row + 1);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    float wsum = 0.f, wsq = 0.f;
    // Eight entries per round: their ids and weights are fetched as one batch of independent loads,
    // then their row pieces as a second batch, then they are added IN ORDER (the float32 sum order
    // stays the id order).  The entry-at-a-time loop it replaces paid two dependent memory round
    // trips per entry (0.61 of HBM at bags of 8).
    // Every load of a round is UNCONDITIONAL (entries past the segment's end re-read its last entry, bad ids
    // read row 0; both are discarded by selects afterwards): a load inside a branch -- `in ? ids[p] : -1`,
    // `if (ok) x = row` -- makes the number of loads in flight unknown to the compiler, which then waits for
    // every load before issuing the next (vmcnt(0)), and the round is eight serial round trips again.
    constexpr int kU = 8;
    auto rounds = [&](auto has_w) __attribute__((always_inline)) {
      for (int64_t p0 = lo; p0 < hi; p0 += kU) {
        int64_t idv[kU];
        float wv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int64_t p = p0 + u < hi ? p0 + u : hi - 1;
          idv[u] = load_id<IdT>(ids, p);
          wv[u] = decltype(has_w)::value ? weights[p] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (!(p0 + u < hi)) {
            idv[u] = -1;
            wv[u] = 0.0f;
          }
        }
        float ev[kU][VEC];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const bool ok = idv[u] >= 0 && idv[u] < vocab;
          const int64_t src = (ok ? idv[u] : 0) * per_row + c;
          if (VEC == 4) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 x = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(table) + src);
#pragma unroll
            for (int v = 0; v < VEC; ++v) ev[u][v] = x[v];
          } else {
            ev[u][0] = table[src];
          }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (p0 + u < hi) {
            wsum += wv[u];
            wsq = __builtin_fmaf(wv[u], wv[u], wsq);
            if (idv[u] < 0 || idv[u] >= vocab) {
              if (err_flag) *err_flag = 1;
            } else {
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[v] += wv[u] * ev[u][v];
            }
          }
        }
      }
    };
    if (weights) rounds(std::true_type{}); else rounds(std::false_type{});
    float scale = 1.0f;
    if (hi > lo) {
      if (combiner == 1) scale = 1.0f / wsum;
      if (combiner == 2) scale = 1.0f / sqrtf(wsq);
    }
    if (VEC == 4) {
      float4 o;
      if (combiner == 0) {
        o = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
      } else {  // divide (not multiply by the reciprocal) to match acc / sum(w)
        const float den = (hi > lo) ? (combiner == 1 ? wsum : sqrtf(wsq)) : 1.0f;
        o = make_float4(acc[0] / den, acc[1 % VEC] / den, acc[2 % VEC] / den, acc[3 % VEC] / den);
      }
      reinterpret_cast<float4 *>(out)[t] = o;
    } else {
      const float den = (hi > lo && combiner != 0) ? (combiner == 1 ? wsum : sqrtf(wsq)) : 1.0f;
      out[t] = acc[0] / den;
    }
    (void)scale;
  }
}

// ---- segment (combiner) reduce backward ---------------------------------------------------
// d out[b] / d e_p = w_p / den_b for every entry p of segment b (den as in the forward), so the
// gradient rows of the nnz looked-up entries are grad_rows[p] = (grad_out[b] / den_b) * w_p:
// the IndexedSlices form (ids, grad_rows) the scatter-add / sparse Adagrad kernels consume.
// Same lane mapping as the forward (D/4 lanes per segment): grad_out read once per segment,
// nnz*D*4 bytes written.
template <typename IdT, int VEC>
__global__ void __launch_bounds__(256) segment_reduce_bwd_kernel(
    const float *__restrict__ grad_out, int d, const void *__restrict__ row_splits,
    const float *__restrict__ weights, int64_t nrows, int combiner,
    float *__restrict__ grad_rows) {
  const int per_row = d / VEC;
  const int64_t total = nrows * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / per_row;
    const int c = (int)(t - row * per_row);
    const int64_t lo = load_id<IdT>(row_splits, row), hi = load_id<IdT>(row_splits, row + 1);
    if (hi <= lo) continue;
    float den = 1.0f;
    if (combiner != 0) {
      float wsum = 0.f, wsq = 0.f;
      for (int64_t p = lo; p < hi; ++p) {
        const float w = weights ? weights[p] : 1.0f;
        wsum += w;
        wsq = __builtin_fmaf(w, w, wsq);
      }
      den = combiner == 1 ? wsum : sqrtf(wsq);
    }
    float g[VEC];
    if (VEC == 4) {
      const float4 gv = reinterpret_cast<const float4 *>(grad_out)[t];
      g[0] = gv.x / den;
      g[1 % VEC] = gv.y / den;
      g[2 % VEC] = gv.z / den;
      g[3 % VEC] = gv.w / den;
    } else {
      g[0] = grad_out[t] / den;
    }
    for (int64_t p = lo; p < hi; ++p) {
      const float w = weights ? weights[p] : 1.0f;
      if (VEC == 4) {
        reinterpret_cast<float4 *>(grad_rows)[p * per_row + c] =
            make_float4(g[0] * w, g[1 % VEC] * w, g[2 % VEC] * w, g[3 % VEC] * w);
      } else {
        grad_rows[p * per_row + c] = g[0] * w;
      }
    }
  }
}

// ---- scatter-add backward (+ fused Adagrad) --------------------------------------------
// Input: ids sorted ascending (stable) with perm[i] = original position.  A lane group
// owns every position that starts a run of equal ids and sums the run's gradient rows in
// occurrence order: no atomics, bit-reproducible, duplicates summed BEFORE the optimizer
// update as Keras does for IndexedSlices.
template <int VEC>
__global__ void __launch_bounds__(256) scatter_add_kernel(
    const float *__restrict__ grad_out, const int64_t *__restrict__ sorted_ids,
    const int64_t *__restrict__ perm, int64_t n, int d, float *__restrict__ dst,
    float *__restrict__ accum, float lr, float eps, int adagrad) {
  const int per_row = d / VEC;
  const int64_t total = n * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / per_row;
    const int c = (int)(t - i * per_row);
    const int64_t id = sorted_ids[i];
    if (id < 0) continue;                            // padding slot of a sequence feature
    if (i > 0 && sorted_ids[i - 1] == id) continue;  // not the start of a run
    float g[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) g[v] = 0.f;
    for (int64_t p = i; p < n && sorted_ids[p] == id; ++p) {
      const int64_t src = perm[p];
      if (VEC == 4) {
        const float4 e = reinterpret_cast<const float4 *>(grad_out)[src * per_row + c];
        g[0] += e.x;
        g[1 % VEC] += e.y;
        g[2 % VEC] += e.z;
        g[3 % VEC] += e.w;
      } else {
        g[0] += grad_out[src * per_row + c];
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int64_t o = (id * per_row + c) * VEC + v;
      if (adagrad) {
        const float a = accum[o] + g[v] * g[v];
        accum[o] = a;
        dst[o] = dst[o] - lr * g[v] / adagrad_denom(a, eps, adagrad);
      } else {
        dst[o] = g[v];
      }
    }
  }
}

// ---- scatter-add for SMALL vocabularies: one wave per table row -------------------------
// No sort: wave v scans the id list 64 at a time (ballot), and for every position that holds
// id v -- in ascending position, i.e. occurrence order, the order of the sorted path and of
// the oracle -- adds that gradient row (lane = feature).  O(vocab * n / 64) wave-steps: used
// when vocab * n is small (the MovieLens-sized tables of BASELINE configs[0]), where it
// replaces a 40 us radix sort + zero-fill per table with one ~5 us kernel.
constexpr int kRowscanChunk = 4096;   // ids per LDS chunk (int32 in LDS: anything outside [0, 2^31) matches no row: -1)
constexpr int kRowscanHitCap = 128;
// NS = number of 64-feature groups of a row (d <= 64 * NS): a template parameter so that every load of the gradient-row
// fetch is unconditional -- a load under `if (lane + 64 s < d)` makes the number of loads in flight unknown to the
// compiler, which then waits for each one (the ISA of the runtime-d version: 176 loads, at most ONE in flight).
template <typename IdT, int NS>
__device__ __forceinline__ void scatter_rowscan_body_ns(
    const float *__restrict__ grad_out, const void *__restrict__ ids, int64_t n, int d,
    int64_t vocab, float *__restrict__ dst, float *__restrict__ accum, float lr, float eps,
    int adagrad, int64_t block, int32_t *s_ids, int *s_hits) {
  // the id list goes through LDS in chunks shared by the workgroup's 4 rows, so a wave's scan
  // is 64 LDS reads per 4096 ids instead of 64 dependent global loads
  constexpr int kChunk = kRowscanChunk;
  constexpr int kHitCap = kRowscanHitCap;
  int *my_hits = s_hits + (threadIdx.x >> 6) * kHitCap;
  const int lane = threadIdx.x & 63;
  const int64_t v = block * 4 + (threadIdx.x >> 6);
  const bool row_ok = v < vocab;
  float g[NS];  // features lane, lane + 64, ...
#pragma unroll
  for (int s = 0; s < NS; ++s) g[s] = 0.0f;
  int fo[NS];   // clamped feature offsets (a lane beyond d re-reads the row's last feature and drops it)
  bool fok[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    fok[s] = lane + 64 * s < d;
    fo[s] = fok[s] ? lane + 64 * s : d - 1;
  }
  bool touched = false;
  for (int64_t c0 = 0; c0 < n; c0 += kChunk) {
    const int m = (int)((n - c0 < kChunk) ? (n - c0) : kChunk);
    __syncthreads();
    {
      // all 16 loads of a thread in flight before the first LDS write, unconditionally (clamped into the chunk): one
      // memory round trip per chunk
      int64_t t[kChunk / 256];
#pragma unroll
      for (int i = 0; i < kChunk / 256; ++i) {
        const int e = threadIdx.x + i * 256;
        t[i] = load_id<IdT>(ids, c0 + (e < m ? e : m - 1));
      }
#pragma unroll
      for (int i = 0; i < kChunk / 256; ++i) {
        const int e = threadIdx.x + i * 256;
        if (e < m) s_ids[e] = (t[i] >= 0 && t[i] <= 0x7FFFFFFFll) ? (int32_t)t[i] : -1;
      }
    }
    __syncthreads();
    if (!row_ok) continue;
    // Two phases per chunk: the scan only records where this row's id occurs (in order); the
    // gradient rows are then fetched eight at a time as independent loads and added in
    // occurrence order -- one memory latency per eight duplicates instead of one per duplicate.
    int nh = 0;   // wave-uniform
    auto flush = [&]() __attribute__((always_inline)) {
      for (int i0 = 0; i0 < nh; i0 += 8) {
        float r[8][NS];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int hp = (i0 + u < nh) ? my_hits[i0 + u] : my_hits[i0];
          const float *row = grad_out + (c0 + hp) * d;
#pragma unroll
          for (int s = 0; s < NS; ++s) r[u][s] = row[fo[s]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i0 + u < nh) {   // uniform
#pragma unroll
            for (int s = 0; s < NS; ++s) g[s] += r[u][s];
          }
      }
      nh = 0;
    };
    for (int base = 0; base < m; base += 64) {
      const int p = base + lane;
      const bool hit = (p < m) && ((int64_t)s_ids[p] == v);
      const uint64_t mask = __ballot(hit);
      if (mask == 0ull) continue;
      touched = true;
      if (nh + 64 > kHitCap) flush();
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      if (hit) my_hits[nh + (int)below] = p;
      nh += (int)__popcll(mask);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    flush();
  }
  if (!row_ok) return;
  if (adagrad) {
    if (touched) {   // wave-uniform
      float av[NS], pv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        av[s] = accum[v * d + fo[s]];
        pv[s] = dst[v * d + fo[s]];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (fok[s]) {
          const float a = av[s] + g[s] * g[s];
          accum[v * d + fo[s]] = a;
          dst[v * d + fo[s]] = pv[s] - lr * g[s] / adagrad_denom(a, eps, adagrad);
        }
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (fok[s]) dst[v * d + fo[s]] = g[s];  // untouched rows get their zeros here: no separate fill
  }
}

// (the LDS arrays are declared ONCE by the kernel: static __shared__ arrays inside the template would be allocated per
// instantiation -- six copies of 18 KB)
template <typename IdT>
__device__ __forceinline__ void scatter_rowscan_body(
    const float *__restrict__ grad_out, const void *__restrict__ ids, int64_t n, int d,
    int64_t vocab, float *__restrict__ dst, float *__restrict__ accum, float lr, float eps,
    int adagrad, int64_t block, int32_t *s_ids, int *s_hits) {
  if (d <= 64) scatter_rowscan_body_ns<IdT, 1>(grad_out, ids, n, d, vocab, dst, accum, lr, eps, adagrad, block, s_ids, s_hits);
  else if (d <= 128) scatter_rowscan_body_ns<IdT, 2>(grad_out, ids, n, d, vocab, dst, accum, lr, eps, adagrad, block, s_ids, s_hits);
  else scatter_rowscan_body_ns<IdT, 4>(grad_out, ids, n, d, vocab, dst, accum, lr, eps, adagrad, block, s_ids, s_hits);
}

template <typename IdT>
__global__ void __launch_bounds__(256) scatter_rowscan_kernel(
    const float *__restrict__ grad_out, const void *__restrict__ ids, int64_t n, int d,
    int64_t vocab, float *__restrict__ dst, float *__restrict__ accum, float lr, float eps,
    int adagrad) {
  __shared__ int32_t s_ids[kRowscanChunk];
  __shared__ int s_hits[4 * kRowscanHitCap];
  scatter_rowscan_body<IdT>(grad_out, ids, n, d, vocab, dst, accum, lr, eps, adagrad, blockIdx.x, s_ids, s_hits);
}

// Several small tables in ONE launch (the user and item tables of a two-tower step): each table's
// scan is a chain of dependent latencies (ids -> hits -> gradient rows -> row update), so two
// launches back to back cost twice the chain while one launch overlaps them.
struct RowscanTables {
  int ntab;
  int first_block[9];          // table t owns blocks [first_block[t], first_block[t + 1])
  const float *grad_out[8];
  const void *ids[8];
  int64_t n[8];
  int d[8];
  int64_t vocab[8];
  float *dst[8];
  float *accum[8];
  int i64[8];
};
__global__ void __launch_bounds__(256) scatter_rowscan_multi_kernel(const RowscanTables t, float lr, float eps,
                                                                    int adagrad) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < t.ntab && (int)blockIdx.x >= t.first_block[i]) k = i;
  const int64_t block = (int)blockIdx.x - t.first_block[k];
  __shared__ int32_t s_ids[kRowscanChunk];
  __shared__ int s_hits[4 * kRowscanHitCap];
  if (t.i64[k])
    scatter_rowscan_body<int64_t>(t.grad_out[k], t.ids[k], t.n[k], t.d[k], t.vocab[k], t.dst[k], t.accum[k], lr,
                                  eps, adagrad, block, s_ids, s_hits);
  else
    scatter_rowscan_body<int32_t>(t.grad_out[k], t.ids[k], t.n[k], t.d[k], t.vocab[k], t.dst[k], t.accum[k], lr,
                                  eps, adagrad, block, s_ids, s_hits);
}

static unsigned grid_for(int64_t total_threads, int64_t cap = 256 * 8) {
  int64_t blocks = (total_threads + 255) / 256;  // default cap: 8 workgroups per CU, grid-stride beyond
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace tfrs

using namespace tfrs;

extern "C" int tfrs_embedding_gather_fwd(const float *table, int64_t vocab, int d,
                                         const void *ids, int ids_are_i64, int64_t n,
                                         float *out, int32_t *err_flag, void *stream) {
  TFRS_CHECK_ARG(vocab >= 1 && d >= 1 && n >= 0, "embedding_gather: bad shape");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(table && ids && out, "embedding_gather: NULL pointer");
  const bool vec = (d % 4 == 0) && (((uintptr_t)table | (uintptr_t)out) % 16 == 0);
  const int64_t total = n * (vec ? d / 4 : d);
  // Rows are read once and the output is written once: non-temporal loads/stores and a
  // grid of up to 64 workgroups per CU measured 6.4 TB/s (read + written) on 26M x 128
  // against 5.4 TB/s for cached accesses with 8 workgroups per CU (tools/exp_gather.py).
  const dim3 grid(grid_for((total + 3) / 4, 256 * 64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (ids_are_i64) {
    if (vec)
      hipLaunchKernelGGL((gather_kernel<int64_t, 4, 4, true>), grid, block, 0, s, table, vocab, d, ids, n, out, err_flag);
    else
      hipLaunchKernelGGL((gather_kernel<int64_t, 1>), grid, block, 0, s, table, vocab, d, ids, n, out, err_flag);
  } else {
    if (vec)
      hipLaunchKernelGGL((gather_kernel<int32_t, 4, 4, true>), grid, block, 0, s, table, vocab, d, ids, n, out, err_flag);
    else
      hipLaunchKernelGGL((gather_kernel<int32_t, 1>), grid, block, 0, s, table, vocab, d, ids, n, out, err_flag);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_embedding_segment_reduce_fwd(const float *table, int64_t vocab, int d,
                                                 const void *ids, const void *row_splits,
                                                 int ids_are_i64, const float *weights,
                                                 int64_t nrows, int combiner, float *out,
                                                 int32_t *err_flag, void *stream) {
  TFRS_CHECK_ARG(vocab >= 1 && d >= 1 && nrows >= 0, "embedding_segment_reduce: bad shape");
  TFRS_CHECK_ARG(combiner >= 0 && combiner <= 2,
                 "embedding_segment_reduce: combiner must be 0 (sum), 1 (mean) or 2 (sqrtn)");
  if (nrows == 0) return TFRS_OK;
  TFRS_CHECK_ARG(table && row_splits && out, "embedding_segment_reduce: NULL pointer");
  const bool vec = (d % 4 == 0) && (((uintptr_t)table | (uintptr_t)out) % 16 == 0);
  const int64_t total = nrows * (vec ? d / 4 : d);
  const dim3 grid(grid_for(total, 256 * 64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (ids_are_i64) {
    if (vec)
      hipLaunchKernelGGL((segment_reduce_kernel<int64_t, 4>), grid, block, 0, s, table, vocab, d, ids, row_splits, weights, nrows, combiner, out, err_flag);
    else
      hipLaunchKernelGGL((segment_reduce_kernel<int64_t, 1>), grid, block, 0, s, table, vocab, d, ids, row_splits, weights, nrows, combiner, out, err_flag);
  } else {
    if (vec)
      hipLaunchKernelGGL((segment_reduce_kernel<int32_t, 4>), grid, block, 0, s, table, vocab, d, ids, row_splits, weights, nrows, combiner, out, err_flag);
    else
      hipLaunchKernelGGL((segment_reduce_kernel<int32_t, 1>), grid, block, 0, s, table, vocab, d, ids, row_splits, weights, nrows, combiner, out, err_flag);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_embedding_segment_reduce_bwd(const float *grad_out, int d,
                                                 const void *row_splits, int splits_are_i64,
                                                 const float *weights, int64_t nrows,
                                                 int combiner, float *grad_rows, void *stream) {
  TFRS_CHECK_ARG(d >= 1 && nrows >= 0, "embedding_segment_reduce_bwd: bad shape");
  TFRS_CHECK_ARG(combiner >= 0 && combiner <= 2,
                 "embedding_segment_reduce_bwd: combiner must be 0 (sum), 1 (mean) or 2 (sqrtn)");
  if (nrows == 0) return TFRS_OK;
  TFRS_CHECK_ARG(grad_out && row_splits && grad_rows, "embedding_segment_reduce_bwd: NULL pointer");
  const bool vec = (d % 4 == 0) && (((uintptr_t)grad_out | (uintptr_t)grad_rows) % 16 == 0);
  const int64_t total = nrows * (vec ? d / 4 : d);
  const dim3 grid(grid_for(total, 256 * 64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (splits_are_i64) {
    if (vec)
      hipLaunchKernelGGL((segment_reduce_bwd_kernel<int64_t, 4>), grid, block, 0, s, grad_out, d, row_splits, weights, nrows, combiner, grad_rows);
    else
      hipLaunchKernelGGL((segment_reduce_bwd_kernel<int64_t, 1>), grid, block, 0, s, grad_out, d, row_splits, weights, nrows, combiner, grad_rows);
  } else {
    if (vec)
      hipLaunchKernelGGL((segment_reduce_bwd_kernel<int32_t, 4>), grid, block, 0, s, grad_out, d, row_splits, weights, nrows, combiner, grad_rows);
    else
      hipLaunchKernelGGL((segment_reduce_bwd_kernel<int32_t, 1>), grid, block, 0, s, grad_out, d, row_splits, weights, nrows, combiner, grad_rows);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_embedding_scatter_add_bwd(const float *grad_out, const int64_t *sorted_ids,
                                              const int64_t *perm, int64_t n, int d,
                                              float *grad_table_or_table, float *accum,
                                              float lr, float eps, int adagrad, void *stream) {
  TFRS_CHECK_ARG(n >= 0 && d >= 1, "embedding_scatter_add: bad shape");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(grad_out && sorted_ids && perm && grad_table_or_table,
                 "embedding_scatter_add: NULL pointer");
  TFRS_CHECK_ARG(!adagrad || accum, "embedding_scatter_add: Adagrad needs an accumulator");
  const bool vec = (d % 4 == 0) && (((uintptr_t)grad_out) % 16 == 0);
  const int64_t total = n * (vec ? d / 4 : d);
  const dim3 grid(grid_for(total)), block(256);
  if (vec)
    hipLaunchKernelGGL((scatter_add_kernel<4>), grid, block, 0, (hipStream_t)stream, grad_out, sorted_ids, perm, n, d, grad_table_or_table, accum, lr, eps, adagrad);
  else
    hipLaunchKernelGGL((scatter_add_kernel<1>), grid, block, 0, (hipStream_t)stream, grad_out, sorted_ids, perm, n, d, grad_table_or_table, accum, lr, eps, adagrad);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// ------------------------------------------------------------------------------------------------
// Own stable LSD radix sort of (id, position) pairs for the large-vocabulary scatter-add
// (replaces torch.sort / rocPRIM on the backward path of models/base.py:77-78).
//   keys   uint32 ids; ids outside [0, vocab) -- the padding slots of sequence features and
//          anything invalid -- become 0xFFFFFFFF: they sort last and the scatter skips them, so
//          an out-of-range id can never write outside the table (the gather reads it as zeros)
//   passes 8, 9 or 10 bits each (more than 8 when that saves a pass: 26 M rows need 26 bits = 3 x 9 instead
//          of 4 x 8, 100 M rows 28 = 3 x 10; a pass is four launch-latency-bound kernels, 47 us at 1.7 M keys); every pass = tile histograms ->
//          exclusive scan (digit-major) -> stable scatter
//   tile   4096 keys per 256-thread workgroup; wave w owns keys [1024 w, 1024 w + 1024) of the
//          tile and walks them 64 at a time IN ORDER: equal digits of one step are ranked with
//          8 .. 10 ballots (lanes with the same digit form a mask; rank = popcount below the lane), the
//          wave's running per-digit offsets live in LDS.  Stable by construction.
// Integer work, HBM-trivial (16 bytes per key and pass); launch-latency bound below ~1M keys.
// ------------------------------------------------------------------------------------------------
namespace tfrs {
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4 *p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(const float4 &x, float4 *p) {
  nt_f4 v = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(v, reinterpret_cast<nt_f4 *>(p));
}
constexpr int kSortTile = 4096;

template <int BITS>
__device__ __forceinline__ uint64_t same_digit_mask(uint32_t digit) {
  uint64_t m = ~0ull;
#pragma unroll
  for (int bit = 0; bit < BITS; ++bit) {
    const uint64_t bal = __ballot((digit >> bit) & 1u);
    m &= ((digit >> bit) & 1u) ? bal : ~bal;
  }
  return m;
}

// keys_out[i] = id or 0xFFFFFFFF, vals_out[i] = i  (pass 0 reads these)
__global__ void __launch_bounds__(256) sort_init_kernel(const void *__restrict__ ids, int i64, int64_t n,
                                                        int64_t vocab, uint32_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t id = i64 ? static_cast<const int64_t *>(ids)[i] : (int64_t)static_cast<const int32_t *>(ids)[i];
  keys[i] = (id >= 0 && id < vocab) ? (uint32_t)id : 0xFFFFFFFFu;
  vals[i] = (uint32_t)i;
}

// hist[tile][wave][digit]
template <int BITS>
__global__ void __launch_bounds__(256) sort_hist_kernel(const uint32_t *__restrict__ keys, int64_t n,
                                                        int shift, uint32_t *__restrict__ hist) {
  constexpr int NB = 1 << BITS;
  __shared__ uint32_t h[4][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 4 * NB; e += 256) (&h[0][0])[e] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile + wave * 1024;
  uint32_t kv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {   // unconditional, at a clamped index: 16 loads in flight (guarded, each was awaited)
    const int64_t i = base + r * 64 + lane;
    kv[r] = keys[i < n ? i : n - 1];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (base + r * 64 + lane < n) atomicAdd(&h[wave][(kv[r] >> shift) & (uint32_t)(NB - 1)], 1u);
  __syncthreads();
  for (int e = tid; e < 4 * NB; e += 256) hist[(int64_t)blockIdx.x * (4 * NB) + e] = (&h[0][0])[e];
}

// Exclusive scan of the segment histograms in digit-major order, two levels (thread = digit):
//   scan1: workgroup c scans its chunk of 64 segments -> offs[segment][digit] (chunk-local) and
//          chunk_tot[c][digit];
//   scan2: one workgroup turns chunk_tot into chunk_base[c][digit] = keys with a smaller digit
//          anywhere + keys with the same digit in earlier chunks.
// The scatter kernel adds the two.
constexpr int kScanChunk = 64;
template <int BITS>
__global__ void __launch_bounds__(1 << BITS) sort_scan1_kernel(const uint32_t *__restrict__ hist, int64_t nseg,
                                                               uint32_t *__restrict__ offs,
                                                               uint32_t *__restrict__ chunk_tot) {
  constexpr int NB = 1 << BITS;
  const int dgt = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * kScanChunk;
  const int64_t s1 = s0 + kScanChunk < nseg ? s0 + kScanChunk : nseg;
  uint32_t run = 0;
#pragma unroll 8
  for (int64_t sgm = s0; sgm < s1; ++sgm) {
    const uint32_t c = hist[sgm * NB + dgt];
    offs[sgm * NB + dgt] = run;
    run += c;
  }
  chunk_tot[(int64_t)blockIdx.x * NB + dgt] = run;
}
template <int BITS>
__global__ void __launch_bounds__(1 << BITS) sort_scan2_kernel(uint32_t *__restrict__ chunk_tot, int64_t nchunk) {
  constexpr int NB = 1 << BITS;
  __shared__ uint32_t tot[NB];
  const int dgt = threadIdx.x;
  uint32_t run = 0;
#pragma unroll 8
  for (int64_t c = 0; c < nchunk; ++c) {
    const uint32_t v = chunk_tot[c * NB + dgt];
    chunk_tot[c * NB + dgt] = run;
    run += v;
  }
  tot[dgt] = run;
  __syncthreads();
  uint32_t before = 0;
  for (int e = 0; e < dgt; ++e) before += tot[e];
#pragma unroll 8
  for (int64_t c = 0; c < nchunk; ++c) chunk_tot[c * NB + dgt] += before;
}

template <int BITS>
__global__ void __launch_bounds__(256) sort_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                           const uint32_t *__restrict__ vals_in, int64_t n,
                                                           int shift, const uint32_t *__restrict__ offs,
                                                           const uint32_t *__restrict__ chunk_base,
                                                           uint32_t *__restrict__ keys_out,
                                                           uint32_t *__restrict__ vals_out) {
  constexpr int NB = 1 << BITS;
  __shared__ uint32_t run[4][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int64_t sgm = (int64_t)blockIdx.x * 4 + wave;
    for (int e = lane; e < NB; e += 64)
      run[wave][e] = offs[sgm * NB + e] + chunk_base[(sgm / kScanChunk) * NB + e];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * kSortTile + wave * 1024;
  const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // the wave's 1024 keys and values: 32 unconditional loads (clamped index) issued up front -- loaded round by
  // round behind `i < n ? ... : 0`, each round paid its own memory round trip between two LDS synchronisations
  // (28 us per pass for 1.7 M keys)
  uint32_t kk[16], vv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t i = base + r * 64 + lane;
    kk[r] = keys_in[i < n ? i : n - 1];
    vv[r] = vals_in[i < n ? i : n - 1];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool ok = i < n;
    const uint32_t key = ok ? kk[r] : 0u;
    const uint32_t val = ok ? vv[r] : 0u;
    // inactive tail lanes get a digit of their own class so that they never rank among real keys
    const uint32_t digit = (key >> shift) & (uint32_t)(NB - 1);
    const uint64_t act = __ballot(ok);
    const uint64_t same = same_digit_mask<BITS>(digit) & act;
    if (ok) {
      const uint32_t rank = (uint32_t)__builtin_popcountll(same & below);
      const uint32_t dst = run[wave][digit] + rank;
      keys_out[dst] = key;
      vals_out[dst] = val;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ok && (same & below) == 0ull) run[wave][digit] += (uint32_t)__builtin_popcountll(same);   // leader
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// scatter-add over uint32 sorted keys / positions (see scatter_add_kernel); keys >= vocab are the
// invalid / padding ids and sort last
// Long runs of one id (a hot item of a Zipf-distributed feature, a padding id) are cut at
// multiples of `piece` positions (the first cut at least `piece` positions into the run): scatter_add_pieces_kernel sums every piece that CONTINUES a run
// across such a boundary into part[boundary / piece] (in parallel), and the run's first thread
// below adds its own first piece and then those partial sums.  Without this the whole run is one
// thread's serial loop: 1.5 M gradients for one row took 580 ms, 1500 per row 1.8 ms instead of 0.4.
template <int VEC>
__global__ void __launch_bounds__(256) scatter_add_pieces_kernel(
    const float *__restrict__ grad_out, const uint32_t *__restrict__ sorted_ids,
    const uint32_t *__restrict__ perm, int64_t n, int d, uint32_t vocab, int piece,
    float *__restrict__ part) {
  const int per_row = d / VEC;
  const int64_t nslots = (n + piece - 1) / piece;
  const int64_t total = nslots * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / per_row;
    const int c = (int)(t - b * per_row);
    const int64_t j = b * piece;
    if (b == 0 || j >= n) continue;
    const uint32_t id = sorted_ids[j];
    // a piece starts here only for a run that began at least `piece` positions earlier (ids are
    // sorted: equal ends mean an equal stretch), so runs shorter than `piece` are still summed by
    // ONE thread in position order -- bit-identical to the sequential oracle
    if (id >= vocab || sorted_ids[j - piece] != id) continue;
    float g[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) g[v] = 0.f;
    const int64_t end = (j + piece < n) ? j + piece : n;
    for (int64_t p = j; p < end && sorted_ids[p] == id; ++p) {
      const int64_t src = perm[p];
      if (VEC == 4) {
        const float4 e = reinterpret_cast<const float4 *>(grad_out)[src * per_row + c];
        g[0] += e.x;
        g[1 % VEC] += e.y;
        g[2 % VEC] += e.z;
        g[3 % VEC] += e.w;
      } else {
        g[0] += grad_out[src * per_row + c];
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[((b - 1) * per_row + c) * VEC + v] = g[v];   // slot b - 1: boundary 0 continues nothing
  }
}

// NT: the gradient rows, the table / accumulator rows and their stores carry the non-temporal hint -- every one of them
// is touched once per launch, and a table beyond the last-level cache (the launcher asks for > 1 GiB) gains nothing from
// keeping them: 26 M x 128, 1.7 M ids, same box, alternating: 0.960 -> 0.933 ms (the loads alone 0.943, the stores alone +-0).
template <int VEC, bool NT = false>
__global__ void __launch_bounds__(256) scatter_add_u32_kernel(
    const float *__restrict__ grad_out, const uint32_t *__restrict__ sorted_ids,
    const uint32_t *__restrict__ perm, int64_t n, int d, uint32_t vocab, float *__restrict__ dst,
    float *__restrict__ accum, float lr, float eps, int adagrad, int piece,
    const float *__restrict__ part) {
  const int per_row = d / VEC;
  const int64_t total = n * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / per_row;
    const int c = (int)(t - i * per_row);
    // everything the common case (a run of one) needs goes out in TWO rounds of independent loads:
    // {id, its neighbours, the position} then {gradient piece, weights, accumulator}
    const uint32_t id = sorted_ids[i];
    const uint32_t id_prev = sorted_ids[i > 0 ? i - 1 : 0];          // (clamped: the loads are unconditional)
    const uint32_t id_next = sorted_ids[i + 1 < n ? i + 1 : n - 1];
    const int64_t src0 = perm[i];
    if (id >= vocab) continue;                       // invalid / padding id
    if (i > 0 && id_prev == id) continue;            // not the start of a run
    float g[VEC];
    float4 a_pre = make_float4(0.f, 0.f, 0.f, 0.f), w_pre = a_pre;
    if (VEC == 4 && adagrad) {
      const int64_t o4 = (int64_t)id * per_row + c;
      if (NT) {
        a_pre = nt_load4(reinterpret_cast<const float4 *>(accum) + o4);
        w_pre = nt_load4(reinterpret_cast<const float4 *>(dst) + o4);
      } else {
        a_pre = reinterpret_cast<const float4 *>(accum)[o4];
        w_pre = reinterpret_cast<const float4 *>(dst)[o4];
      }
    }
    if (VEC == 4) {
      const float4 e = NT ? nt_load4(reinterpret_cast<const float4 *>(grad_out) + src0 * per_row + c)
                                             : reinterpret_cast<const float4 *>(grad_out)[src0 * per_row + c];
      g[0] = 0.f + e.x;          // (0 + x, not x: the sum of a run starts from +0 like the oracle's, -0 gradients included)
      g[1 % VEC] = 0.f + e.y;
      g[2 % VEC] = 0.f + e.z;
      g[3 % VEC] = 0.f + e.w;
    } else {
      g[0] = 0.f + grad_out[src0 * per_row + c];
    }
    // the run's first piece: up to the first multiple of `piece` that is >= i + piece ...
    int64_t p = i + 1;
    const int64_t first_end = ((i + piece - 1) / piece + 1) * (int64_t)piece;
    if (p < n && id_next == id) {
      for (; p < n && p < first_end && sorted_ids[p] == id; ++p) {
        const int64_t src = perm[p];
        if (VEC == 4) {
          const float4 e = reinterpret_cast<const float4 *>(grad_out)[src * per_row + c];
          g[0] += e.x;
          g[1 % VEC] += e.y;
          g[2 % VEC] += e.z;
          g[3 % VEC] += e.w;
        } else {
          g[0] += grad_out[src * per_row + c];
        }
      }
    }
    // ... then the partial sums of the pieces that continue it (scatter_add_pieces_kernel)
    if (p == first_end) {
      for (int64_t b = first_end / piece; b * piece < n && sorted_ids[b * piece] == id; ++b) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) g[v] += part[((b - 1) * per_row + c) * VEC + v];
      }
    }
    if (VEC == 4) {
      const int64_t o4 = (int64_t)id * per_row + c;
      float4 *d4 = reinterpret_cast<float4 *>(dst) + o4;
      if (adagrad) {
        float4 *a4 = reinterpret_cast<float4 *>(accum) + o4;
        float4 a = a_pre, w = w_pre;
        a.x += g[0] * g[0]; a.y += g[1 % VEC] * g[1 % VEC]; a.z += g[2 % VEC] * g[2 % VEC]; a.w += g[3 % VEC] * g[3 % VEC];
        w.x -= lr * g[0] / adagrad_denom(a.x, eps, adagrad); w.y -= lr * g[1 % VEC] / adagrad_denom(a.y, eps, adagrad);
        w.z -= lr * g[2 % VEC] / adagrad_denom(a.z, eps, adagrad); w.w -= lr * g[3 % VEC] / adagrad_denom(a.w, eps, adagrad);
        if (NT) {
          nt_store4(a, a4);
          nt_store4(w, d4);
        } else {
          *a4 = a;
          *d4 = w;
        }
      } else {
        *d4 = make_float4(g[0], g[1 % VEC], g[2 % VEC], g[3 % VEC]);
      }
    } else {
      const int64_t o = (int64_t)id * per_row + c;
      if (adagrad) {
        const float a = accum[o] + g[0] * g[0];
        accum[o] = a;
        dst[o] = dst[o] - lr * g[0] / adagrad_denom(a, eps, adagrad);
      } else {
        dst[o] = g[0];
      }
    }
  }
}

static inline size_t sort_al(size_t x) { return (x + 255) / 256 * 256; }
}  // namespace tfrs

extern "C" size_t tfrs_embedding_scatter_add_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  const size_t tiles = (size_t)((n + tfrs::kSortTile - 1) / tfrs::kSortTile);
  // (histograms and offsets of 4 waves x 1024 digits per tile, chunk totals of 1024 digits: the 10-bit passes)
  return 4 * tfrs::sort_al((size_t)n * 4) + 2 * tfrs::sort_al(tiles * 4096 * 4) +
         tfrs::sort_al((tiles * 4 / tfrs::kScanChunk + 1) * 1024 * 4);
}

// Backward of gather from UNSORTED ids: own radix sort + the segmented scatter-add / fused
// Adagrad above.  ids outside [0, vocab) are ignored (they read as zero rows in the forward).
extern "C" int tfrs_embedding_scatter_add_unsorted(const float *grad_out, const void *ids,
                                                   int ids_are_i64, int64_t n, int d, int64_t vocab,
                                                   float *grad_table_or_table, float *accum, float lr,
                                                   float eps, int adagrad, void *workspace,
                                                   size_t workspace_bytes, void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(n >= 0 && d >= 1 && vocab >= 1, "embedding_scatter_add_unsorted: bad shape");
  TFRS_CHECK_ARG(vocab < 0xFFFFFFFFll && n < 0xFFFFFFFFll,
                 "embedding_scatter_add_unsorted: vocab / n must fit 32 bits");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(grad_out && ids && grad_table_or_table && workspace,
                 "embedding_scatter_add_unsorted: NULL pointer");
  TFRS_CHECK_ARG(!adagrad || accum, "embedding_scatter_add_unsorted: Adagrad needs an accumulator");
  if (workspace_bytes < tfrs_embedding_scatter_add_workspace_bytes(n)) {
    set_error("embedding_scatter_add_unsorted: workspace too small");
    return TFRS_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  char *w = static_cast<char *>(workspace);
  const size_t kb = sort_al((size_t)n * 4);
  uint32_t *keys[2] = {reinterpret_cast<uint32_t *>(w), reinterpret_cast<uint32_t *>(w + kb)};
  uint32_t *vals[2] = {reinterpret_cast<uint32_t *>(w + 2 * kb), reinterpret_cast<uint32_t *>(w + 3 * kb)};
  const int64_t tiles = (n + kSortTile - 1) / kSortTile;
  uint32_t *hist = reinterpret_cast<uint32_t *>(w + 4 * kb);
  uint32_t *offs = reinterpret_cast<uint32_t *>(w + 4 * kb + sort_al((size_t)tiles * 4096 * 4));
  uint32_t *chunk = reinterpret_cast<uint32_t *>(w + 4 * kb + 2 * sort_al((size_t)tiles * 4096 * 4));
  const int64_t nseg = tiles * 4, nchunk = (nseg + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(sort_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, ids_are_i64, n,
                     vocab, keys[0], vals[0]);
  // digits that can differ: the bits of vocab (0xFFFFFFFF of invalid ids needs the top pass too,
  // which the last valid pass provides as long as it covers a bit above vocab - 1)
  int bits = 1;
  while (bits < 32 && (1ll << bits) <= vocab) ++bits;   // 2^bits > vocab: invalid keys have bit `bits`.. set
  int passes = (bits + 1 + 7) / 8;
  if (passes > 4) passes = 4;
  // 9 or 10 bits per pass where that saves a whole pass (26 significant bits: 3 x 9; 28 .. 30: 3 x 10)
  int digit_bits = 8;
  for (int b = 9; b <= 10; ++b)
    if ((bits + 1 + b - 1) / b < passes) {
      passes = (bits + 1 + b - 1) / b;
      digit_bits = b;
    }
  int cur = 0;
  auto one_pass = [&](auto bc, int p) {
    constexpr int B = decltype(bc)::value;
    const int shift = B * p;
    hipLaunchKernelGGL(sort_hist_kernel<B>, dim3((unsigned)tiles), dim3(256), 0, s, keys[cur], n, shift, hist);
    hipLaunchKernelGGL(sort_scan1_kernel<B>, dim3((unsigned)nchunk), dim3(1 << B), 0, s, hist, nseg, offs, chunk);
    hipLaunchKernelGGL(sort_scan2_kernel<B>, dim3(1), dim3(1 << B), 0, s, chunk, nchunk);
    hipLaunchKernelGGL(sort_scatter_kernel<B>, dim3((unsigned)tiles), dim3(256), 0, s, keys[cur], vals[cur], n,
                       shift, offs, chunk, keys[cur ^ 1], vals[cur ^ 1]);
    cur ^= 1;
  };
  for (int p = 0; p < passes; ++p) {
    if (digit_bits == 10) one_pass(std::integral_constant<int, 10>{}, p);
    else if (digit_bits == 9) one_pass(std::integral_constant<int, 9>{}, p);
    else one_pass(std::integral_constant<int, 8>{}, p);
  }
  TFRS_LAUNCH_CHECK();
  const bool vec = (d % 4 == 0) && (((uintptr_t)grad_out) % 16 == 0) &&
                   (((uintptr_t)grad_table_or_table) % 16 == 0) && (!accum || ((uintptr_t)accum) % 16 == 0);
  const int64_t total = n * (vec ? d / 4 : d);
  const dim3 grid(grid_for(total, 256 * 64)), block(256);
  // pieces of `piece` >= d positions: slot b - 1 (b >= 1, b * piece < n) ends at b * d <= b * piece < n
  // floats, i.e. inside the n floats of the sort's free ping-pong key buffer
  int piece = 32;
  while (piece < d) piece *= 2;
  float *part = reinterpret_cast<float *>(keys[cur ^ 1]);
  const int64_t ptotal = ((n + piece - 1) / piece) * (vec ? d / 4 : d);
  const dim3 pgrid(grid_for(ptotal, 256 * 64));
  if (vec) {
    hipLaunchKernelGGL((scatter_add_pieces_kernel<4>), pgrid, block, 0, s, grad_out, keys[cur], vals[cur], n, d,
                       (uint32_t)vocab, piece, part);
    const char *nte = option("TFRS_SCATTER_NT");
    if (vocab * (int64_t)d * 4 > (1ll << 30) && !(nte && nte[0] == '0'))
      hipLaunchKernelGGL((scatter_add_u32_kernel<4, true>), grid, block, 0, s, grad_out, keys[cur], vals[cur], n, d,
                         (uint32_t)vocab, grad_table_or_table, accum, lr, eps, adagrad, piece, part);
    else
      hipLaunchKernelGGL((scatter_add_u32_kernel<4>), grid, block, 0, s, grad_out, keys[cur], vals[cur], n, d,
                         (uint32_t)vocab, grad_table_or_table, accum, lr, eps, adagrad, piece, part);
  } else {
    hipLaunchKernelGGL((scatter_add_pieces_kernel<1>), pgrid, block, 0, s, grad_out, keys[cur], vals[cur], n, d,
                       (uint32_t)vocab, piece, part);
    hipLaunchKernelGGL((scatter_add_u32_kernel<1>), grid, block, 0, s, grad_out, keys[cur], vals[cur], n, d,
                       (uint32_t)vocab, grad_table_or_table, accum, lr, eps, adagrad, piece, part);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// ---- dense Adagrad of several parameters in ONE launch -------------------------------------------------------------
// The dense parameters of a ranking model (Cross kernels, MLP kernels and biases: 18 tensors at configs[3]) were
// updated by four torch kernels each -- addcmul, add, sqrt, addcdiv: 72 launches and 0.43 ms of a 54 ms step, most of
// them a few KB.  Here tensor t owns blocks [first_block[t], first_block[t + 1]) of 256 threads x 4 x float4; same
// arithmetic as the sparse rows (adagrad_denom): acc += g * g; p -= lr * g / denom(acc).
namespace tfrs {
struct DenseAdagradTensors {
  int ntensors;
  int first_block[33];
  float *p[32];
  float *acc[32];
  const float *g[32];
  int64_t n[32];
};
constexpr int kDenseAdagradPerBlock = 256 * 16;
__global__ void __launch_bounds__(256) adagrad_dense_multi_kernel(const DenseAdagradTensors t, float lr, float eps,
                                                                  int mode) {
  int k = 0;
  while (k + 1 < t.ntensors && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
  float *__restrict__ p = t.p[k];
  float *__restrict__ acc = t.acc[k];
  const float *__restrict__ g = t.g[k];
  const int64_t n = t.n[k];
  const int64_t base = (int64_t)((int)blockIdx.x - t.first_block[k]) * kDenseAdagradPerBlock;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
  if (vec && base + kDenseAdagradPerBlock <= n) {
    float4 gv[4], av[4], pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + (int64_t)(u * 256 + threadIdx.x) * 4;
      gv[u] = *reinterpret_cast<const float4 *>(g + i);
      av[u] = *reinterpret_cast<const float4 *>(acc + i);
      pv[u] = *reinterpret_cast<const float4 *>(p + i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + (int64_t)(u * 256 + threadIdx.x) * 4;
      av[u].x += gv[u].x * gv[u].x; av[u].y += gv[u].y * gv[u].y; av[u].z += gv[u].z * gv[u].z; av[u].w += gv[u].w * gv[u].w;
      pv[u].x -= lr * gv[u].x / adagrad_denom(av[u].x, eps, mode);
      pv[u].y -= lr * gv[u].y / adagrad_denom(av[u].y, eps, mode);
      pv[u].z -= lr * gv[u].z / adagrad_denom(av[u].z, eps, mode);
      pv[u].w -= lr * gv[u].w / adagrad_denom(av[u].w, eps, mode);
      *reinterpret_cast<float4 *>(acc + i) = av[u];
      *reinterpret_cast<float4 *>(p + i) = pv[u];
    }
    return;
  }
  for (int64_t i = base + threadIdx.x; i < n && i < base + kDenseAdagradPerBlock; i += 256) {
    const float gi = g[i];
    const float a = acc[i] + gi * gi;
    acc[i] = a;
    p[i] = p[i] - lr * gi / adagrad_denom(a, eps, mode);
  }
}
}  // namespace tfrs

extern "C" int tfrs_adagrad_dense_multi(int ntensors, float *const *params_h, float *const *accum_h,
                                        const float *const *grads_h, const int64_t *n_h, float lr, float eps,
                                        int mode, void *stream) {
  TFRS_CHECK_ARG(ntensors >= 1 && ntensors <= 32, "adagrad_dense_multi: 1..32 tensors");
  TFRS_CHECK_ARG(params_h && accum_h && grads_h && n_h, "adagrad_dense_multi: NULL argument array");
  TFRS_CHECK_ARG(mode == 1 || mode == 2, "adagrad_dense_multi: mode must be 1 (sqrt(acc + eps)) or 2 (sqrt(acc) + eps)");
  tfrs::DenseAdagradTensors t = {};
  t.ntensors = ntensors;
  int64_t blocks = 0;
  for (int i = 0; i < ntensors; ++i) {
    TFRS_CHECK_ARG(n_h[i] >= 0 && (n_h[i] == 0 || (params_h[i] && accum_h[i] && grads_h[i])),
                   "adagrad_dense_multi: bad tensor %d", i);
    t.first_block[i] = (int)blocks;
    blocks += (n_h[i] + tfrs::kDenseAdagradPerBlock - 1) / tfrs::kDenseAdagradPerBlock;
    TFRS_CHECK_ARG(blocks < (1ll << 31), "adagrad_dense_multi: too many elements for one launch");
    t.p[i] = params_h[i]; t.acc[i] = accum_h[i]; t.g[i] = grads_h[i]; t.n[i] = n_h[i];
  }
  t.first_block[ntensors] = (int)blocks;
  if (blocks == 0) return TFRS_OK;
  hipLaunchKernelGGL(tfrs::adagrad_dense_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, lr,
                     eps, mode);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// ---- a batch's input tensors -> the static buffers of a captured step, ONE launch ----------------------------------
// `Model.fit` replays a captured train step on static input buffers; the batch (two id vectors of 32 KB at the quickstart
// shapes) was copied in by torch._foreach_copy_: 6.0-6.4 us of a 113 us step for 64 KB.  Buffer t owns blocks
// [first_block[t], first_block[t + 1]) of 256 threads x 4 x 16 bytes.
namespace tfrs {
struct CopyBuffers {
  int n;
  int first_block[17];
  char *dst[16];
  const char *src[16];
  int64_t bytes[16];
};
constexpr int kCopyPerBlock = 256 * 64;
__global__ void __launch_bounds__(256) copy_multi_kernel(const CopyBuffers t) {
  int k = 0;
  while (k + 1 < t.n && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
  char *__restrict__ dst = t.dst[k];
  const char *__restrict__ src = t.src[k];
  const int64_t n = t.bytes[k];
  const int64_t base = (int64_t)((int)blockIdx.x - t.first_block[k]) * kCopyPerBlock;
  const bool vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 && n >= 16;
  if (vec) {
    // unconditional (clamped) loads first, then the stores: one memory round trip per block
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t o = base + (int64_t)(u * 256 + threadIdx.x) * 16;
      v[u] = *reinterpret_cast<const uint4 *>(src + (o + 16 <= n ? o : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t o = base + (int64_t)(u * 256 + threadIdx.x) * 16;
      if (o + 16 <= n) *reinterpret_cast<uint4 *>(dst + o) = v[u];
    }
    // the last n % 16 bytes of the buffer: the block that owns them
    const int64_t tail = n & ~(int64_t)15;
    if (tail >= base && tail < base + kCopyPerBlock && (int64_t)threadIdx.x < n - tail) dst[tail + threadIdx.x] = src[tail + threadIdx.x];
    return;
  }
  for (int64_t o = base + threadIdx.x; o < n && o < base + kCopyPerBlock; o += 256) dst[o] = src[o];
}
}  // namespace tfrs

extern "C" int tfrs_copy_multi(int nbuffers, void *const *dst_h, const void *const *src_h, const int64_t *bytes_h,
                               void *stream) {
  TFRS_CHECK_ARG(nbuffers >= 1 && nbuffers <= 16, "copy_multi: 1..16 buffers");
  TFRS_CHECK_ARG(dst_h && src_h && bytes_h, "copy_multi: NULL argument array");
  tfrs::CopyBuffers t = {};
  t.n = nbuffers;
  int64_t blocks = 0;
  for (int i = 0; i < nbuffers; ++i) {
    TFRS_CHECK_ARG(bytes_h[i] >= 0 && (bytes_h[i] == 0 || (dst_h[i] && src_h[i])), "copy_multi: bad buffer %d", i);
    t.first_block[i] = (int)blocks;
    blocks += (bytes_h[i] + tfrs::kCopyPerBlock - 1) / tfrs::kCopyPerBlock;
    TFRS_CHECK_ARG(blocks < (1ll << 31), "copy_multi: too many bytes for one launch");
    t.dst[i] = static_cast<char *>(dst_h[i]); t.src[i] = static_cast<const char *>(src_h[i]); t.bytes[i] = bytes_h[i];
  }
  t.first_block[nbuffers] = (int)blocks;
  if (blocks == 0) return TFRS_OK;
  hipLaunchKernelGGL(tfrs::copy_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_embedding_scatter_add_rowscan_multi(int ntables, const float *const *grad_out_h,
                                                        const void *const *ids_h,
                                                        const int *ids_are_i64_h, const int64_t *n_h,
                                                        const int *d_h, const int64_t *vocab_h,
                                                        float *const *tables_h, float *const *accum_h,
                                                        float lr, float eps, int adagrad,
                                                        void *stream) {
  TFRS_CHECK_ARG(ntables >= 1 && ntables <= 8, "embedding_scatter_add_rowscan_multi: 1..8 tables");
  TFRS_CHECK_ARG(grad_out_h && ids_h && ids_are_i64_h && n_h && d_h && vocab_h && tables_h,
                 "embedding_scatter_add_rowscan_multi: NULL argument array");
  tfrs::RowscanTables t = {};
  t.ntab = ntables;
  int blocks = 0;
  for (int i = 0; i < ntables; ++i) {
    TFRS_CHECK_ARG(n_h[i] >= 0 && d_h[i] >= 1 && d_h[i] <= 256 && vocab_h[i] >= 1,
                   "embedding_scatter_add_rowscan_multi: bad shape of table %d", i);
    TFRS_CHECK_ARG(tables_h[i] && (n_h[i] == 0 || (grad_out_h[i] && ids_h[i])) && (!adagrad || (accum_h && accum_h[i])),
                   "embedding_scatter_add_rowscan_multi: NULL pointer for table %d", i);
    t.first_block[i] = blocks;
    blocks += (int)((vocab_h[i] + 3) / 4);
    t.grad_out[i] = grad_out_h[i]; t.ids[i] = ids_h[i]; t.n[i] = n_h[i]; t.d[i] = d_h[i];
    t.vocab[i] = vocab_h[i]; t.dst[i] = tables_h[i]; t.accum[i] = accum_h ? accum_h[i] : nullptr;
    t.i64[i] = ids_are_i64_h[i];
  }
  t.first_block[ntables] = blocks;
  hipLaunchKernelGGL(tfrs::scatter_rowscan_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     t, lr, eps, adagrad);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_embedding_scatter_add_rowscan(const float *grad_out, const void *ids,
                                                  int ids_are_i64, int64_t n, int d,
                                                  int64_t vocab, float *grad_table_or_table,
                                                  float *accum, float lr, float eps, int adagrad,
                                                  void *stream) {
  TFRS_CHECK_ARG(n >= 0 && d >= 1 && vocab >= 1, "embedding_scatter_add_rowscan: bad shape");
  TFRS_CHECK_ARG(d <= 256, "embedding_scatter_add_rowscan: d=%d > 256 (use the sorted path)", d);
  TFRS_CHECK_ARG((n == 0 || (grad_out && ids)) && grad_table_or_table,
                 "embedding_scatter_add_rowscan: NULL pointer");
  TFRS_CHECK_ARG(!adagrad || accum, "embedding_scatter_add_rowscan: Adagrad needs an accumulator");
  const dim3 grid((unsigned)((vocab + 3) / 4)), block(256);
  if (ids_are_i64)
    hipLaunchKernelGGL((scatter_rowscan_kernel<int64_t>), grid, block, 0, (hipStream_t)stream, grad_out, ids, n, d, vocab, grad_table_or_table, accum, lr, eps, adagrad);
  else
    hipLaunchKernelGGL((scatter_rowscan_kernel<int32_t>), grid, block, 0, (hipStream_t)stream, grad_out, ids, n, d, vocab, grad_table_or_table, accum, lr, eps, adagrad);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
