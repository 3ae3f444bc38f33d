// softmax.hip -- in-batch sampled-softmax loss of tfrs.tasks.Retrieval, forward and
// backward, without materialising the [nq, nc] logits.
//
// Replaces (tasks/retrieval.py) the scores matmul :178-180, labels=eye :185, temperature
// :187-188, SamplingProbablityCorrection :190-192 (layers/loss.py:150-158),
// RemoveAccidentalHits :194-200 (layers/loss.py:114-147), score_mask :202-203 and the Keras
// CategoricalCrossentropy(from_logits, SUM) :210, plus their gradients
// (models/base.py:77).
//
//   S_bc  = (q_b . c_c) * inv_t - corr_c  [+ MIN_FLOAT if ids_c == ids_b, c != b]
//           [= MIN_FLOAT where !mask_bc]
//   loss  = sum_b w_b (logsumexp_c S_bc - S_bb)
//   G_bc  = gloss * w_b * (exp(S_bc - lse_b) - [b == c]) * inv_t      (0 where masked)
//   dq_b  = sum_c G_bc c_c ,   dc_c = sum_b G_bc q_b
//
// Structure (flash-attention like): a wave owns 32 rows of one side as the MFMA B operand
// and streams 32-row tiles of the other side as the A operand straight from L2 (both
// embedding matrices of a batch are a few MB).  Lanes index the owned rows, accumulator
// registers index the streamed rows, so the online max/sum of a row is lane-local.  In
// the backward the tile of G stays in the accumulator layout and is fed back as the B
// operand of the second GEMM (out^T[feature][row] += X^T G) -- no LDS, no transposes.
// The streamed side is split across waves for occupancy; partial (max, sum) pairs and
// partial gradients are combined by small deterministic reduce kernels (no float atomics).
//
// Roofline: MFMA-bound, 2*nq*nc*d flop forward and 8*nq*nc*d backward (S is recomputed
// once per gradient); at the MovieLens batch (4096 x 4096 x 64) the whole step is a few
// tens of microseconds, i.e. launch-latency territory.
#include <algorithm>

#include <stdlib.h>

#include "mfma_tile.h"

namespace tfrs {

constexpr float kMinFloat = -3.4028234663852886e36f;  // np.finfo(float32).min / 100

struct SoftmaxArgs {
  const float *q, *c;
  int64_t nq, nc;
  int d;
  const float *w;       // [nq] sample weights or NULL
  float inv_t;          // 1 / temperature
  const float *corr;    // [nc] log(clip(p, 1e-6, 1)) or NULL
  const int64_t *ids;   // [nc] candidate ids (accidental-hit removal) or NULL
  const uint8_t *mask;  // [nq, nc] score_mask or NULL
  int nsplit;
  int64_t split_len;    // multiple of 32
  float *pm, *pl;       // [nsplit, nq] partial max / sum-exp
  float *ppos;          // [nq] positive logit
  const float *lse;     // [nq]
  const float *gloss;   // device scalar or NULL (= 1)
  float *partial;       // [nsplit, rows, d] partial gradients
  uint32_t *ticket;     // finalize kernel's arrival counter (re-armed by the forward kernel)
};

__device__ __forceinline__ float make_logit(float dot, int64_t query, int64_t cand,
                                            const SoftmaxArgs &a, float corr_c,
                                            int64_t id_q, int64_t id_c, bool *masked) {
  float v = dot * a.inv_t;
  if (a.corr) v -= corr_c;
  if (a.ids && cand != query && id_c == id_q) v += kMinFloat;
  *masked = false;
  if (a.mask && !a.mask[query * a.nc + cand]) {
    v = kMinFloat;
    *masked = true;
  }
  return v;
}

// PLAIN: no sampling-probability correction, no accidental-hit removal, no score mask (the
// default Retrieval configuration): those branches are compiled out of the tile epilogue.
template <int DP, bool PLAIN>
__global__ void __launch_bounds__(256) softmax_fwd_kernel(const SoftmaxArgs a_in) {
  SoftmaxArgs a = a_in;
  if (PLAIN) {
    a.corr = nullptr;
    a.ids = nullptr;
    a.mask = nullptr;
  }
  // re-arms the finalize kernel's ticket: it runs strictly after this kernel (a hipMemsetAsync
  // node is not reliably ordered against kernel nodes when the step replays from a HIP graph)
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.ticket = 0u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nqb = (a.nq + 31) / 32;
  if (wid >= nqb * a.nsplit) return;
  const int64_t qb = wid / a.nsplit;
  const int sp = (int)(wid - qb * a.nsplit);
  const int j = lane & 31, h = lane >> 5;
  const int64_t row = qb * 32 + j;
  const bool rvalid = row < a.nq;
  const bool vec_ok = (a.d == DP) && ((((uintptr_t)a.q) | ((uintptr_t)a.c)) % 16 == 0);

  float bq[DP / 2];
  load_row_frag<DP>(bq, a.q, row, rvalid, a.d, h, vec_ok);
  const int64_t id_q = (a.ids && rvalid) ? a.ids[row] : 0;

  float m = -__builtin_inff(), l = 0.0f, pos = 0.0f;
  bool haspos = false;
  const int64_t c_lo = (int64_t)sp * a.split_len;
  int64_t c_hi = c_lo + a.split_len;
  if (c_hi > a.nc) c_hi = a.nc;

  // the next tile's fragment is fetched while the current one is multiplied
  float af_next[DP / 2];
  load_row_frag<DP>(af_next, a.c, c_lo + j, c_lo + j < a.nc && c_lo < c_hi, a.d, h, vec_ok);
  for (int64_t c0 = c_lo; c0 < c_hi; c0 += 32) {
    float af[DP / 2];
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) af[s] = af_next[s];
    if (c0 + 32 < c_hi) load_row_frag<DP>(af_next, a.c, c0 + 32 + j, c0 + 32 + j < a.nc, a.d, h, vec_ok);
    const f32x16 acc = tile_dot<DP>(af, bq);

    float s[16];
    float tmax = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t cand = c0 + tile_row_of_reg(r, h);
      const bool valid = rvalid && cand < c_hi;
      float v = -__builtin_inff();
      if (valid) {
        bool masked;
        v = make_logit(acc[r], row, cand, a, a.corr ? a.corr[cand] : 0.0f, id_q,
                       a.ids ? a.ids[cand] : 0, &masked);
        if (cand == row) {
          pos = v;
          haspos = true;
        }
      }
      s[r] = v;
      tmax = fmaxf(tmax, v);
    }
    if (tmax > m) {
      l *= __expf(m - tmax);  // m = -inf on the first tile: exp(-inf) = 0, l = 0
      m = tmax;
    }
    if (m > -__builtin_inff()) {
#pragma unroll
      for (int r = 0; r < 16; ++r) l += __expf(s[r] - m);  // s = -inf contributes 0
    }
  }

  // the two lane halves hold disjoint candidates of the same query
  const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32);
  const float mm = fmaxf(m, m2);
  float ll = 0.0f;
  if (m > -__builtin_inff()) ll += l * __expf(m - mm);
  if (m2 > -__builtin_inff()) ll += l2 * __expf(m2 - mm);
  if (h == 0 && rvalid) {
    a.pm[(int64_t)sp * a.nq + row] = mm;
    a.pl[(int64_t)sp * a.nq + row] = ll;
  }
  if (haspos) a.ppos[row] = pos;  // exactly one lane of one split sees c == b
}

// Combines the per-split (max, sum) pairs, writes lse/pos and the weighted loss.
// One thread per query row; every block leaves a partial loss in block_part[] and the LAST
// block to finish (ticket counter) adds the partials in block order, so the result does not
// depend on scheduling.  `ticket` must be zero on entry and is reset for the next launch.
__global__ void __launch_bounds__(256) softmax_finalize_kernel(const SoftmaxArgs a, float *out_loss,
                                                               float *out_lse, float *out_pos,
                                                               double *block_part, uint32_t *ticket) {
  __shared__ double red[256];
  __shared__ bool is_last;
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double local = 0.0;
  if (row < a.nq) {
    float mm = -__builtin_inff();
    for (int sp = 0; sp < a.nsplit; ++sp) mm = fmaxf(mm, a.pm[(int64_t)sp * a.nq + row]);
    float ll = 0.0f;
    for (int sp = 0; sp < a.nsplit; ++sp) {
      const float pm = a.pm[(int64_t)sp * a.nq + row];
      if (pm > -__builtin_inff()) ll += a.pl[(int64_t)sp * a.nq + row] * expf(pm - mm);
    }
    const float lse = mm + logf(ll);
    const float pos = a.ppos[row];
    out_lse[row] = lse;
    out_pos[row] = pos;
    const float w = a.w ? a.w[row] : 1.0f;
    local = (double)w * ((double)lse - (double)pos);
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&block_part[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (is_last && threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double total = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b)
      total += __hip_atomic_load(&block_part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out_loss = (float)total;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ROWS_ARE_QUERIES = true : wave owns 32 queries, streams candidates, emits partial dq.
// ROWS_ARE_QUERIES = false: wave owns 32 candidates, streams queries, emits partial dc.
template <int DP, bool ROWS_ARE_QUERIES, bool PLAIN>
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const SoftmaxArgs a_in) {
  SoftmaxArgs a = a_in;
  if (PLAIN) {
    a.corr = nullptr;
    a.ids = nullptr;
    a.mask = nullptr;
  }
  constexpr int NFB = (DP + 31) / 32;  // 32-feature output blocks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_r = ROWS_ARE_QUERIES ? a.nq : a.nc;
  const int64_t n_s = ROWS_ARE_QUERIES ? a.nc : a.nq;
  const float *rdata = ROWS_ARE_QUERIES ? a.q : a.c;
  const float *sdata = ROWS_ARE_QUERIES ? a.c : a.q;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nrb = (n_r + 31) / 32;
  if (wid >= nrb * a.nsplit) return;
  const int64_t rb = wid / a.nsplit;
  const int sp = (int)(wid - rb * a.nsplit);
  const int j = lane & 31, h = lane >> 5;
  const int64_t rrow = rb * 32 + j;
  const bool rvalid = rrow < n_r;
  const bool vec_ok = (a.d == DP) && ((((uintptr_t)a.q) | ((uintptr_t)a.c)) % 16 == 0);

  float br[DP / 2];
  load_row_frag<DP>(br, rdata, rrow, rvalid, a.d, h, vec_ok);

  // per-lane constants of the owned row
  float lse_r = 0.0f, w_r = 1.0f, corr_r = 0.0f;
  int64_t id_r = 0;
  if (rvalid) {
    if (ROWS_ARE_QUERIES) {
      lse_r = a.lse[rrow];
      if (a.w) w_r = a.w[rrow];
    } else {
      if (a.corr) corr_r = a.corr[rrow];
    }
    if (a.ids) id_r = a.ids[rrow];  // nq <= nc, so a query row is also a valid candidate row
  }
  const float gl = (a.gloss ? *a.gloss : 1.0f) * a.inv_t;

  f32x16 outacc[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
    for (int r = 0; r < 16; ++r) outacc[fb][r] = 0.0f;

  const int64_t s_lo = (int64_t)sp * a.split_len;
  int64_t s_hi = s_lo + a.split_len;
  if (s_hi > n_s) s_hi = n_s;

  // Streamed tile (32 rows x DP features): fetched one tile ahead into registers (row j, half h
  // = the first GEMM's A fragment) and mirrored into this wave's LDS slab so that the second
  // GEMM can read it transposed (lane = feature) with conflict-free ds_read_b32.
  constexpr int kLd = DP + 4;  // floats per LDS row (+16 B: rows start in different bank groups)
  extern __shared__ __attribute__((aligned(16))) float smem_sm[];
  float *slab = smem_sm + (size_t)wave * 32 * kLd;
  float af_next[DP / 2];
  load_row_frag<DP>(af_next, sdata, s_lo + j, s_lo + j < n_s && s_lo < s_hi, a.d, h, vec_ok);

  for (int64_t s0 = s_lo; s0 < s_hi; s0 += 32) {
    float af[DP / 2];
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) af[s] = af_next[s];
    if (s0 + 32 < s_hi) load_row_frag<DP>(af_next, sdata, s0 + 32 + j, s0 + 32 + j < n_s, a.d, h, vec_ok);
#pragma unroll
    for (int m4 = 0; m4 < DP / 8; ++m4)
      *reinterpret_cast<float4 *>(slab + j * kLd + h * (DP / 2) + 4 * m4) =
          make_float4(af[4 * m4], af[4 * m4 + 1], af[4 * m4 + 2], af[4 * m4 + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x16 acc = tile_dot<DP>(af, br);

    float g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t srow = s0 + tile_row_of_reg(r, h);
      const bool valid = rvalid && srow < s_hi;
      float gg = 0.0f;
      if (valid) {
        const int64_t query = ROWS_ARE_QUERIES ? rrow : srow;
        const int64_t cand = ROWS_ARE_QUERIES ? srow : rrow;
        const float corr_c = ROWS_ARE_QUERIES ? (a.corr ? a.corr[cand] : 0.0f) : corr_r;
        const int64_t id_q = ROWS_ARE_QUERIES ? id_r : (a.ids ? a.ids[query] : 0);
        const int64_t id_c = ROWS_ARE_QUERIES ? (a.ids ? a.ids[cand] : 0) : id_r;
        bool masked;
        const float v = make_logit(acc[r], query, cand, a, corr_c, id_q, id_c, &masked);
        const float lse_q = ROWS_ARE_QUERIES ? lse_r : a.lse[query];
        const float w_q = ROWS_ARE_QUERIES ? w_r : (a.w ? a.w[query] : 1.0f);
        const float p = __expf(v - lse_q);
        gg = masked ? 0.0f : w_q * (p - (cand == query ? 1.0f : 0.0f)) * gl;
      }
      g[r] = gg;
    }

    // out^T[feature][owned row] += sum over the 32 streamed rows of X[srow][feature] * G[srow][row]
    // MFMA step r contracts the streamed-row pair (tile_row_of_reg(r,0), tile_row_of_reg(r,1)),
    // which is exactly where g[r] lives in lane halves 0 / 1.  Rows past the end and padded
    // features are zero in the slab.
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *xrow = slab + tile_row_of_reg(r, h) * kLd;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        const int feat = fb * 32 + j;
        const float av = (feat < DP) ? xrow[feat] : 0.0f;
        outacc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, g[r], outacc[fb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the slab is rewritten by the next tile
  }

  if (rvalid) {
    float *dst = a.partial + ((int64_t)sp * n_r + rrow) * a.d;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int feat = fb * 32 + tile_row_of_reg(r, h);
        if (feat < a.d) dst[feat] = outacc[fb][r];
      }
  }
}

__global__ void __launch_bounds__(256) reduce_partials_kernel(const float *partial, int nsplit,
                                                              int64_t count, float *out) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count;
       t += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) acc += partial[(int64_t)sp * count + t];
    out[t] = acc;
  }
}

static void plan(int64_t n_rows, int64_t n_stream, int *nsplit, int64_t *split_len) {
  const int64_t row_blocks = (n_rows + 31) / 32;
  const int64_t tiles = (n_stream + 31) / 32;
  static const int64_t target_waves = [] {
    const char *v = option("TFRS_SOFTMAX_WAVES");
    return (v && *v) ? (int64_t)atoll(v) : (int64_t)2048;  // ~2 waves per SIMD on 256 CUs
  }();
  int64_t want = (target_waves + row_blocks - 1) / row_blocks;
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  const int64_t per = (tiles + want - 1) / want;
  *split_len = per * 32;
  *nsplit = (int)((tiles + per - 1) / per);
}

static size_t al(size_t x) { return (x + 255) / 256 * 256; }

template <int DP>
static void launch_fwd(const SoftmaxArgs &a, hipStream_t s) {
  const int64_t waves = ((a.nq + 31) / 32) * a.nsplit;
  if (!a.corr && !a.ids && !a.mask)
    hipLaunchKernelGGL((softmax_fwd_kernel<DP, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((softmax_fwd_kernel<DP, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}

template <int DP, bool RQ>
static void launch_bwd(const SoftmaxArgs &a, hipStream_t s) {
  const int64_t rows = RQ ? a.nq : a.nc;
  const int64_t waves = ((rows + 31) / 32) * a.nsplit;
  const size_t lds = (size_t)4 * 32 * (DP + 4) * sizeof(float);
  if (!a.corr && !a.ids && !a.mask)
    hipLaunchKernelGGL((softmax_bwd_kernel<DP, RQ, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((softmax_bwd_kernel<DP, RQ, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, s, a);
}

// softmax16.hip: the split-fp16 path
size_t softmax16_workspace_bytes(int64_t nq, int64_t nc, int d);
int softmax16_forward(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                      float inv_t, float *out_loss, float *out_lse, float *out_pos, void *ws,
                      hipStream_t s);
int softmax16_backward(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                       float inv_t, const float *lse, const float *gloss, float *dq, float *dc,
                       void *ws, int reuse, hipStream_t s);

// TFRS_SOFTMAX_MODE=f32 keeps everything on the f32-MFMA kernels; default: the split-fp16 path
// whenever no logit option that needs per-element side inputs is set.
static bool use_f16_path(const float *corr, const int64_t *ids, const uint8_t *mask) {
  const char *v = option("TFRS_SOFTMAX_MODE");   // read per call: tests switch it
  const bool f32_only = v && v[0] == 'f' && v[1] == '3';
  return !f32_only && !corr && !ids && !mask;
}

}  // namespace tfrs

using namespace tfrs;

static int check_common(const float *q, const float *c, int64_t nq, int64_t nc, int d,
                        const char *who) {
  TFRS_CHECK_ARG(nq >= 1 && nc >= 1 && d >= 1, "%s: bad shape", who);
  TFRS_CHECK_ARG(nc >= nq, "%s: needs num_candidates >= num_queries (labels = eye)", who);
  TFRS_CHECK_ARG(q && c, "%s: NULL pointer", who);
  if (d > 128) {
    set_error("%s: embedding dim %d > 128 is not implemented", who, d);
    return TFRS_ENOTIMPL;
  }
  return TFRS_OK;
}

extern "C" size_t tfrs_inbatch_softmax_workspace_bytes(int64_t nq, int64_t nc, int d) {
  if (nq < 1 || nc < 1 || d < 1) return 256;
  int nsf, nsq, nsc;
  int64_t len;
  plan(nq, nc, &nsf, &len);
  plan(nq, nc, &nsq, &len);
  plan(nc, nq, &nsc, &len);
  const size_t fwd = 2 * al((size_t)nsf * nq * 4) + al((size_t)nq * 4) +
                     al((size_t)((nq + 255) / 256) * 8) + al(4);  // + finalize partials, ticket
  const size_t bwd = al((size_t)nsq * nq * d * 4) + al((size_t)nsc * nc * d * 4);
  const size_t f32 = fwd > bwd ? fwd : bwd;
  const size_t f16 = d <= 128 ? softmax16_workspace_bytes(nq, nc, d) : 0;
  return f32 > f16 ? f32 : f16;
}

extern "C" int tfrs_inbatch_softmax_ce_fwd(const float *q, const float *c, int64_t nq, int64_t nc,
                                           int d, const float *sample_weight,
                                           float inv_temperature, const float *log_q_correction,
                                           const int64_t *cand_ids, const uint8_t *score_mask,
                                           float *out_loss, float *out_lse, float *out_pos,
                                           void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_common(q, c, nq, nc, d, "inbatch_softmax_ce_fwd");
  if (rc != TFRS_OK) return rc;
  TFRS_CHECK_ARG(out_loss && out_lse && out_pos && workspace, "inbatch_softmax_ce_fwd: NULL output");
  if (workspace_bytes < tfrs_inbatch_softmax_workspace_bytes(nq, nc, d)) {
    set_error("inbatch_softmax_ce_fwd: workspace too small");
    return TFRS_ENOMEM;
  }
  if (use_f16_path(log_q_correction, cand_ids, score_mask))
    return softmax16_forward(q, c, nq, nc, d, sample_weight, inv_temperature, out_loss, out_lse,
                             out_pos, workspace, (hipStream_t)stream);
  SoftmaxArgs a = {};
  a.q = q; a.c = c; a.nq = nq; a.nc = nc; a.d = d;
  a.w = sample_weight; a.inv_t = inv_temperature; a.corr = log_q_correction;
  a.ids = cand_ids; a.mask = score_mask;
  plan(nq, nc, &a.nsplit, &a.split_len);
  char *p = static_cast<char *>(workspace);
  a.pm = reinterpret_cast<float *>(p); p += al((size_t)a.nsplit * nq * 4);
  a.pl = reinterpret_cast<float *>(p); p += al((size_t)a.nsplit * nq * 4);
  a.ppos = reinterpret_cast<float *>(p); p += al((size_t)nq * 4);
  const unsigned fin_blocks = (unsigned)((nq + 255) / 256);
  double *block_part = reinterpret_cast<double *>(p); p += al((size_t)fin_blocks * 8);
  uint32_t *ticket = reinterpret_cast<uint32_t *>(p);
  a.ticket = ticket;
  hipStream_t s = (hipStream_t)stream;
  switch (softmax_padded_dim(d)) {
    case 8: launch_fwd<8>(a, s); break;
    case 16: launch_fwd<16>(a, s); break;
    case 32: launch_fwd<32>(a, s); break;
    case 64: launch_fwd<64>(a, s); break;
    default: launch_fwd<128>(a, s); break;
  }
  TFRS_LAUNCH_CHECK();
  hipLaunchKernelGGL(softmax_finalize_kernel, dim3(fin_blocks), dim3(256), 0, s, a, out_loss, out_lse,
                     out_pos, block_part, ticket);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_inbatch_softmax_ce_bwd(const float *q, const float *c, int64_t nq, int64_t nc,
                                           int d, const float *sample_weight,
                                           float inv_temperature, const float *log_q_correction,
                                           const int64_t *cand_ids, const uint8_t *score_mask,
                                           const float *lse, const float *gloss, float *dq,
                                           float *dc, void *workspace, size_t workspace_bytes,
                                           int reuse_forward_workspace, void *stream) {
  int rc = check_common(q, c, nq, nc, d, "inbatch_softmax_ce_bwd");
  if (rc != TFRS_OK) return rc;
  TFRS_CHECK_ARG(lse && dq && dc && workspace, "inbatch_softmax_ce_bwd: NULL pointer");
  if (workspace_bytes < tfrs_inbatch_softmax_workspace_bytes(nq, nc, d)) {
    set_error("inbatch_softmax_ce_bwd: workspace too small");
    return TFRS_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  if (use_f16_path(log_q_correction, cand_ids, score_mask))
    return softmax16_backward(q, c, nq, nc, d, sample_weight, inv_temperature, lse, gloss, dq, dc,
                              workspace, reuse_forward_workspace, s);
  SoftmaxArgs a = {};
  a.q = q; a.c = c; a.nq = nq; a.nc = nc; a.d = d;
  a.w = sample_weight; a.inv_t = inv_temperature; a.corr = log_q_correction;
  a.ids = cand_ids; a.mask = score_mask; a.lse = lse; a.gloss = gloss;
  char *p = static_cast<char *>(workspace);

  // dq: waves own queries, stream candidates
  plan(nq, nc, &a.nsplit, &a.split_len);
  a.partial = reinterpret_cast<float *>(p);
  const int nsq = a.nsplit;
  switch (softmax_padded_dim(d)) {
    case 8: launch_bwd<8, true>(a, s); break;
    case 16: launch_bwd<16, true>(a, s); break;
    case 32: launch_bwd<32, true>(a, s); break;
    case 64: launch_bwd<64, true>(a, s); break;
    default: launch_bwd<128, true>(a, s); break;
  }
  TFRS_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<int64_t>((nq * d + 255) / 256, 2048)),
                     dim3(256), 0, s, a.partial, nsq, nq * d, dq);
  TFRS_LAUNCH_CHECK();

  // dc: waves own candidates, stream queries
  p += al((size_t)nsq * nq * d * 4);
  plan(nc, nq, &a.nsplit, &a.split_len);
  a.partial = reinterpret_cast<float *>(p);
  switch (softmax_padded_dim(d)) {
    case 8: launch_bwd<8, false>(a, s); break;
    case 16: launch_bwd<16, false>(a, s); break;
    case 32: launch_bwd<32, false>(a, s); break;
    case 64: launch_bwd<64, false>(a, s); break;
    default: launch_bwd<128, false>(a, s); break;
  }
  TFRS_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<int64_t>((nc * d + 255) / 256, 2048)),
                     dim3(256), 0, s, a.partial, a.nsplit, nc * d, dc);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
