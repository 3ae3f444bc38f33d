// topk_scan.hip -- the hot kernel: S = Q x Cand^T on the f32 matrix cores with the
// top-K selection fused behind it, so the [nq, n] score matrix never exists.
//
// Replaces tf.matmul + tf.math.top_k of BruteForce.call
// (layers/factorized_top_k.py:603-605) and of Streaming's top_scores (:429-436).
//
// Work decomposition (one launch = one "round" over candidate rows [c_begin, c_end)):
//   grid      = n_splits x n_qtiles workgroups of 512 threads (8 waves, 2 per SIMD);
//               XCD-aware remap so the 32 workgroups resident on one XCD stream the
//               SAME candidate split and share it through that XCD's L2.
//   workgroup = 256 queries x one candidate split; candidate stages of 128 packed
//               rows are double-buffered in LDS and shared by all 8 waves.
//   wave      = 32 queries, held for the whole kernel as the MFMA B operand in
//               DP/2 VGPRs; for each 32-candidate sub-tile it issues DP/2
//               v_mfma_f32_32x32x2_f32 (A = candidates from LDS via ds_read_b128).
//               Operands are swapped (A = candidates, B = queries) so that every
//               accumulator register of a lane belongs to ONE query: the query's
//               threshold is a single VGPR and the common case "nothing in this
//               32x32 tile beats the current K-th score" costs 8 v_max3 + 1 compare
//               per 16 scores.
//   filter    = scores > thr[query] are appended to a per-wave LDS queue
//               (ballot/mbcnt compaction, no atomics) that is drained to the
//               per-query global lists with one atomic slot grab per entry.
//   exactness = the accumulator is bit-for-bit an fmaf chain over d = 0..D-1
//               (packed layout feeds features 2s / 2s+1 to lanes 0-31 / 32-63 of step
//               s), so scores compare == with oracle/c/oracle_core.c.
//
// Roofline: MFMA-bound (f32 matrix rate 157.3 TFLOP/s): 2*DP flop per score;
// LDS traffic is 8 KiB per wave per 32*DP/2 MFMAs (<15 % of the LDS rate).
#include "common.h"

namespace tfrs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs

constexpr int kWaves = 8;          // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kQueriesPerWg = kWaves * 32;
constexpr int kQueueCap = 256;     // entries per wave queue (8 B each)

template <int DP>
struct ScanGeom {
  static constexpr int kSteps = DP / 2;
  static constexpr int kRowB = DP * 4 + 16;
  static constexpr int kStageB = kTileN * kRowB;
  static constexpr int kChunks = kStageB / 16;                        // 16-B chunks per stage
  static constexpr int kLoads = (kChunks + kThreads - 1) / kThreads;  // per thread
  static constexpr int kLdsBytes = 2 * kStageB + kWaves * kQueueCap * 8;
};

__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Drains this wave's LDS queue into the per-query global lists.
__device__ __forceinline__ void flush_queue(const uint2 *queue, uint32_t qcnt, int lane,
                                            int64_t q_row0, int64_t c0,
                                            const ScanArgs &a) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t e = lane; e < qcnt; e += 64) {
    const uint2 ent = queue[e];
    const int64_t row = q_row0 + (ent.y >> 27);
    const uint32_t idx = (uint32_t)(c0 + (int64_t)(ent.y & 0x07FFFFFFu));
    const uint32_t slot = atomicAdd(&a.cnt[row], 1u);
    if (slot < a.cap) {
      a.buf[row * (int64_t)a.cap + slot] = make_uint2(ent.x, idx);
    } else {
      a.overflow[row] = 1u;  // select_kernel recomputes this row exactly
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int DP, bool MATERIALIZE>
__global__ void __launch_bounds__(kThreads, 2) scan_kernel(const ScanArgs a) {
  using G = ScanGeom<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;   // query column of this lane
  const int h = lane >> 5;   // feature parity plane / upper candidate half

  // ---- XCD-aware workgroup remap (bijective for any grid size) -----------------
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = bid & 7, pos = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
  const int split = logical / a.n_qtiles;
  const int qt = logical - split * a.n_qtiles;

  const int64_t c0 = a.c_begin + (int64_t)split * a.split_len;
  int64_t c1 = c0 + a.split_len;
  if (c1 > a.c_end) c1 = a.c_end;
  if (c0 >= c1) return;
  const int nstages = (int)((c1 - c0 + kTileN - 1) / kTileN);

  // ---- this wave's 32 queries -> MFMA B operand, resident for the whole kernel ---
  const int64_t q_row0 = (int64_t)qt * kQueriesPerWg + wave * 32;
  const int64_t qrow = q_row0 + j;
  const bool qvalid = qrow < a.nq;
  float bq[G::kSteps];
#pragma unroll
  for (int s = 0; s < G::kSteps; ++s) {
    const int k = 2 * s + h;
    bq[s] = (qvalid && k < a.d) ? a.q[qrow * a.d + k] : 0.0f;
  }
  float thr = 0.0f;
  if (!MATERIALIZE) thr = qvalid ? a.thr[qrow] : __builtin_inff();

  uint2 *queue = reinterpret_cast<uint2 *>(smem + 2 * G::kStageB) + wave * kQueueCap;
  uint32_t qcnt = 0;  // wave-uniform

  // ---- stage 0 -> LDS ------------------------------------------------------------
  const char *gsrc = a.packed + c0 * (int64_t)G::kRowB;
  // Every thread moves kLoads 16-byte chunks per stage; chunk numbers past the stage end are
  // clamped for the load (harmless re-read) and skipped for the LDS write, which keeps
  // the staging registers unconditionally defined (no scratch).
  f32x4 stg[G::kLoads];
#pragma unroll
  for (int i = 0; i < G::kLoads; ++i) {
    const int ch = min(tid + i * kThreads, G::kChunks - 1);
    stg[i] = *reinterpret_cast<const f32x4 *>(gsrc + ch * 16);
  }
#pragma unroll
  for (int i = 0; i < G::kLoads; ++i) {
    const int ch = tid + i * kThreads;
    if (i + 1 < G::kLoads || ch < G::kChunks) *reinterpret_cast<f32x4 *>(smem + ch * 16) = stg[i];
  }
  __syncthreads();

  for (int st = 0; st < nstages; ++st) {
    const char *tile = smem + (st & 1) * G::kStageB;
    const bool more = (st + 1 < nstages);
    if (more) {  // prefetch the next stage into registers; written to LDS after compute
      const char *g = gsrc + (int64_t)(st + 1) * G::kStageB;
#pragma unroll
      for (int i = 0; i < G::kLoads; ++i) {
        const int ch = min(tid + i * kThreads, G::kChunks - 1);
        stg[i] = *reinterpret_cast<const f32x4 *>(g + ch * 16);
      }
    }

    const int64_t stage_c = c0 + (int64_t)st * kTileN;  // first candidate row of this stage
#pragma unroll 1
    for (int sub = 0; sub < kTileN / 32; ++sub) {
      const int64_t sub_c = stage_c + sub * 32;
      if (sub_c >= c1) break;  // wave-uniform: nothing valid left in this stage

      // A operand: candidate row (sub*32 + j), plane h: DP/8 slots of 4 consecutive steps
      const char *ap = tile + (sub * 32 + j) * G::kRowB + h * (DP / 8) * 16;
      f32x4 a4[G::kSteps / 4];
#pragma unroll
      for (int m = 0; m < G::kSteps / 4; ++m)
        a4[m] = *reinterpret_cast<const f32x4 *>(ap + m * 16);

      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int m = 0; m < G::kSteps / 4; ++m) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].x, bq[4 * m + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].y, bq[4 * m + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].z, bq[4 * m + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].w, bq[4 * m + 3], acc, 0, 0, 0);
      }
      // acc[r] = score(query j, candidate sub_c + (r&3) + 8*(r>>2) + 4*h)

      if (MATERIALIZE) {
        if (qvalid) {
          float *drow = a.dense + qrow * a.ld_dense + (sub_c - a.c_begin) + 4 * h;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            *reinterpret_cast<float4 *>(drow + 8 * g4) =
                make_float4(acc[4 * g4 + 0], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
          }
        }
      } else {
        float m0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
        float m1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
        float m2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]);
        float m3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
        float m4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]);
        m0 = fmaxf(fmaxf(m0, m1), m2);
        m3 = fmaxf(fmaxf(m3, m4), acc[15]);
        m0 = fmaxf(m0, m3);
        if (__ballot(m0 > thr) != 0ull) {  // rare after warm-up: something may enter the top-K
          const bool ragged = (sub_c + 32 > c1);
          const uint32_t off0 = (uint32_t)(sub_c - c0) + 4u * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t off = off0 + (r & 3) + 8 * (r >> 2);
            bool p = acc[r] > thr;
            if (ragged) p = p && (c0 + (int64_t)off < c1);
            const uint64_t mask = __ballot(p);
            if (mask != 0ull) {
              if (qcnt + 64 > kQueueCap) {
                flush_queue(queue, qcnt, lane, q_row0, c0, a);
                qcnt = 0;
              }
              if (p) {
                queue[qcnt + mbcnt64(mask)] =
                    make_uint2(__float_as_uint(acc[r]), ((uint32_t)j << 27) | off);
              }
              qcnt += (uint32_t)__popcll(mask);
            }
          }
        }
      }
    }

    if (more) {
      char *dst = smem + ((st + 1) & 1) * G::kStageB;
#pragma unroll
      for (int i = 0; i < G::kLoads; ++i) {
        const int ch = tid + i * kThreads;
        if (i + 1 < G::kLoads || ch < G::kChunks) *reinterpret_cast<f32x4 *>(dst + ch * 16) = stg[i];
      }
    }
    __syncthreads();
  }

  if (!MATERIALIZE) {
    if (qcnt) flush_queue(queue, qcnt, lane, q_row0, c0, a);
  }
}

template <int DP>
static int launch_scan_dp(const ScanArgs &a, bool materialize, hipStream_t stream) {
  using G = ScanGeom<DP>;
  static bool attr_set[2] = {false, false};
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  if (materialize) {
    if (!attr_set[0]) {
      TFRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<DP, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   G::kLdsBytes));
      attr_set[0] = true;
    }
    hipLaunchKernelGGL((scan_kernel<DP, true>), grid, dim3(kThreads), G::kLdsBytes, stream, a);
  } else {
    if (!attr_set[1]) {
      TFRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<DP, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   G::kLdsBytes));
      attr_set[1] = true;
    }
    hipLaunchKernelGGL((scan_kernel<DP, false>), grid, dim3(kThreads), G::kLdsBytes, stream, a);
  }
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_scan(const ScanArgs &a, bool materialize, hipStream_t stream) {
  if (a.nq <= 0 || a.c_end <= a.c_begin) return TFRS_OK;
  TFRS_CHECK_ARG(a.c_begin % kTileN == 0 && a.split_len % kTileN == 0,
                 "scan: c_begin/split_len must be multiples of %d", kTileN);
  TFRS_CHECK_ARG(a.split_len <= (1 << 27), "scan: split too long");
  switch (padded_dim(a.d)) {
    case 8: return launch_scan_dp<8>(a, materialize, stream);
    case 16: return launch_scan_dp<16>(a, materialize, stream);
    case 32: return launch_scan_dp<32>(a, materialize, stream);
    case 64: return launch_scan_dp<64>(a, materialize, stream);
    case 128: return launch_scan_dp<128>(a, materialize, stream);
  }
  set_error("scan: unsupported dim %d", a.d);
  return TFRS_ENOTIMPL;
}

}  // namespace tfrs
