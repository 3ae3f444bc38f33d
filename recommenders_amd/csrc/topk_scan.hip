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
//   filter    = scores > thr[query] are appended by the owning lane to its PRIVATE segment
//               of the query's survivor list in global memory (counter in a VGPR):
//               plain fire-and-forget stores -- no atomics, no LDS queue, no memset,
//               nothing that makes the wave wait on a memory round trip.
//   exactness = the accumulator is bit-for-bit an fmaf chain over d = 0..D-1
//               (packed layout feeds features 2s / 2s+1 to lanes 0-31 / 32-63 of step
//               s), so scores compare == with oracle/c/oracle_core.c.
//
// Roofline: MFMA-bound (f32 matrix rate 157.3 TFLOP/s): 2*DP flop per score;
// LDS traffic is 8 KiB per wave per 32*DP/2 MFMAs (<15 % of the LDS rate).
#include <stdlib.h>

#include "common.h"

namespace tfrs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs

constexpr int kWaves = 8;          // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kQueriesPerWg = kWaves * 32;

template <int DP>
struct ScanGeom {
  static constexpr int kSteps = DP / 2;
  static constexpr int kRowB = DP * 4 + 16;                           // row stride in LDS (lds_row_bytes: odd slot count)
  static constexpr int kRowG = DP * 4;                                // row stride of the image in memory (row_bytes)
  static constexpr int kSlotsL = kRowB / 16;                          // 16-B slots per LDS row, the last one a pad
  static constexpr int kStageB = kTileN * kRowB;
  static constexpr int kStageG = kTileN * kRowG;
  static constexpr int kChunks = kStageB / 16;                        // 16-B chunks per stage IN LDS
  static constexpr int kLoads = (kChunks + kThreads - 1) / kThreads;  // per thread
  static constexpr int kLdsBytes = 2 * kStageB;
};

// Stage copy HBM/L2 -> LDS.  GLDS = true uses the gfx950 direct-to-LDS load
// (global_load_lds_dwordx4: per-lane global address, LDS destination = wave-uniform base +
// lane * 16): no staging VGPRs and no ds_write pass.  GLDS = false stages through registers
// (global_load_dwordx4 -> ds_write_b128).  Either way chunk P of the LDS stage (row P / kSlotsL, slot P % kSlotsL of
// the padded LDS rows) comes from slot min(P % kSlotsL, last) of the unpadded row in memory: the 16 lanes of a row
// still read one contiguous 4 DP bytes (the pad slot re-reads the row's last slot; nobody reads it back).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int DP>
__device__ __forceinline__ int stage_src_offset(int chunk) {
  constexpr int kSlotsL = DP / 4 + 1;
  const int r = chunk / kSlotsL;
  const int sl = chunk - r * kSlotsL;
  return r * (DP * 4) + (sl < kSlotsL - 1 ? sl : kSlotsL - 2) * 16;
}

template <int DP, int CHUNKS, int LOADS>
__device__ __forceinline__ void stage_glds(const char *gsrc, char *lds_dst, int tid, int wave) {
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int ch0 = i * kThreads + wave * 64;  // first chunk of this wave's instruction
    if (ch0 < CHUNKS) {                         // wave-uniform (CHUNKS is a multiple of 64)
      __builtin_amdgcn_global_load_lds((gbl_void_t *)(gsrc + stage_src_offset<DP>(i * kThreads + tid)),
                                       (lds_void_t *)(lds_dst + ch0 * 16), 16, 0, 0);
    }
  }
}

template <int DP, bool MATERIALIZE, bool GLDS>
__global__ void __launch_bounds__(kThreads, DP <= 64 ? 4 : 2) scan_kernel(const ScanArgs a) {
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
  if (!MATERIALIZE) {
    thr = qvalid ? a.thr[qrow] : __builtin_inff();
    // shuffled index: `score > nextbelow(thr)` keeps the ties with the K-th best (they are decided
    // later by original row numbers)
    if (a.tie_ge && thr > -__builtin_inff() && thr < __builtin_inff()) {
      const uint32_t u = f32_orderable(thr);
      float below = f32_from_orderable(u - 1u);
      if (below == thr) below = f32_from_orderable(u - 2u);   // +0 -> -0 compares equal
      thr = below;
    }
  }

  // paged search: rows already returned by an earlier page score above the query's ceiling
  const float ceil = (!MATERIALIZE && a.ceil_score && qvalid) ? a.ceil_score[qrow] : __builtin_inff();

  // this lane's private survivor segment
  const int64_t seg = qrow * a.nseg + 2 * split + h;
  // entry-major list: entry e of this segment lives at buf[(qrow * cap_l + e) * nseg + seg]
  uint2 *mybuf = MATERIALIZE ? nullptr : a.buf + (qrow * (int64_t)a.cap_l) * a.nseg + (2 * split + h);
  uint32_t mycnt = 0;

  // ---- stage 0 -> LDS ------------------------------------------------------------
  const char *gsrc = a.packed + c0 * (int64_t)G::kRowG;
  // Every thread moves kLoads 16-byte chunks per stage; chunk numbers past the stage end are
  // clamped for the load (harmless re-read) and skipped for the LDS write, which keeps
  // the staging registers unconditionally defined (no scratch).
  static_assert(G::kChunks % 64 == 0, "stage size must be a whole number of wave copies");
  f32x4 stg[GLDS ? 1 : G::kLoads];
  if (GLDS) {
    stage_glds<DP, G::kChunks, G::kLoads>(gsrc, smem, tid, wave);
  } else {
#pragma unroll
    for (int i = 0; i < G::kLoads; ++i) {
      const int ch = min(tid + i * kThreads, G::kChunks - 1);
      stg[i] = *reinterpret_cast<const f32x4 *>(gsrc + stage_src_offset<DP>(ch));
    }
#pragma unroll
    for (int i = 0; i < G::kLoads; ++i) {
      const int ch = tid + i * kThreads;
      if (i + 1 < G::kLoads || ch < G::kChunks) *reinterpret_cast<f32x4 *>(smem + ch * 16) = stg[i];
    }
  }
  __syncthreads();

  for (int st = 0; st < nstages; ++st) {
    const char *tile = smem + (st & 1) * G::kStageB;
    const bool more = (st + 1 < nstages);
    if (more) {  // prefetch the next stage (other LDS buffer: its readers passed the last barrier)
      const char *g = gsrc + (int64_t)(st + 1) * G::kStageG;
      if (GLDS) {
        stage_glds<DP, G::kChunks, G::kLoads>(g, smem + ((st + 1) & 1) * G::kStageB, tid, wave);
      } else {
#pragma unroll
        for (int i = 0; i < G::kLoads; ++i) {
          const int ch = min(tid + i * kThreads, G::kChunks - 1);
          stg[i] = *reinterpret_cast<const f32x4 *>(g + stage_src_offset<DP>(ch));
        }
      }
    }

    const int64_t stage_c = c0 + (int64_t)st * kTileN;  // first candidate row of this stage

    // A operand of sub-tile 0: candidate row j, plane h: DP/8 slots of 4 consecutive steps.
    // The fragment registers are refilled IN PLACE with the next sub-tile's slots as soon as
    // the four MFMAs that consume a slot have issued, so the LDS reads of sub-tile t+1 run
    // under the MFMA chain of sub-tile t at no extra register cost.
    const char *ap = tile + j * G::kRowB + h * (DP / 8) * 16;
    f32x4 a4[G::kSteps / 4];
#pragma unroll
    for (int m = 0; m < G::kSteps / 4; ++m) a4[m] = *reinterpret_cast<const f32x4 *>(ap + m * 16);

#pragma unroll 1
    for (int sub = 0; sub < kTileN / 32; ++sub) {
      const int64_t sub_c = stage_c + sub * 32;
      if (sub_c >= c1) break;  // wave-uniform: nothing valid left in this stage
      // refill source: the next sub-tile, or (harmlessly) this one again on the last pass
      const bool has_next = (sub + 1 < kTileN / 32) && (sub_c + 32 < c1);
      const char *an = ap + (has_next ? 32 * G::kRowB : 0);

      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int m = 0; m < G::kSteps / 4; ++m) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][0], bq[4 * m + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][1], bq[4 * m + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][2], bq[4 * m + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][3], bq[4 * m + 3], acc, 0, 0, 0);
        a4[m] = *reinterpret_cast<const f32x4 *>(an + m * 16);
      }
      ap = an;
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
        // group maxima (registers 4g..4g+3 = 4 consecutive candidates of this lane)
        const float g0 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
        const float g1 = fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]));
        const float g2 = fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11]));
        const float g3 = fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]));
        const float m0 = fmaxf(fmaxf(g0, g1), fmaxf(g2, g3));
        if (__ballot(m0 > thr) != 0ull) {  // rare after warm-up: something may enter the top-K
          const bool ragged = (sub_c + 32 > c1);
          const uint32_t off0 = (uint32_t)(sub_c - c0) + 4u * h;
          const float gm[4] = {g0, g1, g2, g3};
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            if (__ballot(gm[g4] > thr) == 0ull) continue;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int r = 4 * g4 + rr;
              const uint32_t off = off0 + rr + 8 * g4;
              bool p = acc[r] > thr && acc[r] <= ceil;
              if (ragged) p = p && (c0 + (int64_t)off < c1);
              if (p) {
                if (mycnt < a.cap_l)
                  mybuf[(size_t)mycnt * a.nseg] =
                      make_uint2(__float_as_uint(acc[r]), (uint32_t)(c0 + (int64_t)off));
                ++mycnt;
              }
            }
          }
        }
      }
    }

    if (more && !GLDS) {
      char *dst = smem + ((st + 1) & 1) * G::kStageB;
#pragma unroll
      for (int i = 0; i < G::kLoads; ++i) {
        const int ch = tid + i * kThreads;
        if (i + 1 < G::kLoads || ch < G::kChunks) *reinterpret_cast<f32x4 *>(dst + ch * 16) = stg[i];
      }
    }
    __syncthreads();
  }

  if (!MATERIALIZE && qvalid) a.cnt[seg] = mycnt;  // every segment is written: no memset needed
}

template <int DP, bool MAT, bool GLDS>
static int launch_scan_variant(const ScanArgs &a, hipStream_t stream) {
  using G = ScanGeom<DP>;
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&scan_kernel<DP, MAT, GLDS>), G::kLdsBytes));
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  hipLaunchKernelGGL((scan_kernel<DP, MAT, GLDS>), grid, dim3(kThreads), G::kLdsBytes, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

static int env_int(const char *name, int dflt) {
  const char *v = option(name);
  return (v && *v) ? atoi(v) : dflt;
}

template <int DP>
static int launch_scan_dp(const ScanArgs &a, bool materialize, hipStream_t stream) {
  const bool glds = env_int("TFRS_SCAN_GLDS", 1) != 0;
  if (materialize) {
    return glds ? launch_scan_variant<DP, true, true>(a, stream)
                : launch_scan_variant<DP, true, false>(a, stream);
  }
  return glds ? launch_scan_variant<DP, false, true>(a, stream)
              : launch_scan_variant<DP, false, false>(a, stream);
}

int launch_scan(const ScanArgs &a, bool materialize, hipStream_t stream) {
  if (a.nq <= 0 || a.c_end <= a.c_begin) return TFRS_OK;
  TFRS_CHECK_ARG(a.c_begin % kTileN == 0 && a.split_len % kTileN == 0,
                 "scan: c_begin/split_len must be multiples of %d", kTileN);
  TFRS_CHECK_ARG(materialize || a.nseg == 2 * a.n_splits, "scan: nseg must be 2 * n_splits");
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
