// topk_scan16.hip -- fp16 PREFILTER scan: S~ = fp16(Q) x fp16(Cand)^T on the 16-bit matrix
// cores (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate) with the top-K filter fused
// behind it.  Nothing computed here is ever returned: a prefilter score only decides whether
// a candidate can still reach a query's top-K, under the rigorous error bound of common.h
//   |s~ - s| <= ||q|| * ||c|| * kappa (+tiny),
// and every survivor is later re-scored with the exact f32 fma chain (select kernel).
// Same role in BruteForce.call (layers/factorized_top_k.py:603-605) as topk_scan.hip.
//
// Work decomposition (one launch covers a list of 128-row stages: stage0 + i * stride):
//   grid      = n_splits x n_qtiles workgroups of 512 threads (8 waves), XCD-aware remap so
//               that the workgroups resident on one XCD stream the same candidate split.
//   workgroup = 512 queries x one split of the stage list; stages (18 KiB at D=64) are
//               double-buffered in LDS by global_load_lds and shared by the 8 waves.
//   wave      = 2 groups of 32 queries, resident as MFMA B operands (DP/16 x 4 VGPRs each);
//               per 32-candidate sub-tile: DP/16 ds_read_b128 (A operand, shared by both
//               groups) and 2 x DP/16 MFMAs.  A = candidates, B = queries, so a lane's 16
//               accumulator registers belong to ONE query: the per-tile test is a
//               v_max3 tree (8 VALU) + one compare per 16 scores.
// Modes:
//   BINMAX      (threshold pass over a SAMPLE of the stages, no thresholds, no branches): the
//               maximum prefilter score per (query, bin_stages stages, lane half) -> binmax.
//               Bin maxima belong to distinct candidates, so the K-th largest of them minus
//               eps is a proven lower bound of the query's final K-th score.
//   FILTER      keep s~ > lower[q] - qk[q] * norm[stage] - tiny, appended by the owning lane
//               to its private segment of the query's survivor list (no atomics).
//   MATERIALIZE raw prefilter scores (test hook).
// The image holds x / scale[stage] and the queries q / qscale[q] (powers of two, so fp16's
// narrow exponent range is never the limit): the scales are folded into the threshold once
// per stage, the MFMA results are compared as they are.
//
// Roofline: fp16/bf16 MFMA (2.5 PFLOP/s dense): 2*DP flop per score.
// Built with -fno-honor-nans (build.py): fmaxf trees become bare v_max3_f32.
#include <stdlib.h>

#include <string.h>

#include "common.h"

// Timing ablations of scan16f_kernel (tools/ab_variants.sh builds one library per value; results are WRONG with
// any of them set -- they answer "what does this part of the loop cost"): 1 = no per-stage s_barrier,
// 2 = survivors ignored (no queue writes), 4 = A fragments read once per stage (no ds_reads in the sub-tile loop),
// 8 = stage copies skipped after the first two (no L2 / HBM traffic), 16 = no max tree / compare at all.
#ifndef TFRS_SCAN16_ABLATE
#define TFRS_SCAN16_ABLATE 0
#endif
#if TFRS_SCAN16_ABLATE
// An ablation build computes WRONG results by design: it needs -DTFRS_ALLOW_ABLATION next to -DTFRS_SCAN16_ABLATE=..., and the
// marker symbol below makes recommenders_amd/_lib.py refuse the library unless TFRS_ALLOW_ABLATION=1 is set.
#ifndef TFRS_ALLOW_ABLATION
#error "TFRS_SCAN16_ABLATE != 0 is a measurement build with wrong results: add -DTFRS_ALLOW_ABLATION to confirm"
#endif
extern "C" int tfrs_ablation_build_scan16(void) { return TFRS_SCAN16_ABLATE; }
#endif
// The queue append of check() with the common case peeled out of the loop (every hot lane fits: one ballot, no
// per-lane capacity compare): 0 = the loop only, 1 = peeled, 2 (default) = peeled + the MFMA -> VALU wait states in
// front of the max tree written by hand.  Value 1 is kept because it reproduces a compiler fault: with the append as a
// short side branch, the hazard recognizer of this ROCm's LLVM emits 1-6 wait states where gfx950 needs 12 between a
// v_mfma_f32_32x32x16_f16 and the first VALU read of its result, the max tree reads a stale accumulator now and then
// and survivors of the wave's last query group are lost (profiles/r05_scan16f_peel.txt;
// tools/check_mfma_hazards.py finds it on the assembly, tests/test_host_cpu.py runs that over every MFMA kernel).
#ifndef TFRS_SCAN16_PEEL
#define TFRS_SCAN16_PEEL 2
#endif

namespace tfrs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves16 = 8;
constexpr int kThreads16 = kWaves16 * 64;
constexpr int kQG = 2;  // query groups (of 32) per wave
static_assert(kWaves16 * kQG * 32 == kScan16QueriesPerWg, "queries per workgroup");

enum Scan16Mode { kModeFilter = 0, kModeMaterialize = 1, kModeBinMax = 2 };

template <int DP>
struct Scan16Geom {
  static constexpr int kSteps = DP / 16;
  static constexpr int kRowB = DP * 2 + 16;
  static constexpr int kStageB = kTileN * kRowB;
  static constexpr int kChunks = kStageB / 16;
  static constexpr int kLoads = (kChunks + kThreads16 - 1) / kThreads16;
  static constexpr int kMetaOff = 2 * kStageB;      // two 16-byte StageMeta slots after the tiles
  static constexpr int kLdsBytes = 2 * kStageB + 32;
};


// Direct-to-LDS copy of 16 bytes per lane (LDS address = wave base + lane * 16).  Issued from
// inline assembly on purpose: for the builtin the compiler cannot prove that the copy into one
// stage buffer is independent of the ds_reads of the other and of neighbouring loads, and
// drains it with s_waitcnt vmcnt(0) right after issue -- the prefetch then never overlaps the
// MFMAs of the current stage.  Completion is tracked by hand: `wait_dma()` before the barrier
// that ends a stage.
// NT: the copy carries the non-temporal hint -- for launches with ONE query tile, where a stage is read once, by one
// workgroup (Streaming's block-fed scans, topk_raw.hip raw_glds_copy16, have the measurements behind the hint).  Here:
// BruteForce over 12.5 M x 128 / 25 M x 64, 1 and 64 queries 0.718 / 0.774 and 0.700 / 0.721 -> 0.695 / 0.746 and 0.667 /
// 0.695 ms (-3.3 ... -4.7 %; 256 / 512 queries -0.5 %); with several query tiles the later ones read the stage from L2 and the
// hint costs 3 % (8192 queries), so it is a template parameter chosen by the launchers (TFRS_SCAN16_NT=0: never).
template <bool NT = false>
__device__ __forceinline__ void glds_copy16(const char *gsrc_lane, const char *lds_wave_base) {
  const uint32_t m0v = (uint32_t)(uintptr_t)(
      __attribute__((address_space(3))) const char *)lds_wave_base;
  // M0 (the LDS base of the copy) is a reserved register the compiler does not track through
  // inline assembly: it is saved and restored around the instruction.
  uint32_t m0_saved;
  if constexpr (NT) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved)
                 : "v"(gsrc_lane), "s"(m0v)
                 : "memory");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved)
                 : "v"(gsrc_lane), "s"(m0v)
                 : "memory");
  }
}
__device__ __forceinline__ void wait_dma() {   // s_waitcnt vmcnt(0), other counters untouched
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8) | (0 << 14));
}

// Linear stage copy HBM/L2 -> LDS (the fp16 image already has its LDS layout): every
// wave-instruction moves 1 KiB; the stage's 16-byte StageMeta rides along into its LDS slot.
template <int CHUNKS, int LOADS, bool NT = false>
__device__ __forceinline__ void stage16_glds(const char *gsrc, char *lds_dst, const StageMeta *meta,
                                             char *lds_meta, int tid, int wave) {
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int ch0 = i * kThreads16 + wave * 64;
    if (ch0 < CHUNKS)  // wave-uniform (CHUNKS is a multiple of 64)
      glds_copy16<NT>(gsrc + (size_t)(i * kThreads16 + tid) * 16, lds_dst + ch0 * 16);
  }
  if (tid == 0) glds_copy16(reinterpret_cast<const char *>(meta), lds_meta);
}

__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) {
  union {
    u32x4 u;
    f16x8 b;
  } x;
  x.u = v;
  return x.b;
}
__device__ __forceinline__ uint32_t cvt_f16x2(float lo, float hi) {
  union {
    f16x2 h;
    uint32_t u;
  } v;
  v.h[0] = (_Float16)lo;  // v_cvt_pk_f16_f32 (round to nearest even)
  v.h[1] = (_Float16)hi;
  return v.u;
}
__device__ __forceinline__ float mx3(float a, float b, float c) {
  return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}
__device__ __forceinline__ float max16(const f32x16 &c) {
  const float a = mx3(c[0], c[1], c[2]), b = mx3(c[3], c[4], c[5]), d = mx3(c[6], c[7], c[8]),
              e = mx3(c[9], c[10], c[11]), f = mx3(c[12], c[13], c[14]);
  return __builtin_fmaxf(mx3(a, b, d), mx3(e, f, c[15]));
}

template <int DP, int MODE, bool NT = false>
__global__ void __launch_bounds__(kThreads16, DP <= 64 ? 4 : 2) scan16_kernel(const Scan16Args a) {
  using G = Scan16Geom<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (MODE == kModeFilter && a.zero_word && blockIdx.x == 0 && tid == 0) *a.zero_word = 0u;
  if (MODE == kModeFilter && a.zero_aux && blockIdx.x == 0 && tid < 4) a.zero_aux[tid] = 0u;
  const int j = lane & 31;  // query column of this lane
  const int h = lane >> 5;  // k half of the MFMA step / upper candidate half of the C layout

  // ---- XCD-aware workgroup remap (bijective for any grid size) -----------------
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = bid & 7, pos = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
  const int split = logical / a.n_qtiles;
  const int qt = logical - split * a.n_qtiles;

  // this split's slice of the stage list
  const int i0 = split * a.stages_per_split;
  int i1 = i0 + a.stages_per_split;
  if (i1 > a.n_stages) i1 = a.n_stages;
  if (i0 >= i1) return;
  const int nst = i1 - i0;
  const int64_t stride = a.stage_stride;
  const int64_t first_stage = a.stage0 + (int64_t)i0 * stride;

  // ---- this wave's 2 x 32 queries -> fp16 MFMA B operands (resident) -------------
  f16x8 bq[kQG][G::kSteps];
  float lower[kQG], qk[kQG], qs[kQG], qinv[kQG];
  bool qvalid[kQG];
  int64_t qrow[kQG];
  uint2 *wp[kQG];  // FILTER: next free slot of this lane's segment (stride nseg entries)
  uint32_t mycnt[kQG];
  // Every load of this prologue is UNCONDITIONAL (a padding lane re-reads the last query and drops it) and all of them
  // are issued before the first conversion: loads under `if (qvalid)` / inside the per-step `if (vec_ok)` left the
  // number of loads in flight unknown to the compiler, which waited for each pair -- eight serial round trips per wave
  // where one does (the ISA of this prologue had 34 vmcnt(0) waits).
  const bool vec_ok = (a.d == DP) && ((reinterpret_cast<uintptr_t>(a.q) & 15) == 0);  // uniform
  float4 qlo[kQG][G::kSteps], qhi[kQG][G::kSteps];
  int64_t qclampv[kQG];
#pragma unroll
  for (int g = 0; g < kQG; ++g) {
    qrow[g] = (int64_t)qt * kScan16QueriesPerWg + wave * (kQG * 32) + g * 32 + j;
    qvalid[g] = qrow[g] < a.nq;
    qclampv[g] = qvalid[g] ? qrow[g] : a.nq - 1;
    qs[g] = a.qscale[qclampv[g]];
    if (vec_ok) {
      const float *qp = a.q + qclampv[g] * a.d;
#pragma unroll
      for (int m = 0; m < G::kSteps; ++m) {  // 8 consecutive features = two 16-byte loads
        qlo[g][m] = *reinterpret_cast<const float4 *>(qp + 16 * m + 8 * h);
        qhi[g][m] = *reinterpret_cast<const float4 *>(qp + 16 * m + 8 * h + 4);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < kQG; ++g) {
    const int64_t qclamp = qclampv[g];
    const float *qp = a.q + qclamp * a.d;
    if (!qvalid[g]) qs[g] = 1.0f;
    qinv[g] = 1.0f / qs[g];  // exact: power of two
#pragma unroll
    for (int m = 0; m < G::kSteps; ++m) {
      float x[8];
      if (vec_ok) {
        float4 lo = qlo[g][m], hi = qhi[g][m];
        if (!qvalid[g]) lo = hi = make_float4(0.f, 0.f, 0.f, 0.f);
        x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
        x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = 16 * m + 8 * h + i;
          x[i] = (qvalid[g] && k < a.d) ? qp[k] : 0.0f;
        }
      }
      u32x4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = cvt_f16x2(x[2 * i] * qinv[g], x[2 * i + 1] * qinv[g]);
      bq[g][m] = as_f16x8(w);
    }
    lower[g] = __builtin_inff();
    qk[g] = 0.0f;
    if (MODE == kModeFilter) {
      const float lw = a.lower[qclamp], kq = a.qk[qclamp];
      lower[g] = qvalid[g] ? lw : __builtin_inff();
      qk[g] = qvalid[g] ? kq : 0.0f;
    }
    // survivor lists are entry-major: buf[(q * cap_l + e) * nseg + seg], seg = 2 * split + h
    wp[g] = (MODE == kModeFilter)
                ? a.buf + (qrow[g] * (int64_t)a.cap_l) * a.nseg + (2 * split + h)
                : nullptr;
    mycnt[g] = 0;
  }
  const uint32_t row_limit = (uint32_t)a.row_limit;
  const bool wave_active =
      (int64_t)qt * kScan16QueriesPerWg + wave * (kQG * 32) < a.nq;   // wave-uniform

  // ---- stage 0 -> LDS ------------------------------------------------------------
  static_assert(G::kChunks % 64 == 0, "stage size must be a whole number of wave copies");
  const char *gsrc = a.packed16 + first_stage * (int64_t)G::kStageB;
  const int64_t gstep = stride * (int64_t)G::kStageB;
  const StageMeta *mp = a.meta + first_stage;
  stage16_glds<G::kChunks, G::kLoads, NT>(gsrc, smem, mp, smem + G::kMetaOff, tid, wave);
  wait_dma();
  __syncthreads();

  float binmax[kQG] = {-__builtin_inff(), -__builtin_inff()};
  for (int st = 0; st < nst; ++st) {
    const char *tile = smem + (st & 1) * G::kStageB;
    const bool more = (st + 1 < nst);
    if (more) {  // prefetch the next stage into the other buffer (its readers passed the barrier)
      stage16_glds<G::kChunks, G::kLoads, NT>(gsrc + (int64_t)(st + 1) * gstep,
                                          smem + ((st + 1) & 1) * G::kStageB,
                                          mp + (int64_t)(st + 1) * stride,
                                          smem + G::kMetaOff + ((st + 1) & 1) * 16, tid, wave);
    }
    const StageMeta sm = *reinterpret_cast<const StageMeta *>(smem + G::kMetaOff + (st & 1) * 16);
    // MFMA result = true prefilter score / (qscale * stage scale): compare against the
    // threshold divided by the same powers of two; `unscale` restores survivors' scores.
    float thr[kQG], unscale[kQG];
#pragma unroll
    for (int g = 0; g < kQG; ++g) {
      thr[g] = (__builtin_fmaf(-qk[g], sm.norm, lower[g]) - kF16Tiny) * qinv[g] * sm.inv_scale;
      unscale[g] = qs[g] * sm.scale;
    }
    // BINMAX: one bin = bin_stages consecutive stages of the list x lane half h
    if (MODE == kModeBinMax && (st % a.bin_stages) == 0) {
#pragma unroll
      for (int g = 0; g < kQG; ++g) binmax[g] = -__builtin_inff();
    }

    float stagemax[kQG];
#pragma unroll
    for (int g = 0; g < kQG; ++g) stagemax[g] = -__builtin_inff();
    const uint32_t stage_row = (uint32_t)((first_stage + (int64_t)st * stride) * kTileN);
    const char *ap = tile + j * G::kRowB + h * 16;

    // Small batches: a wave whose 64 query slots are all padding only helps to stage the tiles
    // (it still issues its copies and meets the barriers) and skips the scoring, so the stage
    // rate of a B <= 64 call is set by the HBM stream, not by MFMAs on padding.
    if (!wave_active) goto stage_done;
    {
    // A fragments are double-buffered in registers: the ds_reads of sub-tile t+1 are issued
    // before the MFMAs of sub-tile t, so no MFMA waits on LDS latency inside a stage.
    u32x4 af[2][G::kSteps];
#pragma unroll
    for (int m = 0; m < G::kSteps; ++m) af[0][m] = *reinterpret_cast<const u32x4 *>(ap + m * 32);

    // The two query groups of a wave are skewed by half a step: while the MFMA chain of one
    // group's tile runs, the VALU work on the OTHER group's finished tile (max tree, threshold
    // compare) is issued under it -- chain(g1, s) || check(g0, s), chain(g0, s+1) || check(g1, s).
    // In straight order (both chains, then both checks) a wave issues nothing to the matrix
    // pipe while it reduces its tiles, and the waves of a SIMD fall into that rhythm together.
    f32x16 acc[kQG];
    // acc[g][r] = s~(query g*32 + j, candidate stage_row + sub*32 + (r&3) + 8*(r>>2) + 4*h)
    auto chain = [&](int g, int sub) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
#pragma unroll
      for (int m = 0; m < G::kSteps; ++m)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(af[sub & 1][m]), bq[g][m], acc[g], 0, 0, 0);
    };
    auto check = [&](int g, int sub) __attribute__((always_inline)) {
      const f32x16 &c = acc[g];
      if (MODE == kModeMaterialize) {
        if (qvalid[g]) {
          float *drow = a.dense + qrow[g] * a.ld_dense +
                        ((int64_t)(i0 + st) * kTileN + sub * 32 + 4 * h);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<float4 *>(drow + 8 * g4) =
                make_float4(c[4 * g4 + 0] * unscale[g], c[4 * g4 + 1] * unscale[g],
                            c[4 * g4 + 2] * unscale[g], c[4 * g4 + 3] * unscale[g]);
        }
        return;
      }
      const float m0 = max16(c);
      if (MODE == kModeBinMax) {
        stagemax[g] = __builtin_fmaxf(stagemax[g], m0);
        // reduce NOW (the optimiser would otherwise sink the max trees to the end of the
        // stage and keep all accumulator tiles alive -> scratch spills)
        asm volatile("" : "+v"(stagemax[g]));
        return;
      }
      if (__ballot(m0 > thr[g]) != 0ull) {  // some lane of the wave has a survivor in this tile
        const uint32_t rbase = stage_row + sub * 32 + 4u * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float gm = __builtin_fmaxf(mx3(c[4 * g4], c[4 * g4 + 1], c[4 * g4 + 2]), c[4 * g4 + 3]);
          if (__ballot(gm > thr[g]) == 0ull) continue;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float v = c[4 * g4 + rr];
            const uint32_t row = rbase + rr + 8 * g4;
            if (v > thr[g] && row < row_limit) {
              if (mycnt[g] < a.cap_l) *wp[g] = make_uint2(__float_as_uint(v * unscale[g]), row);
              wp[g] += a.nseg;
              ++mycnt[g];
            }
          }
        }
      }
    };
    // one MFMA, then a few VALU instructions under it (the max tree is 8, the compare 2)
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < G::kSteps; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
    };

    chain(0, 0);
#pragma unroll
    for (int sub = 0; sub < kTileN / 32; ++sub) {
      if (sub + 1 < kTileN / 32) {
#pragma unroll
        for (int m = 0; m < G::kSteps; ++m)
          af[(sub + 1) & 1][m] =
              *reinterpret_cast<const u32x4 *>(ap + (sub + 1) * 32 * G::kRowB + m * 32);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this sub-tile's MFMAs
      chain(1, sub);
      check(0, sub);
      if (MODE != kModeMaterialize) interleave();
      if (sub + 1 < kTileN / 32) {
        chain(0, sub + 1);
        check(1, sub);
        if (MODE != kModeMaterialize) interleave();
      } else {
        check(1, sub);
      }
    }
    }
  stage_done:
    if (MODE == kModeBinMax && wave_active) {
      // scores of different stages are compared in true units
#pragma unroll
      for (int g = 0; g < kQG; ++g) binmax[g] = __builtin_fmaxf(binmax[g], stagemax[g] * unscale[g]);
      if ((st % a.bin_stages) == a.bin_stages - 1 || st == nst - 1) {
#pragma unroll
        for (int g = 0; g < kQG; ++g)
          if (qvalid[g])
            a.binmax[qrow[g] * a.ld_binmax + 2 * ((i0 + st) / a.bin_stages) + h] = binmax[g];
      }
    }
    wait_dma();      // this wave's share of the next stage has landed ...
    __syncthreads();  // ... and so has everybody else's; the current buffer is free again
  }

  if (MODE == kModeFilter) {
#pragma unroll
    for (int g = 0; g < kQG; ++g)
      if (qvalid[g]) a.cnt[qrow[g] * a.nseg + 2 * split + h] = mycnt[g];  // every segment is written
  }
}


// ---------------------------------------------------------------------------------------------
// FILTER pass, second generation (default; TFRS_SCAN16_V=1 selects the kernel above).
//
// Same decomposition, operand layout, skewed two-group schedule and filter rule as
// scan16_kernel<DP, FILTER>; what changed is everything AROUND the matrix stream:
//   * survivors leave the scoring loop through a per-wave LDS queue ("dump & drain"): a lane
//     whose 16-score column holds a survivor appends its 16 accumulators + {threshold, unscale,
//     row base, source lane} (80 bytes, five ds_write_b128, no search, no global store) and
//     the wave goes straight back to its MFMAs.  At the end of the stage the queue is drained
//     with all 64 lanes working: lane L tests score L%16 of entry L/16, survivors take a slot
//     from the per-(query, lane half) counter (LDS atomic) and are stored 4 entries per pass.
//     The first-generation kernel searched the 32x32 tile with one lane active (ballot ladder
//     + element loop, ~54 instructions and 6 taken branches per survivor) and issued one
//     8-byte store per survivor: measured +147 us (search) and +118 us (stores) per batch.
//   * no global store, scratch access or pointer bookkeeping is left in the scoring loop, so the
//     only VMEM operations in flight there are the stage copies: the per-stage
//     `s_waitcnt vmcnt(0)` waits for copies issued a whole stage earlier (the first-generation
//     kernel spilled 7 VGPRs, and every scratch reload forced a vmcnt(0) right behind the
//     prefetch it had just issued).  Per-query write offsets / counts live in LDS.
// A tile that does not fit into the queue any more (adversarial data: near-duplicate clusters)
// takes a direct per-element path with the same counters; nothing is ever dropped silently (counts
// beyond cap_l flag the query for the exact redo exactly as before).
template <int DP, int NW, int QG, int SPB = 1>
struct Scan16FGeom : Scan16Geom<DP> {
  using B = Scan16Geom<DP>;
  // TILES query tiles of kScan16QueriesPerWg queries share one workgroup -- and with it ONE copy of every stage
  // (<16, 2>: the two tiles that otherwise sit on a CU as two workgroups, each copying the stage from L2)
  static constexpr int kTiles = NW * QG * 32 / kScan16QueriesPerWg;
  static constexpr int kWavesPerTile = NW / kTiles;
  static_assert(kTiles >= 1 && kTiles * kScan16QueriesPerWg == NW * QG * 32, "queries per workgroup");
  static constexpr int kThreads = NW * 64;
  static constexpr int kLoadsF = (B::kChunks + kThreads - 1) / kThreads;
  static constexpr int kQCap = 32;                       // queue entries per wave
  static constexpr int kEntB = 80;                       // 16 scores + 16-byte header
  // queue + QG * 64 segment counters + QG * 64 x {flo, fqk, qscale, segment base} filter constants
  static constexpr int kWaveB = kQCap * kEntB + QG * 64 * 4 + QG * 64 * 16;
  // SPB stages per barrier period, double-buffered: 2 * SPB stage slots + as many 16-byte StageMeta slots
  static constexpr int kSlots = 2 * SPB;
  static constexpr int kMetaOffF = kSlots * B::kStageB;
  static constexpr int kQueueOff = (kMetaOffF + kSlots * 16 + 15) / 16 * 16;
  static constexpr int kLdsBytesF = kQueueOff + NW * kWaveB;
  static_assert(kLdsBytesF <= 160 * 1024, "LDS budget");
};

// stage copy with NW waves (see stage16_glds)
template <int CHUNKS, int LOADS, int THREADS, bool NT = false>
__device__ __forceinline__ void stage16f_glds(const char *gsrc, char *lds_dst, const StageMeta *meta,
                                              char *lds_meta, int tid, int wave) {
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int ch0 = i * THREADS + wave * 64;
    if (ch0 < CHUNKS)  // wave-uniform (CHUNKS is a multiple of 64)
      glds_copy16<NT>(gsrc + (size_t)(i * THREADS + tid) * 16, lds_dst + ch0 * 16);
  }
  if (tid == 0) glds_copy16(reinterpret_cast<const char *>(meta), lds_meta);
}

__device__ __forceinline__ uint32_t lds_atomic_inc(uint32_t *p) {
  return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}


// NW waves x QG query groups of 32 per wave (NW * QG * 32 = 512 queries per workgroup):
//   <8, 2>  four waves per SIMD at <= 128 VGPRs (two workgroups per CU)
//   <4, 4>  two waves per SIMD at <= 256 VGPRs: one set of A fragments (4 ds_read_b128) feeds 16
//           MFMAs instead of 8 -- half the LDS reads per flop, half the waves per barrier
//   <16, 2> one workgroup of 16 waves per CU = two query tiles on one stage buffer: half the L2 -> LDS copies
// SPB = stages per barrier period (the LDS a <16, 2> workgroup saves by sharing its stage buffers pays for two)
// (Tried and dropped, round 5: one skewed sub-tile pipeline ACROSS the two stages of a period -- next stage's first A
// fragments fetched under the current stage's last chains, stage constants switched per group at the boundary:
// 0.931 ms against 0.895 for the plain two-stage period on the same box, profiles/r05_scan16f_shapes.txt.)
template <int DP, int NW, int QG, int SPB = 1, bool NT = false>
__global__ void __launch_bounds__(NW * 64, NW * QG >= 32 ? 1 : (DP <= 64 ? 2 : 1) * NW / 4) scan16f_kernel(const Scan16Args a) {
  using G = Scan16FGeom<DP, NW, QG, SPB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.zero_word && blockIdx.x == 0 && tid == 0) *a.zero_word = 0u;
  if (a.zero_aux && blockIdx.x == 0 && tid < 4) a.zero_aux[tid] = 0u;
  const int j = lane & 31;
  const int h = lane >> 5;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = bid & 7, pos = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
  const int n_qgroups = (a.n_qtiles + G::kTiles - 1) / G::kTiles;      // workgroups per split
  const int split = logical / n_qgroups;
  // this wave's query tile (a tile beyond n_qtiles starts at a query >= nq: its waves only help to copy)
  const int qt = (logical - split * n_qgroups) * G::kTiles + wave / G::kWavesPerTile;
  const int wave_in_tile = wave % G::kWavesPerTile;

  const int i0 = split * a.stages_per_split;
  int i1 = i0 + a.stages_per_split;
  if (i1 > a.n_stages) i1 = a.n_stages;
  if (i0 >= i1) return;
  const int nst = i1 - i0;
  const int64_t first_stage = a.stage0 + (int64_t)i0 * a.stage_stride;

  // per-wave LDS: survivor queue + the 128 (group, lane) segment counters
  char *const qbase = smem + G::kQueueOff + wave * G::kWaveB;
  uint32_t *const wcnt = reinterpret_cast<uint32_t *>(qbase + G::kQCap * G::kEntB);
  #pragma unroll
  for (int g = 0; g < QG; ++g) wcnt[g * 64 + lane] = 0u;
  const int64_t q0 = (int64_t)qt * kScan16QueriesPerWg + wave_in_tile * (QG * 32);   // wave's first query

  // ---- this wave's 2 x 32 queries -> fp16 MFMA B operands (resident) -------------
  f16x8 bq[QG][G::kSteps];
  // per (group, lane): {(lower - tiny) / qscale, qk / qscale, qscale} parked in LDS (re-read
  // once per stage: three more resident VGPR pairs do not fit under 128)
  float4 *const qconst = reinterpret_cast<float4 *>(qbase + G::kQCap * G::kEntB + QG * 64 * 4);
  // all loads of the prologue unconditional and issued before the first conversion: see scan16_kernel
  const bool vec_ok = (a.d == DP) && ((reinterpret_cast<uintptr_t>(a.q) & 15) == 0);  // uniform
  float4 qlo[QG][G::kSteps], qhi[QG][G::kSteps];
  float qscv[QG], lowv[QG], qkv[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    const int64_t qrow = q0 + g * 32 + j;
    const int64_t qclamp = qrow < a.nq ? qrow : a.nq - 1;
    qscv[g] = a.qscale[qclamp];
    lowv[g] = a.lower[qclamp];
    qkv[g] = a.qk[qclamp];
    if (vec_ok) {
      const float *qp = a.q + qclamp * a.d;
#pragma unroll
      for (int m = 0; m < G::kSteps; ++m) {
        qlo[g][m] = *reinterpret_cast<const float4 *>(qp + 16 * m + 8 * h);
        qhi[g][m] = *reinterpret_cast<const float4 *>(qp + 16 * m + 8 * h + 4);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    const int64_t qrow = q0 + g * 32 + j;
    const bool qvalid = qrow < a.nq;
    const int64_t qclamp = qvalid ? qrow : a.nq - 1;
    const float *qp = a.q + qclamp * a.d;
    const float qsc = qvalid ? qscv[g] : 1.0f;
    const float lower_q = lowv[g], qk_q = qkv[g];
    const float qinv = 1.0f / qsc;  // exact: power of two
#pragma unroll
    for (int m = 0; m < G::kSteps; ++m) {
      float x[8];
      if (vec_ok) {
        float4 lo = qlo[g][m], hi = qhi[g][m];
        if (!qvalid) lo = hi = make_float4(0.f, 0.f, 0.f, 0.f);
        x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
        x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = 16 * m + 8 * h + i;
          x[i] = (qvalid && k < a.d) ? qp[k] : 0.0f;
        }
      }
      u32x4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = cvt_f16x2(x[2 * i] * qinv, x[2 * i + 1] * qinv);
      bq[g][m] = as_f16x8(w);
    }
    // thr = ((lower - qk * norm) - tiny) / (qscale * stage scale); divisions by powers of two
    // commute with the roundings, so this is the first-generation threshold up to the order of
    // the two subtractions (tiny is far below one ulp of lower unless lower is ~0)
    // .w: entry index of this (query, lane half, split) segment in the survivor buffer (the
    // launcher guarantees that the whole buffer is indexable with 32 bits)
    const uint32_t seg_base = (uint32_t)((qrow * (int64_t)a.cap_l) * a.nseg + (2 * split + h));
    qconst[g * 64 + lane] = make_float4(qvalid ? (lower_q - kF16Tiny) * qinv : __builtin_inff(),
                                        qvalid ? qk_q * qinv : 0.0f, qsc,
                                        __uint_as_float(seg_base));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t row_limit = (uint32_t)a.row_limit;
  const bool wave_active = q0 < a.nq;   // wave-uniform
  int qtail = 0;                        // wave-uniform: entries in the queue

  // drain: four lanes per entry, lane L tests scores 4 * (L % 4) .. + 3 of entry p0 + L / 4, so up
  // to 16 entries cost one pass (two dependent LDS round trips); a column rarely holds more than
  // one survivor, so the per-lane loop over its hits usually runs once
  auto drain = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t s4 = (uint32_t)(lane & 3);
    for (int p0 = 0; p0 < qtail; p0 += 16) {
      const int e = p0 + (lane >> 2);
      if (e < qtail) {
        const char *ep = qbase + e * G::kEntB;
        const float4 v4 = *reinterpret_cast<const float4 *>(ep + 16 * s4);
        const uint4 hd = *reinterpret_cast<const uint4 *>(ep + 64);
        const float thr_e = __uint_as_float(hd.x);
        const uint32_t row0 = hd.z + 8u * s4;   // accumulator register 4 s + i holds row 8 s + i (+ 4 h)
        uint32_t hits = (v4.x > thr_e ? 1u : 0u) | (v4.y > thr_e ? 2u : 0u) |
                        (v4.z > thr_e ? 4u : 0u) | (v4.w > thr_e ? 8u : 0u);
        if (row0 + 3u >= row_limit) {   // the index's last, partly filled stage
#pragma unroll
          for (uint32_t i = 0; i < 4u; ++i)
            if (row0 + i >= row_limit) hits &= ~(1u << i);
        }
        if (hits) {
          const uint32_t src = hd.w;   // g * 64 + source lane
          const float2 zc = *reinterpret_cast<const float2 *>(&qconst[src].z);
          const float un = zc.x * __uint_as_float(hd.y);   // qscale * stage scale
          do {
            const uint32_t i = (uint32_t)__builtin_ctz(hits);
            hits &= hits - 1u;
            const float v = i == 0u ? v4.x : i == 1u ? v4.y : i == 2u ? v4.z : v4.w;
            const uint32_t row = row0 + i;
            const uint32_t slot = lds_atomic_inc(&wcnt[src]);
            if (slot < a.cap_l) {
              a.buf[(uint64_t)(__float_as_uint(zc.y) + slot * (uint32_t)a.nseg)] =
                  make_uint2(__float_as_uint(v * un), row);
            } else if (a.ovf_cnt && slot - a.cap_l < kOvfPerSeg) {
              // the segment is full (rows ordered by cluster: a query's survivors sit in one or two
              // splits): per-query overflow list, one global atomic per such survivor
              const int64_t qrow = q0 + (src >> 6) * 32 + (src & 31);
              const uint32_t o = atomicAdd(&a.ovf_cnt[qrow], 1u);
              if (o < a.ovf_cap) a.ovf_buf[qrow * (int64_t)a.ovf_cap + o] = make_uint2(__float_as_uint(v * un), row);
            }
          } while (hits);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    qtail = 0;
  };

  // ---- the first period's stages -> LDS ----------------------------------------------
  static_assert(G::kChunks % 64 == 0, "stage size must be a whole number of wave copies");
  const char *gsrc = a.packed16 + first_stage * (int64_t)G::kStageB;
  const int64_t gstep = (int64_t)a.stage_stride * (int64_t)G::kStageB;
  const StageMeta *mp = a.meta + first_stage;
  // stage `st` of this split's list lives in slot (period & 1) * SPB + st % SPB
  auto copy_stage = [&](int st) __attribute__((always_inline)) {
    const int slot = ((st / SPB) & 1) * SPB + (st % SPB);
    stage16f_glds<G::kChunks, G::kLoadsF, G::kThreads, NT>(gsrc + (int64_t)st * gstep, smem + slot * G::kStageB,
                                                       mp + (int64_t)st * a.stage_stride,
                                                       smem + G::kMetaOffF + slot * 16, tid, wave);
  };
#pragma unroll
  for (int u = 0; u < SPB; ++u)
    if (u < nst) copy_stage(u);
  wait_dma();
  __syncthreads();

  const int n_periods = (nst + SPB - 1) / SPB;
  for (int per = 0; per < n_periods; ++per) {
    // prefetch the next period's stages into the other half of the slots (their readers passed the barrier)
    if (!((TFRS_SCAN16_ABLATE & 8) && per >= 1)) {
#pragma unroll
      for (int u = 0; u < SPB; ++u)
        if ((per + 1) * SPB + u < nst) copy_stage((per + 1) * SPB + u);
    }
#pragma unroll
    for (int u = 0; u < SPB; ++u) {
    const int st = per * SPB + u;
    if (SPB > 1 && st >= nst) break;
    const int slot = (per & 1) * SPB + u;
    const char *tile = smem + slot * G::kStageB;
    if (wave_active) {
    const StageMeta sm = *reinterpret_cast<const StageMeta *>(smem + G::kMetaOffF + slot * 16);
    // wave-uniform stage constants -> SGPRs
    const float s_norm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sm.norm)));
    const uint32_t s_scale_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(sm.scale));
    const float s_inv = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sm.inv_scale)));
    float thr[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const float4 qc = qconst[g * 64 + lane];
      thr[g] = __builtin_fmaf(-qc.y, s_norm, qc.x) * s_inv;
    }
    const uint32_t stage_row = (uint32_t)((first_stage + (int64_t)st * a.stage_stride) * kTileN);
    const char *ap = tile + j * G::kRowB + h * 16;

    u32x4 af[2][G::kSteps];
#pragma unroll
    for (int m = 0; m < G::kSteps; ++m) af[0][m] = *reinterpret_cast<const u32x4 *>(ap + m * 32);

    f32x16 acc[QG];
    if (TFRS_SCAN16_ABLATE & 32) __builtin_amdgcn_s_setprio(2);
    auto chain = [&](int g, int sub) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
#pragma unroll
      for (int m = 0; m < G::kSteps; ++m)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(af[sub & 1][m]), bq[g][m], acc[g], 0, 0, 0);
    };
    auto check = [&](int g, int sub) __attribute__((always_inline)) {
      // 10 wait states + the >= 2 instructions between the chain's last MFMA and this point in every instantiation
      // (checked on the assembly): the 12 the result needs before a VALU instruction may read it
      if (TFRS_SCAN16_PEEL == 2) asm volatile("s_nop 9" : "+v"(acc[g]));
      const f32x16 &c = acc[g];
      if (TFRS_SCAN16_ABLATE & 16) { asm volatile("" :: "v"(c[0]), "v"(c[15])); return; }
      const float m0 = max16(c);
      const bool hot = m0 > thr[g];
      const uint64_t hm = __ballot(hot);
      if (TFRS_SCAN16_ABLATE & 2) { asm volatile("" :: "s"(hm)); return; }
      if (__builtin_expect(hm != 0ull, 0)) {   // wave-uniform: some lane's 16-score column holds a survivor
        const uint32_t rbase = stage_row + sub * 32 + 4u * h;
        if (TFRS_SCAN16_PEEL) {
          const int nhot = __builtin_popcountll(hm);
          if (__builtin_expect(qtail + nhot <= G::kQCap, 1)) {   // wave-uniform: every hot lane fits
            if (hot) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
              char *e = qbase + (qtail + rank) * G::kEntB;
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4 *>(e + 16 * g4) =
                    make_float4(c[4 * g4], c[4 * g4 + 1], c[4 * g4 + 2], c[4 * g4 + 3]);
              *reinterpret_cast<uint4 *>(e + 64) =
                  make_uint4(__float_as_uint(thr[g]), s_scale_bits, rbase, (uint32_t)(g * 64 + lane));
            }
            qtail += nhot;
            return;
          }
        }
        uint64_t rem = hm;
        do {   // one round unless the queue is full (bursts of near-duplicates, identical queries)
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(rem >> 32),
                               __builtin_amdgcn_mbcnt_lo((uint32_t)rem, 0u));
          const bool mine = ((rem >> lane) & 1ull) != 0ull && qtail + rank < G::kQCap;
          if (mine) {
            char *e = qbase + (qtail + rank) * G::kEntB;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
              *reinterpret_cast<float4 *>(e + 16 * g4) =
                  make_float4(c[4 * g4], c[4 * g4 + 1], c[4 * g4 + 2], c[4 * g4 + 3]);
            *reinterpret_cast<uint4 *>(e + 64) =
                make_uint4(__float_as_uint(thr[g]), s_scale_bits, rbase, (uint32_t)(g * 64 + lane));
          }
          const uint64_t taken = __ballot(mine);
          qtail += __builtin_popcountll(taken);
          rem &= ~taken;
          if (__builtin_expect(rem != 0ull, 0)) drain();
        } while (__builtin_expect(rem != 0ull, 0));
      }
    };
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < G::kSteps; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
    };

    if constexpr (QG == 2) {
    // two groups, skewed by half a step: chain(g1, s) || check(g0, s), chain(g0, s + 1) || check(g1, s)
    chain(0, 0);
#pragma unroll
    for (int sub = 0; sub < kTileN / 32; ++sub) {
      if (sub + 1 < kTileN / 32) {
#pragma unroll
        for (int m = 0; m < G::kSteps; ++m)
          if (TFRS_SCAN16_ABLATE & 4) af[(sub + 1) & 1][m] = af[sub & 1][m];
          else
          af[(sub + 1) & 1][m] =
              *reinterpret_cast<const u32x4 *>(ap + (sub + 1) * 32 * G::kRowB + m * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
      chain(1, sub);
      check(0, sub);
      interleave();
      if (sub + 1 < kTileN / 32) {
        chain(0, sub + 1);
        check(1, sub);
        interleave();
      } else {
        check(1, sub);
      }
    }
    } else {
    // Software pipeline over the (sub-tile, group) pairs in order: the MFMA chain of pair p is
    // issued, then the VALU work on the finished tile of pair p - 1 runs under it.  The A
    // fragments of sub-tile s + 1 are fetched when the chains of sub-tile s start.
#pragma unroll
    for (int sub = 0; sub < kTileN / 32; ++sub) {
      if (sub + 1 < kTileN / 32) {
#pragma unroll
        for (int m = 0; m < G::kSteps; ++m)
          af[(sub + 1) & 1][m] =
              *reinterpret_cast<const u32x4 *>(ap + (sub + 1) * 32 * G::kRowB + m * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        chain(g, sub);
        if (g > 0) {
          check(g - 1, sub);
          interleave();
        } else if (sub > 0) {
          check(QG - 1, sub - 1);
          interleave();
        }
      }
    }
    check(QG - 1, kTileN / 32 - 1);
    }
    }
    }   // stages of the period
    // The stage copies were issued a whole period ago and the previous drain's stores before
    // them: nothing recent is outstanding here.
    if (TFRS_SCAN16_ABLATE & 32) __builtin_amdgcn_s_setprio(0);
    wait_dma();
    if (qtail > 0 && (qtail >= a.drain_min || per == n_periods - 1 || ((per + 1) % a.drain_every) == 0)) drain();
    if (!(TFRS_SCAN16_ABLATE & 1)) __builtin_amdgcn_s_barrier();
  }

  // every segment's count is written (counts beyond cap_l flag the query for the exact redo)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    const int64_t qrow = q0 + g * 32 + j;
    if (qrow < a.nq) a.cnt[qrow * a.nseg + 2 * split + h] = wcnt[g * 64 + lane];
  }
}


// one query tile: every stage is read once, by one workgroup -> the copies carry the non-temporal hint (glds_copy16)
static bool scan16_nt_copies(const Scan16Args &a) {
  const char *e = option("TFRS_SCAN16_NT");
  return a.n_qtiles == 1 && !(e && e[0] == '0');
}

template <int DP, int NW, int QG, int SPB = 1, bool NT = false>
static int launch_scan16f(const Scan16Args &a, hipStream_t stream) {
  using G = Scan16FGeom<DP, NW, QG, SPB>;
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&scan16f_kernel<DP, NW, QG, SPB, NT>), G::kLdsBytesF));
  const dim3 grid((unsigned)((a.n_qtiles + G::kTiles - 1) / G::kTiles * a.n_splits));
  hipLaunchKernelGGL((scan16f_kernel<DP, NW, QG, SPB, NT>), grid, dim3(NW * 64), G::kLdsBytesF, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// TFRS_SCAN16_SHAPE = 16x2s2 | 16x2 | 8x2 | 4x4 | 8x4.  Default (round 5): batches of at least two query tiles take
// the 16-wave workgroup (two tiles on one stage buffer), with two stages per barrier period where the LDS allows it
// (dims <= 64); one tile (<= 512 queries) keeps <8, 2>.  Same-box measurements: profiles/r05_scan16f_shapes.txt.
template <int DP>
static int launch_scan16f_shape(const Scan16Args &a, hipStream_t stream) {
  const char *e = option("TFRS_SCAN16_SHAPE");
  const bool pair = a.n_qtiles >= 2;
  auto is = [&](const char *v) { return e && strcmp(e, v) == 0; };   // whole-string matches only
  if (e && !(is("16x2s2") || is("16x2") || is("8x2") || is("4x4") || is("8x4"))) {
    set_error("TFRS_SCAN16_SHAPE=%s: expected one of 16x2s2, 16x2, 8x2, 4x4, 8x4", e);
    return TFRS_EINVAL;
  }
  if (is("8x2")) return launch_scan16f<DP, 8, 2>(a, stream);
  if constexpr (DP <= 64) {
    if (is("4x4")) return launch_scan16f<DP, 4, 4>(a, stream);
    if (is("8x4") && pair) return launch_scan16f<DP, 8, 4>(a, stream);
    if (is("16x2") && pair) return launch_scan16f<DP, 16, 2>(a, stream);
    if (pair) return launch_scan16f<DP, 16, 2, 2>(a, stream);
  }
  // (dim 128: the 16-wave instantiation needs more than 128 registers -- 15.0 ms against 3.4 -- and stays out)
  if (!e && scan16_nt_copies(a)) return launch_scan16f<DP, 8, 2, 1, true>(a, stream);
  return launch_scan16f<DP, 8, 2>(a, stream);
}

template <int DP, int MODE, bool NT = false>
static int launch_scan16_variant(const Scan16Args &a, hipStream_t stream) {
  using G = Scan16Geom<DP>;
  if constexpr (MODE == kModeBinMax && !NT) {   // (the threshold pass of a one-tile batch: the sampled stages are read once)
    if (scan16_nt_copies(a)) return launch_scan16_variant<DP, MODE, true>(a, stream);
  }
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&scan16_kernel<DP, MODE, NT>), G::kLdsBytes));
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  hipLaunchKernelGGL((scan16_kernel<DP, MODE, NT>), grid, dim3(kThreads16), G::kLdsBytes, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

template <int DP>
static int launch_scan16_dp(const Scan16Args &a, hipStream_t stream) {
  if (a.dense) return launch_scan16_variant<DP, kModeMaterialize>(a, stream);
  if (a.binmax) return launch_scan16_variant<DP, kModeBinMax>(a, stream);
  return launch_scan16_variant<DP, kModeFilter>(a, stream);
}

static int scan16_generation() {
  const char *e = option("TFRS_SCAN16_V");   // read per call: one process can compare both
  return (e && e[0] == '1') ? 1 : 2;
}

int launch_scan16(const Scan16Args &a_in, hipStream_t stream) {
  Scan16Args a = a_in;
  if (a.nq <= 0 || a.n_stages <= 0) return TFRS_OK;
  if (a.drain_min < 1) a.drain_min = 1;
  if (a.drain_every < 1) a.drain_every = 1;
  TFRS_CHECK_ARG(a.stage_stride >= 1 && a.stages_per_split >= 1 &&
                     (int64_t)a.n_splits * a.stages_per_split >= a.n_stages,
                 "scan16: bad stage split");
  TFRS_CHECK_ARG(a.dense || a.binmax || a.nseg == 2 * a.n_splits, "scan16: nseg must be 2 * n_splits");
  TFRS_CHECK_ARG(!a.binmax || (a.bin_stages >= 1 && a.stages_per_split % a.bin_stages == 0),
                 "scan16: stages_per_split must be a multiple of bin_stages");
  // (the second-generation kernel indexes the survivor buffer with 32 bits)
  const bool fits32 = (uint64_t)a.nq * a.cap_l * (uint64_t)a.nseg < (1ull << 32);
  if (!a.dense && !a.binmax && fits32 && scan16_generation() == 2) {
    switch (padded_dim16(a.d)) {
      case 16: return launch_scan16f_shape<16>(a, stream);
      case 32: return launch_scan16f_shape<32>(a, stream);
      case 64: return launch_scan16f_shape<64>(a, stream);
      case 128: return launch_scan16f_shape<128>(a, stream);
    }
  }
  switch (padded_dim16(a.d)) {
    case 16: return launch_scan16_dp<16>(a, stream);
    case 32: return launch_scan16_dp<32>(a, stream);
    case 64: return launch_scan16_dp<64>(a, stream);
    case 128: return launch_scan16_dp<128>(a, stream);
  }
  set_error("scan16: unsupported dim %d", a.d);
  return TFRS_ENOTIMPL;
}

}  // namespace tfrs
