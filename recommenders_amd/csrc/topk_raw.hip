// topk_raw.hip -- Streaming.call over candidate blocks that are read IN PLACE.
//
// The reference's Streaming layer keeps only a reference to its candidate dataset and re-reads it
// on every call (layers/factorized_top_k.py:384-390, :496-507): per block tf.matmul + tf.math.top_k
// (:424-438) and a concat + top_k fold (:440-472).  Here the blocks of a group of consecutive
// dataset elements (row-major f32 [rows, d] in HBM, described by a RawTable, common.h) are scored
// where they lie:
//
//   rawscan_kernel     the first rows of every stream (dense round) and up to 32 queries below dim 128:
//                      exact f32 scores on the f32 matrix cores (v_mfma_f32_32x32x2_f32 == the d-ordered
//                      fma chain of oracle/c/oracle_core.c) with the top-K filter fused behind them.  A
//                      candidate byte is read from HBM once and nothing is written but the survivors.
//   rawscan16_kernel   up to 256 queries (HBM-bound, 0.73 of the spec): the same one pass over the f32
//                      blocks, every wave converting its 32 rows of a stage to fp16 in registers and
//                      scoring them on the 16-bit matrix cores; a prefilter under the bound of common.h,
//                      survivors re-scored exactly from the blocks (list_topk16_kernel, raw_score).
//   pack16_raw_kernel  large query batches (MFMA-bound): the fp16 prefilter image of the group
//                      (topk_scan16.hip consumes it) is built straight from the blocks -- the f32
//                      packed image of the BruteForce index is never written; survivors are
//                      re-scored exactly from the blocks (raw_score, common.h).
//
// Layout of a raw stage in LDS.  A stage is 128 consecutive rows = DP/2 KiB; the copy is
// global_load_lds_dwordx4 (LDS address = wave-uniform base + lane * 16, so one instruction lands
// 1 KiB of consecutive rows contiguously: a "chunk" of 256 / DP rows).  Chunks are placed 1040 bytes
// apart (one 16-byte pad): consecutive chunks then start in consecutive 16-byte bank groups, and
// the lane -> row assignment of the MFMA A operand walks the CHUNKS first,
//     t = 32 * wave + j  ->  row (t % chunks) * rows_per_chunk + t / chunks,
// so the 16 lanes of a ds_read_b128 quarter hit 16 different bank groups (DP >= 32).  Any
// assignment is exact: a score's value does not depend on which MFMA row computes it, and the
// survivor carries its row number.
// A row keeps its natural feature order.  v_mfma_f32_32x32x2_f32 wants feature 2s from lanes
// 0-31 and 2s+1 from lanes 32-63 at step s: lane half h reads the 16-byte piece 2m + h
// (features 8m + 4h ..) and two v_permlane32_swap turn the four registers into the steps
// 4m .. 4m+3 = features (8m, 8m+1), (8m+2, 8m+3), (8m+4, 8m+5), (8m+6, 8m+7).
#include <stdlib.h>

#include "common.h"

namespace tfrs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRawWaves = 4;
constexpr int kRawThreads = kRawWaves * 64;
constexpr int kRawChunkB = 1040;   // 1 KiB of rows + 16 bytes of padding

// NWV waves per workgroup = 32 * NWV rows per stage (kTileN with the default four waves; rawscan16_kernel also runs as
// TWO waves on 64-row stages: two workgroups per CU at dim 128, see there)
template <int DP, int NWV = kRawWaves>
struct RawGeom {
  static constexpr int kTile = 32 * NWV;                      // rows per stage
  static constexpr int kRowB = DP * 4;
  static constexpr int kRowsPerChunk = 1024 / kRowB;          // 2 (DP = 128) .. 32 (DP = 8)
  static constexpr int kChunks = kTile / kRowsPerChunk;       // DP / 2 with four waves
  static constexpr int kCopies = kChunks / NWV;               // copy instructions per wave and stage
  static constexpr int kStageB = kChunks * kRawChunkB;
  static constexpr int kCntOff = 2 * kStageB;                 // uint32 [64] workgroup survivor counters
  static constexpr int kLdsBytes = 2 * kStageB + 64 * 4;
  static_assert(kChunks % NWV == 0 && kChunks >= NWV, "every wave issues the same number of stage copies");
};

// Direct-to-LDS copy of 16 bytes per lane, issued from inline assembly so that the prefetch of the
// next stage stays in flight under the MFMAs of the current one (see topk_scan16.hip); completion
// is awaited by hand (raw_wait_dma) before the barrier that ends a stage.
// NT: the copy carries the non-temporal hint.  The guide's LDS-DMA measurements (MI355X_MICROARCH.md, "nt-weights") put
// issued -> landed 18-19 % lower for a stream that ONE CU reads once, and these kernels hold ONE stage of prefetch per CU
// -- their rate is 64 KiB per memory latency -- so the hint is worth what it takes off that latency: 12.5 M x 128, one
// query tile (same-box A/B, profiles/r06_stream_nt.txt): 1 / 32 / 64 / 128 queries 1.32 / 1.37 / 1.43 / 1.57 -> 1.23 /
// 1.29 / 1.35 / 1.51-1.54 ms.  With TWO query tiles per split (129 .. 256 queries at dim 128) the second tile re-reads
// the rows from the XCD's L2 and the hint costs 3 %: the launchers take NT = (n_qtiles == 1).  A compile-time choice: as
// a run-time (wave-uniform) branch around the two instructions the SAME hint made 128 / 256 queries 5-11 % slower.
// TFRS_RAW_NT=0 (library switch): never.
template <bool NT>
__device__ __forceinline__ void raw_glds_copy16(const char *gsrc_lane, const char *lds_wave_base) {
  const uint32_t m0v = (uint32_t)(uintptr_t)(
      __attribute__((address_space(3))) const char *)lds_wave_base;
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
__device__ __forceinline__ void raw_wait_dma() {   // s_waitcnt vmcnt(0), other counters untouched
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8) | (0 << 14));
}

__device__ __forceinline__ float raw_max16(const f32x16 &c) {
  const float a = fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]));
  const float b = fmaxf(fmaxf(c[4], c[5]), fmaxf(c[6], c[7]));
  const float d = fmaxf(fmaxf(c[8], c[9]), fmaxf(c[10], c[11]));
  const float e = fmaxf(fmaxf(c[12], c[13]), fmaxf(c[14], c[15]));
  return fmaxf(fmaxf(a, b), fmaxf(d, e));
}
// The same as maxima of three (v_max3_f32; this file is built with -fno-honor-nans like topk_scan16.hip: without it
// fmaxf() of an MFMA result costs a canonicalising `v_max_f32 x, x` per input first -- 31 instructions for 16 values
// instead of 8.  A NaN score fails `> threshold` either way.  NOT inline assembly: the compiler's hazard recognizer
// does not look at the operands of an asm statement, and a v_max3 written that way read the accumulators right behind
// the chain's last MFMA -- tools/check_mfma_hazards.py found it on the listing.)
__device__ __forceinline__ float raw_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float raw_max8f(const f32x16 &c, int o) {   // max of c[o .. o + 7]
  return raw_max3(raw_max3(c[o], c[o + 1], c[o + 2]), raw_max3(c[o + 3], c[o + 4], c[o + 5]), fmaxf(c[o + 6], c[o + 7]));
}
__device__ __forceinline__ float raw_max16f(const f32x16 &c) {
  const float a = raw_max3(c[0], c[1], c[2]), b = raw_max3(c[3], c[4], c[5]), d = raw_max3(c[6], c[7], c[8]);
  const float e = raw_max3(c[9], c[10], c[11]), f = raw_max3(c[12], c[13], c[14]);
  return fmaxf(raw_max3(a, b, d), raw_max3(e, f, c[15]));
}

__global__ void raw_table_write_kernel(const RawTable t, RawTable *dst) {
  // (word-wise copy of the by-value argument; 3096 bytes)
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&t);
  uint32_t *out = reinterpret_cast<uint32_t *>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(RawTable) / 4); i += blockDim.x) out[i] = src[i];
}

int launch_raw_table_write(const RawTable &table_h, RawTable *table_dev, hipStream_t stream) {
  hipLaunchKernelGGL(raw_table_write_kernel, dim3(1), dim3(256), 0, stream, table_h, table_dev);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

template <int DP, int QG, bool MATERIALIZE, bool NT>
__global__ void __launch_bounds__(kRawThreads) rawscan_kernel(const RawScanArgs a) {
  using G = RawGeom<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *wg_cnt = reinterpret_cast<uint32_t *>(smem + G::kCntOff);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;   // query column of this lane / A-tile row it supplies
  const int h = lane >> 5;   // feature half of the MFMA step / upper candidate half of the C layout

  // ---- XCD-aware workgroup remap (bijective for any grid size), as in topk_scan.hip ------------
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
  const int nstages = c0 < c1 ? (int)((c1 - c0 + kTileN - 1) / kTileN) : 0;

  // ---- the tile's queries -> MFMA B operands, resident in every wave for the whole kernel ------
  float bq[QG][DP / 2];
  float thr[QG];
  int64_t qrow[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    qrow[g] = ((int64_t)qt * QG + g) * 32 + j;
    const bool qvalid = qrow[g] < a.nq;
    // unconditional 16-byte loads at a clamped row, selected afterwards: a load under `if (qvalid)` is
    // awaited on its own (one memory round trip per feature pair)
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.q + (qvalid ? qrow[g] : 0) * DP);
#pragma unroll
    for (int u = 0; u < DP / 4; ++u) {
      const f32x4 v = q4[u];   // features 4u .. 4u+3; step 2u takes (4u | 4u+1), step 2u+1 takes (4u+2 | 4u+3)
      bq[g][2 * u] = qvalid ? (h ? v[1] : v[0]) : 0.0f;
      bq[g][2 * u + 1] = qvalid ? (h ? v[3] : v[2]) : 0.0f;
    }
    thr[g] = __builtin_inff();
    if (!MATERIALIZE) {
      const float tv = a.thr[qvalid ? qrow[g] : 0];
      thr[g] = qvalid ? tv : __builtin_inff();
    }
  }
  if (tid < 64) wg_cnt[tid] = 0u;

  // ---- block cursor (wave-uniform): the block that holds the stage's first row -----------------
  const RawTable *T = a.table;
  const int nblk = T->n_blocks;
  int blk = nstages > 0 ? __builtin_amdgcn_readfirstlane(raw_find_block(T, c0)) : 0;
  int64_t blk_lo = T->row_start[blk], blk_hi = T->row_start[blk + 1];
  const char *blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);

  // Prefetch of stage `st` into `lds`.  The copies of a stage that lies in one block (the common case)
  // are NOT issued here: the caller spreads them over its MFMA loop, one per feature step -- a wave that
  // issues all of them back to back sits in the issue stage until the LDS-DMA queue has room, i.e. for
  // most of a memory latency, before its first MFMA (measured: copy time + MFMA time per stage instead of
  // their maximum).  Returns the lane's source address of chunk `wave` (NULL: the copies were issued here).
  auto begin_stage = [&](int st, char *lds) -> const char * {
    const int64_t v0 = c0 + (int64_t)st * kTileN;
    while (v0 >= blk_hi && blk + 1 < nblk) {
      ++blk;
      blk_lo = blk_hi;
      blk_hi = T->row_start[blk + 1];
      blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
    }
    if (v0 + kTileN <= blk_hi && v0 + kTileN <= c1)   // the whole stage lies in one block: a linear copy
      return blk_ptr + (v0 - blk_lo) * (int64_t)G::kRowB + wave * 1024 + lane * 16;
    // block boundary or the last, partly filled stage: every lane looks its row up; rows at or
    // beyond c1 re-read the last valid row (their scores are never used)
#pragma unroll 1
    for (int i = 0; i < G::kCopies; ++i) {
      const int ch = wave + kRawWaves * i;
      const int byte = ch * 1024 + lane * 16;
      const int r = byte / G::kRowB;
      int64_t row = v0 + r;
      if (row > c1 - 1) row = c1 - 1;
      const char *p = reinterpret_cast<const char *>(raw_row_ptr(T, row, DP)) + (byte - r * G::kRowB);
      raw_glds_copy16<NT>(p, lds + ch * kRawChunkB);
    }
    return nullptr;
  };

  if (nstages > 0) {
    const char *src0 = begin_stage(0, smem);
    if (src0 != nullptr) {
#pragma unroll
      for (int i = 0; i < G::kCopies; ++i)
        raw_glds_copy16<NT>(src0 + i * (kRawWaves * 1024), smem + (wave + kRawWaves * i) * kRawChunkB);
    }
  }
  raw_wait_dma();
  __syncthreads();

  // this lane's A-tile row: stage row of t = 32 * wave + j
  const int t_row = 32 * wave + j;
  const int a_off = (t_row % G::kChunks) * kRawChunkB + (t_row / G::kChunks) * G::kRowB + h * 16;
  static_assert(G::kCopies == DP / 8, "one stage copy per feature step of the MFMA loop");

  for (int st = 0; st < nstages; ++st) {
    const char *tile = smem + (st & 1) * G::kStageB;
    const bool more = st + 1 < nstages;
    char *next_lds = smem + ((st + 1) & 1) * G::kStageB + wave * kRawChunkB;
    const char *next_src = more ? begin_stage(st + 1, smem + ((st + 1) & 1) * G::kStageB) : nullptr;
    const int64_t stage_c = c0 + (int64_t)st * kTileN;

    f32x16 acc[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
    const char *ap = tile + a_off;
    // every A fragment of the wave's 32 rows is fetched before the first MFMA: the copies below are inline
    // assembly with a memory clobber, which the compiler will not move an LDS read across -- read inside
    // the loop, each fragment was awaited right where it was issued (one LDS latency per feature step)
    f32x4 av[DP / 8];
#pragma unroll
    for (int m = 0; m < DP / 8; ++m) av[m] = *reinterpret_cast<const f32x4 *>(ap + m * 32);
#pragma unroll
    for (int m = 0; m < DP / 8; ++m) {
      if (next_src != nullptr)   // (wave-uniform) chunk wave + 4 m of the next stage
        raw_glds_copy16<NT>(next_src + m * (kRawWaves * 1024), next_lds + m * (kRawWaves * kRawChunkB));
      const f32x4 v = av[m];
      // (lo | hi) lanes: v = (d0|d4, d1|d5, d2|d6, d3|d7) -> steps (d0|d1), (d2|d3), (d4|d5), (d6|d7)
      const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
      const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
      const float a0 = __uint_as_float(s01[0]), a2 = __uint_as_float(s01[1]);
      const float a1 = __uint_as_float(s23[0]), a3 = __uint_as_float(s23[1]);
#pragma unroll
      for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[g][4 * m + 0], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[g][4 * m + 1], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bq[g][4 * m + 2], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, bq[g][4 * m + 3], acc[g], 0, 0, 0);
    }
    // acc[g][r] = score(query qrow[g], stage row of t = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h)

#pragma unroll
    for (int g = 0; g < QG; ++g) {
      if (MATERIALIZE) {
        if (qrow[g] < a.nq) {
          float *drow = a.dense + qrow[g] * a.ld_dense + (stage_c - a.c_begin);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int srow = (t % G::kChunks) * G::kRowsPerChunk + t / G::kChunks;
            if (stage_c + srow < c1) drow[srow] = acc[g][r];
          }
        }
      } else {
        const float m0 = raw_max16(acc[g]);
        if (__ballot(m0 > thr[g]) != 0ull) {   // rare once the threshold is warm
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int srow = (t % G::kChunks) * G::kRowsPerChunk + t / G::kChunks;
            if (acc[g][r] > thr[g] && stage_c + srow < c1) {
              const uint32_t e = atomicAdd(&wg_cnt[g * 32 + j], 1u);
              if (e < a.cap_l)
                a.buf[(qrow[g] * (int64_t)a.cap_l + e) * a.nseg + split] =
                    make_uint2(__float_as_uint(acc[g][r]), (uint32_t)(stage_c + srow));
            }
          }
        }
      }
    }
    if (more) raw_wait_dma();
    __syncthreads();
  }

  // every (query, split) count is written: no memset needed
  if (!MATERIALIZE && tid < QG * 32) {
    const int64_t qr = ((int64_t)qt * QG + (tid >> 5)) * 32 + (tid & 31);
    if (qr < a.nq) a.cnt[qr * a.nseg + split] = wg_cnt[tid];
  }
}

// the stage copies carry the non-temporal hint when every row is read once, by one workgroup (raw_glds_copy16)
static bool raw_nt_copies(const RawScanArgs &a) {
  const char *e = option("TFRS_RAW_NT");
  return a.n_qtiles == 1 && !(e && e[0] == '0');
}

template <int DP, int QG, bool MAT, bool NT>
static int launch_rawscan_variant_nt(const RawScanArgs &a, hipStream_t stream) {
  using G = RawGeom<DP>;
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&rawscan_kernel<DP, QG, MAT, NT>), G::kLdsBytes));
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  hipLaunchKernelGGL((rawscan_kernel<DP, QG, MAT, NT>), grid, dim3(kRawThreads), G::kLdsBytes, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
template <int DP, int QG, bool MAT>
static int launch_rawscan_variant(const RawScanArgs &a, hipStream_t stream) {
  if constexpr (!MAT) {   // (the dense round is a few stages per workgroup: no second instantiation for it)
    if (raw_nt_copies(a)) return launch_rawscan_variant_nt<DP, QG, MAT, true>(a, stream);
  }
  return launch_rawscan_variant_nt<DP, QG, MAT, false>(a, stream);
}

template <int DP>
static int launch_rawscan_dp(const RawScanArgs &a, bool materialize, hipStream_t stream) {
  if (a.qg == 1)
    return materialize ? launch_rawscan_variant<DP, 1, true>(a, stream)
                       : launch_rawscan_variant<DP, 1, false>(a, stream);
  return materialize ? launch_rawscan_variant<DP, 2, true>(a, stream)
                     : launch_rawscan_variant<DP, 2, false>(a, stream);
}

int launch_rawscan(const RawScanArgs &a, bool materialize, hipStream_t stream) {
  if (a.nq <= 0 || a.c_end <= a.c_begin) return TFRS_OK;
  TFRS_CHECK_ARG(a.c_begin % kTileN == 0 && a.split_len % kTileN == 0 && a.split_len > 0,
                 "rawscan: c_begin/split_len must be multiples of %d", kTileN);
  TFRS_CHECK_ARG(a.qg == 1 || a.qg == 2, "rawscan: qg must be 1 or 2");
  TFRS_CHECK_ARG(materialize || a.nseg == a.n_splits, "rawscan: nseg must equal n_splits");
  TFRS_CHECK_ARG(a.d == padded_dim(a.d), "rawscan: dim %d is not one of 8, 16, 32, 64, 128", a.d);
  switch (a.d) {
    case 8: return launch_rawscan_dp<8>(a, materialize, stream);
    case 16: return launch_rawscan_dp<16>(a, materialize, stream);
    case 32: return launch_rawscan_dp<32>(a, materialize, stream);
    case 64: return launch_rawscan_dp<64>(a, materialize, stream);
    case 128: return launch_rawscan_dp<128>(a, materialize, stream);
  }
  set_error("rawscan: unsupported dim %d", a.d);
  return TFRS_ENOTIMPL;
}

typedef _Float16 rf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t raw_pack_f16x2(float lo, float hi) {
  union {
    rf16x2_t h;
    uint32_t u;
  } v;
  v.h[0] = (_Float16)lo;   // v_cvt_pk_f16_f32, round to nearest even
  v.h[1] = (_Float16)hi;
  return v.u;
}

// ---- fp16-PREFILTERED scan straight from the blocks (query batches up to 256) ---------------------
//
// The exact kernel above spends 64 matrix-core cycles per feature pair and group of 32 queries: at
// D = 128 a workgroup needs 4096 cycles per stage for ONE group, about what the stage's 64 KiB take to
// arrive from HBM -- and twice that for 64 queries.  Here the stage arrives the same way (row-major f32,
// direct-to-LDS), every wave converts ITS 32 rows to fp16 in registers (x / s with s = 2^ceil(log2 max|x|)
// of those 32 rows, the contract of the fp16 image of topk_pack.hip with a "stage" of 32 rows) and scores
// them with v_mfma_f32_32x32x16_f16: 32 cycles per 16 features and group, so that four groups (128
// queries) cost less than the copy and eight (256) about as much.  Nothing computed here is returned: a prefilter score only
// decides whether a row can still reach a query's top-K under the bound of common.h,
//     |s~ - s| <= ||q|| * N * kappa (+ tiny),   N = largest row norm of the wave's 32 rows,
// against thr[q], a proven lower bound of the query's final K-th score (the carried state's exact K-th
// score); survivors are re-scored exactly from the blocks by list_topk16_kernel (raw_score).
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

template <int DP, int QG, bool NT, int NWV = kRawWaves>
__global__ void __launch_bounds__(NWV * 64) rawscan16_kernel(const RawScanArgs a) {
  using G = RawGeom<DP, NWV>;
  constexpr int kTile = G::kTile;   // rows per stage
  constexpr int KS = DP >= 16 ? DP / 16 : 1;   // MFMA steps of 16 features
  constexpr int NV = DP >= 16 ? DP / 8 : 2;    // 16-byte pieces a lane holds: half of its row (DP = 8: the row, lane half 0)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *wg_cnt = reinterpret_cast<uint32_t *>(smem + G::kCntOff);
  static_assert(QG * 32 <= 256, "wg_cnt holds 256 counters");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  if (a.zero_word && blockIdx.x == 0 && tid == 0) *a.zero_word = 0u;
  if (a.zero_aux && blockIdx.x == 0 && tid < 4) a.zero_aux[tid] = 0u;

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
  const int nstages = c0 < c1 ? (int)((c1 - c0 + kTile - 1) / kTile) : 0;

  // ---- the tile's queries -> fp16 MFMA B operands (q / qscale), resident -------------------------
  f16x8r bq[QG][KS];
  float flo[QG], fqk[QG], qsc[QG];   // (lower - tiny) / qscale, qk / qscale, qscale
  int64_t qrow[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    qrow[g] = ((int64_t)qt * QG + g) * 32 + j;
    const bool qvalid = qrow[g] < a.nq;
    const int64_t qr = qvalid ? qrow[g] : 0;
    const float qs_ld = a.qscale[qr], thr_ld = a.thr[qr * (a.thr_stride > 0 ? a.thr_stride : 1)], qk_ld = a.qk[qr];   // unconditional: counted loads
    const float qs = qvalid ? qs_ld : 1.0f;
    const float qinv = 1.0f / qs;   // exact: power of two
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.q + qr * DP);
#pragma unroll
    for (int m = 0; m < KS; ++m) {
      // features 16 m + 8 h .. + 7 (DP = 8: the whole row in lane half 0, zeros in half 1)
      const int p0 = DP >= 16 ? 4 * m + 2 * h : 0;
      const f32x4 lo = q4[p0], hi = q4[p0 + 1];
      const bool on = qvalid && (DP >= 16 || h == 0);
      u32x4r w;
      w[0] = raw_pack_f16x2(on ? lo[0] * qinv : 0.0f, on ? lo[1] * qinv : 0.0f);
      w[1] = raw_pack_f16x2(on ? lo[2] * qinv : 0.0f, on ? lo[3] * qinv : 0.0f);
      w[2] = raw_pack_f16x2(on ? hi[0] * qinv : 0.0f, on ? hi[1] * qinv : 0.0f);
      w[3] = raw_pack_f16x2(on ? hi[2] * qinv : 0.0f, on ? hi[3] * qinv : 0.0f);
      bq[g][m] = __builtin_bit_cast(f16x8r, w);
    }
    flo[g] = qvalid ? (thr_ld - kF16Tiny) * qinv : __builtin_inff();
    fqk[g] = qvalid ? qk_ld * qinv : 0.0f;
    qsc[g] = qs;
  }
  for (int e = tid; e < 256; e += NWV * 64) wg_cnt[e] = 0u;   // 256 counters

  // ---- block cursor and stage prefetch: as in rawscan_kernel ------------------------------------
  const RawTable *T = a.table;
  const int nblk = T->n_blocks;
  int blk = nstages > 0 ? __builtin_amdgcn_readfirstlane(raw_find_block(T, c0)) : 0;
  int64_t blk_lo = T->row_start[blk], blk_hi = T->row_start[blk + 1];
  const char *blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
  // The copies of a stage that lies in one block (the common case) are NOT issued here: a wave that issues
  // all of them back to back sits in the issue stage until the LDS-DMA queue has room, and its arithmetic
  // starts only then (measured with four query groups: 382 us per 3.1 M rows against 272 us with one).
  // The caller spreads them over the phases of the stage (copy_part).  Returns the lane's source address
  // of chunk `wave` (NULL: the copies were issued here).
  auto begin_stage = [&](int st, char *lds) -> const char * {
    const int64_t v0 = c0 + (int64_t)st * kTile;
    while (v0 >= blk_hi && blk + 1 < nblk) {
      ++blk;
      blk_lo = blk_hi;
      blk_hi = T->row_start[blk + 1];
      blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
    }
    if (v0 + kTile <= blk_hi && v0 + kTile <= c1)   // the whole stage lies in one block: a linear copy
      return blk_ptr + (v0 - blk_lo) * (int64_t)G::kRowB + wave * 1024 + lane * 16;
    // block boundary or the last, partly filled stage: every lane looks its row up; rows at or
    // beyond c1 re-read the last valid row (their scores are never used)
#pragma unroll 1
    for (int i = 0; i < G::kCopies; ++i) {
      const int ch = wave + NWV * i;
      const int byte = ch * 1024 + lane * 16;
      const int r = byte / G::kRowB;
      int64_t row = v0 + r;
      if (row > c1 - 1) row = c1 - 1;
      const char *p = reinterpret_cast<const char *>(raw_row_ptr(T, row, DP)) + (byte - r * G::kRowB);
      raw_glds_copy16<NT>(p, lds + ch * kRawChunkB);
    }
    return nullptr;
  };
  // quarter `part` (0 .. 3) of the wave's copies of a stage
  auto copy_part = [&](const char *src, char *lds_wave, int part) __attribute__((always_inline)) {
    if (src == nullptr) return;   // (wave-uniform)
    constexpr int kPer = (G::kCopies + 3) / 4;
#pragma unroll
    for (int i = part * kPer; i < (part + 1) * kPer && i < G::kCopies; ++i)
      raw_glds_copy16<NT>(src + i * (NWV * 1024), lds_wave + i * (NWV * kRawChunkB));
  };
  if (nstages > 0) {
    const char *src0 = begin_stage(0, smem);
#pragma unroll
    for (int part = 0; part < 4; ++part) copy_part(src0, smem + wave * kRawChunkB, part);
  }
  raw_wait_dma();
  __syncthreads();

  // this lane's A-tile row: stage row of t = 32 * wave + j; its half h of the row (32 bytes per step)
  const int t_row = 32 * wave + j;
  const int a_off = (t_row % G::kChunks) * kRawChunkB + (t_row / G::kChunks) * G::kRowB + (DP >= 16 ? h * 32 : 0);
  float norm_run = 0.0f;   // largest row norm this wave has met
  const uint32_t rows_here = (uint32_t)(c1 > c0 ? c1 - c0 : 0);   // rows of this split (< 2^31)
  const uint32_t row0 = (uint32_t)c0;                              // (group-local row numbers fit 32 bits)

  for (int st = 0; st < nstages; ++st) {
    const char *tile = smem + (st & 1) * G::kStageB;
    const bool more = st + 1 < nstages;
    char *next_lds = smem + ((st + 1) & 1) * G::kStageB + wave * kRawChunkB;
    const char *next_src = more ? begin_stage(st + 1, smem + ((st + 1) & 1) * G::kStageB) : nullptr;
    const int64_t stage_c = c0 + (int64_t)st * kTile;

    // ---- the lane's half row; norm and max |x| of the wave's 32 rows ------------------------------
    const char *ap = tile + a_off;
    f32x4 av[NV];
#pragma unroll
    for (int m = 0; m < KS; ++m) {
      av[2 * m] = *reinterpret_cast<const f32x4 *>(ap + m * 64);
      av[2 * m + 1] = *reinterpret_cast<const f32x4 *>(ap + m * 64 + 16);
    }
    copy_part(next_src, next_lds, 0);
    if (DP < 16 && h == 1) {
      av[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      av[1] = av[0];
    }
    float ss = 0.0f, am = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ss = __builtin_fmaf(av[i][c], av[i][c], ss);
        am = fmaxf(am, __builtin_fabsf(av[i][c]));
      }
    ss += __shfl_xor(ss, 32);   // the row's other half
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      if (off < 32) ss = fmaxf(ss, __shfl_xor(ss, off));
      am = fmaxf(am, __shfl_xor(am, off));
    }
    const float nrm = __builtin_sqrtf(ss) * kNormSlack;   // (upper bound of the 32 row norms)
    const float cs = pow2_ceil(am);
    const float inv = 1.0f / cs;                          // exact: power of two
    norm_run = fmaxf(norm_run, nrm);
    copy_part(next_src, next_lds, 1);

    f16x8r af[KS];
#pragma unroll
    for (int m = 0; m < KS; ++m) {
      u32x4r w;
      w[0] = raw_pack_f16x2(av[2 * m][0] * inv, av[2 * m][1] * inv);
      w[1] = raw_pack_f16x2(av[2 * m][2] * inv, av[2 * m][3] * inv);
      w[2] = raw_pack_f16x2(av[2 * m + 1][0] * inv, av[2 * m + 1][1] * inv);
      w[3] = raw_pack_f16x2(av[2 * m + 1][2] * inv, av[2 * m + 1][3] * inv);
      af[m] = __builtin_bit_cast(f16x8r, w);
    }

    copy_part(next_src, next_lds, 2);

    // ---- prefilter scores and the filter ----------------------------------------------------------
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      if (g == (QG > 1 ? 1 : 0)) copy_part(next_src, next_lds, 3);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m], bq[g][m], acc, 0, 0, 0);
      // acc[r] = prefilter score of (query qrow[g], stage row of t = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h),
      // in units of qscale * cs.  keep s~ > lower - qk * N - tiny, the scales folded into the threshold
      const float thr = __builtin_fmaf(-fqk[g], nrm, flo[g]) * inv;
      const float m0 = raw_max16f(acc);
      if (__ballot(m0 > thr) != 0ull) {   // rare once the bound is warm
        // (round 5, as in rawscan16w_kernel: 32-bit row arithmetic relative to the split and ONE counter round trip
        // per lane and hot tile instead of sixteen dependent ones)
        const float un = qsc[g] * cs;
        const uint32_t rel_stage = (uint32_t)st * kTile;
        uint32_t hits = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) hits |= acc[r] > thr ? (1u << r) : 0u;
        if (rel_stage + kTile > rows_here) {   // (uniform) the split's last, partly filled stage: rows beyond it score nothing
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            const uint32_t srow = (uint32_t)((t % G::kChunks) * G::kRowsPerChunk + t / G::kChunks);
            if (rel_stage + srow >= rows_here) hits &= ~(1u << r);
          }
        }
        if (hits) {
          uint32_t e = atomicAdd(&wg_cnt[g * 32 + j], (uint32_t)__builtin_popcount(hits));
          uint2 *const seg = a.buf + (qrow[g] * (int64_t)a.cap_l) * a.nseg + split;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (hits & (1u << r)) {
              const int t = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
              const uint32_t srow = (uint32_t)((t % G::kChunks) * G::kRowsPerChunk + t / G::kChunks);
              if (e < a.cap_l)
                seg[(uint64_t)e * (uint32_t)a.nseg] = make_uint2(__float_as_uint(acc[r] * un), row0 + rel_stage + srow);
              ++e;
            }
          }
        }
      }
    }
    if (more) raw_wait_dma();
    __syncthreads();
  }

  // every (query, split) count is written: no memset needed (counts beyond cap_l flag the query for
  // the exact redo, as the image-fed filter kernel does)
  for (int e = tid; e < QG * 32; e += NWV * 64) {   // (the stage loop ends with a barrier)
    const int64_t qr = ((int64_t)qt * QG + (e >> 5)) * 32 + (e & 31);
    if (qr < a.nq) a.cnt[qr * a.nseg + split] = wg_cnt[e];
  }
  if (nstages > 0 && lane == 0 && a.norm_max)
    atomicMax(reinterpret_cast<uint32_t *>(a.norm_max), __float_as_uint(norm_run));   // >= 0
}

template <int DP, int QG, bool NT, int NWV = kRawWaves>
static int launch_rawscan16_variant_nt(const RawScanArgs &a, hipStream_t stream) {
  using G = RawGeom<DP, NWV>;
  static_assert(G::kLdsBytes + 768 <= 160 * 1024, "two raw stages + the counters must fit the LDS");
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&rawscan16_kernel<DP, QG, NT, NWV>), G::kLdsBytes + 768));
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  hipLaunchKernelGGL((rawscan16_kernel<DP, QG, NT, NWV>), grid, dim3(NWV * 64), G::kLdsBytes + 768, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
// Dim 128: two 64 KiB stages fill the LDS -- ONE workgroup of four waves per CU, whose copies of the next stage are all
// issued in the first third of a stage and awaited at its end.  Two workgroups of TWO waves on 64-row stages (2 x 33 KB
// each) run out of step.  Measured on 12.5 M x 128 (same box, alternating): with four query groups per wave (65 .. 128
// queries) 96 / 128 queries 1.49 / 1.55 -> 1.455 / 1.51 ms; with one or two groups the four-wave form is 2-3 % FASTER
// (1.239 / 1.317 / 1.385 against 1.27 / 1.357 / 1.412 ms at 1 / 32 / 64 queries: twice the splits = twice the survivor
// segments for the list kernel, and the scan was at the copy engine's rate already) -- so only four groups take it
// (topk_api.hip doubles the number of splits for these launches; TFRS_RAW16_HALF=0: never).
bool rawscan16_half(int d, int qg, int n_qtiles) {
  const char *e = option("TFRS_RAW16_HALF");
  return d == 128 && qg == 4 && n_qtiles == 1 && !(e && e[0] == '0');
}
template <int DP, int QG>
static int launch_rawscan16_variant(const RawScanArgs &a, hipStream_t stream) {
  if constexpr (DP == 128 && QG == 4) {
    if (rawscan16_half(a.d, a.qg, a.n_qtiles))
      return raw_nt_copies(a) ? launch_rawscan16_variant_nt<DP, QG, true, 2>(a, stream)
                              : launch_rawscan16_variant_nt<DP, QG, false, 2>(a, stream);
  }
  return raw_nt_copies(a) ? launch_rawscan16_variant_nt<DP, QG, true>(a, stream)
                          : launch_rawscan16_variant_nt<DP, QG, false>(a, stream);
}

// ---- the same filter for query batches of 257 .. 2048 (round 5): ONE conversion of a stage per workgroup -------
//
// rawscan16_kernel keeps the tile's queries resident in EVERY wave and lets wave w score its 32 rows: the query tile
// is bounded by one wave's registers (128 queries at dim 128, 256 below), and larger batches went through one fp16
// image of the whole group (pack16_raw_kernel: 1.9 ms for 12.5 M x 128 -- 6.4 GB read, 3.4 GB written, then read
// again by the filter passes), which dominated calls of 257 .. ~2000 queries (512 queries: 3.7 ms against 1.7 for a
// resident index).  Here the WAVES split the queries instead: 8 waves x 2 groups of 32 = 512 queries per workgroup,
// and the stage is converted to fp16 ONCE per workgroup:
//   load     wave w reads rows [16 w, 16 w + 16) of a 128-row stage straight into registers (lane l takes the 16-byte
//            pieces l, l + 64, .. of those 16 rows: 1 KiB per instruction, whole lines), TWO stages ahead of the
//            one being scored -- two stages of 64 KiB in flight per CU, no f32 buffer in LDS (a first version
//            staged the f32 rows through LDS with one stage in flight: 2.5 TB/s);
//   convert  the wave's 16 rows share one power-of-two scale 2^ceil(log2 max|x|) and one norm bound (the largest of
//            the 16 row norms): the contract of the fp16 image with a "stage" of 16 rows; fp16 rows go to one of TWO
//            fp16 tiles in LDS (row pitch 2 DP + 16 bytes: the conflict-free A-operand layout of topk_scan16.hip),
//            {1 / scale, scale, norm} of the 16-row group to a small table;
//   score    after ONE barrier per stage every wave reads the A fragments of all 128 rows and runs them against
//            its own two query groups; a 32 x 32 tile spans two 16-row groups = accumulator registers 0 .. 7 and
//            8 .. 15 of a lane, each half tested against its own threshold.
// Per stage and SIMD 2 x 64 MFMAs = 4096 matrix-core cycles for 64 KiB of rows.  LDS: 2 x 34.8 KB + counters at dim 128.
// MEASURED (round 5, 12.5 M x 128, 512 queries; TFRS_RAWW_ABLATE builds): loads + conversion alone 1.16 ms per call =
// 5.5 TB/s, the whole kernel 2.8-2.96 ms (2.6-2.75 with the trimmed check of the later part of round 5): the scoring phase runs at about a third of the matrix pipe's rate (waves
// parked 48 % of their cycles: two waves per SIMD of ONE workgroup that converts, meets its barrier and scores in
// lock step), so the kernel pays only between 257 and ~700 queries (topk_api.hip: TFRS_STREAM_RAW16_MAX_NQ = 640).
// Round 6 tried the remedy the lock step suggests -- ONE query group per wave (256 queries per workgroup, <= 128 registers),
// TWO workgroups per CU so that one converts while the other scores: SLOWER at every size (12.5 M x 128: 257 / 384 / 512 /
// 640 queries 3.19 / 3.13 / 3.38 / 8.2 ms against 2.54 / 2.67 / 2.87 / 5.0; 25 M x 64: 3.45 / 3.56 against 3.03 / 3.40) --
// every stage is then loaded and converted twice per CU (the conversion is the phase that was to be hidden), and dim 128
// needs 25 registers of scratch at 128.  Not kept.
// Tried on the way, each within +-3 %: rows staged through LDS by DMA (one stage in flight), two stages of rows in
// flight in registers, the chains of the two groups in sequence instead of step by step, one counter round trip per
// hot tile instead of sixteen, no cross-lane reductions at all (ablation).
// timing ablations of rawscan16w_kernel (tools/ab_variants.sh; results WRONG): 1 = no cross-lane reductions in the
// conversion, 2 = no scoring, 4 = no conversion at all, 8 = no loads after the first stage
// 1 = a second A-fragment set at dim 128 too (the next sub-tile's fragments fetched under the current chains): needs
// 5-9 registers more than the 256 a lane has next to the stage of rows in flight -- scratch in the stage loop; not built
#ifndef TFRS_RAWW_AFB128
#define TFRS_RAWW_AFB128 0
#endif
#ifndef TFRS_RAWW_ABLATE
#define TFRS_RAWW_ABLATE 0
#endif
#if TFRS_RAWW_ABLATE
// An ablation build computes WRONG results by design: it needs -DTFRS_ALLOW_ABLATION next to -DTFRS_RAWW_ABLATE=..., and the
// marker symbol below makes recommenders_amd/_lib.py refuse the library unless TFRS_ALLOW_ABLATION=1 is set.
#ifndef TFRS_ALLOW_ABLATION
#error "TFRS_RAWW_ABLATE != 0 is a measurement build with wrong results: add -DTFRS_ALLOW_ABLATION to confirm"
#endif
extern "C" int tfrs_ablation_build_raww(void) { return TFRS_RAWW_ABLATE; }
#endif

__device__ __forceinline__ void raw_lds_barrier() {
  // s_barrier is IntrNoMem for LLVM: without a compiler-level fence the optimizer may move LDS loads / stores across
  // it.  The empty asm statements with a "memory" clobber pin every memory access on its side of the
  // barrier without emitting anything -- a workgroup-scope __builtin_amdgcn_fence would bring the vmcnt(0) back.
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));   // s_waitcnt lgkmcnt(0), vmcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

constexpr int kRawWWaves = 8;
constexpr int kRawWQG = 2;                                  // query groups of 32 per wave
constexpr int kRawWQueries = kRawWWaves * kRawWQG * 32;     // 512 per workgroup

template <int DP>
struct RawWGeom {
  static constexpr int kRowB = DP * 4;
  static constexpr int kPieces = 16 * kRowB / 16 / 64;        // 16-byte pieces per lane and stage (the wave's 16 rows)
  static constexpr int kRow16B = DP * 2 + 16;
  static constexpr int kTile16B = kTileN * kRow16B;
  static constexpr int kMetaOff = 2 * kTile16B;               // [2][8] x float4 {1 / scale, scale, norm, -}
  static constexpr int kCntOff = kMetaOff + 2 * kRawWWaves * 16;
  static constexpr int kLdsBytes = kCntOff + kRawWQueries * 4;
  static_assert(kPieces >= 1 && kPieces * 64 * 16 == 16 * kRowB, "a wave's 16 rows are whole 1 KiB pieces");
};

template <int DP>
__global__ void __launch_bounds__(kRawWWaves * 64) rawscan16w_kernel(const RawScanArgs a) {
  using G = RawWGeom<DP>;
  constexpr int KS = DP / 16;           // MFMA steps of 16 features
  constexpr int PPR = DP / 4;           // 16-byte pieces per f32 row
  constexpr int NP = G::kPieces;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *wg_cnt = reinterpret_cast<uint32_t *>(smem + G::kCntOff);
  float4 *meta = reinterpret_cast<float4 *>(smem + G::kMetaOff);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  if (a.zero_word && blockIdx.x == 0 && tid == 0) *a.zero_word = 0u;
  if (a.zero_aux && blockIdx.x == 0 && tid < 4) a.zero_aux[tid] = 0u;

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
  const int nstages = c0 < c1 ? (int)((c1 - c0 + kTileN - 1) / kTileN) : 0;

  // ---- this WAVE's 2 x 32 queries -> fp16 MFMA B operands (q / qscale), resident ------------------
  f16x8r bq[kRawWQG][KS];
  float flo[kRawWQG], fqk[kRawWQG], qsc[kRawWQG];
  // (one 32-bit query number per lane, group g adds 32 g: the batch holds < 2^31 queries; two 64-bit row numbers
  // were four registers of a budget that the second A-fragment set needs)
  const uint32_t qrow0 = (uint32_t)qt * kRawWQueries + (uint32_t)(wave * kRawWQG) * 32u + (uint32_t)j;
#pragma unroll
  for (int g = 0; g < kRawWQG; ++g) {
    const int64_t qrow_g = (int64_t)qrow0 + 32 * g;
    const bool qvalid = qrow_g < a.nq;
    const int64_t qr = qvalid ? qrow_g : 0;
    const float qs_ld = a.qscale[qr], thr_ld = a.thr[qr * (a.thr_stride > 0 ? a.thr_stride : 1)], qk_ld = a.qk[qr];   // unconditional: counted loads
    const float qs = qvalid ? qs_ld : 1.0f;
    const float qinv = 1.0f / qs;   // exact: power of two
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.q + qr * DP);
#pragma unroll
    for (int m = 0; m < KS; ++m) {
      const f32x4 lo = q4[4 * m + 2 * h], hi = q4[4 * m + 2 * h + 1];   // features 16 m + 8 h .. + 7
      u32x4r w;
      w[0] = raw_pack_f16x2(qvalid ? lo[0] * qinv : 0.0f, qvalid ? lo[1] * qinv : 0.0f);
      w[1] = raw_pack_f16x2(qvalid ? lo[2] * qinv : 0.0f, qvalid ? lo[3] * qinv : 0.0f);
      w[2] = raw_pack_f16x2(qvalid ? hi[0] * qinv : 0.0f, qvalid ? hi[1] * qinv : 0.0f);
      w[3] = raw_pack_f16x2(qvalid ? hi[2] * qinv : 0.0f, qvalid ? hi[3] * qinv : 0.0f);
      bq[g][m] = __builtin_bit_cast(f16x8r, w);
    }
    flo[g] = qvalid ? (thr_ld - kF16Tiny) * qinv : __builtin_inff();
    fqk[g] = qvalid ? qk_ld * qinv : 0.0f;
    qsc[g] = qs;
  }
  wg_cnt[tid] = 0u;   // 512 threads, 512 counters
  const bool wave_active = (int64_t)qt * kRawWQueries + wave * (kRawWQG * 32) < a.nq;   // (uniform)

  // ---- block cursor; the wave's loads of a stage (its own 16 rows) ----------------------------------
  const RawTable *T = a.table;
  const int nblk = T->n_blocks;
  int blk = nstages > 0 ? __builtin_amdgcn_readfirstlane(raw_find_block(T, c0)) : 0;
  int64_t blk_lo = T->row_start[blk], blk_hi = T->row_start[blk + 1];
  const char *blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
  const uint32_t rows_here = (uint32_t)(c1 > c0 ? c1 - c0 : 0);   // rows of this split (a split holds < 2^31 rows)
  const uint32_t row0 = (uint32_t)c0;                              // (group-local row numbers fit 32 bits)
  auto load_stage = [&](int st, f32x4 (&dst)[NP]) __attribute__((always_inline)) {
    const int64_t v0 = c0 + (int64_t)st * kTileN;
    while (v0 >= blk_hi && blk + 1 < nblk) {
      ++blk;
      blk_lo = blk_hi;
      blk_hi = T->row_start[blk + 1];
      blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
    }
    if (v0 + kTileN <= blk_hi && v0 + kTileN <= c1) {   // the whole stage lies in one block: linear loads
      const char *src = blk_ptr + ((v0 - blk_lo) + 16 * wave) * (int64_t)G::kRowB + lane * 16;
#pragma unroll
      for (int i = 0; i < NP; ++i) dst[i] = *reinterpret_cast<const f32x4 *>(src + i * 1024);
      return;
    }
    // block boundary or the last, partly filled stage: every lane looks its row up; rows at or beyond c1
    // re-read the last valid row (their scores are never used)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = i * 64 + lane;
      int64_t row = v0 + 16 * wave + piece / PPR;
      if (row > c1 - 1) row = c1 - 1;
      dst[i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(raw_row_ptr(T, row, DP)) +
                                                (piece % PPR) * 16);
    }
  };
  float norm_run = 0.0f;
  // the wave's 16 rows (registers) -> fp16 tile `buf`, group constants -> meta[buf][wave]
  auto convert = [&](const f32x4 (&av)[NP], int buf) __attribute__((always_inline)) {
    float am = 0.0f, nmax = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      float ss = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ss = __builtin_fmaf(av[i][c], av[i][c], ss);
        am = fmaxf(am, __builtin_fabsf(av[i][c]));
      }
      // piece i * 64 + lane of the wave's 16 rows: PPR consecutive lanes hold one row
      if (!(TFRS_RAWW_ABLATE & 1)) {
#pragma unroll
      for (int off = 1; off < PPR && off < 64; off <<= 1) ss += __shfl_xor(ss, off);
      }
      nmax = fmaxf(nmax, ss);
    }
    if (!(TFRS_RAWW_ABLATE & 1)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      nmax = fmaxf(nmax, __shfl_xor(nmax, off));
      am = fmaxf(am, __shfl_xor(am, off));
    }
    }
    const float nrm = __builtin_sqrtf(nmax) * kNormSlack;   // (upper bound of the 16 row norms)
    const float cs = pow2_ceil(am);
    const float inv = 1.0f / cs;                            // exact: power of two
    norm_run = fmaxf(norm_run, nrm);
    char *t16 = smem + buf * G::kTile16B;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = i * 64 + lane;
      const int row = 16 * wave + piece / PPR, col = piece % PPR;
      uint2 w;
      w.x = raw_pack_f16x2(av[i][0] * inv, av[i][1] * inv);
      w.y = raw_pack_f16x2(av[i][2] * inv, av[i][3] * inv);
      *reinterpret_cast<uint2 *>(t16 + row * G::kRow16B + col * 8) = w;
    }
    if (lane == 0) meta[buf * kRawWWaves + wave] = make_float4(inv, cs, nrm, 0.0f);
  };

  // the rows of stage s + 1 wait in registers while stage s is scored (one stage = 64 KiB per CU in flight; a second
  // set -- two stages ahead -- measured the same and its 32 registers are better spent on the second accumulator)
  f32x4 nx[NP];
  if (nstages > 0) load_stage(0, nx);
  if (nstages > 0) convert(nx, 0);
  __syncthreads();                      // fp16 tile 0 complete (and the counters zeroed)
  if (nstages > 1) load_stage(1, nx);

  for (int st = 0; st < nstages; ++st) {
    const int buf = st & 1;
    const char *t16 = smem + buf * G::kTile16B;
    if (wave_active && !(TFRS_RAWW_ABLATE & 2)) {
      const char *ap = t16 + j * G::kRow16B + h * 16;
      // A fragments of the next sub-tile are fetched under the chains of the current one -- up to dim 64; at dim 128
      // a second set (32 registers next to two stages of rows in flight) spills
      constexpr int AFB = TFRS_RAWW_AFB128 ? 2 : (DP >= 128 ? 1 : 2);
      u32x4r af[AFB][KS];
#pragma unroll
      for (int m = 0; m < KS; ++m) af[0][m] = *reinterpret_cast<const u32x4r *>(ap + m * 32);
#pragma unroll
      for (int sub = 0; sub < kTileN / 32; ++sub) {
        if (AFB == 2 && sub + 1 < kTileN / 32) {
#pragma unroll
          for (int m = 0; m < KS; ++m)
            af[(sub + 1) % AFB][m] = *reinterpret_cast<const u32x4r *>(ap + (sub + 1) * 32 * G::kRow16B + m * 32);
        }
        const float4 mA = meta[buf * kRawWWaves + 2 * sub], mB = meta[buf * kRawWWaves + 2 * sub + 1];
        // the chains of the two groups step by step in turn: consecutive MFMAs never depend on each other
        f32x16 acc[kRawWQG];
#pragma unroll
        for (int g = 0; g < kRawWQG; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
#pragma unroll
        for (int m = 0; m < KS; ++m)
#pragma unroll
          for (int g = 0; g < kRawWQG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8r, af[sub % AFB][m]), bq[g][m], acc[g], 0, 0, 0);
        if (AFB == 1 && sub + 1 < kTileN / 32) {   // dim 128: the one fragment set is refilled under the chains' tail
#pragma unroll
          for (int m = 0; m < KS; ++m)
            af[0][m] = *reinterpret_cast<const u32x4r *>(ap + (sub + 1) * 32 * G::kRow16B + m * 32);
        }
#pragma unroll
        for (int g = 0; g < kRawWQG; ++g) {
          // acc[g][r] = prefilter score of (query qrow[g], stage row 32 sub + (r & 3) + 8 (r >> 2) + 4 h) in units of
          // qscale * (scale of the row's 16-row group): registers 0 .. 7 are rows of group 2 sub, 8 .. 15 of 2 sub + 1
          const f32x16 &c = acc[g];
          const float thrA = __builtin_fmaf(-fqk[g], mA.z, flo[g]) * mA.x;
          const float thrB = __builtin_fmaf(-fqk[g], mB.z, flo[g]) * mB.x;
          const float xa = raw_max8f(c, 0), xb = raw_max8f(c, 8);
          if (__ballot(xa > thrA || xb > thrB) != 0ull) {   // rare once the bound is warm
            // (32-bit row arithmetic relative to the split: sixteen 64-bit compares per tile were precomputed
            // and spilled -- 42 scratch stores per stage)
            const uint32_t rel0 = (uint32_t)st * kTileN + 32u * sub + 4u * h;
            uint2 *const seg = a.buf + (((int64_t)qrow0 + 32 * g) * (int64_t)a.cap_l) * a.nseg + split;
            const float unA = qsc[g] * mA.y, unB = qsc[g] * mB.y;
            // ONE counter round trip per lane and hot tile: the lane counts its survivors, reserves that many
            // slots, then stores them (sixteen dependent ds_add_rtn round trips per hot tile made the scoring
            // phase 3x its matrix-core time: in a stream's early ranges every tile is hot)
            uint32_t hits = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) hits |= c[r] > (r < 8 ? thrA : thrB) ? (1u << r) : 0u;
            if ((uint32_t)(st + 1) * kTileN > rows_here) {   // (uniform) the split's last, partly filled stage
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (rel0 + (r & 3) + 8 * (r >> 2) >= rows_here) hits &= ~(1u << r);
            }
            if (hits) {
              uint32_t e = atomicAdd(&wg_cnt[(wave * kRawWQG + g) * 32 + j], (uint32_t)__builtin_popcount(hits));
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                if (hits & (1u << r)) {
                  if (e < a.cap_l)
                    seg[(uint64_t)e * (uint32_t)a.nseg] =
                        make_uint2(__float_as_uint(c[r] * (r < 8 ? unA : unB)), row0 + rel0 + (r & 3) + 8 * (r >> 2));
                  ++e;
                }
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // one sub-tile at a time (the chains of all four issued first keep
                                             // 128 accumulators alive)
      }
    }
    // the next stage's rows of this wave have landed long ago: convert them, then refill their registers with the
    // rows of the stage after next
    if (st + 1 < nstages && !(TFRS_RAWW_ABLATE & 4)) convert(nx, buf ^ 1);
    if (st + 2 < nstages && !(TFRS_RAWW_ABLATE & 8)) load_stage(st + 2, nx);
    // everybody scored stage st and converted its part of st + 1.  NOT __syncthreads(): its workgroup fence waits for
    // vmcnt(0), i.e. for the loads of stage st + 2 issued one line above -- the whole HBM latency exposed once per
    // stage (7.4 us per stage, 2.2 TB/s).  Only this wave's LDS writes have to be done before the barrier.
    raw_lds_barrier();
  }

  {   // every (query, split) count is written (counts beyond cap_l flag the query for the exact redo)
    const int64_t qr = (int64_t)qt * kRawWQueries + tid;
    if (qr < a.nq) a.cnt[qr * a.nseg + split] = wg_cnt[tid];
  }
  if (nstages > 0 && lane == 0 && a.norm_max)
    atomicMax(reinterpret_cast<uint32_t *>(a.norm_max), __float_as_uint(norm_run));   // >= 0
}

// ---- the wide filter, producer / consumer form (round 6) ---------------------------------------------------------------
// rawscan16w_kernel above converts, meets its barrier and scores in lock step: all eight waves do the same thing at the
// same time, the matrix pipe idles while the stage is converted and the vector units idle while it is scored (waves
// parked 48 % of their cycles; two such workgroups per CU were measured and are slower: every stage converted twice).
// Here the ROLES are split inside one workgroup, the structure of dot_interaction_fwd_pc_kernel:
//   waves 4-7  producers: wave p loads rows [32 p, 32 p + 32) of a 128-row stage into registers TWO stages ahead (two
//              register sets: 2 x 64 KiB per CU in flight), converts them to fp16 with one power-of-two scale and one
//              norm bound per 16-row group (the same contract and LDS layout as above) and writes tile (st + 1) & 1;
//   waves 0-3  consumers: each keeps FOUR groups of 32 queries resident (512 queries per workgroup as before) and scores
//              all 128 rows of tile st & 1: 4 sub-tiles x 4 groups x DP / 16 MFMAs = the same 4096 matrix-core cycles per
//              stage and SIMD, now issued by ONE wave per SIMD while the producer wave of that SIMD converts.
// One barrier per stage for all eight waves.  Same survivors, same lists, same counters as the lock-step kernel.

// CW consumer waves of 16 / CW query groups each + four producer waves.  Shipped: <8> = 12 waves, three per SIMD -- two
// consumers that alternate between their MFMA chains and their checks, as the lock-step kernel's waves do while scoring,
// and one producer converting beside them.  Measured (12.5 M x 128, whole call, same box): 257 / 384 / 512 / 640 queries
// 2.52 / 2.75 / 2.98 / 4.95 ms lock-step -> 2.01-2.04 / 2.27-2.30 / 2.46-2.51 / 4.24; 25 M x 64: 3.05 / 3.38 -> 2.47 / 2.99.
// <4> (8 waves, ONE consumer wave per SIMD with four groups) was built first and is no faster than lock-step (3.10 ms at
// 512 queries): a single wave per SIMD leaves the matrix pipe idle while it reduces its own tiles.
template <int DP, int CW>
struct RawPcGeom {
  static constexpr int kQG = 16 / CW;                         // query groups of 32 per consumer wave
  static constexpr int kThreads = (CW + 4) * 64;
  static constexpr int kRowB = DP * 4;
  static constexpr int kPieces = 32 * kRowB / 16 / 64;        // 16-byte pieces per producer lane and stage (32 rows)
  static constexpr int kRow16B = DP * 2 + 16;
  static constexpr int kTile16B = kTileN * kRow16B;
  static constexpr int kMetaOff = 2 * kTile16B;               // [2][8] x float4 {1 / scale, scale, norm, -}
  static constexpr int kCntOff = kMetaOff + 2 * 8 * 16;
  static constexpr int kQcOff = kCntOff + kRawWQueries * 4;   // per (consumer wave, group, lane): {(bound - tiny) / qscale, qk / qscale, qscale, -}
  static constexpr int kLdsBytes = kQcOff + CW * kQG * 64 * 16;
};

template <int DP, int CW>
__global__ void __launch_bounds__((CW + 4) * 64) rawscan16pc_kernel(const RawScanArgs a) {
  using G = RawPcGeom<DP, CW>;
  constexpr int kRawPcQG = G::kQG;
  constexpr int KS = DP / 16;           // MFMA steps of 16 features
  constexpr int PPR = DP / 4;           // 16-byte pieces per f32 row
  constexpr int NP = G::kPieces;
  constexpr int RPI = 64 / PPR;         // rows covered by one load instruction of a wave (2 / 4 / 8 at dim 128 / 64 / 32)
  static_assert(DP >= 32 && NP * RPI == 32 && 16 % RPI == 0, "a producer wave's 32 rows are whole instructions, groups of 16 rows too");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *wg_cnt = reinterpret_cast<uint32_t *>(smem + G::kCntOff);
  float4 *meta = reinterpret_cast<float4 *>(smem + G::kMetaOff);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  if (a.zero_word && blockIdx.x == 0 && tid == 0) *a.zero_word = 0u;
  if (a.zero_aux && blockIdx.x == 0 && tid < 4) a.zero_aux[tid] = 0u;

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
  const int nstages = c0 < c1 ? (int)((c1 - c0 + kTileN - 1) / kTileN) : 0;
  if (tid < kRawWQueries) wg_cnt[tid] = 0u;   // one counter per query of the workgroup
  const uint32_t rows_here = (uint32_t)(c1 > c0 ? c1 - c0 : 0);   // rows of this split (a split holds < 2^31 rows)
  const uint32_t row0 = (uint32_t)c0;                              // (group-local row numbers fit 32 bits)

  if (wave >= CW) {
    // ------------------------------------------------ producers ------------------------------------------------
    const int pw = wave - CW;
    const RawTable *T = a.table;
    const int nblk = T->n_blocks;
    int blk = nstages > 0 ? __builtin_amdgcn_readfirstlane(raw_find_block(T, c0)) : 0;
    int64_t blk_lo = T->row_start[blk], blk_hi = T->row_start[blk + 1];
    const char *blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
    auto load_stage = [&](int st, f32x4 (&dst)[NP]) __attribute__((always_inline)) {
      const int64_t v0 = c0 + (int64_t)st * kTileN;
      while (v0 >= blk_hi && blk + 1 < nblk) {
        ++blk;
        blk_lo = blk_hi;
        blk_hi = T->row_start[blk + 1];
        blk_ptr = reinterpret_cast<const char *>(T->ptr[blk]);
      }
      if (v0 + kTileN <= blk_hi && v0 + kTileN <= c1) {   // the whole stage lies in one block: linear loads
        const char *src = blk_ptr + ((v0 - blk_lo) + 32 * pw) * (int64_t)G::kRowB + lane * 16;
#pragma unroll
        for (int i = 0; i < NP; ++i) dst[i] = *reinterpret_cast<const f32x4 *>(src + i * 1024);
        return;
      }
      // block boundary or the last, partly filled stage: every lane looks its row up; rows at or beyond c1
      // re-read the last valid row (their scores are never used)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int piece = i * 64 + lane;
        int64_t row = v0 + 32 * pw + piece / PPR;
        if (row > c1 - 1) row = c1 - 1;
        dst[i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(raw_row_ptr(T, row, DP)) +
                                                  (piece % PPR) * 16);
        asm volatile("" ::: "memory");   // one address at a time (sixteen 64-bit addresses next to two stages of rows spilled)
      }
    };
    float norm_run = 0.0f;
    // the wave's 32 rows (registers) -> fp16 tile `buf`; the constants of its two 16-row groups -> meta[buf][2 pw + {0, 1}]
    auto convert = [&](const f32x4 (&av)[NP], int buf) __attribute__((always_inline)) {
      float am[2] = {0.0f, 0.0f}, nmax[2] = {0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        constexpr int kHalf = NP / 2;        // pieces i < NP / 2 are rows 0 .. 15 of the wave, the rest rows 16 .. 31
        const int gi = i >= kHalf ? 1 : 0;
        float ss = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ss = __builtin_fmaf(av[i][c], av[i][c], ss);
          am[gi] = fmaxf(am[gi], __builtin_fabsf(av[i][c]));
        }
#pragma unroll
        for (int off = 1; off < PPR && off < 64; off <<= 1) ss += __shfl_xor(ss, off);   // PPR consecutive lanes = one row
        nmax[gi] = fmaxf(nmax[gi], ss);
      }
      float inv[2];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          nmax[gi] = fmaxf(nmax[gi], __shfl_xor(nmax[gi], off));
          am[gi] = fmaxf(am[gi], __shfl_xor(am[gi], off));
        }
        const float nrm = __builtin_sqrtf(nmax[gi]) * kNormSlack;   // (upper bound of the 16 row norms)
        const float cs = pow2_ceil(am[gi]);
        inv[gi] = 1.0f / cs;                                        // exact: power of two
        norm_run = fmaxf(norm_run, nrm);
        if (lane == 0) meta[buf * 8 + 2 * pw + gi] = make_float4(inv[gi], cs, nrm, 0.0f);
      }
      char *t16 = smem + buf * G::kTile16B;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int piece = i * 64 + lane;
        const int row = 32 * pw + piece / PPR, col = piece % PPR;
        const float sc = inv[i >= NP / 2 ? 1 : 0];
        uint2 w;
        w.x = raw_pack_f16x2(av[i][0] * sc, av[i][1] * sc);
        w.y = raw_pack_f16x2(av[i][2] * sc, av[i][3] * sc);
        *reinterpret_cast<uint2 *>(t16 + row * G::kRow16B + col * 8) = w;
      }
    };
    // two register sets (stages of even / odd index, each loaded TWO stages ahead) where 2 x NP x 4 registers fit; with
    // three waves per SIMD at dim 128 (170 registers a wave) ONE set, loaded one stage ahead -- 64 KiB per CU in flight,
    // what the lock-step kernel has
    constexpr bool kTwoSets = !(CW == 8 && DP >= 128);
    f32x4 ra[NP], rb[kTwoSets ? NP : 1];
    if constexpr (kTwoSets) {
      if (nstages > 0) load_stage(0, ra);
      if (nstages > 1) load_stage(1, rb);
      if (nstages > 0) convert(ra, 0);
      if (nstages > 2) load_stage(2, ra);
      __syncthreads();                    // fp16 tile 0 complete (and the counters zeroed)
      for (int st = 0; st < nstages; st += 2) {
        // consumers score stage st (tile 0) now: stage st + 1 -> tile 1
        if (st + 1 < nstages) convert(rb, 1);
        if (st + 3 < nstages) load_stage(st + 3, rb);
        raw_lds_barrier();
        if (st + 1 >= nstages) break;
        // consumers score stage st + 1 (tile 1): stage st + 2 -> tile 0
        if (st + 2 < nstages) convert(ra, 0);
        if (st + 4 < nstages) load_stage(st + 4, ra);
        raw_lds_barrier();
      }
    } else {
      if (nstages > 0) load_stage(0, ra);
      if (nstages > 0) convert(ra, 0);
      if (nstages > 1) load_stage(1, ra);
      __syncthreads();
      for (int st = 0; st < nstages; ++st) {
        if (st + 1 < nstages) convert(ra, (st + 1) & 1);
        if (st + 2 < nstages) load_stage(st + 2, ra);
        raw_lds_barrier();
      }
    }
    if (nstages > 0 && lane == 0 && a.norm_max)
      atomicMax(reinterpret_cast<uint32_t *>(a.norm_max), __float_as_uint(norm_run));   // >= 0
  } else {
    // ------------------------------------------------ consumers ------------------------------------------------
    f16x8r bq[kRawPcQG][KS];
    // the filter constants of a (group, lane) are parked in LDS and re-read once per sub-tile: twelve resident registers
    // next to four query groups and four accumulators spilled at dim 128
    float4 *const qconst = reinterpret_cast<float4 *>(smem + G::kQcOff) + wave * (kRawPcQG * 64);
    const uint32_t qrow0 = (uint32_t)qt * kRawWQueries + (uint32_t)(wave * kRawPcQG) * 32u + (uint32_t)j;
#pragma unroll
    for (int g = 0; g < kRawPcQG; ++g) {
      const int64_t qrow_g = (int64_t)qrow0 + 32 * g;
      const bool qvalid = qrow_g < a.nq;
      const int64_t qr = qvalid ? qrow_g : 0;
      const float qs_ld = a.qscale[qr], thr_ld = a.thr[qr * (a.thr_stride > 0 ? a.thr_stride : 1)], qk_ld = a.qk[qr];   // unconditional: counted loads
      const float qs = qvalid ? qs_ld : 1.0f;
      const float qinv = 1.0f / qs;   // exact: power of two
      const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.q + qr * DP);
#pragma unroll
      for (int m = 0; m < KS; ++m) {
        const f32x4 lo = q4[4 * m + 2 * h], hi = q4[4 * m + 2 * h + 1];   // features 16 m + 8 h .. + 7
        u32x4r w;
        w[0] = raw_pack_f16x2(qvalid ? lo[0] * qinv : 0.0f, qvalid ? lo[1] * qinv : 0.0f);
        w[1] = raw_pack_f16x2(qvalid ? lo[2] * qinv : 0.0f, qvalid ? lo[3] * qinv : 0.0f);
        w[2] = raw_pack_f16x2(qvalid ? hi[0] * qinv : 0.0f, qvalid ? hi[1] * qinv : 0.0f);
        w[3] = raw_pack_f16x2(qvalid ? hi[2] * qinv : 0.0f, qvalid ? hi[3] * qinv : 0.0f);
        bq[g][m] = __builtin_bit_cast(f16x8r, w);
        if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // eight loads at a time
      }
      qconst[g * 64 + lane] = make_float4(
          qvalid ? (thr_ld - kF16Tiny) * qinv : __builtin_inff(),
          qvalid ? qk_ld * qinv : 0.0f, qs, 0.0f);
      __builtin_amdgcn_sched_barrier(0);   // one group's sixteen loads at a time (all 64 issued up front spilled 50 registers)
    }
    const bool wave_active = (int64_t)qt * kRawWQueries + wave * (kRawPcQG * 32) < a.nq;   // (uniform)
    __syncthreads();                    // fp16 tile 0 complete (and the counters zeroed)
    for (int st = 0; st < nstages; ++st) {
      const int buf = st & 1;
      const char *t16 = smem + buf * G::kTile16B;
      if (wave_active) {
        const char *ap = t16 + j * G::kRow16B + h * 16;
        // A fragments in two halves of KS / 2 steps, each right in front of its MFMAs (measured against whole sets fetched
        // a sub-tile ahead, the lock-step kernel's scheme: 2.46-2.51 ms against 2.79 at dim 128, equal at dim 64 -- the
        // finer interleave of LDS reads and matrix instructions wins, and a whole set spilled 10 registers at dim 128)
        constexpr int KH = KS >= 2 ? KS / 2 : KS;
#pragma unroll
        for (int sub = 0; sub < kTileN / 32; ++sub) {
          const float4 mA = meta[buf * 8 + 2 * sub], mB = meta[buf * 8 + 2 * sub + 1];
          f32x16 acc[kRawPcQG];
#pragma unroll
          for (int g = 0; g < kRawPcQG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
#pragma unroll
          for (int hf = 0; hf < KS / KH; ++hf) {
            u32x4r af[KH];
#pragma unroll
            for (int m = 0; m < KH; ++m)
              af[m] = *reinterpret_cast<const u32x4r *>(ap + sub * 32 * G::kRow16B + (hf * KH + m) * 32);
#pragma unroll
            for (int m = 0; m < KH; ++m)
#pragma unroll
              for (int g = 0; g < kRawPcQG; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8r, af[m]), bq[g][hf * KH + m], acc[g], 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < kRawPcQG; ++g) {
            // acc[g][r] = prefilter score of (query qrow[g], stage row 32 sub + (r & 3) + 8 (r >> 2) + 4 h) in units of
            // qscale * (scale of the row's 16-row group): registers 0 .. 7 are rows of group 2 sub, 8 .. 15 of 2 sub + 1
            const f32x16 &c = acc[g];
            const float4 qc = qconst[g * 64 + lane];
            const float thrA = __builtin_fmaf(-qc.y, mA.z, qc.x) * mA.x;
            const float thrB = __builtin_fmaf(-qc.y, mB.z, qc.x) * mB.x;
            const float xa = raw_max8f(c, 0), xb = raw_max8f(c, 8);
            if (__ballot(xa > thrA || xb > thrB) != 0ull) {   // rare once the bound is warm
              const uint32_t rel0 = (uint32_t)st * kTileN + 32u * sub + 4u * h;
              uint2 *const seg = a.buf + (((int64_t)qrow0 + 32 * g) * (int64_t)a.cap_l) * a.nseg + split;
              const float unA = qc.z * mA.y, unB = qc.z * mB.y;
              uint32_t hits = 0u;
#pragma unroll
              for (int r = 0; r < 16; ++r) hits |= c[r] > (r < 8 ? thrA : thrB) ? (1u << r) : 0u;
              if ((uint32_t)(st + 1) * kTileN > rows_here) {   // (uniform) the split's last, partly filled stage
#pragma unroll
                for (int r = 0; r < 16; ++r)
                  if (rel0 + (r & 3) + 8 * (r >> 2) >= rows_here) hits &= ~(1u << r);
              }
              if (hits) {
                uint32_t e = atomicAdd(&wg_cnt[(wave * kRawPcQG + g) * 32 + j], (uint32_t)__builtin_popcount(hits));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  if (hits & (1u << r)) {
                    if (e < a.cap_l)
                      seg[(uint64_t)e * (uint32_t)a.nseg] =
                          make_uint2(__float_as_uint(c[r] * (r < 8 ? unA : unB)), row0 + rel0 + (r & 3) + 8 * (r >> 2));
                    ++e;
                  }
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // one sub-tile at a time
        }
      }
      raw_lds_barrier();
    }
  }

  __syncthreads();
  {   // every (query, split) count is written (counts beyond cap_l flag the query for the exact redo)
    const int64_t qr = (int64_t)qt * kRawWQueries + tid;
    if (tid < kRawWQueries && qr < a.nq) a.cnt[qr * a.nseg + split] = wg_cnt[tid];
  }
}

template <int DP, int CW>
static int launch_rawscan16pc(const RawScanArgs &a, hipStream_t stream) {
  using G = RawPcGeom<DP, CW>;
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&rawscan16pc_kernel<DP, CW>), G::kLdsBytes));
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  hipLaunchKernelGGL((rawscan16pc_kernel<DP, CW>), grid, dim3(G::kThreads), G::kLdsBytes, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

template <int DP>
static int launch_rawscan16w(const RawScanArgs &a, hipStream_t stream) {
  using G = RawWGeom<DP>;
  TFRS_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&rawscan16w_kernel<DP>), G::kLdsBytes));
  const dim3 grid((unsigned)(a.n_qtiles * a.n_splits));
  hipLaunchKernelGGL((rawscan16w_kernel<DP>), grid, dim3(kRawWWaves * 64), G::kLdsBytes, stream, a);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

template <int DP>
static int launch_rawscan16_dp(const RawScanArgs &a, hipStream_t stream) {
  switch (a.qg) {
    case 1: return launch_rawscan16_variant<DP, 1>(a, stream);
    case 2: return launch_rawscan16_variant<DP, 2>(a, stream);
    case 4: return launch_rawscan16_variant<DP, 4>(a, stream);
    case 8:   // (eight resident query groups of dim 128 do not fit the register file: 1.6 KB of scratch)
      if constexpr (DP <= 64) return launch_rawscan16_variant<DP, 8>(a, stream);
      break;
    case 16:  // 512 queries per workgroup: the waves split the queries, one conversion per workgroup
      if constexpr (DP >= 32) {
        const char *wv = option("TFRS_STREAM_RAW16_WIDE");    // pc (producer / consumer waves, default) | lockstep
        if (wv && wv[0] == 'l') return launch_rawscan16w<DP>(a, stream);
        return launch_rawscan16pc<DP, 8>(a, stream);
      }
      break;
  }
  set_error("rawscan16: %d query groups per workgroup (1, 2, 4; 8 up to dim 64)", a.qg);
  return TFRS_EINVAL;
}

int launch_rawscan16(const RawScanArgs &a, hipStream_t stream) {
  if (a.nq <= 0 || a.c_end <= a.c_begin) return TFRS_OK;
  TFRS_CHECK_ARG(a.d == padded_dim(a.d) && a.d >= 8 && a.d <= 128, "rawscan16: dim %d is not one of 8 .. 128", a.d);
  TFRS_CHECK_ARG(a.split_len >= kTileN && a.split_len % kTileN == 0 && a.n_splits >= 1 && a.nseg == a.n_splits,
                 "rawscan16: bad split");
  TFRS_CHECK_ARG(a.thr && a.qk && a.qscale && a.cnt && a.buf, "rawscan16: NULL pointer");
  switch (a.d) {
    case 8: return launch_rawscan16_dp<8>(a, stream);
    case 16: return launch_rawscan16_dp<16>(a, stream);
    case 32: return launch_rawscan16_dp<32>(a, stream);
    case 64: return launch_rawscan16_dp<64>(a, stream);
    default: return launch_rawscan16_dp<128>(a, stream);
  }
}

// ---- fp16 prefilter image straight from the blocks ----------------------------------------------

// One workgroup (256 threads) per stage of kTileN rows (same contract as pack16_stage_kernel,
// topk_pack.hip: x / scale with scale = 2^ceil(log2 max |x|), StageMeta, global max row norm).
// Every candidate byte is read ONCE, coalesced: thread t holds the 16-byte pieces t, t + 256, ... of
// the stage (DP / 8 of them, all loads in flight together); a row's pieces sit in DP / 4 neighbouring
// lanes, so its norm and max |x| are xor-shuffle reductions; after the workgroup has agreed on the
// scale each thread converts the pieces it holds and stores 8 bytes of the image.  (The first version
// walked each row with two threads and re-read the stage for the conversion: 2.4 TB/s.)
// Rows at or beyond n_rows are written as zeros.
template <int DP>
__global__ void __launch_bounds__(256) pack16_raw_kernel(const RawTable *__restrict__ T, int64_t n_rows,
                                                         char *__restrict__ packed16,
                                                         StageMeta *__restrict__ meta,
                                                         float *__restrict__ norm_max) {
  constexpr int kPieces = DP / 8;          // 16-byte pieces per thread
  constexpr int kRowPieces = DP / 4;       // 16-byte pieces per row
  constexpr int kDp16 = DP < 16 ? 16 : DP; // padded_dim16
  constexpr int kRowB16 = kDp16 * 2 + 16;
  __shared__ const float *s_row[kTileN];
  __shared__ float s_red[2][4];
  __shared__ float s_scale;
  const int64_t stage = blockIdx.x;
  const int64_t row0 = stage * kTileN;
  const int tid = threadIdx.x;
  if (tid < kTileN) {
    // (rows past the end borrow the stage's first row -- always valid -- so that every load below is
    // unconditional; their values are zeroed after the load)
    const int64_t row = row0 + tid;
    s_row[tid] = raw_row_ptr(T, row < n_rows ? row : row0, DP);
  }
  __syncthreads();
  f32x4 v[kPieces];
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int x = i * 256 + tid;
    const int r = x / kRowPieces, c4 = x - r * kRowPieces;
    v[i] = reinterpret_cast<const f32x4 *>(s_row[r])[c4];
  }
  float nm2 = 0.0f, am = 0.0f;   // max over this thread's rows of (sum of squares, max |x|)
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int r = (i * 256 + tid) / kRowPieces;
    if (row0 + r >= n_rows) v[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float ssq = 0.0f;
    ssq = __builtin_fmaf(v[i][0], v[i][0], ssq);
    ssq = __builtin_fmaf(v[i][1], v[i][1], ssq);
    ssq = __builtin_fmaf(v[i][2], v[i][2], ssq);
    ssq = __builtin_fmaf(v[i][3], v[i][3], ssq);
    float amax = fmaxf(fmaxf(__builtin_fabsf(v[i][0]), __builtin_fabsf(v[i][1])),
                       fmaxf(__builtin_fabsf(v[i][2]), __builtin_fabsf(v[i][3])));
#pragma unroll
    for (int off = 1; off < kRowPieces; off <<= 1) {   // the row's pieces: kRowPieces neighbouring lanes
      ssq += __shfl_xor(ssq, off);
      amax = fmaxf(amax, __shfl_xor(amax, off));
    }
    nm2 = fmaxf(nm2, ssq);
    am = fmaxf(am, amax);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nm2 = fmaxf(nm2, __shfl_xor(nm2, off));
    am = fmaxf(am, __shfl_xor(am, off));
  }
  if ((tid & 63) == 0) {
    s_red[0][tid >> 6] = nm2;
    s_red[1][tid >> 6] = am;
  }
  __syncthreads();
  if (tid == 0) {
    const float n2 = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
    const float a_ = fmaxf(fmaxf(s_red[1][0], s_red[1][1]), fmaxf(s_red[1][2], s_red[1][3]));
    // (the sum of squares of a row is accumulated in a different order than pack16_stage_kernel does:
    // kNormSlack covers the rounding of either; the norm only has to be an upper bound)
    const float nm = __builtin_sqrtf(n2) * kNormSlack;
    const float sc = pow2_ceil(a_);
    s_scale = sc;
    meta[stage].norm = nm;
    meta[stage].scale = sc;
    meta[stage].inv_scale = 1.0f / sc;
    meta[stage].pad_ = 0.0f;
    atomicMax(reinterpret_cast<uint32_t *>(norm_max), __float_as_uint(nm));   // nm >= 0
  }
  __syncthreads();
  const float inv = 1.0f / s_scale;   // exact: power of two
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int x = i * 256 + tid;
    const int r = x / kRowPieces, c4 = x - r * kRowPieces;
    const uint2 w = make_uint2(raw_pack_f16x2(v[i][0] * inv, v[i][1] * inv), raw_pack_f16x2(v[i][2] * inv, v[i][3] * inv));
    *reinterpret_cast<uint2 *>(packed16 + (row0 + r) * (int64_t)kRowB16 + c4 * 8) = w;
  }
  if (tid < kTileN) {   // the zero tail of every row: padding features (DP = 8) and the 16-byte pad slot
    char *tail = packed16 + (row0 + tid) * (int64_t)kRowB16 + DP * 2;
#pragma unroll
    for (int b = 0; b < kRowB16 - DP * 2; b += 16) *reinterpret_cast<uint4 *>(tail + b) = make_uint4(0u, 0u, 0u, 0u);
  }
}

template <int DP>
static int launch_pack16_raw_dp(const RawTable *table, int64_t n_rows, char *packed16, StageMeta *meta,
                                float *norm_max, hipStream_t stream) {
  const int64_t stages = (n_rows + kTileN - 1) / kTileN;
  hipLaunchKernelGGL((pack16_raw_kernel<DP>), dim3((unsigned)stages), dim3(256), 0, stream, table, n_rows,
                     packed16, meta, norm_max);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

int launch_pack16_raw(const RawTable *table, int64_t n_rows, int d, char *packed16, StageMeta *meta,
                      float *norm_max, hipStream_t stream) {
  if (n_rows <= 0) return TFRS_OK;
  switch (d) {
    case 8: return launch_pack16_raw_dp<8>(table, n_rows, packed16, meta, norm_max, stream);
    case 16: return launch_pack16_raw_dp<16>(table, n_rows, packed16, meta, norm_max, stream);
    case 32: return launch_pack16_raw_dp<32>(table, n_rows, packed16, meta, norm_max, stream);
    case 64: return launch_pack16_raw_dp<64>(table, n_rows, packed16, meta, norm_max, stream);
    case 128: return launch_pack16_raw_dp<128>(table, n_rows, packed16, meta, norm_max, stream);
  }
  set_error("pack16_raw: dim %d is not one of 8, 16, 32, 64, 128", d);
  return TFRS_EINVAL;
}

}  // namespace tfrs
