// softmax16.hip -- the default in-batch sampled-softmax path: split-fp16 MFMA GEMMs.
//
// Same contract as softmax.hip (tasks/retrieval.py:172-210 and its gradient, models/base.py:77)
// for the default Retrieval configuration (optional sample weights and temperature; the
// sampling-probability correction, accidental-hit removal and score mask stay on the f32
// kernels).  The f32-input MFMA runs at 157 TFLOP/s, the fp16 one at 2.5 PFLOP/s: every f32
// operand x is therefore split once per step into two fp16 numbers, x * 2^a = hi + lo with
// |x * 2^a - hi - lo| <= 2^-22 |x * 2^a| (a = per-row power of two that puts the row's largest
// magnitude in [2^9, 2^10), so nothing under- or overflows in fp16), and every product of the
// three GEMMs of a step is formed as  hi*hi + hi*lo + lo*hi  on the fp16 matrix cores with f32
// accumulation: f32-level accuracy (the dropped lo*lo term is 2^-22 relative) at a third of
// the fp16 rate, about 5x the f32 MFMA rate.  Power-of-two scales are undone exactly.
//
// Data layout: each matrix becomes an array of 32-row RECORDS that already have the LDS layout
// of one streamed tile,
//   [hi rows | lo rows | inv scale | lse2 | w*inv | transposed hi | transposed lo]
// (row-major rows padded to an odd number of 16-byte slots -> conflict-free ds_read_b128; the
// transposed part is the second GEMM's operand: lane = feature, 8 contiguous halves = the 8
// streamed rows of one MFMA k-group, in accumulator-register order), so a tile is staged with
// straight direct-to-LDS copies (global_load_lds, no registers) into a ring of LDS buffers that
// runs up to 2 tiles ahead of the MFMAs.
//
// Kernels of one step:
//   prep      q, c (f32) -> records, per-block largest scale exponent        (1 launch)
//   fwd       workgroup = 4 waves x 32 owned query rows, streams candidate tiles; S tile =
//             3*D/16 MFMAs; online base-2 max / sum-exp per row in registers
//   finalize  combines splits, writes lse / pos / the weighted loss (deterministic) and the
//             per-query lse2 the backward needs into the query records
//   bwd x2    (rows = queries -> dq, rows = candidates -> dc): recomputes the S tile, forms
//             T = (softmax - onehot) * streamed-row factor in the accumulator layout, splits it
//             to fp16 and feeds it straight back as the B operand of the second GEMM
//             out^T[feature][row] += X^T T  (3*D/16 more MFMAs); no transposes, no atomics
//   reduce    sums the per-split partial gradients in split order             (1 launch)
//
// Roofline: MFMA-bound in the large-batch limit, 3 * (2 nq nc d) MFMA flop forward and
// 3 * (4 nq nc d) per backward kernel, priced against the dense fp16 peak; the ALGORITHMIC
// flop are 2 nq nc d and 8 nq nc d as before.
#include <stdlib.h>

#include <algorithm>

#include "mfma_tile.h"

namespace tfrs {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Byte layout of one 32-row record (= one LDS ring buffer).
template <int DP>
struct Rec16 {
  static constexpr int kRowB = DP * 2 + 16;   // row-major row: odd number of 16-B slots
  static constexpr int kXtB = 64 + 16;        // transposed row: 32 halves + pad
  static constexpr int kHi = 0;
  static constexpr int kLo = 32 * kRowB;
  static constexpr int kInv = 64 * kRowB;     // float[32]  2^-a (undoes the row scale)
  static constexpr int kLse = kInv + 128;     // float[32]  lse * log2(e)      (query records)
  static constexpr int kWq = kLse + 128;      // float[32]  w * 2^-a           (query records)
  static constexpr int kXh = kWq + 128;
  static constexpr int kXl = kXh + DP * kXtB;
  static constexpr int kBytes = kXl + DP * kXtB;
  static constexpr int kFwdBytes = kLse;      // the forward streams [hi | lo | inv] only
  static_assert(kBytes % 64 == 0 && kFwdBytes % 64 == 0, "records split evenly over 4 waves");
};

struct Side16 {
  const char *rec;        // [np / 32] records
  const uint32_t *bmax;   // [np / 32] per record: biased exponent of its largest |2^floor(log2|w|/2) * x| (0: zero record)
  int64_t n, np;
};

struct Sm16Args {
  Side16 q, c;
  int d;
  const float *w;
  float inv_t;
  int nsplit;
  int64_t split_len;
  float *pm, *pl, *ppos;
  const float *lse;
  const float *gloss;
  float *partial;
  char *q_rec_w;          // writable view of the query records (finalize fills lse2)
};

__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// streamed-row position p = kk*16 + h*8 + e of the transposed image <-> row of the 32-row tile
// held by accumulator register r = 8*kk + e of lane half h (tile_row_of_reg)
__host__ __device__ inline int xt_row_of_pos(int p) {
  const int kk = p >> 4, hh = (p >> 3) & 1, e = p & 7;
  return (e & 3) + 8 * ((e >> 2) + 2 * kk) + 4 * hh;
}

// ---- prep ------------------------------------------------------------------------------
struct PrepSide {
  const float *x;
  int64_t n;
  const float *w;
  char *rec;
  uint32_t *bmax;
};

// One launch converts both matrices: blocks [0, q_blocks) take 32-row blocks of q, the rest of c.
template <int DP>
__global__ void __launch_bounds__(256) sm16_prep_kernel(const PrepSide sq, const PrepSide sc,
                                                        int q_blocks, int d,
                                                        uint32_t *__restrict__ ticket) {
  typedef Rec16<DP> RL;
  __shared__ float tile[32][DP + 1];
  __shared__ float s_scale[32], s_xw[32], s_tw[32];
  __shared__ uint32_t s_exp[32];
  // first kernel of the chain: re-arms the finalize kernel's ticket (a hipMemsetAsync node is
  // not reliably ordered against kernel nodes when the step is replayed from a HIP graph)
  if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;
  const bool is_q = (int)blockIdx.x < q_blocks;
  const PrepSide &sd = is_q ? sq : sc;
  const int blk = is_q ? (int)blockIdx.x : (int)blockIdx.x - q_blocks;
  const float *__restrict__ x = sd.x;
  const float *__restrict__ w = sd.w;
  const int64_t n = sd.n;
  char *__restrict__ rec = sd.rec + (int64_t)blk * RL::kBytes;
  const int64_t r0 = (int64_t)blk * 32;
  // All loads of a thread are issued before the first LDS write, and unconditionally (clamped, then selected): the
  // runtime loop this replaces -- `tile = in range ? x[..] : 0` per element -- was compiled to one 4-byte load and one
  // memory round trip per iteration, eight in a row at dim 64 (most of the kernel's 7-8 us at the quickstart shapes).
  if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    constexpr int kIt = (32 * DP / 4 + 255) / 256;
    float4 v[kIt];
#pragma unroll
    for (int i = 0; i < kIt; ++i) {
      const int c = threadIdx.x + i * 256;
      const int row = c / (DP / 4), f4 = c - row * (DP / 4);
      const bool ok = c < 32 * DP / 4 && r0 + row < n && 4 * f4 < d;
      v[i] = *reinterpret_cast<const float4 *>(x + (ok ? (r0 + row) * d + 4 * f4 : 0));
      if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < kIt; ++i) {
      const int c = threadIdx.x + i * 256;
      const int row = c / (DP / 4), f4 = c - row * (DP / 4);
      if (c < 32 * DP / 4) {
        tile[row][4 * f4] = v[i].x; tile[row][4 * f4 + 1] = v[i].y; tile[row][4 * f4 + 2] = v[i].z; tile[row][4 * f4 + 3] = v[i].w;
      }
    }
  } else {
    constexpr int kIt = 32 * DP / 256;
    float v[kIt];
#pragma unroll
    for (int i = 0; i < kIt; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int row = idx / DP, f = idx - row * DP;
      const bool ok = r0 + row < n && f < d;
      v[i] = x[ok ? (r0 + row) * d + f : 0];
      if (!ok) v[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < kIt; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int row = idx / DP, f = idx - row * DP;
      tile[row][f] = v[i];
    }
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 3, part = threadIdx.x & 7;
    float m = 0.0f;
    for (int f = part; f < DP; f += 8) m = fmaxf(m, fabsf(tile[row][f]));
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    m = fmaxf(m, __shfl_xor(m, 4));
    const uint32_t e = (f2u(m) >> 23) & 0xffu;
    float s = 1.0f, iv = 1.0f;   // rows below 2^-100 flush to zero in fp16: treated as zero rows
    if (e >= 27u && e < 255u) {
      s = u2f((263u - e) << 23);   // largest magnitude of the row -> [2^9, 2^10)
      iv = u2f((e - 9u) << 23);
    }
    if (part == 0) {
      s_scale[row] = s;
      const float wr = (w && r0 + row < n) ? w[r0 + row] : 1.0f;
      reinterpret_cast<float *>(rec + RL::kInv)[row] = iv;
      reinterpret_cast<float *>(rec + RL::kLse)[row] = 0.0f;      // finalize overwrites (queries)
      // Second-GEMM operands (the backward's out^T += X^T T, contraction over the STREAMED rows).
      // A factor that varies along the contraction can live in either operand; each fp16 hi + lo
      // pair is accurate to 2^-22 of its own value over ~23 binades only, so the weight is split
      // between the two: X carries the power of two 2^floor(log2|w| / 2), T the rest (|tw| in
      // [1, 4)) -- sample weights spanning 4 decades cost each operand 7 binades instead of one
      // operand 13 -- and the transposed image takes ONE scale per record (below), so that T
      // carries no per-row data magnitude at all.  (Round 3: T = p * w * 2^-a_row under one scale:
      // an entry whose own terms were small was accurate only relative to its neighbours'.)
      const uint32_t ewb = (f2u(wr) >> 23) & 0xffu;
      float xw = 1.0f, tw = wr;
      if (ewb >= 1u && ewb < 255u) {
        const int hw = ((int)ewb - 127) >> 1;                     // floor(log2 |w| / 2)
        xw = u2f((uint32_t)(hw + 127) << 23);
        tw = wr * u2f((uint32_t)(127 - hw) << 23);
      }
      s_xw[row] = xw;
      s_tw[row] = tw;
      uint32_t ev = 0u;
      if (r0 + row < n) ev = (f2u(xw * m) >> 23) & 0xffu;         // exponent of the row's largest |xw * x|
      s_exp[row] = ev;
    }
  }
  __syncthreads();
  {
    // the record's exponent (wave-uniform work done by every thread: 32 LDS reads)
    uint32_t ev = 0u;
    for (int r = 0; r < 32; ++r) ev = s_exp[r] > ev ? s_exp[r] : ev;
    float srec = 0.0f, xs = 0.0f;     // records below 2^-100: zero image, no contribution
    if (ev >= 27u && ev < 255u) {
      srec = u2f((263u - ev) << 23);  // largest |xw * x| of the record -> [2^9, 2^10)
      xs = u2f((ev - 9u) << 23);      // ... and its inverse
    }
    if (threadIdx.x == 0) sd.bmax[blk] = (ev >= 27u && ev < 255u) ? ev : 0u;
    // per streamed row: the factor T takes = (rest of the weight) * (inverse image scale)
    if (threadIdx.x < 32) reinterpret_cast<float *>(rec + RL::kWq)[threadIdx.x] = s_tw[threadIdx.x] * xs;
    for (int idx = threadIdx.x; idx < 32 * DP; idx += 256) {
      const int f = idx >> 5, p = idx & 31;
      const int srow = xt_row_of_pos(p);
      const float v = tile[srow][f] * (s_xw[srow] * srec);
      const _Float16 vh = (_Float16)v;
      reinterpret_cast<_Float16 *>(rec + RL::kXh + f * RL::kXtB)[p] = vh;
      reinterpret_cast<_Float16 *>(rec + RL::kXl + f * RL::kXtB)[p] = (_Float16)(v - (float)vh);
    }
  }
  for (int idx = threadIdx.x; idx < 32 * DP; idx += 256) {
    const int row = idx / DP, f = idx - row * DP;
    const float v = tile[row][f] * s_scale[row];
    const _Float16 vh = (_Float16)v;
    reinterpret_cast<_Float16 *>(rec + RL::kHi + row * RL::kRowB)[f] = vh;
    reinterpret_cast<_Float16 *>(rec + RL::kLo + row * RL::kRowB)[f] = (_Float16)(v - (float)vh);
  }
}

// ---- shared pieces of the streaming kernels -------------------------------------------------
// Direct-to-LDS copy of the first BYTES of a record; every wave moves one NW-th with the same
// number of wave-instructions (1 KiB each, the last one partially masked), so a fixed vmcnt
// tells every wave how many of its copies are still in flight.
template <int BYTES, int NW>
struct Copy16 {
  static constexpr int kPart = BYTES / NW;    // bytes copied by one wave
  static constexpr int kInstr = (kPart + 1023) / 1024;
  static_assert(BYTES % NW == 0 && kPart % 16 == 0, "each wave copies whole 16-byte pieces");
  static_assert(kPart % 1024 != 0, "the masked tail keeps every instruction non-empty");
};

// The copy is issued through inline assembly on purpose: for the builtin the compiler cannot
// prove that the copy into one ring buffer is independent of the ds_reads of another (same
// loop, rotating roles) and serialises them with s_waitcnt vmcnt(0), which would expose the
// full memory latency the ring is there to hide.  Completion is tracked by hand instead
// (wait_tile below): copies of one wave complete in issue order, and every wave issues the
// same number per tile.
__device__ __forceinline__ void glds_copy16(const char *gsrc_lane, const char *lds_wave_base) {
  const uint32_t m0v = (uint32_t)(uintptr_t)(
      __attribute__((address_space(3))) const char *)lds_wave_base;
  // M0 (the LDS base of the copy) is a reserved register the compiler does not track through
  // inline assembly: it is saved and restored around the instruction.
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "v"(gsrc_lane), "s"(m0v)
               : "memory");
}

template <int BYTES, int NW>
__device__ __forceinline__ void stage_glds(const char *src, char *dst, int wave, int lane) {
  typedef Copy16<BYTES, NW> C;
#pragma unroll
  for (int i = 0; i < C::kInstr; ++i) {
    const int off = wave * C::kPart + i * 1024;
    if (i * 1024 + lane * 16 < C::kPart) glds_copy16(src + off + lane * 16, dst + off);
  }
}

// s_waitcnt vmcnt(n) alone (expcnt / lgkmcnt untouched); n < 64
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// waits until at most `newer` later tiles' copies of this wave are in flight, then meets the
// other waves: after the barrier the tile is complete in LDS and the buffer that the next
// prefetch overwrites is no longer being read by anyone
template <int INSTR, int MAXNEWER>
__device__ __forceinline__ void wait_tile(int newer) {
  if (MAXNEWER >= 1 && newer >= 1) {
    wait_vm<INSTR>();
  } else {
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
}

// S tile (32 streamed rows x 32 owned rows) from the LDS image of the streamed rows.
template <int DP>
__device__ __forceinline__ f32x16 tile_dot16(const char *buf, const h8 (&bh)[DP / 16],
                                             const h8 (&bl)[DP / 16], int j, int h) {
  typedef Rec16<DP> RL;
  // all A fragments of the tile are fetched before the first MFMA: the chain then never waits
  // on LDS latency between k-steps
  h8 ah[DP / 16], al[DP / 16];
#pragma unroll
  for (int i = 0; i < DP / 16; ++i) {
    ah[i] = *reinterpret_cast<const h8 *>(buf + RL::kHi + j * RL::kRowB + (2 * i + h) * 16);
    al[i] = *reinterpret_cast<const h8 *>(buf + RL::kLo + j * RL::kRowB + (2 * i + h) * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int i = 0; i < DP / 16; ++i) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[i], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[i], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[i], acc, 0, 0, 0);
  }
  return acc;
}

// owned rows' B operands from their record; returns the row's inverse scale
template <int DP>
__device__ __forceinline__ float load_owned(h8 (&bh)[DP / 16], h8 (&bl)[DP / 16], const Side16 &R,
                                            int64_t row, int h) {
  typedef Rec16<DP> RL;
  const char *rec = R.rec + (row >> 5) * RL::kBytes;
  const int rr = (int)(row & 31);
#pragma unroll
  for (int i = 0; i < DP / 16; ++i) {
    bh[i] = *reinterpret_cast<const h8 *>(rec + RL::kHi + rr * RL::kRowB + (2 * i + h) * 16);
    bl[i] = *reinterpret_cast<const h8 *>(rec + RL::kLo + rr * RL::kRowB + (2 * i + h) * 16);
  }
  return reinterpret_cast<const float *>(rec + RL::kInv)[rr];
}

// ---- forward ------------------------------------------------------------------------------
// NW waves x 32 owned rows per workgroup: 4 for small batches (more workgroups), 8 for large ones
// (every staged tile then serves twice the rows: the streaming traffic per flop halves).
template <int DP, int NW>
__global__ void __launch_bounds__(NW * 64) sm16_fwd_kernel(const Sm16Args a) {
  typedef Rec16<DP> RL;
  constexpr int NB = 3;                               // ring depth: copies run 2 tiles ahead
  constexpr int kInstr = Copy16<RL::kFwdBytes, NW>::kInstr;
  // separate LDS objects: the compiler can then tell that the copy into one buffer does not
  // alias the reads of another and does not serialise them with s_waitcnt vmcnt(0)
  __shared__ __attribute__((aligned(16))) char ring0[RL::kFwdBytes];
  __shared__ __attribute__((aligned(16))) char ring1[RL::kFwdBytes];
  __shared__ __attribute__((aligned(16))) char ring2[RL::kFwdBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int64_t rb = blockIdx.x / a.nsplit;
  const int sp = (int)(blockIdx.x - rb * a.nsplit);
  const int64_t base32 = rb * (NW * 32) + wave * 32;
  const int64_t row = base32 + j;
  const bool rvalid = row < a.q.n;

  const int64_t c_lo = (int64_t)sp * a.split_len;
  int64_t c_hi = c_lo + a.split_len;
  if (c_hi > a.c.n) c_hi = a.c.n;
  const int nt = (int)((c_hi - c_lo + 31) / 32);
  const char *src = a.c.rec + (c_lo >> 5) * RL::kBytes;

  if (nt > 0) stage_glds<RL::kFwdBytes, NW>(src, ring0, wave, lane);
  if (nt > 1) stage_glds<RL::kFwdBytes, NW>(src + RL::kBytes, ring1, wave, lane);

  h8 bh[DP / 16], bl[DP / 16];
  const float rowfac2 = load_owned<DP>(bh, bl, a.q, row, h) * a.inv_t * kLog2e;

  float m = -__builtin_inff(), l = 0.0f, pos2 = 0.0f;
  bool haspos = false;
  // every ordinary load of the prologue lands here, OUTSIDE the loop: a compiler-placed
  // vmcnt(0) at their first use inside the loop body would also drain the ring's copies on
  // every iteration
  wait_vm<0>();

  auto step = [&](const char *cur, char *pre, int t) __attribute__((always_inline)) {
    wait_tile<kInstr, NB - 2>(nt - 1 - t);
    if (t + NB - 1 < nt)
      stage_glds<RL::kFwdBytes, NW>(src + (int64_t)(t + NB - 1) * RL::kBytes, pre, wave, lane);
    const int64_t s0 = c_lo + (int64_t)t * 32;
    const f32x16 acc = tile_dot16<DP>(cur, bh, bl, j, h);
    const float *sinv = reinterpret_cast<const float *>(cur + RL::kInv);
    float v2[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 sv = *reinterpret_cast<const f32x4 *>(sinv + 8 * g + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) v2[4 * g + k] = acc[4 * g + k] * (sv[k] * rowfac2);
    }
    if (s0 + 32 > c_hi) {   // ragged last tile (wave-uniform and rare: kept a real branch)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (s0 + tile_row_of_reg(r, h) >= c_hi) v2[r] = -__builtin_inff();
    }
    if (s0 == base32) {     // the tile that holds the positives of this wave's rows
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (tile_row_of_reg(r, h) == j) {
          pos2 = v2[r];
          haspos = true;
        }
    }
    float tmax = v2[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, v2[r]);
    if (tmax > m) {
      l *= fast_exp2(m - tmax);   // m = -inf on the first tile: exp2(-inf) = 0 and l = 0
      m = tmax;
    }
    if (m > -__builtin_inff()) {
      float l0 = 0.0f, l1 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        l0 += fast_exp2(v2[r] - m);
        l1 += fast_exp2(v2[r + 1] - m);
      }
      l += l0 + l1;
    }
  };
  for (int t = 0; t < nt; t += NB) {
    step(ring0, ring2, t);
    if (t + 1 < nt) step(ring1, ring0, t + 1);
    if (t + 2 < nt) step(ring2, ring1, t + 2);
  }

  const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32);
  const float mm = fmaxf(m, m2);
  float ll = 0.0f;
  if (m > -__builtin_inff()) ll += l * fast_exp2(m - mm);
  if (m2 > -__builtin_inff()) ll += l2 * fast_exp2(m2 - mm);
  if (h == 0 && rvalid) {
    a.pm[(int64_t)sp * a.q.n + row] = mm;     // base-2 domain
    a.pl[(int64_t)sp * a.q.n + row] = ll;
  }
  if (haspos && rvalid) a.ppos[row] = pos2 * kLn2;
}

// Combines the per-split base-2 (max, sum) pairs, writes lse / pos and the weighted loss, and
// leaves lse * log2(e) in the query records for the backward.  One wave per 64 rows; the
// splits' partials are fetched in batches of 16 independent loads (one memory latency per
// batch instead of one per split).  Every block leaves a partial loss in block_part[] and the
// LAST block to arrive (ticket, re-armed by the prep kernel) adds the partials in block order:
// the loss does not depend on scheduling.
template <int DP>
__global__ void __launch_bounds__(64) sm16_finalize_kernel(const Sm16Args a, float *out_loss,
                                                           float *out_lse, float *out_pos,
                                                           double *block_part, uint32_t *ticket) {
  typedef Rec16<DP> RL;
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool valid = row < a.q.n;
  const int64_t r = valid ? row : 0;
  // (the positive and the weight travel with the first batch of partials: behind the loop they were a round trip of their own)
  const float pos_ld = a.ppos[r];
  const float w_ld = a.w ? a.w[r] : 1.0f;
  float mm = -__builtin_inff(), ll = 0.0f;
  for (int s0 = 0; s0 < a.nsplit; s0 += 16) {
    float pm[16], pl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const bool ok = s0 + k < a.nsplit;
      pm[k] = ok ? a.pm[(int64_t)(s0 + k) * a.q.n + r] : -__builtin_inff();
      pl[k] = ok ? a.pl[(int64_t)(s0 + k) * a.q.n + r] : 0.0f;
    }
    float bm = pm[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) bm = fmaxf(bm, pm[k]);
    if (bm > mm) {
      ll *= fast_exp2(mm - bm);
      mm = bm;
    }
    if (mm > -__builtin_inff()) {
#pragma unroll
      for (int k = 0; k < 16; ++k) ll += pl[k] * fast_exp2(pm[k] - mm);   // pm = -inf: + 0
    }
  }
  double local = 0.0;
  if (valid) {
    const float lse2 = mm + log2f(ll);
    const float lse = lse2 * kLn2;
    const float pos = pos_ld;
    out_lse[row] = lse;
    out_pos[row] = pos;
    reinterpret_cast<float *>(a.q_rec_w + (row >> 5) * RL::kBytes + RL::kLse)[row & 31] = lse * kLog2e;
    const float w = w_ld;
    local = (double)w * ((double)lse - (double)pos);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);   // fixed tree: reproducible
  uint32_t t = 0;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&block_part[blockIdx.x], local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  }
  t = __shfl(t, 0);
  if (t == gridDim.x - 1) {
    // last block to arrive: all 64 lanes fetch the partials (64 independent loads per pass, not a
    // serial walk by one lane) and add them in a fixed order -- lane l takes blocks l, l + 64, ...
    // in order, then the fixed xor tree -- so the loss does not depend on scheduling
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double total = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 64)
      total += __hip_atomic_load(&block_part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (threadIdx.x == 0) {
      *out_loss = (float)total;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The backward alone (no forward in this workspace): lse comes from the caller; this fills the
// lse2 slots of the query records that the finalize kernel would have written.
template <int DP>
__global__ void __launch_bounds__(256) sm16_fill_lse_kernel(const float *lse, int64_t nq, char *q_rec) {
  typedef Rec16<DP> RL;
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row < nq)
    reinterpret_cast<float *>(q_rec + (row >> 5) * RL::kBytes + RL::kLse)[row & 31] = lse[row] * kLog2e;
}

// ---- backward -----------------------------------------------------------------------------
// RQ = true : workgroup owns 128 queries, streams candidates, emits partial dq.
// RQ = false: workgroup owns 128 candidates, streams queries, emits partial dc.
template <int DP, bool RQ, int NW>
__device__ __forceinline__ void sm16_bwd_body(const Sm16Args &a, const int block, char *ring0,
                                              char *ring1, char *ring2, uint32_t *s_e,
                                              float *scr_all) {
  typedef Rec16<DP> RL;
  constexpr int NFB = DP / 32;
  constexpr int NB = DP <= 64 ? 3 : 2;
  constexpr int kInstr = Copy16<RL::kBytes, NW>::kInstr;
  const Side16 &R = RQ ? a.q : a.c;
  const Side16 &S = RQ ? a.c : a.q;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int64_t rb = block / a.nsplit;
  const int sp = (int)(block - rb * a.nsplit);
  const int64_t base32 = rb * (NW * 32) + wave * 32;
  const int64_t row = base32 + j;
  const bool rvalid = row < R.n;

  const int64_t s_lo = (int64_t)sp * a.split_len;
  int64_t s_hi = s_lo + a.split_len;
  if (s_hi > S.n) s_hi = S.n;
  const int nt = (int)((s_hi - s_lo + 31) / 32);
  const char *src = S.rec + (s_lo >> 5) * RL::kBytes;

  if (nt > 0) stage_glds<RL::kBytes, NW>(src, ring0, wave, lane);
  if (NB > 2 && nt > 1) stage_glds<RL::kBytes, NW>(src + RL::kBytes, ring1, wave, lane);

  h8 bh[DP / 16], bl[DP / 16];
  const float rowfac2 = load_owned<DP>(bh, bl, R, row, h) * a.inv_t * kLog2e;

  // T = (softmax - onehot) * (streamed-row factor) goes to the second GEMM as fp16 hi + lo: accurate to
  // 2^-22 of a value only while that value sits in the upper ~27 binades of the fp16 range.  The scale
  // is therefore kept PER OWNED ROW (= per lane: a column of the T tile, a factor on the dimension that is
  // not contracted, undone exactly in the epilogue) and follows the largest |T| this row has met so far,
  // like the running maximum of an online softmax: when a tile brings a larger value the row's
  // accumulators are rescaled by the power of two.  An owned row whose terms are ALL tiny -- a candidate
  // no query likes: p ~ 1e-9 in every tile -- is then as accurate relative to its own terms as any
  // other (round 3 used one scale per streamed side: such rows fell into the fp16 subnormals).
  int gcur = 100;           // T is multiplied by 2^gcur (lowered as larger values arrive)
  float lse2_r = 0.0f, w_r = 1.0f;
  if (RQ && rvalid) {
    lse2_r = a.lse[row] * kLog2e;
    if (a.w) w_r = a.w[row];
  }
  const float gl = (a.gloss ? *a.gloss : 1.0f) * a.inv_t;
  wait_vm<0>();   // prologue loads land outside the loop (see the forward kernel)

  f32x16 outacc[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
    for (int r = 0; r < 16; ++r) outacc[fb][r] = 0.0f;

  auto step = [&](const char *cur, char *pre, int t) __attribute__((always_inline)) {
    wait_tile<kInstr, NB - 2>(nt - 1 - t);
    if (t + NB - 1 < nt)
      stage_glds<RL::kBytes, NW>(src + (int64_t)(t + NB - 1) * RL::kBytes, pre, wave, lane);
    const int64_t s0 = s_lo + (int64_t)t * 32;
    const f32x16 acc = tile_dot16<DP>(cur, bh, bl, j, h);
    const float *fl = reinterpret_cast<const float *>(cur + RL::kInv);
    float tp[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 sv = *reinterpret_cast<const f32x4 *>(fl + 8 * g + 4 * h);
      f32x4 ls = {lse2_r, lse2_r, lse2_r, lse2_r};
      if (!RQ) ls = *reinterpret_cast<const f32x4 *>(fl + 32 + 8 * g + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        tp[4 * g + k] = fast_exp2(acc[4 * g + k] * (sv[k] * rowfac2) - ls[k]);
    }
    if (s0 == base32) {     // the tile that holds the positives: softmax - onehot
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (tile_row_of_reg(r, h) == j) tp[r] -= 1.0f;
    }
    if (s0 + 32 > s_hi) {   // ragged last tile: rows past the end contribute nothing
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (s0 + tile_row_of_reg(r, h) >= s_hi) tp[r] = 0.0f;
    }
    // streamed-row factor (prep: rest of the weight of a streamed query, times the inverse scale of
    // the record's transposed image)
    float mx = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 tf = *reinterpret_cast<const f32x4 *>(fl + 64 + 8 * g + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tp[4 * g + k] *= tf[k];
        mx = fmaxf(mx, __builtin_fabsf(tp[4 * g + k]));
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));               // the other 16 streamed rows of this owned row
    {
      // |T| < 2^(need + 1): the scale that keeps it below 2^14 is 2^(13 - need)
      const int need = (int)((f2u(mx) >> 23) & 0xffu) - 127;
      const bool lower = (13 - need) < gcur;
      if (__ballot(lower) != 0ull) {                   // rare after the first tiles of a sweep
        int gnew = 11 - need;                          // two binades of head-room: fewer rescales
        gnew = gnew < -100 ? -100 : gnew;
        // (a drop of more than 126 binades only happens from the initial scale, with nothing accumulated)
        const int dg = gnew - gcur;
        const float fac = lower ? (dg < -126 ? 0.0f : u2f((uint32_t)(dg + 127) << 23)) : 1.0f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int r = 0; r < 16; ++r) outacc[fb][r] *= fac;
        gcur = lower ? gnew : gcur;
      }
    }
    const float tsc = u2f((uint32_t)(gcur + 127) << 23);
    h8 th[2], tl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = tp[r] * tsc;
      const _Float16 vh = (_Float16)v;
      th[r >> 3][r & 7] = vh;
      tl[r >> 3][r & 7] = (_Float16)(v - (float)vh);
    }
    // out^T[feature][owned row] += sum over the tile's streamed rows X'[srow][feature] T[srow][row]
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int off = (fb * 32 + j) * RL::kXtB + (kk * 16 + h * 8) * 2;
        const h8 ah = *reinterpret_cast<const h8 *>(cur + RL::kXh + off);
        const h8 al = *reinterpret_cast<const h8 *>(cur + RL::kXl + off);
        outacc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, th[kk], outacc[fb], 0, 0, 0);
        outacc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, tl[kk], outacc[fb], 0, 0, 0);
        outacc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, th[kk], outacc[fb], 0, 0, 0);
      }
  };
  if (NB > 2) {
    for (int t = 0; t < nt; t += 3) {
      step(ring0, ring2, t);
      if (t + 1 < nt) step(ring1, ring0, t + 1);
      if (t + 2 < nt) step(ring2, ring1, t + 2);
    }
  } else {
    for (int t = 0; t < nt; t += 2) {
      step(ring0, ring1, t);
      if (t + 1 < nt) step(ring1, ring0, t + 1);
    }
  }

  // undo the row's scale; upstream gradient / temperature; the owned query's weight
  const float coef_r = gl * w_r * u2f((uint32_t)(127 - gcur) << 23);

  // Epilogue.  The accumulators hold out^T (lane = owned row, register = feature): written
  // directly that is one 4-byte store per lane with a row stride between lanes.  Each wave
  // instead transposes 32 features at a time through its own LDS scratch and stores whole
  // 128-byte row segments as float4.
  if ((a.d & 3) == 0) {
    constexpr int kLd = 36;                      // floats per LDS row: 32 + 4 (bank spread)
    float *scr = scr_all + wave * 32 * kLd;
    const int pr = lane >> 3, pc = lane & 7;     // 8 lanes x float4 = one 32-feature row segment
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[j * kLd + tile_row_of_reg(r, h)] = outacc[fb][r] * coef_r;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr = it * 8 + pr;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(scr + rr * kLd + pc * 4);
        const int64_t grow = base32 + rr;
        const int feat = fb * 32 + pc * 4;
        if (grow < R.n && feat < a.d)
          *reinterpret_cast<f32x4 *>(a.partial + ((int64_t)sp * R.n + grow) * a.d + feat) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();           // scr is rewritten by the next feature block
    }
  } else if (rvalid) {
    float *dst = a.partial + ((int64_t)sp * R.n + row) * a.d;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int feat = fb * 32 + tile_row_of_reg(r, h);
        if (feat < a.d) dst[feat] = outacc[fb][r] * coef_r;
      }
  }
}

// One launch for both directions: blocks [0, q_blocks) own queries (-> dq partials), the rest own
// candidates (-> dc partials).  The LDS objects are declared once and shared by the two bodies.
template <int DP, int NW>
__global__ void __launch_bounds__(NW * 64) sm16_bwd_kernel(const Sm16Args aq, const Sm16Args ac,
                                                           const int q_blocks) {
  typedef Rec16<DP> RL;
  constexpr int NB = DP <= 64 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char ring0[RL::kBytes];
  __shared__ __attribute__((aligned(16))) char ring1[RL::kBytes];
  __shared__ __attribute__((aligned(16))) char ring2[NB > 2 ? RL::kBytes : 16];
  __shared__ __attribute__((aligned(16))) float scr_all[NW * 32 * 36];
  __shared__ uint32_t s_e[NW];
  if ((int)blockIdx.x < q_blocks)
    sm16_bwd_body<DP, true, NW>(aq, (int)blockIdx.x, ring0, ring1, ring2, s_e, scr_all);
  else
    sm16_bwd_body<DP, false, NW>(ac, (int)blockIdx.x - q_blocks, ring0, ring1, ring2, s_e, scr_all);
}

// Both gradients' partials in one launch (elements [0, count_a) of a, then b).
__global__ void __launch_bounds__(256) sm16_reduce2_kernel(const float *pa, int na, int64_t count_a,
                                                           float *out_a, const float *pb, int nb,
                                                           int64_t count_b, float *out_b) {
  const int64_t total = count_a + count_b;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const bool first = t < count_a;
    const float *p = first ? pa : pb;
    const int64_t i = first ? t : t - count_a, cnt = first ? count_a : count_b;
    const int ns = first ? na : nb;
    float acc = 0.0f;
    for (int sp = 0; sp < ns; ++sp) acc += p[(int64_t)sp * cnt + i];
    (first ? out_a : out_b)[i] = acc;
  }
}

// ---- host side ----------------------------------------------------------------------------
static inline size_t al16(size_t x) { return (x + 255) / 256 * 256; }
static inline int64_t pad128(int64_t n) { return (n + 255) / 256 * 256; }   // records cover whole 256-row blocks
// waves per workgroup for a side with n_rows owned rows
static inline int nw_of(int64_t n_rows) {
  const char *v = option("TFRS_SOFTMAX_NW");   // read per call: tests switch it
  const int forced = (v && *v) ? atoi(v) : 0;
  if (forced == 4 || forced == 8) return forced;
  return n_rows >= 16384 ? 8 : 4;
}
static inline int dp_of(int d) { return d <= 32 ? 32 : (d <= 64 ? 64 : 128); }
static inline size_t rec_bytes(int dp) {
  return dp == 32 ? Rec16<32>::kBytes : (dp == 64 ? Rec16<64>::kBytes : Rec16<128>::kBytes);
}

static void plan16(int64_t n_rows, int64_t n_stream, int *nsplit, int64_t *split_len, bool backward = false) {
  const int64_t per_wg = nw_of(n_rows) * 32;
  const int64_t row_blocks = (n_rows + per_wg - 1) / per_wg;
  const int64_t tiles = (n_stream + 31) / 32;
  static const int64_t target_fwd = [] {
    const char *v = option("TFRS_SOFTMAX_WGS");
    return (v && *v) ? (int64_t)atoll(v) : (int64_t)512;   // 2 workgroups per CU
  }();
  // The backward's two sides share one launch, and LDS holds two of its workgroups per CU: 256 per side are ONE
  // round of resident workgroups (512 per side ran as two rounds, each paying the ~3 us prologue -- first touch of the
  // records, owned rows -- and wrote twice the partial gradients: 4096 x 4096 x 64, bwd + reduce 45.0 -> 41.9 us)
  static const int64_t target_bwd = [] {
    const char *v = option("TFRS_SOFTMAX_WGS_BWD");
    return (v && *v) ? (int64_t)atoll(v) : (int64_t)256;
  }();
  const int64_t target = backward ? target_bwd : target_fwd;
  int64_t want = (target + row_blocks - 1) / row_blocks;
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  const int64_t per = (tiles + want - 1) / want;
  *split_len = per * 32;
  *nsplit = (int)((tiles + per - 1) / per);
}

struct Layout16 {
  size_t header, q_rec, q_bmax, c_rec, c_bmax, scratch, total;
};

static Layout16 layout16(int64_t nq, int64_t nc, int d) {
  const int dp = dp_of(d);
  const int64_t nqp = pad128(nq), ncp = pad128(nc);
  Layout16 L;
  size_t o = 0;
  L.header = o; o += 256;                                    // finalize ticket
  L.q_rec = o; o += al16((size_t)(nqp / 32) * rec_bytes(dp));
  L.q_bmax = o; o += al16((size_t)(nqp / 32) * 4);
  L.c_rec = o; o += al16((size_t)(ncp / 32) * rec_bytes(dp));
  L.c_bmax = o; o += al16((size_t)(ncp / 32) * 4);
  L.scratch = o;
  int nsf, nsq, nsc;
  int64_t len;
  plan16(nq, nc, &nsf, &len);
  plan16(nq, nc, &nsq, &len, true);
  plan16(nc, nq, &nsc, &len, true);
  const size_t fwd = 2 * al16((size_t)nsf * nq * 4) + al16((size_t)nq * 4) +
                     al16((size_t)((nq + 63) / 64) * 8);
  const size_t bwd = al16((size_t)nsq * nq * d * 4) + al16((size_t)nsc * nc * d * 4);
  L.total = o + (fwd > bwd ? fwd : bwd);
  return L;
}

size_t softmax16_workspace_bytes(int64_t nq, int64_t nc, int d) { return layout16(nq, nc, d).total; }

static void fill_sides(Sm16Args *a, char *ws, const Layout16 &L, int64_t nq, int64_t nc) {
  a->q = {ws + L.q_rec, reinterpret_cast<const uint32_t *>(ws + L.q_bmax), nq, pad128(nq)};
  a->c = {ws + L.c_rec, reinterpret_cast<const uint32_t *>(ws + L.c_bmax), nc, pad128(nc)};
  a->q_rec_w = ws + L.q_rec;
}

template <int DP>
static int prep16(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                  char *ws, const Layout16 &L, hipStream_t s) {
  const PrepSide sq = {q, nq, w, ws + L.q_rec, reinterpret_cast<uint32_t *>(ws + L.q_bmax)};
  const PrepSide sc = {c, nc, nullptr, ws + L.c_rec, reinterpret_cast<uint32_t *>(ws + L.c_bmax)};
  const int qb = (int)(pad128(nq) / 32), cb = (int)(pad128(nc) / 32);
  hipLaunchKernelGGL((sm16_prep_kernel<DP>), dim3((unsigned)(qb + cb)), dim3(256), 0, s, sq, sc, qb, d,
                     reinterpret_cast<uint32_t *>(ws + L.header));
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

template <int DP>
static int fwd16(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                 float inv_t, float *out_loss, float *out_lse, float *out_pos, char *ws,
                 hipStream_t s) {
  const Layout16 L = layout16(nq, nc, d);
  int rc = prep16<DP>(q, c, nq, nc, d, w, ws, L, s);
  if (rc != TFRS_OK) return rc;
  Sm16Args a = {};
  fill_sides(&a, ws, L, nq, nc);
  a.d = d; a.w = w; a.inv_t = inv_t;
  plan16(nq, nc, &a.nsplit, &a.split_len);
  char *p = ws + L.scratch;
  a.pm = reinterpret_cast<float *>(p); p += al16((size_t)a.nsplit * nq * 4);
  a.pl = reinterpret_cast<float *>(p); p += al16((size_t)a.nsplit * nq * 4);
  a.ppos = reinterpret_cast<float *>(p); p += al16((size_t)nq * 4);
  double *block_part = reinterpret_cast<double *>(p);
  uint32_t *ticket = reinterpret_cast<uint32_t *>(ws + L.header);
  if (nw_of(nq) == 8) {
    hipLaunchKernelGGL((sm16_fwd_kernel<DP, 8>), dim3((unsigned)(((nq + 255) / 256) * a.nsplit)),
                       dim3(512), 0, s, a);
  } else {
    hipLaunchKernelGGL((sm16_fwd_kernel<DP, 4>), dim3((unsigned)(((nq + 127) / 128) * a.nsplit)),
                       dim3(256), 0, s, a);
  }
  TFRS_LAUNCH_CHECK();
  const unsigned fin_blocks = (unsigned)((nq + 63) / 64);
  hipLaunchKernelGGL((sm16_finalize_kernel<DP>), dim3(fin_blocks), dim3(64), 0, s, a, out_loss, out_lse,
                     out_pos, block_part, ticket);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

template <int DP>
static int bwd16(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                 float inv_t, const float *lse, const float *gloss, float *dq, float *dc, char *ws,
                 int reuse, hipStream_t s) {
  const Layout16 L = layout16(nq, nc, d);
  if (!reuse) {
    int rc = prep16<DP>(q, c, nq, nc, d, w, ws, L, s);
    if (rc != TFRS_OK) return rc;
    hipLaunchKernelGGL((sm16_fill_lse_kernel<DP>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s,
                       lse, nq, ws + L.q_rec);
    TFRS_LAUNCH_CHECK();
  }
  Sm16Args a = {};
  fill_sides(&a, ws, L, nq, nc);
  a.d = d; a.w = w; a.inv_t = inv_t; a.lse = lse; a.gloss = gloss;
  char *p = ws + L.scratch;

  Sm16Args aq = a, ac = a;
  plan16(nq, nc, &aq.nsplit, &aq.split_len, true);
  const int nsq = aq.nsplit;
  float *part_q = nsq == 1 ? dq : reinterpret_cast<float *>(p);
  aq.partial = part_q;
  p += al16((size_t)nsq * nq * d * 4);
  plan16(nc, nq, &ac.nsplit, &ac.split_len, true);
  const int nsc = ac.nsplit;
  float *part_c = nsc == 1 ? dc : reinterpret_cast<float *>(p);
  ac.partial = part_c;
  // both directions in one launch (a second launch only if the two sides differ in workgroup size)
  const int nwq = nw_of(nq), nwc = nw_of(nc);
  const int qb = (int)(((nq + nwq * 32 - 1) / (nwq * 32)) * nsq);
  const int cb = (int)(((nc + nwc * 32 - 1) / (nwc * 32)) * nsc);
  if (nwq == nwc) {
    if (nwq == 8)
      hipLaunchKernelGGL((sm16_bwd_kernel<DP, 8>), dim3((unsigned)(qb + cb)), dim3(512), 0, s, aq, ac, qb);
    else
      hipLaunchKernelGGL((sm16_bwd_kernel<DP, 4>), dim3((unsigned)(qb + cb)), dim3(256), 0, s, aq, ac, qb);
  } else {
    if (nwq == 8)
      hipLaunchKernelGGL((sm16_bwd_kernel<DP, 8>), dim3((unsigned)qb), dim3(512), 0, s, aq, ac, qb);
    else
      hipLaunchKernelGGL((sm16_bwd_kernel<DP, 4>), dim3((unsigned)qb), dim3(256), 0, s, aq, ac, qb);
    if (nwc == 8)
      hipLaunchKernelGGL((sm16_bwd_kernel<DP, 8>), dim3((unsigned)cb), dim3(512), 0, s, aq, ac, 0);
    else
      hipLaunchKernelGGL((sm16_bwd_kernel<DP, 4>), dim3((unsigned)cb), dim3(256), 0, s, aq, ac, 0);
  }
  TFRS_LAUNCH_CHECK();
  // per-split partial gradients -> dq, dc (a side with a single split wrote its output directly)
  const int64_t cq = nsq > 1 ? nq * d : 0, cc = nsc > 1 ? nc * d : 0;
  if (cq + cc > 0) {
    hipLaunchKernelGGL(sm16_reduce2_kernel, dim3((unsigned)std::min<int64_t>((cq + cc + 255) / 256, 4096)),
                       dim3(256), 0, s, part_q, nsq, cq, dq, part_c, nsc, cc, dc);
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}

int softmax16_forward(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                      float inv_t, float *out_loss, float *out_lse, float *out_pos, void *ws,
                      hipStream_t s) {
  char *p = static_cast<char *>(ws);
  switch (dp_of(d)) {
    case 32: return fwd16<32>(q, c, nq, nc, d, w, inv_t, out_loss, out_lse, out_pos, p, s);
    case 64: return fwd16<64>(q, c, nq, nc, d, w, inv_t, out_loss, out_lse, out_pos, p, s);
    default: return fwd16<128>(q, c, nq, nc, d, w, inv_t, out_loss, out_lse, out_pos, p, s);
  }
}

int softmax16_backward(const float *q, const float *c, int64_t nq, int64_t nc, int d, const float *w,
                       float inv_t, const float *lse, const float *gloss, float *dq, float *dc,
                       void *ws, int reuse, hipStream_t s) {
  char *p = static_cast<char *>(ws);
  switch (dp_of(d)) {
    case 32: return bwd16<32>(q, c, nq, nc, d, w, inv_t, lse, gloss, dq, dc, p, reuse, s);
    case 64: return bwd16<64>(q, c, nq, nc, d, w, inv_t, lse, gloss, dq, dc, p, reuse, s);
    default: return bwd16<128>(q, c, nq, nc, d, w, inv_t, lse, gloss, dq, dc, p, reuse, s);
  }
}

}  // namespace tfrs
