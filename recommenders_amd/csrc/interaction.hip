// interaction.hip -- feature-interaction layers: DCN Cross and DLRM DotInteraction.
//
// Cross.call (layers/feature_interaction/dcn.py:151-186):
//     y = x0 * (x @ kernel + bias + diag_scale * x) + x
//   one LDS-tiled f32-MFMA GEMM (Keras Dense kernel layout [in, out]) with the whole
//   cross formula fused into the epilogue: the [B, d] product is never written to HBM, so
//   the layer moves 3*B*d*4 + d*d*4 bytes instead of the reference's GEMM + three
//   elementwise passes.  MFMA-bound for the DCN-v2 shapes (2*B*d*d flop).
//   The same GEMM with a bias-only epilogue is exported as tfrs_dense_fwd (low-rank U/V
//   projections, MultiLayerDCN multi_layer_dcn.py:147-153).
//
// DotInteraction.call (layers/feature_interaction/dot_interaction.py:53-104):
//   per sample X[F, D] -> lower triangle of X X^T in row-major order (or the full F*F with
//   the upper part zeroed for skip_gather).  One wave per sample, X staged in LDS, pairs
//   spread over lanes; the output is written exactly once, coalesced (HBM-bound target:
//   B*F*D*4 read + B*out_dim*4 written, vs 4-5x that for the reference's
//   ones_like/band_part/boolean_mask temporaries).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "mfma_tile.h"

namespace tfrs {

// ------------------------------------------------------------------------------------------
// GEMM  C[M, N] = A[M, K] @ B[K, N]  (+ epilogue), all row-major f32.
// Block tile 128 x 128 x 16, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA 32x32 tiles.
// k order inside a 16-wide K step is the "half split" (lane half h owns k in [8h, 8h+8)),
// so a lane's A fragment is 8 contiguous floats of an LDS row (two ds_read_b128).
// ------------------------------------------------------------------------------------------
constexpr int kBM = 128, kBN = 128, kBK = 16;
constexpr int kLdA = kBK + 4;   // 20 floats = 5 x 16 B: odd number of 16-B slots per row
constexpr int kLdB = kBN + 4;   // row of B tile

// epilogues on v = accumulator + bias (e0 = x0 field, e1 = x field), same set as gemm16.hip:
//   Bias     out = v
//   Cross    out = e0 * (v + diag * e1) + e1          e0 = x0, e1 = x   (Cross.call)
//   CrossDx0 out = e0 * (v + diag * e1)               e0 = dy, e1 = x   (dx0 = dy * z)
//   CrossDx  out = v + e0 + diag * e0 * e1            e0 = dy, e1 = x0  (dx = dz W^T + dy + diag dz)
enum { kEpiBias = 0, kEpiCross = 1, kEpiCrossDx0 = 2, kEpiCrossDx = 3 };

struct GemmArgs {
  const float *a, *b;
  int64_t m;
  int n, k;
  const float *bias;  // [n] or NULL
  // epilogue operands
  const float *x0;    // [m, n]
  const float *x;     // [m, n]
  float diag;
  float *out;         // [m, n]
  // optional elementwise multipliers, same shape/layout as a / b (dz = dy * x0 is never stored)
  const float *amul, *bmul;
  // split-K (AT && !BT only: K is the row index of both operands): slice blockIdx.y covers
  // k in [y * kper, min(k, (y + 1) * kper)) and writes its partial product to out + y * m * n
  // (no bias); kper == 0: no split.
  int kper;
  // fused activation of v = product + bias (common.h kAct*), applied before the epilogue formula; `pre`
  // (optional, [m, n]) receives v itself -- what the backward differentiates the activation at
  int act;
  float *pre;
};

// AT: `a` holds A^T ([K, M] row-major); BT: `b` holds B^T ([N, K] row-major).  The tiles are
// transposed on their way into LDS, so dW = x^T dz and dx = dz W^T read x, dz and W as they lie.
template <int EPI, bool AT, bool BT>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float as[2][kBM * kLdA];
  __shared__ __attribute__((aligned(16))) float bs[2][kBK * kLdB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // wave position in the 2 x 2 grid
  const int j = lane & 31, h = lane >> 5;

  // block coordinates; consecutive blocks share the same A rows (x) for L2 reuse
  const int nbn = (g.n + kBN - 1) / kBN;
  const int64_t bm = (int64_t)(blockIdx.x / nbn) * kBM;
  const int bn = (int)(blockIdx.x % nbn) * kBN;

  const bool a_vec = ((AT ? g.m : (int64_t)g.k) % 4 == 0) && (((uintptr_t)g.a) % 16 == 0) &&
                     (!g.amul || ((uintptr_t)g.amul) % 16 == 0);
  const bool b_vec = ((BT ? g.k : g.n) % 4 == 0) && (((uintptr_t)g.b) % 16 == 0) &&
                     (!g.bmul || ((uintptr_t)g.bmul) % 16 == 0);
  // 4 consecutive elements of a row-major [rows, ld] array (zero outside), optional multiplier
  auto load4 = [&](const float *p, const float *mul, int64_t r, int64_t c, int64_t rows, int64_t ld,
                   bool vec) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      if (vec && c + 3 < ld) {
        v = *reinterpret_cast<const f32x4 *>(p + r * ld + c);
        if (mul) v = v * *reinterpret_cast<const f32x4 *>(mul + r * ld + c);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (c + t < ld) v[t] = mul ? p[r * ld + c + t] * mul[r * ld + c + t] : p[r * ld + c + t];
      }
    }
    return v;
  };

  // K range of this workgroup (split-K slices exist for the AT && !BT product only)
  const bool split = AT && !BT && g.kper > 0;
  const int k_begin = split ? (int)blockIdx.y * g.kper : 0;
  const int k_end = split ? min(g.k, k_begin + g.kper) : g.k;

  // staging: A tile 128 x 16 = 512 float4 -> 2 per thread; B tile 16 x 128 = 512 float4 -> 2
  f32x4 sa[2], sb[2];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256;      // 0..511
      if (!AT) {   // A tile 128 (m) x 16 (k): row e >> 2, k (e & 3) * 4 .. + 3
        sa[i] = load4(g.a, g.amul, bm + (e >> 2), k0 + (e & 3) * 4, g.m, g.k, a_vec);
      } else {     // A^T: k row e >> 5, m (e & 31) * 4 .. + 3
        sa[i] = load4(g.a, g.amul, k0 + (e >> 5), bm + (e & 31) * 4, k_end, g.m, a_vec);
      }
      if (!BT) {   // B tile 16 (k) x 128 (n): k row e >> 5, n (e & 31) * 4 .. + 3
        sb[i] = load4(g.b, g.bmul, k0 + (e >> 5), bn + (e & 31) * 4, BT ? g.k : k_end, g.n, b_vec);
      } else {     // B^T: n row e >> 2, k (e & 3) * 4 .. + 3
        sb[i] = load4(g.b, g.bmul, bn + (e >> 2), k0 + (e & 3) * 4, g.n, g.k, b_vec);
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256;
      if (!AT) {
        *reinterpret_cast<f32x4 *>(&as[buf][(e >> 2) * kLdA + (e & 3) * 4]) = sa[i];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) as[buf][((e & 31) * 4 + t) * kLdA + (e >> 5)] = sa[i][t];
      }
      if (!BT) {
        *reinterpret_cast<f32x4 *>(&bs[buf][(e >> 5) * kLdB + (e & 31) * 4]) = sb[i];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) bs[buf][((e & 3) * 4 + t) * kLdB + (e >> 2)] = sb[i][t];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;

  const int nk = (k_end - k_begin + kBK - 1) / kBK;
  load_tiles(k_begin);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(k_begin + (kt + 1) * kBK);

    // fragments: A rows (wm*64 + i*32 + j), k = 8h .. 8h+7; B cols (wn*64 + jn*32 + j)
    f32x4 af[2][2];
    float bf[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float *ap = &as[cur][(wm * 64 + i * 32 + j) * kLdA + 8 * h];
      af[i][0] = *reinterpret_cast<const f32x4 *>(ap);
      af[i][1] = *reinterpret_cast<const f32x4 *>(ap + 4);
    }
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int s = 0; s < 8; ++s) bf[jn][s] = bs[cur][(8 * h + s) * kLdB + wn * 64 + jn * 32 + j];

#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s >> 2][s & 3], bf[jn][s],
                                                            acc[i][jn], 0, 0, 0);

    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // epilogue: acc[i][jn][r] = C[row = bm + wm*64 + i*32 + tile_row_of_reg(r, h)][col = bn + wn*64 + jn*32 + j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int col = bn + wn * 64 + jn * 32 + j;
      if (col >= g.n) continue;
      const float bias = (g.bias && !split) ? g.bias[col] : 0.0f;
      float *const outp = split ? g.out + (int64_t)blockIdx.y * g.m * g.n : g.out;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = bm + wm * 64 + i * 32 + tile_row_of_reg(r, h);
        if (row >= g.m) continue;
        float v = acc[i][jn][r] + bias;
        const int64_t o = row * g.n + col;
        if (g.pre) g.pre[o] = v;
        if (g.act) v = act_apply(g.act, v);
        if (EPI == kEpiCross) {
          const float xv = g.x[o];
          v = g.x0[o] * (v + g.diag * xv) + xv;
        } else if (EPI == kEpiCrossDx0) {
          v = g.x0[o] * (v + g.diag * g.x[o]);
        } else if (EPI == kEpiCrossDx) {
          const float dyv = g.x0[o];
          v = v + dyv + g.diag * dyv * g.x[o];
        }
        outp[o] = v;
      }
    }
}

// ------------------------------------------------------------------------------------------
// DotInteraction
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_of_index(int p, bool self, int *pi, int *pj) {
  // row-major lower triangle: rows i = 0.., columns j <= i (self) or j < i.
  // self:  p = i(i+1)/2 + j ;  no self: p = i(i-1)/2 + j
  int i = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
  while ((i + 1) * (i + 2) / 2 <= p) ++i;   // fix float rounding
  while (i * (i + 1) / 2 > p) --i;
  const int j = p - i * (i + 1) / 2;
  if (self) {
    *pi = i;
    *pj = j;
  } else {
    *pi = i + 1;
    *pj = j;
  }
}

__global__ void __launch_bounds__(256) dot_interaction_fwd_kernel(const float *__restrict__ x,
                                                                  int64_t batch, int f, int d,
                                                                  int self, int skip_gather,
                                                                  float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= batch) return;
  const int ld = d + 1;  // odd-ish stride: lanes read different rows at the same column
  float *xs = smem_f + (size_t)wave * f * ld;
  const float *xb = x + b * (int64_t)f * d;
  for (int e = lane; e < f * d; e += 64) xs[(e / d) * ld + (e % d)] = xb[e];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  if (skip_gather) {
    float *ob = out + b * (int64_t)f * f;
    for (int p = lane; p < f * f; p += 64) {
      const int i = p / f, jx = p - i * f;
      float acc = 0.0f;
      if (jx < i || (self && jx == i)) {
        for (int k = 0; k < d; ++k) acc = __builtin_fmaf(xs[i * ld + k], xs[jx * ld + k], acc);
      }
      ob[p] = acc;
    }
  } else {
    const int npairs = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
    float *ob = out + b * (int64_t)npairs;
    for (int p = lane; p < npairs; p += 64) {
      int i, jx;
      pair_of_index(p, self != 0, &i, &jx);
      float acc = 0.0f;
      for (int k = 0; k < d; ++k) acc = __builtin_fmaf(xs[i * ld + k], xs[jx * ld + k], acc);
      ob[p] = acc;
    }
  }
}

// dX[i] = sum_{j != i} dY[pair(i, j)] X[j]  (+ 2 dY[i, i] X[i] with self interaction)
__global__ void __launch_bounds__(256) dot_interaction_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ dout, int64_t batch, int f, int d,
    int self, int skip_gather, float *__restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= batch) return;
  float *xs = smem_f + (size_t)wave * f * d;
  const float *xb = x + b * (int64_t)f * d;
  for (int e = lane; e < f * d; e += 64) xs[e] = xb[e];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int npairs = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const float *dy = dout + b * (int64_t)(skip_gather ? f * f : npairs);
  // one (feature row i, column k) element per lane-iteration; xs[j][k] reads are
  // conflict-free across consecutive k
  for (int e = lane; e < f * d; e += 64) {
    const int i = e / d, k = e - i * d;
    float acc = 0.0f;
    for (int jx = 0; jx < f; ++jx) {
      if (jx == i && !self) continue;
      const int hi = jx > i ? jx : i, lo = jx > i ? i : jx;
      float gy;
      if (skip_gather) {
        gy = dy[hi * f + lo];
      } else {
        gy = self ? dy[hi * (hi + 1) / 2 + lo] : dy[hi * (hi - 1) / 2 + lo];
      }
      if (jx == i) gy *= 2.0f;
      acc = __builtin_fmaf(gy, xs[jx * d + k], acc);
    }
    dx[b * (int64_t)f * d + e] = acc;
  }
}

// MFMA form of the forward pass.  A wave owns one sample at a time (grid-stride over the
// batch): the sample's whole feature matrix X[F, D] lives in registers as NB row-block
// fragments (lane l: row 32*rb + (l & 31), feature half l >> 5; the next sample's fragments
// are fetched under the copy-out of the current one), every 32x32 block of the lower
// triangle of G = X X^T is one chain of DP/2 v_mfma_f32_32x32x2_f32 (exact f32).  The packed
// row-major lower triangle is assembled in LDS (ds_write_b32 per accumulator register) and
// leaves as one linear stream of 8-byte stores, so HBM sees full lines only.
// Algorithmic bytes: B*(F*D + out_dim)*4.  (skip_gather keeps direct stores: F*F is too
// large to stage and the mode is rare.)
// (staged launches use ONE wave per workgroup so that LDS -- 20 KiB per sample at F = 101 --
// limits residency per wave, not per 4-wave group)
template <int DP, int NB, bool STAGE>
__global__ void __launch_bounds__(STAGE ? 64 : 256, 2) dot_interaction_mfma_kernel(const float *__restrict__ x,
                                                                   int64_t batch, int f, int d,
                                                                   int self, int skip_gather_arg,
                                                                   float *__restrict__ out) {
  const bool skip_gather = STAGE ? false : (skip_gather_arg != 0);  // staged launches never skip
  extern __shared__ __attribute__((aligned(16))) float smem_dot[];
  constexpr int kWavesPerWg = STAGE ? 1 : 4;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const bool vec_ok = (d == DP) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((f * d) % 4 == 0);
  const int out_dim = skip_gather ? f * f : (self ? f * (f + 1) / 2 : f * (f - 1) / 2);
  const int stage_len = ((out_dim + 3) & ~3) + 64;  // + one dummy slot per lane
  float *stage = smem_dot + (size_t)wave * stage_len;
  const int dummy = stage_len - 64 + lane;
  const int64_t wave_stride = (int64_t)gridDim.x * kWavesPerWg;

  int64_t b = (int64_t)blockIdx.x * kWavesPerWg + wave;
  float frag[NB][DP / 2];
  if (b < batch) {
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
      load_row_frag<DP>(frag[rb], x + b * (int64_t)f * d, rb * 32 + j, rb * 32 + j < f, d, h, vec_ok);
  }
  for (; b < batch; b += wave_stride) {
    float *ob = out + b * (int64_t)out_dim;
    float *dst = STAGE ? stage : ob;
    // Opaque zero: keeps the ~160 store predicates / offsets of one sample from being hoisted
    // out of the sample loop (they would occupy hundreds of SGPRs/VGPRs for the whole kernel).
    int lz = 0;
    asm volatile("" : "+v"(lz));
    const int jv = j + lz, hv = h + lz;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (bi * 32 >= f) continue;  // uniform
      // packed offset of (row, col): tri(row) + col with tri(row) = row(row -+ 1)/2.  For the
      // lane's 16 rows row0 + dr (dr compile-time) tri(row0 + dr) = tri(row0) + dr*row0 +
      // dr(dr -+ 1)/2: one multiply per block, one mad per element, 32-bit (out_dim < 2^15).
      const int row0 = bi * 32 + 4 * hv;
      const int tri0 = self ? row0 * (row0 + 1) / 2 : row0 * (row0 - 1) / 2;
      const int row_lim = f - row0;          // rows row0 + dr with dr < row_lim exist
      const int diag_t = jv - 4 * hv;        // diagonal block: col < row  <=>  diag_t < dr
      int pre[16];                           // tri(row0 + dr) - tri(row0)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        pre[r] = dr * row0 + (self ? dr * (dr + 1) / 2 : dr * (dr - 1) / 2);
      }
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) {
        f32x16 acc = tile_dot<DP>(frag[bi], frag[bj]);
        // The 18 wait states a v_mfma_f32_32x32x2_f32 result needs before a store may read it, by hand: behind the
        // chain the direct-store variant is a web of predicated side blocks, and the compiler's hazard recognizer
        // leaves them out on the short paths through it (14 instead of 18: tools/check_mfma_hazards.py)
        if (!STAGE) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" : "+v"(acc));
        // acc[r] = G[row0 + dr(r)][bj*32 + j],  dr(r) = (r & 3) + 8 * (r >> 2)
        const int col = bj * 32 + jv;
        const int a0 = tri0 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (skip_gather) {
            const int row = row0 + dr;
            if (row < f && col < f) {
              const bool in_tri = self ? (col <= row) : (col < row);
              dst[row * f + col] = in_tri ? acc[r] : 0.0f;
            }
          } else {
            // off-diagonal blocks lie entirely below the diagonal; only the last row block
            // can hold rows >= f
            bool keep = (bi < NB - 1) || (dr < row_lim);
            if (bj == bi) keep = keep && (self ? diag_t <= dr : diag_t < dr);
            if (STAGE) {  // branch-free: rejected elements land in a per-lane dummy slot
              stage[keep ? a0 + pre[r] : dummy] = acc[r];
            } else if (keep) {
              dst[a0 + pre[r]] = acc[r];
            }
          }
        }
      }
      if (skip_gather) {  // blocks right of the diagonal block: zeros
#pragma unroll
        for (int bj = bi + 1; bj < NB; ++bj) {
          const int col = bj * 32 + j;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            if (row < f && col < f) dst[row * f + col] = 0.0f;
          }
        }
      }
    }
    // the fragments are dead: fetch the next sample of this wave under the copy-out
    const int64_t bn = b + wave_stride;
    if (bn < batch) {
#pragma unroll
      for (int rb = 0; rb < NB; ++rb)
        load_row_frag<DP>(frag[rb], x + bn * (int64_t)f * d, rb * 32 + j, rb * 32 + j < f, d, h, vec_ok);
    }
    if (STAGE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // linear copy-out: sample bases are 8-byte aligned when out_dim is even, else dwords
      if ((out_dim & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)) {
#pragma unroll 8
        for (int e = 2 * lane; e < out_dim; e += 128)
          *reinterpret_cast<float2 *>(ob + e) = *reinterpret_cast<const float2 *>(stage + e);
      } else {
#pragma unroll 8
        for (int e = lane; e < out_dim; e += 64) ob[e] = stage[e];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the staging buffer is reused by the next sample
    }
  }
}

// Split-fp16 form of the staged forward pass (TFRS_DOT_FWD=staged; the direct-store kernel below is
// the default).  The f32 MFMA chain
// above costs DP/2 x 64 cycles per 32x32 block -- 10240 matrix-core cycles per sample at F = 101,
// D = 32, i.e. 0.55 ms of pure MFMA time per 131072 samples, as much as the HBM traffic costs.
// Here every operand is split x * s = hi + lo (two fp16 values, s a per-sample power of two that
// puts max |x| at 2^11..2^12 so that neither half leaves fp16's normal range for the elements
// that matter) and a block is DP/16 steps of three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo +
// lo*hi, f32 accumulation; the dropped lo*lo term is 2^-22 relative): 1920 cycles per sample.
// Representation error 2^-22 |x| per element, i.e. the dot products carry ~2^-21 sum |x_i x_j|
// -- the same order as the f32 rounding of the chain it replaces.  Staging, copy-out and the
// prefetch of the next sample are those of dot_interaction_mfma_kernel<.., true>; the 1/s^2 is
// applied (exactly) on the way out.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  union {
    h16x2 h;
    uint32_t u;
  } v;
  v.h[0] = (_Float16)a;   // round to nearest even
  v.h[1] = (_Float16)b;
  return v.u;
}

// max over the wave of a non-negative float (positive floats order like their bit patterns), by
// DPP lane permutations: six VALU operations, no LDS round trips (__shfl_xor is ds_bpermute).
__device__ __forceinline__ float wave_max_nonneg(float v) {
  int x = __float_as_int(v);
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));   // row_half_mirror
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));   // row_mirror
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xF, 0xF, false));   // row_bcast15
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xF, 0xF, false));   // row_bcast31
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

template <int DP, int NB>
__global__ void __launch_bounds__(64, 2) dot_interaction_f16x3_kernel(const float *__restrict__ x,
                                                                      int64_t batch, int f, int d, int self,
                                                                      float *__restrict__ out) {
  static_assert(DP % 16 == 0, "whole 32x32x16 steps");
  constexpr int KS = DP / 16;   // MFMA k-steps: lane half h supplies features h*DP/2 + 8t .. +7 in step t
  extern __shared__ __attribute__((aligned(16))) float smem_dot[];
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const bool vec_ok = (d == DP) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((f * d) % 4 == 0);
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const int stage_len = ((out_dim + 3) & ~3) + 64;  // + one dummy slot per lane
  float *stage = smem_dot;
  const int dummy = stage_len - 64 + lane;
  const int64_t wave_stride = gridDim.x;

  int64_t b = blockIdx.x;
  float raw[NB][DP / 2];
  if (b < batch) {
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
      load_row_frag<DP>(raw[rb], x + b * (int64_t)f * d, rb * 32 + j, rb * 32 + j < f, d, h, vec_ok);
  }
  for (; b < batch; b += wave_stride) {
    float *ob = out + b * (int64_t)out_dim;
    // ---- per-sample scale and the hi / lo operands ----------------------------------------
    float m = 0.0f;
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
      for (int e = 0; e < DP / 2; ++e) m = fmaxf(m, __builtin_fabsf(raw[rb][e]));
    m = wave_max_nonneg(m);
    // s = 2^k with k = 12 - floor(log2 m), clamped so that s, s^2 and their inverses are normal
    int k = 139 - (int)(__float_as_uint(m) >> 23);
    k = (m > 0.0f && m < __builtin_inff()) ? min(max(k, -60), 60) : 0;
    const float sc = __uint_as_float((uint32_t)(127 + k) << 23);
    const float inv2 = __uint_as_float((uint32_t)(127 - 2 * k) << 23);
    h16x8 hi[NB][KS], lo[NB][KS];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        union {
          uint32_t u[4];
          h16x8 v;
        } ph, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = raw[rb][8 * t + 2 * e] * sc, a1 = raw[rb][8 * t + 2 * e + 1] * sc;
          const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
          union {
            h16x2 hh;
            uint32_t u;
          } q;
          q.hh[0] = h0;
          q.hh[1] = h1;
          ph.u[e] = q.u;
          pl.u[e] = pack_h2(a0 - (float)h0, a1 - (float)h1);
        }
        hi[rb][t] = ph.v;
        lo[rb][t] = pl.v;
      }
    // Opaque zero: keeps the ~160 store predicates / offsets of one sample from being hoisted
    // out of the sample loop (see dot_interaction_mfma_kernel).
    int lz = 0;
    asm volatile("" : "+v"(lz));
    const int jv = j + lz, hv = h + lz;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (bi * 32 >= f) continue;  // uniform
      const int row0 = bi * 32 + 4 * hv;
      const int tri0 = self ? row0 * (row0 + 1) / 2 : row0 * (row0 - 1) / 2;
      const int row_lim = f - row0;          // rows row0 + dr with dr < row_lim exist
      const int diag_t = jv - 4 * hv;        // diagonal block: col < row  <=>  diag_t < dr
      int pre[16];                           // (tri(row0 + dr) - tri(row0)) * 4: byte offsets
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        pre[r] = 4 * (dr * row0 + (self ? dr * (dr + 1) / 2 : dr * (dr - 1) / 2));
      }
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[bi][t], hi[bj][t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[bi][t], lo[bj][t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo[bi][t], hi[bj][t], acc, 0, 0, 0);
        }
        const int a0 = 4 * (tri0 + bj * 32 + jv);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          bool keep = (bi < NB - 1) || (dr < row_lim);
          if (bj == bi) keep = keep && (self ? diag_t <= dr : diag_t < dr);
          // rejected elements land in a per-lane dummy slot
          *reinterpret_cast<float *>(reinterpret_cast<char *>(stage) + (keep ? a0 + pre[r] : 4 * dummy)) = acc[r];
        }
      }
    }
    // the operands are dead: fetch the next sample under the copy-out
    const int64_t bn = b + wave_stride;
    if (bn < batch) {
#pragma unroll
      for (int rb = 0; rb < NB; ++rb)
        load_row_frag<DP>(raw[rb], x + bn * (int64_t)f * d, rb * 32 + j, rb * 32 + j < f, d, h, vec_ok);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((out_dim & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)) {
#pragma unroll 8
      for (int e = 2 * lane; e < out_dim; e += 128) {
        const float2 v = *reinterpret_cast<const float2 *>(stage + e);
        *reinterpret_cast<float2 *>(ob + e) = make_float2(v.x * inv2, v.y * inv2);
      }
    } else {
#pragma unroll 8
      for (int e = lane; e < out_dim; e += 64) ob[e] = stage[e] * inv2;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the staging buffer is reused by the next sample
  }
}

// Split-fp16 forward without LDS staging (default; TFRS_DOT_FWD=staged selects the kernel above):
// accumulator register r of a block is one row of 32 consecutive packed positions per lane half,
// so a store instruction writes two 128-byte runs.  No staging buffer and no operand prefetch --
// occupancy (VGPR-bound, BLOCKS workgroups of four waves per CU) hides the latencies instead.
// The power-of-two scale is per 32-row block here (a block's operands are converted as soon as
// its own rows have arrived); block (bi, bj) is unscaled by 1 / (s_bi s_bj).
template <int DP, int NB, int BLOCKS>
__global__ void __launch_bounds__(256, BLOCKS) dot_interaction_f16x3_direct_kernel(const float *__restrict__ x,
                                                                                   int64_t batch, int f, int d, int self,
                                                                                   float *__restrict__ out, int64_t out_stride) {
  static_assert(DP % 16 == 0, "whole 32x32x16 steps");
  constexpr int KS = DP / 16;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const bool vec_ok = (d == DP) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((f * d) % 4 == 0);
  const int64_t wave_stride = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < batch; b += wave_stride) {
    float *ob = out + b * out_stride;   // (out_stride >= out_dim: rows of a wider matrix, tfrs_dot_interaction_fwd_strided)
    h16x8 hi[NB][KS], lo[NB][KS];
    float inv_s[NB];   // wave-uniform
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
      float raw[DP / 2];
      load_row_frag<DP>(raw, x + b * (int64_t)f * d, rb * 32 + j, rb * 32 + j < f, d, h, vec_ok);
      float m = 0.0f;
#pragma unroll
      for (int e = 0; e < DP / 2; ++e) m = fmaxf(m, __builtin_fabsf(raw[e]));
      m = wave_max_nonneg(m);
      // s = 2^k with k = 12 - floor(log2 m), clamped so that s and 1/s stay normal
      int k = 139 - (int)(__float_as_uint(m) >> 23);
      k = (m > 0.0f && m < __builtin_inff()) ? min(max(k, -60), 60) : 0;
      const float sc = __uint_as_float((uint32_t)(127 + k) << 23);
      inv_s[rb] = __uint_as_float((uint32_t)(127 - k) << 23);
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        union {
          uint32_t u[4];
          h16x8 v;
        } ph, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = raw[8 * t + 2 * e] * sc, a1 = raw[8 * t + 2 * e + 1] * sc;
          const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
          union {
            h16x2 hh;
            uint32_t u;
          } q;
          q.hh[0] = h0;
          q.hh[1] = h1;
          ph.u[e] = q.u;
          pl.u[e] = pack_h2(a0 - (float)h0, a1 - (float)h1);
        }
        hi[rb][t] = ph.v;
        lo[rb][t] = pl.v;
      }
    }
    int lz = 0;   // opaque zero: see dot_interaction_mfma_kernel
    asm volatile("" : "+v"(lz));
    const int jv = j + lz, hv = h + lz;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (bi * 32 >= f) continue;  // uniform
      const int row0 = bi * 32 + 4 * hv;
      const int tri0 = self ? row0 * (row0 + 1) / 2 : row0 * (row0 - 1) / 2;
      const int row_lim = f - row0;
      const int diag_t = jv - 4 * hv;
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[bi][t], hi[bj][t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[bi][t], lo[bj][t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo[bi][t], hi[bj][t], acc, 0, 0, 0);
        }
        const float unscale = inv_s[bi] * inv_s[bj];
        float *dst = ob + tri0 + bj * 32 + jv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          bool keep = (bi < NB - 1) || (dr < row_lim);
          if (bj == bi) keep = keep && (self ? diag_t <= dr : diag_t < dr);
          // tri(row0 + dr) - tri(row0) = dr * row0 + dr (dr -+ 1) / 2
          if (keep) dst[dr * row0 + (self ? dr * (dr + 1) / 2 : dr * (dr - 1) / 2)] = acc[r] * unscale;
        }
      }
    }
  }
}

// Forward, producer / consumer form (d == 32, default): the split-fp16 products of the kernel above with
// the loads and the stores on DIFFERENT waves.  In the direct kernel every wave loads its sample, waits,
// multiplies and stores: gfx950 has ONE in-order counter (vmcnt) for a wave's loads and stores, and loads
// and stores complete out of order with respect to each other, so a wave that has stores in flight
// cannot wait for a prefetched load without waiting for all of its stores -- the direct kernel has no
// prefetch for that reason and each of its 12 waves per CU pays the HBM round trip of its sample with
// nothing else to do (0.50-0.54 of HBM).  Here (the structure of dot_interaction_bwd_h16_kernel): waves 4-7
// only load -- NSET samples in registers, 16-byte loads that are never awaited for two iterations --
// convert sample n + 1 to {hi, lo} fp16 with ONE power-of-two scale per sample (wave max by DPP, one
// ds_max per wave, reduced one iteration earlier) and write it to LDS as operand fragments; waves 0-3
// only multiply and store: they read ready 16-byte fragments (row r: eight 16-byte pieces, 4 hi + 4 lo,
// piece p stored at slot p ^ ((r >> 1) & 7) so that the 16 lanes of a ds_read_b128 group hit 16 different
// bank groups), run the 32 x 32 blocks of the lower triangle (block k = tri(bi) + bj on wave k mod 4) and
// store from the accumulators exactly as the direct kernel does.
template <int NB, int NSET>
__global__ void __launch_bounds__(512, 4) dot_interaction_fwd_pc_kernel(const float *__restrict__ x, int64_t batch,
                                                                        int f, int self, float *__restrict__ out,
                                                                        int64_t out_stride) {
  extern __shared__ __attribute__((aligned(16))) char f_lds[];
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  constexpr int D = 32;
  constexpr int NEX = (NB * 32 * D / 4 + 255) / 256;   // 16-byte chunks of X per producer thread
  constexpr int BUFB = NB * 32 * 128;                  // bytes per buffer: 128 per row (hi plane, lo plane)
  const int stage_bytes = ((((self ? f * (f + 1) / 2 : f * (f - 1) / 2) * 4) + 15) & ~15) + 1024;   // row image + dummy words
  uint32_t *const slots = reinterpret_cast<uint32_t *>(f_lds + 2 * BUFB);   // bits of max |x| of samples n % 4
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xn = f * D;
  const int64_t stride = gridDim.x;
  const int64_t b0 = blockIdx.x;
  for (int e = tid; e < (2 * BUFB + 16) / 4; e += 512) reinterpret_cast<uint32_t *>(f_lds)[e] = 0u;   // (the row images need no init)
  __syncthreads();
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto pow2_exponent = [](uint32_t mbits) __attribute__((always_inline)) -> int {
    int k = 139 - (int)(mbits >> 23);          // max * 2^k in [2^12, 2^13)
    return (mbits > 0u && mbits < 0x7F800000u) ? min(max(k, -60), 60) : 0;
  };
  const int64_t total = (batch - b0 + stride - 1) / stride;   // samples of this workgroup (>= 1)

  if (wave >= 4) {
    // ---------------------------------- producers ----------------------------------
    const int ptid = tid & 255;
    f32x4 rx[NSET][NEX];
    // loads are unconditional (clamped into the sample) so that the compiler can count them: see
    // dot_interaction_bwd_h16_kernel
    auto load_sample = [&](f32x4 (&gx)[NEX], int64_t b) __attribute__((always_inline)) {
      const float *xb = x + (b < batch ? b : batch - 1) * (int64_t)xn;
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        const int p0 = 4 * (ptid + 256 * e);
        gx[e] = *reinterpret_cast<const f4u *>(xb + min(p0, xn - 4));
      }
    };
    auto max_sample = [&](f32x4 (&gx)[NEX], int slot) __attribute__((always_inline)) {
      float mx = 0.0f;
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        if (4 * (ptid + 256 * e) >= xn) gx[e] = f32x4{0.f, 0.f, 0.f, 0.f};   // chunks beyond the sample
#pragma unroll
        for (int c = 0; c < 4; ++c) mx = fmaxf(mx, __builtin_fabsf(gx[e][c]));
      }
      mx = wave_max_nonneg(mx);
      if (lane == 0) atomicMax(&slots[slot], __float_as_uint(mx));
    };
    // chunk c = ptid + 256 e is row c >> 3, dims 4 (c & 7) ..+3: 8 bytes of the hi plane and 8 of the lo plane
    const int d0 = (ptid & 7) * 4, sw = (ptid >> 4) & 7;   // (row >> 1) & 7 does not depend on e: rows advance by 32
    const int wr_hi = (ptid >> 3) * 128 + ((((d0 >> 3)) ^ sw) << 4) + ((d0 >> 2) & 1) * 8;
    const int wr_lo = (ptid >> 3) * 128 + (((4 + (d0 >> 3)) ^ sw) << 4) + ((d0 >> 2) & 1) * 8;
    auto convert_sample = [&](const f32x4 (&gx)[NEX], int slot, char *buf) __attribute__((always_inline)) {
      const float sx = __uint_as_float((uint32_t)(127 + pow2_exponent(slots[slot])) << 23);
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        if (4 * (ptid + 256 * e) < xn) {
          float a[4];
          _Float16 hh[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            a[c] = gx[e][c] * sx;
            hh[c] = (_Float16)a[c];
          }
          u32x2 vh, vl;
          union {
            h16x2 h;
            uint32_t u;
          } q0, q1;
          q0.h[0] = hh[0];
          q0.h[1] = hh[1];
          q1.h[0] = hh[2];
          q1.h[1] = hh[3];
          vh[0] = q0.u;
          vh[1] = q1.u;
          vl[0] = pack_h2(a[0] - (float)hh[0], a[1] - (float)hh[1]);
          vl[1] = pack_h2(a[2] - (float)hh[2], a[3] - (float)hh[3]);
          *reinterpret_cast<u32x2 *>(buf + wr_hi + e * 32 * 128) = vh;
          *reinterpret_cast<u32x2 *>(buf + wr_lo + e * 32 * 128) = vl;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NSET; ++s) load_sample(rx[s], b0 + s * stride);
    max_sample(rx[0], 0);
    lds_barrier();
    convert_sample(rx[0], 0, f_lds);
    load_sample(rx[0], b0 + NSET * stride);
    max_sample(rx[1 % NSET], 1);
    lds_barrier();
    auto iteration = [&](auto uc, int64_t n0) __attribute__((always_inline)) {
      constexpr int U = decltype(uc)::value;
      const int64_t n = n0 + U;
      const int sl = (int)(n & 3);
      convert_sample(rx[(U + 1) % NSET], (sl + 1) & 3, f_lds + ((n + 1) & 1) * BUFB);
      load_sample(rx[(U + 1) % NSET], b0 + (n + NSET + 1) * stride);
      max_sample(rx[(U + 2) % NSET], (sl + 2) & 3);
      if (ptid == 0) slots[(sl + 3) & 3] = 0u;     // the slot of sample n + 3: last read two iterations ago
      lds_barrier();
    };
    int64_t n = 0;
    for (; n + NSET <= total; n += NSET) {
      iteration(std::integral_constant<int, 0>{}, n);
      iteration(std::integral_constant<int, 1>{}, n);
      iteration(std::integral_constant<int, 2>{}, n);
      if constexpr (NSET == 4) iteration(std::integral_constant<int, 3>{}, n);
    }
    if (n < total) iteration(std::integral_constant<int, 0>{}, n);
    if (n + 1 < total) iteration(std::integral_constant<int, 1>{}, n);
    if constexpr (NSET == 4) {
      if (n + 2 < total) iteration(std::integral_constant<int, 2>{}, n);
    }
    return;
  }

  // ---------------------------------- consumers ----------------------------------
  const int r = lane & 31, h = lane >> 5;
  const int rsw = (r >> 1) & 7;
  int frag_off[2][2];                            // [k step t][hi, lo]: byte offset of the lane's 16-byte piece in its row
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    frag_off[t][0] = r * 128 + (((2 * t + h) ^ rsw) << 4);
    frag_off[t][1] = r * 128 + (((4 + 2 * t + h) ^ rsw) << 4);
  }
  // The packed row of a sample leaves through LDS.  A 64-lane 4-byte store moves 256 bytes for the same
  // address work as a 16-byte store that moves 1 KiB, and the 160 4-byte stores per sample of the direct
  // kernel cost about as much as everything else together (measured on this kernel: 1.22 ms with them,
  // 0.74 without).  The accumulators are written into an LDS image of the row (lanes are consecutive
  // columns: conflict-free 128-byte runs; a lane whose element does not exist -- above the diagonal, rows
  // >= f -- writes to a per-lane dummy word, so there is no exec masking), and at the NEXT iteration the
  // four waves stream the finished image out with 16-byte stores, 1 KiB per instruction.  The per-lane LDS
  // offsets of a wave's blocks do not depend on the sample and are computed once.
  constexpr int NBLK = NB * (NB + 1) / 2, MB = (NBLK + 3) / 4;
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const int n4 = out_dim >> 2;                   // whole 16-byte chunks of a packed row
  char *const stage0 = f_lds + 2 * BUFB + 16;
  auto consumer_loop = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value;
    int soff[MB][16];                            // byte offset of acc[q] of the wave's idx-th block in the row image
    uint32_t anyv[MB];                           // bit q: some lane stores acc[q]
    {
      int idx = 0;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj) {
          if (((bi * (bi + 1) / 2 + bj) & 3) != W) continue;
          const int row0 = bi * 32 + 4 * h;
          const int tri0 = self ? row0 * (row0 + 1) / 2 : row0 * (row0 - 1) / 2;
          uint32_t any = 0u;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int dr = (q & 3) + 8 * (q >> 2);
            const int i = row0 + dr, j = bj * 32 + r;
            const bool keep = i < f && (self ? j <= i : j < i);
            // tri(row0 + dr) - tri(row0) = dr * row0 + dr (dr -+ 1) / 2
            soff[idx][q] = keep ? 4 * (tri0 + j + dr * row0 + (self ? dr * (dr + 1) / 2 : dr * (dr - 1) / 2))
                                : stage_bytes - 1024 + 4 * (W * 64 + lane);
            any |= (__ballot(keep) != 0ull ? 1u : 0u) << q;
          }
          anyv[idx] = any;
          ++idx;
        }
    }
    // sample m's finished image -> global: chunk c = 256 e + 64 W + lane
    // sample m's finished image -> global, in rounds of three 16-byte chunks per lane (chunk c = 768 k + 256 e +
    // 64 W + lane): a round's LDS reads are issued before a block's fragment reads and its stores after
    // the block's MFMAs, so that the store queue never holds up the products
    auto copy_reads = [&](const char *stage, int k, f32x4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 3; ++e)
        v[e] = *reinterpret_cast<const f32x4 *>(stage + 16 * min(768 * k + 256 * e + 64 * W + lane, n4 - 1));
    };
    auto copy_stores = [&](float *ob, int k, const f32x4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int c = 768 * k + 256 * e + 64 * W + lane;
        if (c < n4) *reinterpret_cast<f4u *>(ob + 4 * c) = v[e];
      }
    };
    auto copy_tail = [&](const char *stage, float *ob) __attribute__((always_inline)) {
      if ((out_dim & 3) != 0 && W == ((n4 >> 6) & 3)) {   // the last 1-3 elements: one lane
        if (lane == (n4 & 63)) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (4 * n4 + c < out_dim) ob[4 * n4 + c] = *reinterpret_cast<const float *>(stage + 16 * n4 + 4 * c);
        }
      }
    };
    auto copy_out = [&](const char *stage, int64_t m) __attribute__((always_inline)) {
      float *ob = out + (b0 + m * stride) * out_stride;
      for (int k = 0; 768 * k < n4; ++k) {
        f32x4 v[3];
        copy_reads(stage, k, v);
        copy_stores(ob, k, v);
      }
      copy_tail(stage, ob);
    };
    auto body = [&](auto pc, int64_t n) __attribute__((always_inline)) {
      constexpr int PAR = decltype(pc)::value;
      const char *buf = f_lds + PAR * BUFB;
      char *stage = stage0 + PAR * stage_bytes;
      // (spreading the copy rounds between the blocks -- reads before a block's fragments, stores after its
      // MFMAs -- was measured 30 % slower than copying first)
      if (n > 0) copy_out(stage0 + (1 - PAR) * stage_bytes, n - 1);
      const int k2 = 2 * pow2_exponent(slots[n & 3]);
      const float unscale = __uint_as_float((uint32_t)(127 - k2) << 23);
      int idx = 0;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj) {
          if (((bi * (bi + 1) / 2 + bj) & 3) != W) continue;   // compile time: this wave's blocks
          const uint32_t any = anyv[idx];
          if (any != 0u) {                                     // uniform (row blocks >= f have nothing to store)
            h16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) {
                fa[t][pl] = *reinterpret_cast<const h16x8 *>(buf + bi * 32 * 128 + frag_off[t][pl]);
                fb[t][pl] = *reinterpret_cast<const h16x8 *>(buf + bj * 32 * 128 + frag_off[t][pl]);
              }
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t][0], fb[t][0], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t][0], fb[t][1], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t][1], fb[t][0], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (any & (1u << q)) *reinterpret_cast<float *>(stage + soff[idx][q]) = acc[q] * unscale;
          }
          ++idx;
        }
      lds_barrier();
    };
    lds_barrier();
    lds_barrier();
    for (int64_t n = 0; n < total; n += 2) {
      body(std::integral_constant<int, 0>{}, n);
      if (n + 1 < total) body(std::integral_constant<int, 1>{}, n + 1);
    }
    copy_out(stage0 + ((total - 1) & 1) * stage_bytes, total - 1);
  };
  switch (wave) {
    case 0: consumer_loop(std::integral_constant<int, 0>{}); break;
    case 1: consumer_loop(std::integral_constant<int, 1>{}); break;
    case 2: consumer_loop(std::integral_constant<int, 2>{}); break;
    default: consumer_loop(std::integral_constant<int, 3>{}); break;
  }
}

template <int NB>
static void launch_dot_fwd_pc(const float *x, int64_t batch, int f, int self, float *out, hipStream_t s,
                              int64_t out_stride) {
  const int out_dim_ = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const size_t lds = (size_t)2 * NB * 32 * 128 + 16 + (size_t)2 * (((out_dim_ * 4 + 15) & ~15) + 1024);
  (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_fwd_pc_kernel<NB, 4>), 80 * 1024);
  const char *gv = option("TFRS_DOT_FWD_GRID");
  const dim3 grid((unsigned)std::min<int64_t>(batch, gv ? atoi(gv) : 512));   // two workgroups per CU
  hipLaunchKernelGGL((dot_interaction_fwd_pc_kernel<NB, 4>), grid, dim3(512), lds, s, x, batch, f, self, out, out_stride);
}

template <int DP, int NB>
static bool launch_dot_mfma_nb(const float *x, int64_t batch, int f, int d, int self, int skip,
                               float *out, hipStream_t s, int64_t out_stride) {
  const int out_dim = skip ? f * f : (self ? f * (f + 1) / 2 : f * (f - 1) / 2);
  const bool strided = out_stride != 0 && out_stride != out_dim;   // only the direct-store kernel takes a row stride
  const size_t lds = (size_t)(((out_dim + 3) & ~3) + 64) * sizeof(float);  // per wave
  // TFRS_DOT_STAGE=0: direct 128-byte-run stores from the accumulators instead of the LDS-staged
  // linear copy-out (measurement switch)
  const char *sv = option("TFRS_DOT_STAGE");
  const bool stage = !skip && lds <= 64 * 1024 && !(sv && sv[0] == '0');
  if (stage) {
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_mfma_kernel<DP, NB, true>), 64 * 1024);
    const int64_t per_cu = std::min<int64_t>(8, std::max<int64_t>(1, (int64_t)(160 * 1024) / (int64_t)lds));
    const dim3 grid((unsigned)std::min<int64_t>(batch, 256 * per_cu * 2));
    // TFRS_DOT_FWD=f32 keeps the exact-f32 MFMA chain (measurement / comparison switch)
    const char *fv = option("TFRS_DOT_FWD");
    if constexpr (DP % 16 == 0 && NB * (DP / 2) <= 96) {   // (beyond that the raw + hi + lo operands spill)
      if constexpr (DP == 32) {
        // default at d == 32: loads and stores on different waves; TFRS_DOT_FWD=direct / staged / f32 select
        // the earlier generations (measurement switches)
        // (one row block, F <= 32, would leave three of the four product waves idle: 0.205 against 0.141 ms at F = 27)
        if (!fv && d == 32 && NB >= 2 && batch >= 512) {
          launch_dot_fwd_pc<NB>(x, batch, f, self, out, s, strided ? out_stride : (int64_t)out_dim);
          return true;
        }
      }
      if (!(fv && (fv[0] == 'f' || fv[0] == 's'))) {   // direct stores
        const dim3 gd((unsigned)std::min<int64_t>((batch + 3) / 4, 256 * 8));
        // three workgroups per CU: 142 VGPRs, no scratch (four would spill 16 registers: 1.18 vs 1.06 ms)
        hipLaunchKernelGGL((dot_interaction_f16x3_direct_kernel<DP, NB, 3>), gd, dim3(256), 0, s, x, batch, f, d, self, out,
                           strided ? out_stride : (int64_t)out_dim);
        return true;
      }
      if (strided) return false;
      if (!(fv && fv[0] == 'f' && fv[1] == '3')) {
        (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_f16x3_kernel<DP, NB>), 64 * 1024);
        hipLaunchKernelGGL((dot_interaction_f16x3_kernel<DP, NB>), grid, dim3(64), lds, s, x, batch, f, d, self, out);
        return true;
      }
    }
    if (strided) return false;
    hipLaunchKernelGGL((dot_interaction_mfma_kernel<DP, NB, true>), grid, dim3(64), lds, s, x, batch, f, d,
                       self, skip, out);
  } else {
    if (strided) return false;
    const dim3 grid((unsigned)std::min<int64_t>((batch + 3) / 4, 256 * 16));
    hipLaunchKernelGGL((dot_interaction_mfma_kernel<DP, NB, false>), grid, dim3(256), 0, s, x, batch, f, d,
                       self, skip, out);
  }
  return true;
}

template <int DP>
static bool launch_dot_mfma_dp(const float *x, int64_t batch, int f, int d, int self, int skip,
                               float *out, hipStream_t s, int64_t out_stride) {
  const int nb = (f + 31) / 32;
  if (nb * (DP / 2) > 128) return false;  // register budget of the resident X
  switch (nb) {
    case 1: return launch_dot_mfma_nb<DP, 1>(x, batch, f, d, self, skip, out, s, out_stride);
    case 2: return launch_dot_mfma_nb<DP, 2>(x, batch, f, d, self, skip, out, s, out_stride);
    case 3: return launch_dot_mfma_nb<DP, 3>(x, batch, f, d, self, skip, out, s, out_stride);
    case 4: return launch_dot_mfma_nb<DP, 4>(x, batch, f, d, self, skip, out, s, out_stride);
    default: return false;
  }
}

// MFMA form of the backward pass (packed-triangle output only): with S = L + L^T, L the
// lower-triangular matrix holding the incoming gradient of the packed pairs (diagonal kept
// for self_interaction, so S_ii = 2 dG_ii), dX = S X.  One wave per sample (one-wave
// workgroups, grid-stride): the packed gradient (out_dim floats) is staged in LDS with
// linear coalesced loads; the B operand (X, lane = feature) is read once from global into
// registers; the A operand S[i][k] is gathered from the packed LDS copy with incremental
// triangular offsets; every 32-row block of dX is NB*16 v_mfma_f32_32x32x2_f32 per 32
// features; rows leave as 128-byte runs per half-wave.
template <int DP, int NB>
__global__ void __launch_bounds__(64, 2) dot_interaction_bwd_mfma_kernel(
    const float *__restrict__ x, const float *__restrict__ dout, int64_t batch, int f, int d,
    int self, float *__restrict__ dx) {
  constexpr int NFB = (DP + 31) / 32;   // 32-feature output blocks
  constexpr int KS = NB * 16;           // MFMA steps over k = 0 .. 32*NB - 1 (2 per step)
  extern __shared__ __attribute__((aligned(16))) float smem_db[];
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  float *ls = smem_db;  // packed gradient, out_dim floats
  for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
    const float *dy = dout + b * (int64_t)out_dim;
    const float *xb = x + b * (int64_t)f * d;
    // Opaque zero: keeps per-sample index math from being hoisted out of the sample loop.
    int lz = 0;
    asm volatile("" : "+v"(lz));
    const int jv = j + lz, hv = h + lz;
    // B operand straight from global (lanes = consecutive features: 128-byte runs), resident
    // in registers for the whole sample: X[k = 2s + h][feature fb*32 + j]
    float bx[NFB][KS];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int row = 2 * s + hv, ft = fb * 32 + jv;
        bx[fb][s] = (row < f && ft < d) ? xb[row * d + ft] : 0.0f;
      }
    for (int e = lane; e < out_dim; e += 64) ls[e] = dy[e];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (bi * 32 >= f) continue;  // uniform
      const int i = bi * 32 + jv;  // A-operand row of this lane
      const int tri_i = self ? i * (i + 1) / 2 : i * (i - 1) / 2;
      const bool i_ok = i < f;
      f32x16 acc[NFB];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fb][r] = 0.0f;
      // k = 2s + h; tri(k) advances by 2k + (self ? 3 : 1) per step.  The A operand S[i][k] is
      // gathered branch-free from the packed LDS copy, 8 steps at a time ahead of their MFMAs.
      int lzb = 0;  // (opaque zero per row block: the k / tri(k) sequences are recomputed, not
      asm volatile("" : "+v"(lzb));  //  kept alive across the four blocks)
      int k = hv + lzb;
      int tri_k = self ? k * (k + 1) / 2 : k * (k - 1) / 2;
#pragma unroll
      for (int s0 = 0; s0 < KS; s0 += 8) {
        float av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = i_ok && k < f && (self || k != i);
          const int idx = ok ? (k < i ? tri_i + k : tri_k + i) : 0;
          float a = ls[idx];
          a = ok ? a : 0.0f;
          av[u] = (k == i) ? 2.0f * a : a;
          tri_k += 2 * k + (self ? 3 : 1);
          k += 2;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb)
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bx[fb][s0 + u], acc[fb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // one chunk of gathers in flight, not all of them
      }
      // acc[fb][r] = dX[bi*32 + tile_row_of_reg(r, h)][fb*32 + j]
      float *db = dx + b * (int64_t)f * d;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        const int ft = fb * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = bi * 32 + tile_row_of_reg(r, h);
          if (row < f && ft < d) db[row * d + ft] = acc[fb][r];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // LDS is refilled by the next sample
  }
}

// Backward, third generation (default for packed-triangle outputs): dX = S X with the symmetric
// S = L + L^T unpacked ONCE per sample into a dense LDS tile.
//   workgroup = 4 waves = one sample at a time (grid-stride); 2-3 workgroups per CU run out of
//               phase, so one's unpack / stores overlap another's MFMAs.
//   unpack    : the packed gradient is read linearly from HBM (coalesced dwords), every element
//               dy[p] = L[i][j] is written to S[i][j] and S[j][i] (diagonal doubled for
//               self_interaction).  S rows are KH2 + 4 floats (an ODD number of 16-byte slots),
//               the diagonal (no self) and the k >= f pad columns stay zero from kernel start.
//   products  : wave w owns row block w (32 rows of dX).  The k order is the "half split":
//               lane half h covers k in [h * KH, (h + 1) * KH), so a lane's A operand is KH
//               CONTIGUOUS floats of its S row -- KH / 4 conflict-free ds_read_b128, every byte
//               used (the second-generation kernel gathered 4-byte elements through triangular
//               index arithmetic: 256 scattered LDS reads per row block).  The B operand
//               X[k][feature] comes straight from global (128-byte runs per half wave; the four
//               waves' re-reads hit L2).  KH steps of v_mfma_f32_32x32x2_f32 per 32 features.
//   stores    : dX rows leave as 128-byte runs.
// Bytes: B * (F*D*2 + out_dim) * 4 algorithmic; MFMA floor at F = 101, D = 32: 52 steps x 4
// blocks x 64 cycles per sample = 0.71 ms for 131072 samples -- next to the 0.76 ms HBM floor.
template <int NFB, int MAXE>
__global__ void __launch_bounds__(256, 2) dot_interaction_bwd_dense_kernel(
    const float *__restrict__ x, const float *__restrict__ dout, int64_t batch, int f, int d,
    int self, int kh, float *__restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int ld = 2 * kh + 4;                 // floats per S row: (2 kh + 4) / 4 is odd
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  for (int e = tid; e < f * ld; e += 256) s_lds[e] = 0.0f;   // diagonal + pad columns stay zero
  // Thread t owns the packed elements p = t + 256 e of EVERY sample: their two S positions are
  // computed once (low half: i * ld + jx, high half: jx * ld + i; equal on the diagonal).
  // Elements past out_dim go to a per-thread dummy slot behind the tile (branch-free unpack).
  uint32_t pos[MAXE];
  uint64_t diag_bits = 0;                    // bit e: element e lies on the diagonal (value doubled)
  const uint32_t dummy = (uint32_t)(f * ld + tid);
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int p = tid + 256 * e;
    int i = 0, jx = 0;
    if (p < out_dim) pair_of_index(p, self != 0, &i, &jx);
    pos[e] = (p < out_dim) ? ((uint32_t)(i * ld + jx) | ((uint32_t)(jx * ld + i) << 16)) : (dummy | (dummy << 16));
    if (p < out_dim && i == jx) diag_bits |= 1ull << e;
  }
  __syncthreads();
  const int row = wave * 32 + j;             // this lane's S row (A operand) in the product phase
  const bool row_ok = row < f;
  const bool active = wave * 32 < f;         // wave-uniform
  const int nquad = kh >> 2;
  const int ngrp = (nquad + 3) >> 2;

  // the packed gradient of a sample, MAXE coalesced dwords per thread, held in registers: the
  // NEXT sample's loads are in flight during the products of the current one
  float gy[MAXE];
  auto load_dy = [&](int64_t b) __attribute__((always_inline)) {
    const float *dy = dout + b * (int64_t)out_dim;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int p = tid + 256 * e;
      gy[e] = (p < out_dim) ? dy[p] : 0.0f;
    }
  };
  int64_t b = blockIdx.x;
  if (b < batch) load_dy(b);
  for (; b < batch; b += gridDim.x) {
    const float *xb = x + b * (int64_t)f * d;
    // B operand X[k][feature] straight from global in groups of 16 k-steps, double-buffered in
    // registers: group g + 1 is in flight under the 16 MFMAs (1024 pipe cycles) of group g; the
    // first group is issued before the unpack.
    auto load_b = [&](float (&bv)[16], int fb, int grp) __attribute__((always_inline)) {
      const int ft = fb * 32 + j;
      const float *col = xb + ft;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = h * kh + grp * 16 + u;
        bv[u] = (k < f && ft < d) ? col[(int64_t)k * d] : 0.0f;   // (k >= f also covers the pad steps)
      }
    };
    float b0[16], b1[16];
    if (active) load_b(b0, 0, 0);
    // ---- unpack (S was released by the barrier that ended the previous sample) ----------------
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const float v = ((diag_bits >> e) & 1ull) ? 2.0f * gy[e] : gy[e];
      s_lds[pos[e] & 0xFFFFu] = v;
      s_lds[pos[e] >> 16] = v;
    }
    __syncthreads();
    if (b + gridDim.x < batch) load_dy(b + gridDim.x);
    // ---- products: dX[row block `wave`] = S[rows, :] X ----------------------------------------
    if (active) {
      const float *srow = s_lds + (row_ok ? row : 0) * ld + h * kh;
      auto mfma16 = [&](f32x16 &acc, const float (&bv)[16], int grp) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (grp * 4 + q < nquad) {         // uniform
            f32x4 a4 = *reinterpret_cast<const f32x4 *>(srow + grp * 16 + 4 * q);
            if (!row_ok) a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], bv[4 * q + u], acc, 0, 0, 0);
          }
        }
      };
#pragma unroll 1
      for (int fb = 0; fb < NFB; ++fb) {
        if (fb * 32 >= d) break;
        if (fb > 0) load_b(b0, fb, 0);
        const int ft = fb * 32 + j;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
        for (int grp = 0; grp < ngrp; grp += 2) {
          if (grp + 1 < ngrp) load_b(b1, fb, grp + 1);
          mfma16(acc, b0, grp);
          if (grp + 1 < ngrp) {
            if (grp + 2 < ngrp) load_b(b0, fb, grp + 2);
            mfma16(acc, b1, grp + 1);
          }
        }
        float *db = dx + b * (int64_t)f * d;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int orow = wave * 32 + tile_row_of_reg(r, h);
          if (orow < f && ft < d) db[orow * d + ft] = acc[r];
        }
      }
    }
    __syncthreads();   // S is rewritten by the next sample
  }
}

// Backward, fourth generation (d <= 32): the dense-S kernel above with the memory side and the
// products on DIFFERENT waves.  One workgroup of 8 waves per CU: waves 4-7 (producers) bring
// sample t + 1 into LDS -- the packed gradient unpacked into one half of a double-buffered S tile,
// X copied into one half of a double-buffered [2 kh, 32] tile, their global loads for sample t + 2
// already in flight in registers -- while waves 0-3 (consumers, one per SIMD) run the 52 MFMA
// steps of sample t with BOTH operands read from LDS, so the matrix pipe only pauses for the one
// barrier per sample and never waits for HBM.  (In the single-role kernel every wave alternated
// unpack -> barrier -> products -> barrier and fetched its B operand from global one MFMA group
// ahead: the MFMA pipe was busy 29 % of the time.)
template <int MAXE>
__global__ void __launch_bounds__(512) dot_interaction_bwd_pc_kernel(
    const float *__restrict__ x, const float *__restrict__ dout, int64_t batch, int f, int d,
    int self, int kh, float *__restrict__ dx, int64_t dout_stride) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  constexpr int MAXX = 16;                   // X elements per producer thread: f * d <= 128 * 32
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;           // wave-uniform
  const int j = lane & 31, h = lane >> 5;
  const int ld = 2 * kh + 4;                 // floats per S row: (2 kh + 4) / 4 is odd
  const int tile = f * ld + 256;             // one S buffer: tile + 256 dummy slots
  const int xtile = 2 * kh * 32;             // one X buffer: rows >= f and columns >= d stay zero
  float *const xs_base = s_lds + 2 * tile;
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const int xn = f * d;
  for (int e = tid; e < 2 * tile + 2 * xtile; e += 512) s_lds[e] = 0.0f;
  const int ptid = tid & 255;                // producer thread index
  uint32_t pos[MAXE];
  uint64_t diag_bits = 0;
  float gy0[MAXE], gx0[MAXX];   // the next sample, in flight during the current one's products
  const uint32_t dummy = (uint32_t)(f * ld + ptid);
  if (producer) {
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int p = ptid + 256 * e;
      int i = 0, jx = 0;
      if (p < out_dim) pair_of_index(p, self != 0, &i, &jx);
      pos[e] = (p < out_dim) ? ((uint32_t)(i * ld + jx) | ((uint32_t)(jx * ld + i) << 16)) : (dummy | (dummy << 16));
      if (p < out_dim && i == jx) diag_bits |= 1ull << e;
    }
  }
  uint16_t xpos[MAXX];                        // X element p = ptid + 256 e -> row * 32 + column
#pragma unroll
  for (int e = 0; e < MAXX; ++e) {
    const int p = ptid + 256 * e;
    xpos[e] = (uint16_t)((p / d) * 32 + (p % d));
  }
  auto load_sample = [&](float (&gy)[MAXE], float (&gx)[MAXX], int64_t b) __attribute__((always_inline)) {
    const float *dy = dout + b * dout_stride;   // (rows of a wider matrix: tfrs_dot_interaction_bwd_strided)
    const float *xb = x + b * (int64_t)xn;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int p = ptid + 256 * e;
      gy[e] = (p < out_dim) ? dy[p] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < MAXX; ++e) {
      const int p = ptid + 256 * e;
      gx[e] = (p < xn) ? xb[p] : 0.0f;
    }
  };
  auto unpack = [&](const float (&gy)[MAXE], const float (&gx)[MAXX], float *sb, float *xsb)
      __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const float v = ((diag_bits >> e) & 1ull) ? 2.0f * gy[e] : gy[e];
      sb[pos[e] & 0xFFFFu] = v;
      sb[pos[e] >> 16] = v;
    }
#pragma unroll
    for (int e = 0; e < MAXX; ++e) {
      const int p = ptid + 256 * e;
      if (p < xn) xsb[xpos[e]] = gx[e];
    }
  };
  const int row = (wave & 3) * 32 + j;       // consumer: this lane's S row (A operand)
  const bool row_ok = row < f;
  const bool active = !producer && wave * 32 < f;
  const int nquad = kh >> 2;
  const int64_t stride = gridDim.x;
  __syncthreads();                           // zero fill done
  int64_t b = blockIdx.x;
  if (b < batch && producer) {
    load_sample(gy0, gx0, b);
    unpack(gy0, gx0, s_lds, xs_base);
    if (b + stride < batch) load_sample(gy0, gx0, b + stride);
  }
  __syncthreads();
  for (int t = 0; b < batch; b += stride, ++t) {
    if (producer) {
      if (b + stride < batch) {
        unpack(gy0, gx0, s_lds + ((t + 1) & 1) * tile, xs_base + ((t + 1) & 1) * xtile);
        if (b + 2 * stride < batch) load_sample(gy0, gx0, b + 2 * stride);
      }
    } else if (active) {
      const float *srow = s_lds + (t & 1) * tile + (row_ok ? row : 0) * ld + h * kh;
      const float *xcol = xs_base + (t & 1) * xtile + (h * kh) * 32 + j;   // X[h * kh + s][j]
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      // one quad (4 k-steps) ahead: the LDS reads of quad q + 1 are issued before the 4 MFMAs
      // (256 pipe cycles) of quad q
      f32x4 a_cur = *reinterpret_cast<const f32x4 *>(srow);
      float b_cur[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b_cur[u] = xcol[u * 32];
#pragma unroll 2
      for (int q = 0; q < nquad; ++q) {
        f32x4 a_nxt = a_cur;
        float b_nxt[4] = {0.f, 0.f, 0.f, 0.f};
        if (q + 1 < nquad) {                 // uniform
          a_nxt = *reinterpret_cast<const f32x4 *>(srow + 4 * (q + 1));
#pragma unroll
          for (int u = 0; u < 4; ++u) b_nxt[u] = xcol[(4 * (q + 1) + u) * 32];
        }
        f32x4 a4 = a_cur;
        if (!row_ok) a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b_cur[u], acc, 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int u = 0; u < 4; ++u) b_cur[u] = b_nxt[u];
      }
      float *db = dx + b * (int64_t)xn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = wave * 32 + tile_row_of_reg(r, h);
        if (orow < f && j < d) db[orow * d + j] = acc[r];
      }
    }
    __syncthreads();
  }
}


// Backward, fifth generation (d == 32, default): split-fp16 matrix cores on the PACKED gradient.
//
// What round 3 measured on the producer / consumer kernel above (tools/exp_dotbwd.py, 2.0 ms box):
// the consumers alone (no loads, no unpack) take 1.57 ms -- 52 dependent v_mfma_f32_32x32x2_f32 per
// 32-row block with the LDS reads of the next quad between them run at ~100 cycles each, not 64 --
// and the producers alone 1.2 ms: one sample of loads in flight per CU, issued right before the
// barrier and awaited right after it, so every sample pays a full HBM round trip.  Both are removed:
//
// * operands are split fp16 (x s = hi + lo, s a power of two PER SAMPLE from the sample's max |dy| and
//   max |x|): a 32 x 32 x 16 step is three v_mfma_f32_32x32x16_f16 (hi hi + hi lo + lo hi), 96 pipe cycles
//   for what took 512; 21 MFMAs per row block and sample instead of 52 four-times-slower ones;
// * the gradient stays PACKED in LDS (row-major lower triangle, as it arrives): element p of the
//   sample becomes ONE 32-bit word {hi, lo} at word p -- one ds_write_b128 per four elements, no
//   transposed second copy (half of the old kernel's LDS cycles were its bank conflicts), 25 KB per
//   sample instead of 45.  The wave that owns rows [32 rb, 32 rb + 32) of dX = (L + L^T) X walks the 16-wide
//   k steps s of S = L + L^T: below the diagonal block (s < 2 rb) its A operand is 8 consecutive words of
//   its row of L; right of it (s >= 2 rb + 2) it is 8 words of COLUMN i of L -- lanes are consecutive
//   columns, so those reads are conflict-free too; the two steps on the diagonal take both, masked,
//   added as packed halves (which also doubles the diagonal for self_interaction);
// * the producers keep NSET samples in registers: at iteration n they convert sample n + 1 (its scale
//   was reduced into LDS during iteration n - 1: wave max by DPP, one ds_max per wave), issue the
//   loads of sample n + NSET + 1 into the registers that freed, and reduce the max of sample n + 2 --
//   so NSET - 2 samples of 16-byte loads are always in flight and none is awaited for two iterations;
// * the per-iteration barrier waits for LDS only, not for the consumers' dX stores.
//
// The region of an L buffer beyond the packed triangle is zero-filled once and never written: reads of
// rows / columns >= f land there.  X is words [16 NSTEP][32], rows >= f zero.  Error: 2^-22 per
// operand element relative to the element (for elements within 2^-18 of the sample maximum), the
// dropped lo lo term 2^-22: the same bound as the forward's.
template <int NSTEP, int NEY, int NSET>
__global__ void __launch_bounds__(512) dot_interaction_bwd_h16_kernel(
    const float *__restrict__ x, const float *__restrict__ dout, int64_t batch, int f, int self,
    int lwords, float *__restrict__ dx, int64_t dout_stride) {
  extern __shared__ __attribute__((aligned(16))) uint32_t w_lds[];
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int D = 32, NEX = 4;             // f * 32 / 4 / 256 <= 4 chunks of X per producer thread
  constexpr int XW = 16 * NSTEP * D;         // X words per buffer
  const int bufw = lwords + XW;
  uint32_t *const slots = w_lds + 2 * bufw;  // [4][2]: bits of max |dy|, max |x| of samples n % 4
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const int xn = f * D;
  const int64_t stride = gridDim.x;
  const int64_t b0 = blockIdx.x;
  for (int e = tid; e < 2 * bufw + 8; e += 512) w_lds[e] = 0u;
  __syncthreads();
  // LDS-only barrier: every wave's LDS traffic is complete, global stores may still be in flight
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto pow2_exponent = [](uint32_t mbits) __attribute__((always_inline)) -> int {
    // k with max * 2^k in [2^12, 2^13): twice that (the diagonal) stays far below fp16's 65504
    int k = 139 - (int)(mbits >> 23);
    return (mbits > 0u && mbits < 0x7F800000u) ? min(max(k, -60), 60) : 0;
  };

  if (wave >= 4) {
    // ---------------------------------- producers ----------------------------------
    const int ptid = tid & 255;
    f32x4 ry[NSET][NEY], rx[NSET][NEX];
    // Every thread ALWAYS issues its NEY + NEX 16-byte loads, at addresses clamped into the row, and
    // repairs the clamped chunks with selects afterwards: gfx950 counts outstanding loads in order
    // (vmcnt), so a load inside a branch makes the count unknown to the compiler, which then waits
    // for ALL loads (vmcnt(0)) wherever it needs one -- the first version of this kernel guarded its
    // loads with `if (p < out_dim)` and had no prefetch at all (every wait in its ISA was vmcnt(0)).
    // A clamped chunk holds elements [q, q + 4), q = min(p0, n - 4); element p0 + c is lane value c + (p0 - q).
    auto shifted = [](f32x4 v, int sft) __attribute__((always_inline)) -> f32x4 {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = 0.0f;
#pragma unroll
        for (int k = c; k < 4; ++k) t = (sft == k - c) ? v[k] : t;
        o[c] = t;
      }
      return o;
    };
    const int ey_fix = (((out_dim - 4) >> 2) + 1) >> 8;   // first e with chunks that are clamped (uniform)
    const int ex_fix = (xn >> 2) >> 8;
    auto load_sample = [&](f32x4 (&gy)[NEY], f32x4 (&gx)[NEX], int64_t b) __attribute__((always_inline)) {
      const int64_t bl = b < batch ? b : batch - 1;                  // past the end: reload the last sample (unused)
      const float *dy = dout + bl * dout_stride;                     // (rows of a wider matrix: _bwd_strided)
      const float *xb = x + bl * (int64_t)xn;
#pragma unroll
      for (int e = 0; e < NEY; ++e) {
        const int p0 = 4 * (ptid + 256 * e);
        gy[e] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(dy + min(p0, out_dim - 4)));
      }
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        const int p0 = 4 * (ptid + 256 * e);
        gx[e] = *reinterpret_cast<const f4u *>(xb + min(p0, xn - 4));
      }
    };
    // after the loads have landed: chunks cut by (or beyond) the row's end
    auto repair_sample = [&](f32x4 (&gy)[NEY], f32x4 (&gx)[NEX]) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < NEY; ++e) {
        if (e >= ey_fix) {                                           // uniform
          const int p0 = 4 * (ptid + 256 * e);
          gy[e] = shifted(gy[e], p0 - min(p0, out_dim - 4));
        }
      }
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        if (e >= ex_fix) {
          const int p0 = 4 * (ptid + 256 * e);
          gx[e] = shifted(gx[e], p0 - min(p0, xn - 4));
        }
      }
    };
    auto max_sample = [&](f32x4 (&gy)[NEY], f32x4 (&gx)[NEX], int slot) __attribute__((always_inline)) {
      repair_sample(gy, gx);
      float my = 0.0f, mx = 0.0f;
#pragma unroll
      for (int e = 0; e < NEY; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) my = fmaxf(my, __builtin_fabsf(gy[e][c]));
#pragma unroll
      for (int e = 0; e < NEX; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) mx = fmaxf(mx, __builtin_fabsf(gx[e][c]));
      my = wave_max_nonneg(my);
      mx = wave_max_nonneg(mx);
      if (lane == 0) {
        atomicMax(&slots[2 * slot], __float_as_uint(my));
        atomicMax(&slots[2 * slot + 1], __float_as_uint(mx));
      }
    };
    auto split_word = [](float a) __attribute__((always_inline)) -> uint32_t {
      union {
        h16x2 h;
        uint32_t u;
      } w;
      const _Float16 hi = (_Float16)a;      // round to nearest even
      w.h[0] = hi;
      w.h[1] = (_Float16)(a - (float)hi);
      return w.u;
    };
    auto convert_sample = [&](const f32x4 (&gy)[NEY], const f32x4 (&gx)[NEX], int slot, uint32_t *buf)
        __attribute__((always_inline)) {
      const float sy = __uint_as_float((uint32_t)(127 + pow2_exponent(slots[2 * slot])) << 23);
      const float sx = __uint_as_float((uint32_t)(127 + pow2_exponent(slots[2 * slot + 1])) << 23);
#pragma unroll
      for (int e = 0; e < NEY; ++e) {
        const int w0 = 4 * (ptid + 256 * e);   // elements beyond the row's end are 0 (repair_sample)
        if (w0 < out_dim) {
          u32x4 w;
#pragma unroll
          for (int c = 0; c < 4; ++c) w[c] = split_word(gy[e][c] * sy);
          *reinterpret_cast<u32x4 *>(buf + w0) = w;
        }
      }
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        const int p0 = 4 * (ptid + 256 * e);
        if (p0 < xn) {
          u32x4 w;
#pragma unroll
          for (int c = 0; c < 4; ++c) w[c] = split_word(gx[e][c] * sx);
          *reinterpret_cast<u32x4 *>(buf + lwords + p0) = w;
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NSET; ++s) load_sample(ry[s], rx[s], b0 + s * stride);
    max_sample(ry[0], rx[0], 0);
    lds_barrier();
    convert_sample(ry[0], rx[0], 0, w_lds);
    load_sample(ry[0], rx[0], b0 + NSET * stride);
    max_sample(ry[1 % NSET], rx[1 % NSET], 1);
    lds_barrier();
    // iteration n = n0 + U of the steady state; U is a compile-time constant so that the register sets
    // rotate by NAME (no copies) and every load has a fixed position in the vmcnt order
    auto iteration = [&](auto uc, int64_t n0) __attribute__((always_inline)) {
      constexpr int U = decltype(uc)::value;
      const int64_t n = n0 + U;
      const int sl = (int)(n & 3);
      convert_sample(ry[(U + 1) % NSET], rx[(U + 1) % NSET], (sl + 1) & 3, w_lds + ((n + 1) & 1) * bufw);
      load_sample(ry[(U + 1) % NSET], rx[(U + 1) % NSET], b0 + (n + NSET + 1) * stride);
      max_sample(ry[(U + 2) % NSET], rx[(U + 2) % NSET], (sl + 2) & 3);
      if (ptid == 0) {                         // the slot of sample n + 3: last read two iterations ago
        slots[2 * ((sl + 3) & 3)] = 0u;
        slots[2 * ((sl + 3) & 3) + 1] = 0u;
      }
      lds_barrier();
    };
    const int64_t total = (batch - b0 + stride - 1) / stride;   // samples of this workgroup (>= 1)
    int64_t n = 0;
    for (; n + NSET <= total; n += NSET) {
      iteration(std::integral_constant<int, 0>{}, n);
      iteration(std::integral_constant<int, 1>{}, n);
      iteration(std::integral_constant<int, 2>{}, n);
      if constexpr (NSET == 4) iteration(std::integral_constant<int, 3>{}, n);
    }
    if (n < total) iteration(std::integral_constant<int, 0>{}, n);
    if (n + 1 < total) iteration(std::integral_constant<int, 1>{}, n);
    if constexpr (NSET == 4) {
      if (n + 2 < total) iteration(std::integral_constant<int, 2>{}, n);
    }
    return;
  }

  // ---------------------------------- consumers ----------------------------------
  const int r = lane & 31, h = lane >> 5, P = 8 * h;
  const int i = wave * 32 + r;                 // this lane's row of S (A operand) = its column of L
  const int ic = min(i, f);                    // rows >= f read the zero region behind the triangle
  const int ra = (self ? ic * (ic + 1) / 2 : ic * (ic - 1) / 2) + P;   // L[ic][P + m] = word ra + m
  // L[m + P][i] = word tri(m + P) + self (m + P) + i = ca + m cb + tri(m), tri(t) = t (t - 1) / 2
  const int ca = i + P * (P - 1) / 2 + self * P, cb = P + self;
  const int xa = lwords + 32 * P + r;          // X[P + m][r] = word xa + 32 m
  uint32_t mlo[2][8], mup[2][8];               // the two steps on the diagonal: which slots are below / above it
#pragma unroll
  for (int ss = 0; ss < 2; ++ss)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = r - P - 16 * ss;           // slot u is column q of the lane's row, relative to the diagonal
      mlo[ss][u] = (u <= q + self - 1) ? 0xFFFFFFFFu : 0u;
      mup[ss][u] = (u >= q - self + 1) ? 0xFFFFFFFFu : 0u;
    }
  const bool active = wave * 32 < f;
  auto halves = [](const uint32_t (&w)[8], h16x8 *hi, h16x8 *lo) __attribute__((always_inline)) {
    union {
      uint32_t u[4];
      h16x8 v;
    } a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a.u[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x05040100u);   // {hi of 2e, hi of 2e + 1}
      b.u[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x07060302u);   // {lo of 2e, lo of 2e + 1}
    }
    *hi = a.v;
    *lo = b.v;
  };
  auto consume = [&](auto rbc, const uint32_t *xbuf, int slot, int64_t b) __attribute__((always_inline)) {
    constexpr int RB = decltype(rbc)::value;
    const uint32_t *buf = xbuf;
    const int k = pow2_exponent(slots[2 * slot]) + pow2_exponent(slots[2 * slot + 1]);
    const float inv = __uint_as_float((uint32_t)(127 - k) << 23);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    // the words of step s: X[16 s + P + u][r] and S[i][16 s + P + u]
    auto fetch = [&](auto sc, uint32_t (&w)[8], uint32_t (&xw)[8]) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
#pragma unroll
      for (int u = 0; u < 8; ++u) xw[u] = buf[xa + 32 * (16 * s + u)];
      if constexpr (s < 2 * RB) {              // below the diagonal block: 8 consecutive words of row i
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = buf[ra + 16 * s + u];
      } else if constexpr (s < 2 * RB + 2) {   // on it: row part + column part (the diagonal from both)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          constexpr int ss = s - 2 * RB;
          const int m = 16 * s + u;
          union {
            uint32_t u32;
            h16x2 h2;
          } lo_w, up_w, sum;
          lo_w.u32 = buf[ra + m] & mlo[ss][u];
          up_w.u32 = buf[ca + __mul24(m, cb) + m * (m - 1) / 2] & mup[ss][u];
          sum.h2 = lo_w.h2 + up_w.h2;
          w[u] = sum.u32;
        }
      } else {                                 // right of it: 8 words of column i of L (rows 16 s + P + u)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int m = 16 * s + u;
          w[u] = buf[ca + __mul24(m, cb) + m * (m - 1) / 2];
        }
      }
    };
    // one step ahead: the LDS reads of step s + 1 are issued before the permutes and MFMAs of step s
    uint32_t w[2][8], xw[2][8];
    fetch(std::integral_constant<int, 0>{}, w[0], xw[0]);
    auto step = [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 1 < NSTEP) fetch(std::integral_constant<int, s + 1>{}, w[(s + 1) & 1], xw[(s + 1) & 1]);
      h16x8 ah, al, xh, xl;
      halves(w[s & 1], &ah, &al);
      halves(xw[s & 1], &xh, &xl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc, 0, 0, 0);
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (NSTEP > 1) step(std::integral_constant<int, 1>{});
    if constexpr (NSTEP > 2) step(std::integral_constant<int, 2>{});
    if constexpr (NSTEP > 3) step(std::integral_constant<int, 3>{});
    if constexpr (NSTEP > 4) step(std::integral_constant<int, 4>{});
    if constexpr (NSTEP > 5) step(std::integral_constant<int, 5>{});
    if constexpr (NSTEP > 6) step(std::integral_constant<int, 6>{});
    if constexpr (NSTEP > 7) step(std::integral_constant<int, 7>{});
    // lane = dim, register q = row: a store instruction writes two whole 128-byte rows.  (The transposed
    // product X^T S would leave four 16-byte stores per wave instead of sixteen 4-byte ones, but each
    // of them touches 32 rows 32 bytes at a time: measured 1.31 against 1.11 ms.)
    float *db = dx + b * (int64_t)xn;
    if (RB * 32 + 32 <= f) {                   // uniform: a whole row block
#pragma unroll
      for (int q = 0; q < 16; ++q) db[(RB * 32 + tile_row_of_reg(q, h)) * D + r] = acc[q] * inv;
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int orow = RB * 32 + tile_row_of_reg(q, h);
        if (orow < f) db[orow * D + r] = acc[q] * inv;
      }
    }
  };
  auto consumer_loop = [&](auto rbc) __attribute__((always_inline)) {
    lds_barrier();
    lds_barrier();
    const int64_t total = (batch - b0 + stride - 1) / stride;
    for (int64_t n = 0; n < total; ++n) {
      if (active) consume(rbc, w_lds + (n & 1) * bufw, (int)(n & 3), b0 + n * stride);
      lds_barrier();
    }
  };
  switch (wave) {
    case 0: consumer_loop(std::integral_constant<int, 0>{}); break;
    case 1: consumer_loop(std::integral_constant<int, 1>{}); break;
    case 2: consumer_loop(std::integral_constant<int, 2>{}); break;
    default: consumer_loop(std::integral_constant<int, 3>{}); break;
  }
}

// L words per buffer: the packed triangle of 16 NSTEP rows (reads of rows / columns beyond f stay
// inside the zero region) + one 128-word row of slack for columns >= f
static int dot_bwd_h16_lwords(int nstep, int self) {
  const int n = 16 * nstep;
  return ((self ? n * (n + 1) / 2 : n * (n - 1) / 2) + 128 + 3) & ~3;
}

template <int NSTEP, int NEY, int NSET>
static void launch_dot_bwd_h16_v(const float *x, const float *dout, int64_t batch, int f, int self, dim3 grid,
                                 float *dx, hipStream_t s, int64_t dout_stride) {
  const int lwords = dot_bwd_h16_lwords(NSTEP, self);
  const size_t lds = ((size_t)2 * (lwords + 16 * NSTEP * 32) + 8) * sizeof(uint32_t);
  (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_bwd_h16_kernel<NSTEP, NEY, NSET>), 160 * 1024);
  hipLaunchKernelGGL((dot_interaction_bwd_h16_kernel<NSTEP, NEY, NSET>), grid, dim3(512), lds, s, x, dout, batch, f,
                     self, lwords, dx, dout_stride);
}

// d == 32, f <= 128: which instantiation covers (f, self), or false
static bool launch_dot_bwd_h16(const float *x, const float *dout, int64_t batch, int f, int d, int self,
                               float *dx, hipStream_t s, int64_t dout_stride, bool probe_only = false) {
  if (d != 32 || f < 2 || f > 128 || batch < 512) return false;
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  if (out_dim < 4) return false;               // (a 16-byte load must fit inside a packed row)
  const int nstep = (f + 15) / 16, ney = ((out_dim + 3) / 4 + 255) / 256;
  if (ney > 8) return false;                   // f = 128 with self interaction: 8256 pairs
  if (probe_only) return true;
  const dim3 grid((unsigned)std::min<int64_t>(batch, 256));
  const int64_t st = dout_stride ? dout_stride : (int64_t)out_dim;
  if (nstep <= 4 && ney <= 3) launch_dot_bwd_h16_v<4, 3, 4>(x, dout, batch, f, self, grid, dx, s, st);
  else if (nstep <= 7 && ney <= 5) launch_dot_bwd_h16_v<7, 5, 4>(x, dout, batch, f, self, grid, dx, s, st);
  else if (nstep <= 7 && ney <= 6) launch_dot_bwd_h16_v<7, 6, 4>(x, dout, batch, f, self, grid, dx, s, st);
  else launch_dot_bwd_h16_v<8, 8, 3>(x, dout, batch, f, self, grid, dx, s, st);
  return true;
}

template <int MAXE>
static void launch_dot_bwd_pc_v(const float *x, const float *dout, int64_t batch, int f, int d, int self,
                                int kh, size_t lds, dim3 grid, float *dx, hipStream_t s, int64_t dout_stride) {
  (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_bwd_pc_kernel<MAXE>), 160 * 1024);
  hipLaunchKernelGGL((dot_interaction_bwd_pc_kernel<MAXE>), grid, dim3(512), lds, s, x, dout, batch, f, d,
                     self, kh, dx, dout_stride);
}

static void launch_dot_bwd_pc_e(int maxe, const float *x, const float *dout, int64_t batch, int f, int d,
                                int self, int kh, size_t lds, dim3 grid, float *dx, hipStream_t s,
                                int64_t dout_stride) {
  if (maxe <= 12) launch_dot_bwd_pc_v<12>(x, dout, batch, f, d, self, kh, lds, grid, dx, s, dout_stride);
  else if (maxe <= 20) launch_dot_bwd_pc_v<20>(x, dout, batch, f, d, self, kh, lds, grid, dx, s, dout_stride);
  else launch_dot_bwd_pc_v<33>(x, dout, batch, f, d, self, kh, lds, grid, dx, s, dout_stride);
}

template <int NFB, int MAXE>
static void launch_dot_bwd_dense_v(const float *x, const float *dout, int64_t batch, int f, int d,
                                   int self, int kh, size_t lds, dim3 grid, float *dx, hipStream_t s) {
  (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_bwd_dense_kernel<NFB, MAXE>), 64 * 1024);
  hipLaunchKernelGGL((dot_interaction_bwd_dense_kernel<NFB, MAXE>), grid, dim3(256), lds, s, x, dout, batch,
                     f, d, self, kh, dx);
}

template <int NFB>
static void launch_dot_bwd_dense_e(int maxe, const float *x, const float *dout, int64_t batch, int f,
                                   int d, int self, int kh, size_t lds, dim3 grid, float *dx,
                                   hipStream_t s) {
  if (maxe <= 12) launch_dot_bwd_dense_v<NFB, 12>(x, dout, batch, f, d, self, kh, lds, grid, dx, s);
  else if (maxe <= 20) launch_dot_bwd_dense_v<NFB, 20>(x, dout, batch, f, d, self, kh, lds, grid, dx, s);
  else launch_dot_bwd_dense_v<NFB, 33>(x, dout, batch, f, d, self, kh, lds, grid, dx, s);
}

static bool launch_dot_bwd_dense(const float *x, const float *dout, int64_t batch, int f, int d,
                                 int self, float *dx, hipStream_t s, int64_t dout_stride = 0) {
  if (f > 128 || d > 128) return false;
  int kh = (f + 7) / 8 * 4;                  // multiple of 4, 2 * kh >= f
  if (((2 * kh + 4) / 4) % 2 == 0) kh += 4;  // odd number of 16-byte slots per row
  const size_t lds = ((size_t)f * (2 * kh + 4) + 256) * sizeof(float);   // tile + 256 dummy slots
  if (lds > 64 * 1024 || (size_t)f * (2 * kh + 4) + 256 > 0xFFFFu) return false;
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const int maxe = (out_dim + 255) / 256;
  const char *dv = option("TFRS_DOT_BWD");
  // default: split-fp16 kernel on the packed gradient (d == 32); TFRS_DOT_BWD=pc / dense / gather select
  // the earlier generations (measurement switches)
  if (!(dv && (dv[0] == 'd' || dv[0] == 'p')) &&
      launch_dot_bwd_h16(x, dout, batch, f, d, self, dx, s, dout_stride))
    return true;
  const size_t lds_pc = 2 * lds + (size_t)2 * (2 * kh * 32) * sizeof(float);
  if (!(dv && dv[0] == 'd') && d <= 32 && lds_pc <= 160 * 1024 && batch >= 512) {
    // producer / consumer kernel: one 8-wave workgroup per CU, double-buffered S and X tiles
    const dim3 grid_pc((unsigned)std::min<int64_t>(batch, 256));
    launch_dot_bwd_pc_e(maxe, x, dout, batch, f, d, self, kh, lds_pc, grid_pc, dx, s,
                        dout_stride ? dout_stride : (int64_t)out_dim);
    return true;
  }
  if (dout_stride != 0 && dout_stride != out_dim) return false;   // only the producer / consumer kernel takes a row stride
  const int64_t per_cu = std::min<int64_t>(2, std::max<int64_t>(1, (int64_t)(160 * 1024) / (int64_t)lds));
  const dim3 grid((unsigned)std::min<int64_t>(batch, 256 * per_cu));
  const int nfb = (d + 31) / 32;
  if (nfb == 1) launch_dot_bwd_dense_e<1>(maxe, x, dout, batch, f, d, self, kh, lds, grid, dx, s);
  else if (nfb == 2) launch_dot_bwd_dense_e<2>(maxe, x, dout, batch, f, d, self, kh, lds, grid, dx, s);
  else launch_dot_bwd_dense_e<4>(maxe, x, dout, batch, f, d, self, kh, lds, grid, dx, s);
  return true;
}

template <int DP, int NB>
static bool launch_dot_bwd_mfma_nb(const float *x, const float *dout, int64_t batch, int f, int d,
                                   int self, float *dx, hipStream_t s) {
  const int out_dim = self ? f * (f + 1) / 2 : f * (f - 1) / 2;
  const size_t lds = (size_t)((out_dim + 3) & ~3) * sizeof(float);
  if (lds > 64 * 1024) return false;
  (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&dot_interaction_bwd_mfma_kernel<DP, NB>), 64 * 1024);
  const int64_t per_cu = std::min<int64_t>(8, std::max<int64_t>(1, (int64_t)(160 * 1024) / (int64_t)lds));
  const dim3 grid((unsigned)std::min<int64_t>(batch, 256 * per_cu * 2));
  hipLaunchKernelGGL((dot_interaction_bwd_mfma_kernel<DP, NB>), grid, dim3(64), lds, s, x, dout, batch, f,
                     d, self, dx);
  return true;
}

template <int DP>
static bool launch_dot_bwd_mfma_dp(const float *x, const float *dout, int64_t batch, int f, int d,
                                   int self, float *dx, hipStream_t s) {
  const int nb = (f + 31) / 32;
  if (nb * 16 * ((DP + 31) / 32) > 128) return false;  // register budget of the resident X operand
  switch (nb) {
    case 1: return launch_dot_bwd_mfma_nb<DP, 1>(x, dout, batch, f, d, self, dx, s);
    case 2: return launch_dot_bwd_mfma_nb<DP, 2>(x, dout, batch, f, d, self, dx, s);
    case 3: return launch_dot_bwd_mfma_nb<DP, 3>(x, dout, batch, f, d, self, dx, s);
    case 4: return launch_dot_bwd_mfma_nb<DP, 4>(x, dout, batch, f, d, self, dx, s);
    default: return false;
  }
}

static bool launch_dot_bwd_mfma(const float *x, const float *dout, int64_t batch, int f, int d,
                                int self, float *dx, hipStream_t s) {
  if (d > 128 || f > 128) return false;
  switch (softmax_padded_dim(d)) {
    case 8: return launch_dot_bwd_mfma_dp<8>(x, dout, batch, f, d, self, dx, s);
    case 16: return launch_dot_bwd_mfma_dp<16>(x, dout, batch, f, d, self, dx, s);
    case 32: return launch_dot_bwd_mfma_dp<32>(x, dout, batch, f, d, self, dx, s);
    case 64: return launch_dot_bwd_mfma_dp<64>(x, dout, batch, f, d, self, dx, s);
    default: return launch_dot_bwd_mfma_dp<128>(x, dout, batch, f, d, self, dx, s);
  }
}

static bool launch_dot_mfma(const float *x, int64_t batch, int f, int d, int self, int skip,
                            float *out, hipStream_t s, int64_t out_stride = 0) {
  if (d > 128 || f > 128) return false;
  switch (softmax_padded_dim(d)) {
    case 8: return launch_dot_mfma_dp<8>(x, batch, f, d, self, skip, out, s, out_stride);
    case 16: return launch_dot_mfma_dp<16>(x, batch, f, d, self, skip, out, s, out_stride);
    case 32: return launch_dot_mfma_dp<32>(x, batch, f, d, self, skip, out, s, out_stride);
    case 64: return launch_dot_mfma_dp<64>(x, batch, f, d, self, skip, out, s, out_stride);
    default: return launch_dot_mfma_dp<128>(x, batch, f, d, self, skip, out, s, out_stride);
  }
}

}  // namespace tfrs

using namespace tfrs;

// ---- split-K for the weight-gradient products (A^T B, K = batch) ---------------------------
// dW = x^T dy of a layer with a small input or output width is a handful of 128 x 128 output tiles
// with K = batch: 4 workgroups for a 512 -> 1 layer, 11.5 ms at batch 131072 on a 256-CU chip.  The
// K range is therefore cut into slices (grid.y) until tiles x slices covers the chip about four
// times; slices write partial products, splitk_sum_kernel adds them in a fixed order.
constexpr int kSplitMinK = 1024;          // rows of K per slice at least (whole kBK steps; round 6: 4096 left the skinny weight gradients of a
                                          // batch-131072 step -- 13 x 512, 256 x 32, 512 x 1 -- on 128 workgroups, 0.5-0.6 ms each)
static int splitk_slices(int64_t m, int n, int64_t k) {
  const int64_t tiles = ((m + kBM - 1) / kBM) * ((n + kBN - 1) / kBN);
  if (tiles >= 256 || k < 2 * kSplitMinK) return 1;
  int64_t want = (1024 + tiles - 1) / tiles;
  want = std::min<int64_t>(want, k / kSplitMinK);
  // partial products are m * n floats each: keep them under 64 MB
  want = std::min<int64_t>(want, std::max<int64_t>(1, (int64_t)(16 << 20) / std::max<int64_t>(1, m * n)));
  return (int)std::max<int64_t>(1, want);
}
static size_t splitk_bytes(int64_t m, int n, int64_t k) {
  const int sl = splitk_slices(m, n, k);
  return sl > 1 ? (size_t)sl * (size_t)m * (size_t)n * sizeof(float) : 0;
}

__global__ void __launch_bounds__(256) splitk_sum_kernel(const float *__restrict__ part, int slices,
                                                         int64_t count, const float *__restrict__ bias,
                                                         int n, float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = bias ? bias[i % n] : 0.0f;
#pragma unroll 8
    for (int z = 0; z < slices; ++z) v += part[(int64_t)z * count + i];   // (eight loads in flight; fixed order)
    out[i] = v;
  }
}

// `splitk_ws` (optional): scratch of splitk_bytes(m, n, k) for the at && !bt product.
static int launch_gemm(const GemmArgs &g_in, int epi, hipStream_t s, bool at = false, bool bt = false,
                       float *splitk_ws = nullptr) {
  GemmArgs g = g_in;
  g.kper = 0;
  const int64_t nbm = (g.m + kBM - 1) / kBM;
  const int nbn = (g.n + kBN - 1) / kBN;
  const dim3 grid((unsigned)(nbm * nbn));
  if (at) {          // dW = x^T dz
    const int slices = splitk_ws ? splitk_slices(g.m, g.n, g.k) : 1;
    if (slices > 1) {
      g.kper = (int)(((int64_t)g.k + slices - 1) / slices + kBK - 1) / kBK * kBK;
      const int used = (int)(((int64_t)g.k + g.kper - 1) / g.kper);
      float *const final_out = g.out;
      g.out = splitk_ws;
      hipLaunchKernelGGL((gemm_kernel<kEpiBias, true, false>), dim3(grid.x, (unsigned)used), dim3(256), 0, s, g);
      const int64_t count = g.m * (int64_t)g.n;
      hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256),
                         0, s, splitk_ws, used, count, g.bias, g.n, final_out);
    } else {
      hipLaunchKernelGGL((gemm_kernel<kEpiBias, true, false>), grid, dim3(256), 0, s, g);
    }
  } else if (bt)     // dx = dz W^T + dy + diag dz
    hipLaunchKernelGGL((gemm_kernel<kEpiCrossDx, false, true>), grid, dim3(256), 0, s, g);
  else if (epi == kEpiCross)
    hipLaunchKernelGGL((gemm_kernel<kEpiCross, false, false>), grid, dim3(256), 0, s, g);
  else if (epi == kEpiCrossDx0)
    hipLaunchKernelGGL((gemm_kernel<kEpiCrossDx0, false, false>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<kEpiBias, false, false>), grid, dim3(256), 0, s, g);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_cross_fwd_ex(const float *x0, const float *x, const float *a, int ka,
                                 const float *kernel, const float *bias, float diag_scale,
                                 int64_t batch, int d, float *y, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && d >= 1 && ka >= 1, "cross_fwd: bad shape");
  TFRS_CHECK_ARG(diag_scale >= 0.0f, "`diag_scale` should be non-negative. Got `diag_scale` = %g",
                 (double)diag_scale);
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x0 && x && a && kernel && y, "cross_fwd: NULL pointer");
  GemmArgs g = {};
  g.a = a; g.b = kernel; g.m = batch; g.n = d; g.k = ka;
  g.bias = bias; g.x0 = x0; g.x = x; g.diag = diag_scale; g.out = y;
  return launch_gemm(g, kEpiCross, (hipStream_t)stream);
}

extern "C" int tfrs_cross_fwd(const float *x0, const float *x, const float *kernel,
                              const float *bias, float diag_scale, int64_t batch, int d,
                              float *y, void *stream) {
  return tfrs_cross_fwd_ex(x0, x, x, d, kernel, bias, diag_scale, batch, d, y, stream);
}

extern "C" int tfrs_dense_fwd(const float *x, const float *kernel, const float *bias,
                              int64_t batch, int din, int dout, float *out, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && din >= 1 && dout >= 1, "dense_fwd: bad shape");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && kernel && out, "dense_fwd: NULL pointer");
  GemmArgs g = {};
  g.a = x; g.b = kernel; g.m = batch; g.n = dout; g.k = din;
  g.bias = bias; g.out = out;
  return launch_gemm(g, kEpiBias, (hipStream_t)stream);
}

// ---- activations fused into the product's epilogue (round 5; SURVEY 8(b): tfrs_cross_fwd(..., act, ...)) -------
namespace tfrs {
size_t gemm16_workspace_bytes(int64_t m, int n, int k);
int gemm16_run_act(const float *a, const float *b, int64_t m, int n, int k, const float *bias, int act,
                   const float *x0, const float *x, float diag, float *out, float *pre, void *ws,
                   hipStream_t s);
// One pass over the element-wise part of a Cross layer's backward with a (pre)activation, or of a Dense layer's:
//   s   = act(pre)                  (act == 0: s = pre)
//   dx0 = dy * (s + diag * x)                                           [optional]
//   dp  = dy * x0 * act'(pre)       (x0 == NULL: dp = dy * act'(pre))   [optional]
//   dxd = dy * (1 + diag * x0)      the direct terms of dx              [optional]
// ref_is_output: `pre` holds y = act(p) instead (relu / sigmoid / tanh: the derivative follows from y).
__global__ void __launch_bounds__(256) act_pointwise_bwd_kernel(int act, int ref_is_output,
                                                                const float *__restrict__ pre,
                                                                const float *__restrict__ dy,
                                                                const float *__restrict__ x0,
                                                                const float *__restrict__ x, float diag,
                                                                int64_t count, float *__restrict__ dp,
                                                                float *__restrict__ dx0,
                                                                float *__restrict__ dxd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const float p = pre[i], g = dy[i];
    const float x0v = x0 ? x0[i] : 1.0f;
    if (dp) dp[i] = g * x0v * (ref_is_output ? act_grad_from_output(act, p) : act_grad(act, p));
    if (dx0) dx0[i] = g * ((ref_is_output ? p : act_apply(act, p)) + (diag != 0.0f ? diag * x[i] : 0.0f));
    if (dxd) dxd[i] = g * (1.0f + diag * x0v);
  }
}
}  // namespace tfrs

extern "C" int tfrs_act_pointwise_bwd(int act, int ref_is_output, const float *pre, const float *dy,
                                      const float *x0, const float *x, float diag_scale, int64_t count,
                                      float *dp, float *dx0, float *dxd, void *stream) {
  TFRS_CHECK_ARG(act >= kActNone && act <= kActGelu, "act_pointwise_bwd: unknown activation %d", act);
  TFRS_CHECK_ARG(count >= 0, "act_pointwise_bwd: bad count");
  TFRS_CHECK_ARG(!ref_is_output || act == kActNone || act == kActRelu || act == kActSigmoid || act == kActTanh,
                 "act_pointwise_bwd: the derivative of activation %d needs the pre-activation", act);
  if (count == 0) return TFRS_OK;
  TFRS_CHECK_ARG(pre && dy && (dp || dx0 || dxd), "act_pointwise_bwd: NULL pointer");
  TFRS_CHECK_ARG(!(dx0 && diag_scale != 0.0f) || x, "act_pointwise_bwd: diag_scale needs x");
  TFRS_CHECK_ARG(!dxd || x0, "act_pointwise_bwd: dxd needs x0");
  const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(act_pointwise_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, act,
                     ref_is_output, pre, dy, x0, x, diag_scale, count, dp, dx0, dxd);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// out = act(x @ kernel + bias); pre_out (optional) receives x @ kernel + bias.  f16 != 0: split-fp16 MFMA
// (workspace from tfrs_gemm_f16_workspace_bytes(batch, dout, din)).
extern "C" int tfrs_dense_fwd_act(const float *x, const float *kernel, const float *bias, int64_t batch,
                                  int din, int dout, int act, float *out, float *pre_out, int f16,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && din >= 1 && dout >= 1, "dense_fwd_act: bad shape");
  TFRS_CHECK_ARG(act >= kActNone && act <= kActGelu, "dense_fwd_act: unknown activation %d", act);
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && kernel && out, "dense_fwd_act: NULL pointer");
  if (f16) {
    TFRS_CHECK_ARG(workspace != nullptr, "dense_fwd_act: NULL workspace");
    if (workspace_bytes < gemm16_workspace_bytes(batch, dout, din)) {
      set_error("dense_fwd_act: workspace too small");
      return TFRS_ENOMEM;
    }
    return gemm16_run_act(x, kernel, batch, dout, din, bias, act, nullptr, nullptr, 0.0f, out, pre_out,
                          workspace, (hipStream_t)stream);
  }
  GemmArgs g = {};
  g.a = x; g.b = kernel; g.m = batch; g.n = dout; g.k = din;
  g.bias = bias; g.out = out; g.act = act; g.pre = pre_out;
  return launch_gemm(g, kEpiBias, (hipStream_t)stream);
}

// Cross.call with a preactivation and / or a low-rank input (dcn.py:173-186):
//   y = x0 * (act(a @ kernel + bias) + diag_scale * x) + x,   a[batch, ka] = x (full rank, ka == d) or x @ U
// pre_out (optional) receives p = a @ kernel + bias for the backward (tfrs_act_pointwise_bwd + tfrs_dense_bwd[_add]).
extern "C" int tfrs_cross_fwd_act(const float *x0, const float *x, const float *a, int ka,
                                  const float *kernel, const float *bias, float diag_scale, int act,
                                  int64_t batch, int d, float *y, float *pre_out, int f16,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && d >= 1 && ka >= 1, "cross_fwd_act: bad shape");
  TFRS_CHECK_ARG(diag_scale >= 0.0f, "`diag_scale` should be non-negative. Got `diag_scale` = %g",
                 (double)diag_scale);
  TFRS_CHECK_ARG(act >= kActNone && act <= kActGelu, "cross_fwd_act: unknown activation %d", act);
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x0 && x && a && kernel && y, "cross_fwd_act: NULL pointer");
  if (f16) {
    TFRS_CHECK_ARG(workspace != nullptr, "cross_fwd_act: NULL workspace");
    if (workspace_bytes < gemm16_workspace_bytes(batch, d, ka)) {
      set_error("cross_fwd_act: workspace too small");
      return TFRS_ENOMEM;
    }
    return gemm16_run_act(a, kernel, batch, d, ka, bias, act, x0, x, diag_scale, y, pre_out, workspace,
                          (hipStream_t)stream);
  }
  GemmArgs g = {};
  g.a = a; g.b = kernel; g.m = batch; g.n = d; g.k = ka;
  g.bias = bias; g.x0 = x0; g.x = x; g.diag = diag_scale; g.out = y; g.act = act; g.pre = pre_out;
  return launch_gemm(g, kEpiCross, (hipStream_t)stream);
}

// ---- split-fp16 variants (gemm16.hip): same results to f32 accuracy, ~3x the rate on large
// products; the caller provides the workspace for the operand images --------------------------
namespace tfrs {
size_t gemm16_workspace_bytes(int64_t m, int n, int k);
int gemm16_run(const float *a, const float *b, int64_t m, int n, int k, const float *bias,
               const float *x0, const float *x, float diag, float *out, void *ws, hipStream_t s,
               float *aux = nullptr);
size_t gemm16_cross_bwd_workspace_bytes(int64_t batch, int d);
size_t gemm16_dense_bwd_workspace_bytes(int64_t batch, int din, int dout);
int gemm16_scores(const float *q, const float *c, int64_t nq, int nc, int d, float *out, void *ws,
                  hipStream_t s);
int gemm16_dense_bwd(const float *x, const float *kernel, const float *dy, int64_t batch, int din,
                     int dout, float *dx, float *dkernel, float *dbias, void *ws, hipStream_t s,
                     const float *addend = nullptr);
int gemm16_run_act(const float *a, const float *b, int64_t m, int n, int k, const float *bias, int act,
                   const float *x0, const float *x, float diag, float *out, float *pre, void *ws,
                   hipStream_t s);
int gemm16_cross_bwd(const float *x0, const float *x, const float *kernel, const float *bias,
                     float diag, const float *dy, int64_t batch, int d, float *dx0, float *dx,
                     float *dkernel, float *dbias, void *ws, hipStream_t s, const float *u = nullptr,
                     const float *dx0_add = nullptr, int add_dx = 0);

// db[j] = sum_b dy[b, j] * x0[b, j]: partial sums over slabs of 256 rows (one workgroup per
// 64 columns x slab, coalesced rows), then a fixed-order reduction: deterministic, no atomics.
constexpr int kDbSlab = 256;
__global__ void __launch_bounds__(256) colsum_mul_partial_kernel(const float *__restrict__ a,
                                                                 const float *__restrict__ b,
                                                                 int64_t rows, int n,
                                                                 float *__restrict__ part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * kDbSlab;
  const int64_t r1 = r0 + kDbSlab < rows ? r0 + kDbSlab : rows;
  float s = 0.0f;
  if (col < n)
    for (int64_t r = r0 + rg; r < r1; r += 4) s += b ? a[r * n + col] * b[r * n + col] : a[r * n + col];
  red[rg][c] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < n)
    part[(int64_t)blockIdx.y * n + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ void __launch_bounds__(256) colsum_reduce_kernel(const float *__restrict__ part, int nslab,
                                                            int n, float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  float s = 0.0f;
#pragma unroll 8
  for (int i = 0; i < nslab; ++i) s += part[(int64_t)i * n + c];
  out[c] = s;
}
}  // namespace tfrs

// ---- Cross backward (layers/feature_interaction/dcn.py:151-186 under models/base.py:77) ------
// full rank, linear preactivation; z = x W + b + diag x, dz = dy * x0:
//   dx0 = dy * z ; dx = dz W^T + dy + diag dz ; dW = x^T dz ; db = column sums of dz
// Three GEMM launches (z lives only in the first one's epilogue, dz only in operand loads) plus
// the db reduction; nothing is transposed or staged in HBM.
extern "C" size_t tfrs_cross_bwd_workspace_bytes(int64_t batch, int d, int f16) {
  if (batch < 1 || d < 1) return 256;
  if (f16) return gemm16_cross_bwd_workspace_bytes(batch, d);
  // dbias partial sums, then the split-K partial products of dW
  return ((size_t)((batch + kDbSlab - 1) / kDbSlab) * d * 4 + 255) / 256 * 256 + splitk_bytes(d, d, batch) + 256;
}

static int cross_bwd_check(const char *who, const float *x0, const float *x, const float *kernel,
                           float diag, const float *dy, int64_t batch, int d, float *dx0, float *dx,
                           float *dkernel, void *ws) {
  TFRS_CHECK_ARG(batch >= 0 && d >= 1, "%s: bad shape", who);
  TFRS_CHECK_ARG(diag >= 0.0f, "`diag_scale` should be non-negative. Got `diag_scale` = %g", (double)diag);
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x0 && x && kernel && dy && dx0 && dx && dkernel && ws, "%s: NULL pointer", who);
  return TFRS_OK;
}

extern "C" int tfrs_cross_bwd(const float *x0, const float *x, const float *kernel,
                              const float *bias, float diag_scale, const float *dy, int64_t batch,
                              int d, float *dx0, float *dx, float *dkernel, float *dbias,
                              void *workspace, size_t workspace_bytes, void *stream) {
  int rc = cross_bwd_check("cross_bwd", x0, x, kernel, diag_scale, dy, batch, d, dx0, dx, dkernel, workspace);
  if (rc != TFRS_OK || batch == 0) return rc;
  if (workspace_bytes < tfrs_cross_bwd_workspace_bytes(batch, d, 0)) {
    set_error("cross_bwd: workspace too small");
    return TFRS_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g = {};
  // dx0 = dy * (x W + b + diag x)
  g.a = x; g.b = kernel; g.m = batch; g.n = d; g.k = d; g.bias = bias;
  g.x0 = dy; g.x = x; g.diag = diag_scale; g.out = dx0;
  if ((rc = launch_gemm(g, kEpiCrossDx0, s)) != TFRS_OK) return rc;
  // dx = (dy * x0) W^T + dy + diag dy x0
  g = {};
  g.a = dy; g.amul = x0; g.b = kernel; g.m = batch; g.n = d; g.k = d;
  g.x0 = dy; g.x = x0; g.diag = diag_scale; g.out = dx;
  if ((rc = launch_gemm(g, kEpiCrossDx, s, false, true)) != TFRS_OK) return rc;
  // dW = x^T (dy * x0)
  g = {};
  g.a = x; g.b = dy; g.bmul = x0; g.m = d; g.n = d; g.k = (int)batch; g.out = dkernel;
  float *const splitk_ws = reinterpret_cast<float *>(
      static_cast<char *>(workspace) + ((size_t)((batch + kDbSlab - 1) / kDbSlab) * d * 4 + 255) / 256 * 256);
  if ((rc = launch_gemm(g, kEpiBias, s, true, false, splitk_ws)) != TFRS_OK) return rc;
  if (dbias) {
    const int nslab = (int)((batch + kDbSlab - 1) / kDbSlab);
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(colsum_mul_partial_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)nslab),
                       dim3(256), 0, s, dy, x0, batch, d, part);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, s, part,
                       nslab, d, dbias);
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}

// ---- scores = q @ c^T (TopK._compute_score, layers/factorized_top_k.py:320-333: tf.matmul(q, c,
// transpose_b=True); Retrieval.call scores tasks/retrieval.py:172-180) with the candidate matrix read
// in place as the transposed operand.  f16 != 0: split-fp16 MFMA (workspace from
// tfrs_gemm_f16_workspace_bytes(nq, nc, d)); else f32 MFMA, workspace unused. -----------------
extern "C" int tfrs_compute_scores(const float *q, const float *c, int64_t nq, int nc, int d,
                                   float *out, int f16, void *workspace, size_t workspace_bytes,
                                   void *stream) {
  TFRS_CHECK_ARG(nq >= 0 && nc >= 1 && d >= 1, "compute_scores: bad shape");
  if (nq == 0) return TFRS_OK;
  TFRS_CHECK_ARG(q && c && out, "compute_scores: NULL pointer");
  if (f16) {
    TFRS_CHECK_ARG(workspace != nullptr, "compute_scores: NULL workspace");
    if (workspace_bytes < gemm16_workspace_bytes(nq, nc, d)) {
      set_error("compute_scores: workspace too small");
      return TFRS_ENOMEM;
    }
    return gemm16_scores(q, c, nq, nc, d, out, workspace, (hipStream_t)stream);
  }
  GemmArgs g = {};
  g.a = q; g.b = c; g.m = nq; g.n = nc; g.k = d; g.out = out;
  hipLaunchKernelGGL((gemm_kernel<kEpiBias, false, true>),
                     dim3((unsigned)(((nq + kBM - 1) / kBM) * ((nc + kBN - 1) / kBN))), dim3(256), 0,
                     (hipStream_t)stream, g);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// ---- Dense backward (Keras Dense [in, out] kernels of layers/blocks.py:46-61 and the low-rank
// Cross projections dcn.py:176-180): dx = dy W^T, dW = x^T dy, db = column sums of dy, with the
// transposed operands read in place (NULL outputs are skipped) ---------------------------------
extern "C" size_t tfrs_dense_bwd_workspace_bytes(int64_t batch, int din, int dout, int f16) {
  if (batch < 1 || din < 1 || dout < 1) return 256;
  if (f16) return gemm16_dense_bwd_workspace_bytes(batch, din, dout);
  // dbias partial sums, then the split-K partial products of dW
  return ((size_t)((batch + kDbSlab - 1) / kDbSlab) * dout * 4 + 255) / 256 * 256 + splitk_bytes(din, dout, batch) + 256;
}

extern "C" int tfrs_dense_bwd_add(const float *x, const float *kernel, const float *dy, const float *addend,
                                  int64_t batch, int din, int dout, float *dx, float *dkernel, float *dbias,
                                  int f16, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int tfrs_dense_bwd(const float *x, const float *kernel, const float *dy, int64_t batch,
                              int din, int dout, float *dx, float *dkernel, float *dbias,
                              int f16, void *workspace, size_t workspace_bytes, void *stream) {
  return tfrs_dense_bwd_add(x, kernel, dy, nullptr, batch, din, dout, dx, dkernel, dbias, f16, workspace,
                            workspace_bytes, stream);
}

// the same with dx = dy @ kernel^T + addend[batch, din] (the direct terms of a Cross layer's input gradient)
extern "C" int tfrs_dense_bwd_add(const float *x, const float *kernel, const float *dy, const float *addend,
                                  int64_t batch, int din, int dout, float *dx, float *dkernel, float *dbias,
                                  int f16, void *workspace, size_t workspace_bytes, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && din >= 1 && dout >= 1, "dense_bwd: bad shape");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && kernel && dy && workspace, "dense_bwd: NULL pointer");
  TFRS_CHECK_ARG(batch <= 0x7FFFFFFFll, "dense_bwd: batch too large");
  if (workspace_bytes < tfrs_dense_bwd_workspace_bytes(batch, din, dout, f16)) {
    set_error("dense_bwd: workspace too small");
    return TFRS_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  if (f16) return gemm16_dense_bwd(x, kernel, dy, batch, din, dout, dx, dkernel, dbias, workspace, s, addend);
  int rc;
  GemmArgs g = {};
  if (dx) {        // dx[b, i] = sum_j dy[b, j] W[i, j] (+ addend[b, i]: CrossDx epilogue with diag = 0)
    g.a = dy; g.b = kernel; g.m = batch; g.n = din; g.k = dout; g.out = dx;
    const dim3 grid((unsigned)(((batch + kBM - 1) / kBM) * ((din + kBN - 1) / kBN)));
    if (addend) {
      g.x0 = addend; g.x = addend; g.diag = 0.0f;
      hipLaunchKernelGGL((gemm_kernel<kEpiCrossDx, false, true>), grid, dim3(256), 0, s, g);
    } else {
      hipLaunchKernelGGL((gemm_kernel<kEpiBias, false, true>), grid, dim3(256), 0, s, g);
    }
    TFRS_LAUNCH_CHECK();
  }
  if (dkernel) {   // dW[i, j] = sum_b x[b, i] dy[b, j]
    g = {};
    g.a = x; g.b = dy; g.m = din; g.n = dout; g.k = (int)batch; g.out = dkernel;
    float *const splitk_ws = reinterpret_cast<float *>(
        static_cast<char *>(workspace) + ((size_t)((batch + kDbSlab - 1) / kDbSlab) * dout * 4 + 255) / 256 * 256);
    if ((rc = launch_gemm(g, kEpiBias, s, true, false, splitk_ws)) != TFRS_OK) return rc;
  }
  if (dbias) {
    const int nslab = (int)((batch + kDbSlab - 1) / kDbSlab);
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(colsum_mul_partial_kernel, dim3((unsigned)((dout + 63) / 64), (unsigned)nslab),
                       dim3(256), 0, s, dy, (const float *)nullptr, batch, dout, part);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((dout + 255) / 256)), dim3(256), 0, s, part,
                       nslab, dout, dbias);
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}

extern "C" int tfrs_cross_bwd_f16(const float *x0, const float *x, const float *kernel,
                                  const float *bias, float diag_scale, const float *dy,
                                  int64_t batch, int d, float *dx0, float *dx, float *dkernel,
                                  float *dbias, void *workspace, size_t workspace_bytes,
                                  void *stream) {
  int rc = cross_bwd_check("cross_bwd_f16", x0, x, kernel, diag_scale, dy, batch, d, dx0, dx, dkernel,
                           workspace);
  if (rc != TFRS_OK || batch == 0) return rc;
  TFRS_CHECK_ARG(batch <= 0x7FFFFFFFll, "cross_bwd_f16: batch too large");
  if (workspace_bytes < gemm16_cross_bwd_workspace_bytes(batch, d)) {
    set_error("cross_bwd_f16: workspace too small");
    return TFRS_ENOMEM;
  }
  return gemm16_cross_bwd(x0, x, kernel, bias, diag_scale, dy, batch, d, dx0, dx, dkernel, dbias,
                          workspace, (hipStream_t)stream);
}

extern "C" size_t tfrs_gemm_f16_workspace_bytes(int64_t m, int n, int k) {
  if (m < 1 || n < 1 || k < 1) return 256;
  return gemm16_workspace_bytes(m, n, k);
}

extern "C" int tfrs_dense_fwd_f16(const float *x, const float *kernel, const float *bias,
                                  int64_t batch, int din, int dout, float *out, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && din >= 1 && dout >= 1, "dense_fwd_f16: bad shape");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && kernel && out && workspace, "dense_fwd_f16: NULL pointer");
  if (workspace_bytes < gemm16_workspace_bytes(batch, dout, din)) {
    set_error("dense_fwd_f16: workspace too small");
    return TFRS_ENOMEM;
  }
  return gemm16_run(x, kernel, batch, dout, din, bias, nullptr, nullptr, 0.0f, out, workspace,
                    (hipStream_t)stream);
}

extern "C" int tfrs_cross_fwd_f16(const float *x0, const float *x, const float *kernel,
                                  const float *bias, float diag_scale, int64_t batch, int d,
                                  float *y, void *workspace, size_t workspace_bytes, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && d >= 1, "cross_fwd_f16: bad shape");
  TFRS_CHECK_ARG(diag_scale >= 0.0f, "`diag_scale` should be non-negative. Got `diag_scale` = %g",
                 (double)diag_scale);
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x0 && x && kernel && y && workspace, "cross_fwd_f16: NULL pointer");
  if (workspace_bytes < gemm16_workspace_bytes(batch, d, d)) {
    set_error("cross_fwd_f16: workspace too small");
    return TFRS_ENOMEM;
  }
  return gemm16_run(x, kernel, batch, d, d, bias, x0, x, diag_scale, y, workspace,
                    (hipStream_t)stream);
}

// Training forward: also stores u = x W + b + diag x ([batch, d]) for tfrs_cross_bwd_f16_saved.
extern "C" int tfrs_cross_fwd_f16_train(const float *x0, const float *x, const float *kernel,
                                        const float *bias, float diag_scale, int64_t batch, int d,
                                        float *y, float *u_out, void *workspace, size_t workspace_bytes,
                                        void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && d >= 1, "cross_fwd_f16_train: bad shape");
  TFRS_CHECK_ARG(diag_scale >= 0.0f, "`diag_scale` should be non-negative. Got `diag_scale` = %g",
                 (double)diag_scale);
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x0 && x && kernel && y && u_out && workspace, "cross_fwd_f16_train: NULL pointer");
  if (workspace_bytes < gemm16_workspace_bytes(batch, d, d)) {
    set_error("cross_fwd_f16_train: workspace too small");
    return TFRS_ENOMEM;
  }
  return gemm16_run(x, kernel, batch, d, d, bias, x0, x, diag_scale, y, workspace, (hipStream_t)stream, u_out);
}

extern "C" int tfrs_cross_bwd_f16_saved(const float *x0, const float *x, const float *u, const float *kernel,
                                        float diag_scale, const float *dy, int64_t batch, int d,
                                        float *dx0, float *dx, float *dkernel, float *dbias,
                                        void *workspace, size_t workspace_bytes, void *stream) {
  int rc = cross_bwd_check("cross_bwd_f16_saved", x0, x, kernel, diag_scale, dy, batch, d, dx0, dx, dkernel,
                           workspace);
  if (rc != TFRS_OK || batch == 0) return rc;
  TFRS_CHECK_ARG(u, "cross_bwd_f16_saved: NULL pointer");
  TFRS_CHECK_ARG(batch <= 0x7FFFFFFFll, "cross_bwd_f16_saved: batch too large");
  if (workspace_bytes < gemm16_cross_bwd_workspace_bytes(batch, d)) {
    set_error("cross_bwd_f16_saved: workspace too small");
    return TFRS_ENOMEM;
  }
  return gemm16_cross_bwd(x0, x, kernel, nullptr, diag_scale, dy, batch, d, dx0, dx, dkernel, dbias,
                          workspace, (hipStream_t)stream, u);
}

// The same inside a STACK of Cross layers that share x0 (dcn.py:47-56): dx0 = dy * u + dx0_add (dx0_add may be dx0
// itself: x0's gradient accumulates in place across the layers), and for the stack's first layer -- whose x IS x0 --
// add_dx != 0 also folds that layer's dx in: dx0 = dy * u + dx0_add + dx.  dx0_add may be NULL.
extern "C" int tfrs_cross_bwd_f16_saved_acc(const float *x0, const float *x, const float *u, const float *kernel,
                                            float diag_scale, const float *dy, int64_t batch, int d,
                                            const float *dx0_add, int add_dx, float *dx0, float *dx, float *dkernel,
                                            float *dbias, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = cross_bwd_check("cross_bwd_f16_saved_acc", x0, x, kernel, diag_scale, dy, batch, d, dx0, dx, dkernel,
                           workspace);
  if (rc != TFRS_OK || batch == 0) return rc;
  TFRS_CHECK_ARG(u, "cross_bwd_f16_saved_acc: NULL pointer");
  TFRS_CHECK_ARG(batch <= 0x7FFFFFFFll, "cross_bwd_f16_saved_acc: batch too large");
  TFRS_CHECK_ARG(dx != dx0 && dx0_add != dx, "cross_bwd_f16_saved_acc: dx must not alias dx0 / dx0_add");
  if (workspace_bytes < gemm16_cross_bwd_workspace_bytes(batch, d)) {
    set_error("cross_bwd_f16_saved_acc: workspace too small");
    return TFRS_ENOMEM;
  }
  return gemm16_cross_bwd(x0, x, kernel, nullptr, diag_scale, dy, batch, d, dx0, dx, dkernel, dbias,
                          workspace, (hipStream_t)stream, u, dx0_add, add_dx);
}

extern "C" int tfrs_dot_interaction_fwd(const float *x, int64_t batch, int f, int d,
                                        int self_interaction, int skip_gather, float *out,
                                        void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && f >= 1 && d >= 1, "dot_interaction_fwd: bad shape");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && out, "dot_interaction_fwd: NULL pointer");
  if (launch_dot_mfma(x, batch, f, d, self_interaction, skip_gather, out, (hipStream_t)stream)) {
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  const size_t lds = (size_t)4 * f * (d + 1) * sizeof(float);
  if (lds > 64 * 1024) {
    set_error("dot_interaction_fwd: %d features x %d dims do not fit the LDS staging", f, d);
    return TFRS_ENOTIMPL;
  }
  hipLaunchKernelGGL(dot_interaction_fwd_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), lds,
                     (hipStream_t)stream, x, batch, f, d, self_interaction, skip_gather, out);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

// Row-strided variants: the packed pairs are row b of a WIDER matrix (out + b * out_stride), so
// that the layer's output can be written next to the bottom-stack output it is concatenated with
// (experimental/models/ranking.py:225-232) and its gradient read from that matrix's gradient in
// place -- no torch.cat / slice copies of the [batch, F (F - 1) / 2] block.  Packed-triangle output
// only, on the default kernels; TFRS_ENOTIMPL for shapes those do not cover (the caller falls back
// to the contiguous calls).
extern "C" int tfrs_dot_interaction_fwd_strided(const float *x, int64_t batch, int f, int d,
                                                int self_interaction, float *out, int64_t out_stride,
                                                void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && f >= 1 && d >= 1, "dot_interaction_fwd_strided: bad shape");
  const int out_dim = self_interaction ? f * (f + 1) / 2 : f * (f - 1) / 2;
  TFRS_CHECK_ARG(out_stride >= out_dim, "dot_interaction_fwd_strided: out_stride < pairs per sample");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && out, "dot_interaction_fwd_strided: NULL pointer");
  if (launch_dot_mfma(x, batch, f, d, self_interaction, 0, out, (hipStream_t)stream, out_stride)) {
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  set_error("dot_interaction_fwd_strided: %d features x %d dims are not covered by the strided kernel", f, d);
  return TFRS_ENOTIMPL;
}

// 1 when BOTH strided entry points cover the shape (the host class asks before choosing the fused
// concat path: the forward must not succeed where the backward would refuse).  Mirrors the
// conditions of launch_dot_mfma_nb (direct-store kernel) and launch_dot_bwd_dense (producer /
// consumer kernel) including the measurement switches, which route to kernels without a row stride.
extern "C" int tfrs_dot_interaction_strided_supported(int64_t batch, int f, int d, int self_interaction) {
  if (batch < 512 || f < 1 || d < 1 || f > 128 || d > 32 || d % 16 != 0) return 0;
  const int out_dim = self_interaction ? f * (f + 1) / 2 : f * (f - 1) / 2;
  // forward: split-fp16 direct-store kernel
  const int dp = d <= 16 ? 16 : 32, nb = (f + 31) / 32;
  if (nb > 4 || nb * (dp / 2) > 96) return 0;
  if ((size_t)(((out_dim + 3) & ~3) + 64) * sizeof(float) > 64 * 1024) return 0;
  const char *sv = option("TFRS_DOT_STAGE"), *fv = option("TFRS_DOT_FWD"), *dv = option("TFRS_DOT_BWD");
  if ((sv && sv[0] == '0') || (fv && (fv[0] == 'f' || fv[0] == 's')) || (dv && (dv[0] == 'd' || dv[0] == 'g'))) return 0;
  // backward: producer / consumer kernel
  int kh = (f + 7) / 8 * 4;
  if (((2 * kh + 4) / 4) % 2 == 0) kh += 4;
  const size_t lds = ((size_t)f * (2 * kh + 4) + 256) * sizeof(float);
  if (lds > 64 * 1024 || (size_t)f * (2 * kh + 4) + 256 > 0xFFFFu) return 0;
  const size_t lds_pc = 2 * lds + (size_t)2 * (2 * kh * 32) * sizeof(float);
  return lds_pc <= 160 * 1024 ? 1 : 0;
}

extern "C" int tfrs_dot_interaction_bwd_strided(const float *x, const float *dout, int64_t dout_stride,
                                                int64_t batch, int f, int d, int self_interaction,
                                                float *dx, void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && f >= 1 && d >= 1, "dot_interaction_bwd_strided: bad shape");
  const int out_dim = self_interaction ? f * (f + 1) / 2 : f * (f - 1) / 2;
  TFRS_CHECK_ARG(dout_stride >= out_dim, "dot_interaction_bwd_strided: dout_stride < pairs per sample");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && dout && dx, "dot_interaction_bwd_strided: NULL pointer");
  if (launch_dot_bwd_dense(x, dout, batch, f, d, self_interaction, dx, (hipStream_t)stream, dout_stride)) {
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  set_error("dot_interaction_bwd_strided: %d features x %d dims are not covered by the strided kernel", f, d);
  return TFRS_ENOTIMPL;
}

extern "C" int tfrs_dot_interaction_bwd(const float *x, const float *dout, int64_t batch, int f,
                                        int d, int self_interaction, int skip_gather, float *dx,
                                        void *stream) {
  TFRS_CHECK_ARG(batch >= 0 && f >= 1 && d >= 1, "dot_interaction_bwd: bad shape");
  if (batch == 0) return TFRS_OK;
  TFRS_CHECK_ARG(x && dout && dx, "dot_interaction_bwd: NULL pointer");
  // TFRS_DOT_BWD=gather selects the second-generation kernel (A operand gathered per element)
  const char *dv = option("TFRS_DOT_BWD");
  const bool dense_ok = !(dv && dv[0] == 'g');
  if (!skip_gather && dense_ok &&
      launch_dot_bwd_dense(x, dout, batch, f, d, self_interaction, dx, (hipStream_t)stream)) {
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  if (!skip_gather &&
      launch_dot_bwd_mfma(x, dout, batch, f, d, self_interaction, dx, (hipStream_t)stream)) {
    TFRS_LAUNCH_CHECK();
    return TFRS_OK;
  }
  const size_t lds = (size_t)4 * f * d * sizeof(float);
  if (lds > 64 * 1024) {
    set_error("dot_interaction_bwd: %d features x %d dims do not fit the LDS staging", f, d);
    return TFRS_ENOTIMPL;
  }
  hipLaunchKernelGGL(dot_interaction_bwd_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), lds,
                     (hipStream_t)stream, x, dout, batch, f, d, self_interaction, skip_gather, dx);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
