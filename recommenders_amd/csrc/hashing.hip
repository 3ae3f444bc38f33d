// hashing.hip -- salted SipHash-2-4 bucketing of feature ids / strings.
//
// Replaces the tf.keras.layers.Hashing(num_bins, salt=[feature_no, chunk_id]) layers that
// UnifiedEmbedding applies to every feature once per chunk
// (layers/feature_multiplexing/unified_embedding.py:116-119,155-159,198-205).  With a salt,
// Keras hashes tf.as_string(x) (integers) or the string bytes with
// tf.strings.to_hash_bucket_strong = SipHash-2-4 under the key (salt[0], salt[1]), then takes
// the unsigned remainder by num_bins.
//
// Integer work, HBM-bound: 8 B (or 4 B) read + 8 B written per id; one lane per value, the
// decimal digits are produced in registers and fed to the SipHash rounds as 64-bit words, so
// nothing but the id and the bucket ever touches memory.  Strings arrive as one packed byte
// buffer + offsets[n + 1]; a lane walks its own string 8 bytes at a time.
#include "common.h"

namespace tfrs {

struct Sip {
  uint64_t v0, v1, v2, v3;
  __device__ __forceinline__ Sip(uint64_t k0, uint64_t k1)
      : v0(k0 ^ 0x736f6d6570736575ull), v1(k1 ^ 0x646f72616e646f6dull),
        v2(k0 ^ 0x6c7967656e657261ull), v3(k1 ^ 0x7465646279746573ull) {}
  static __device__ __forceinline__ uint64_t rotl(uint64_t x, int b) {
    return (x << b) | (x >> (64 - b));
  }
  __device__ __forceinline__ void round() {
    v0 += v1; v1 = rotl(v1, 13) ^ v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16) ^ v2;
    v0 += v3; v3 = rotl(v3, 21) ^ v0;
    v2 += v1; v1 = rotl(v1, 17) ^ v2; v2 = rotl(v2, 32);
  }
  __device__ __forceinline__ void absorb(uint64_t m) {
    v3 ^= m;
    round();
    round();
    v0 ^= m;
  }
  __device__ __forceinline__ uint64_t finish() {
    v2 ^= 0xff;
    round();
    round();
    round();
    round();
    return v0 ^ v1 ^ v2 ^ v3;
  }
};

// h mod m by Barrett reduction with the host-computed recip = floor((2^64 - 1) / m): the
// estimate q = mulhi(h, recip) is never above and at most 2 below floor(h / m), so two
// conditional subtractions finish it (a runtime 64-bit `%` costs ~10x more instructions).
__device__ __forceinline__ uint64_t mod_barrett(uint64_t h, uint64_t m, uint64_t recip) {
  uint64_t r = h - __umul64hi(h, recip) * m;
  if (r >= m) r -= m;
  if (r >= m) r -= m;
  return r;
}

// Decimal string of a signed 64-bit value (at most 20 characters: '-' + 19 digits) packed
// little-endian into three 64-bit words, exactly the byte stream tf.as_string produces.
__device__ __forceinline__ uint64_t sip_of_int(int64_t x, uint64_t k0, uint64_t k1) {
  const bool neg = x < 0;
  uint64_t mag = neg ? (0ull - (uint64_t)x) : (uint64_t)x;
  // The string's byte 0 is the most significant digit, which the division chain yields
  // last: keep the bytes in a 3-word shift register and prepend each new character.
  uint64_t w0 = 0, w1 = 0, w2 = 0;
  int len = 0;
  do {
    const uint64_t q = mag / 10;
    const uint64_t ch = (uint64_t)('0' + (mag - q * 10));
    w2 = (w2 << 8) | (w1 >> 56);
    w1 = (w1 << 8) | (w0 >> 56);
    w0 = (w0 << 8) | ch;
    mag = q;
    ++len;
  } while (mag);
  if (neg) {
    w2 = (w2 << 8) | (w1 >> 56);
    w1 = (w1 << 8) | (w0 >> 56);
    w0 = (w0 << 8) | (uint64_t)'-';
    ++len;
  }
  Sip s(k0, k1);
  const int full = len >> 3;
  if (full > 0) s.absorb(w0);
  if (full > 1) s.absorb(w1);
  const uint64_t tail = (full == 0 ? w0 : (full == 1 ? w1 : w2)) | ((uint64_t)len << 56);
  s.absorb(tail);
  return s.finish();
}

__device__ __forceinline__ uint64_t sip_of_bytes(const unsigned char *__restrict__ bytes,
                                                 int64_t lo, int64_t hi, uint64_t k0,
                                                 uint64_t k1) {
  const int64_t len = hi - lo;
  Sip s(k0, k1);
  int64_t p = lo;
  for (; p + 8 <= hi; p += 8) {
    uint64_t m = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) m |= (uint64_t)bytes[p + b] << (b * 8);
    s.absorb(m);
  }
  uint64_t m = (uint64_t)(len & 0xff) << 56;
  for (int b = 0; p + b < hi; ++b) m |= (uint64_t)bytes[p + b] << (b * 8);
  s.absorb(m);
  return s.finish();
}

template <typename IdT>
__global__ void __launch_bounds__(256) hash_ids_kernel(const IdT *__restrict__ ids, int64_t n,
                                                       uint64_t num_bins, uint64_t recip,
                                                       uint64_t k0, uint64_t k1,
                                                       int64_t *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int64_t)mod_barrett(sip_of_int((int64_t)ids[i], k0, k1), num_bins, recip);
}

__global__ void __launch_bounds__(256) hash_bytes_kernel(const unsigned char *__restrict__ bytes,
                                                         const int64_t *__restrict__ offsets,
                                                         int64_t n, uint64_t num_bins,
                                                         uint64_t recip, uint64_t k0,
                                                         uint64_t k1, int64_t *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int64_t)mod_barrett(sip_of_bytes(bytes, offsets[i], offsets[i + 1], k0, k1),
                                  num_bins, recip);
}

// ---- fused UnifiedEmbedding lookup ---------------------------------------------------------
// One launch per feature replaces num_chunks x (Hashing -> gather) + tf.concat
// (unified_embedding.py:198-215).  Lookup p = value * C + chunk: lane-private SipHash of the
// value under the chunk's salt gives the bucket; the 64 lookups of a wave are then copied
// cooperatively, D/4 lanes per row in 16-byte pieces (the gather kernel's layout rule), into
// out[p, :] -- and because p enumerates (value, chunk) row-major, out IS the concatenation
// [n, C * D]: no bucket array, no per-chunk activations, no concat pass ever reach HBM.
// Bytes per lookup: D*4 read + D*4 written (+ the id once per value, + 8 if buckets are kept
// for the backward).
constexpr int kUnifiedMaxChunks = 16;   // TFRS_UNIFIED_MAX_CHUNKS in include/tfrs_hip.h

struct UnifiedChunks {
  const float *table[kUnifiedMaxChunks];
  uint64_t k0[kUnifiedMaxChunks], k1[kUnifiedMaxChunks];
};

template <typename IdT, int PER_ROW>   // PER_ROW = D / 4 lanes per row, a power of two <= 64
__global__ void __launch_bounds__(256) unified_lookup_kernel(
    const IdT *__restrict__ ids, const unsigned char *__restrict__ bytes,
    const int64_t *__restrict__ offsets, int64_t n_values, int n_chunks, UnifiedChunks ch,
    uint64_t num_bins, uint64_t recip, float *__restrict__ out, int64_t out_row_stride,
    int64_t out_chunk0,
    int64_t *__restrict__ buckets) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int kRowsPerIter = 64 / PER_ROW;
  __shared__ const f4 *s_src[4][64];
  __shared__ f4 *s_dst[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = n_values * n_chunks;
  const int64_t wave_stride = (int64_t)gridDim.x * 4 * 64;
  const int piece = lane % PER_ROW, sub = lane / PER_ROW;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < total; base += wave_stride) {
    const int64_t p = base + lane;
    const f4 *src = nullptr;
    f4 *dst = nullptr;
    if (p < total) {
      const int64_t v = p / n_chunks;
      const int c = (int)(p - v * n_chunks);
      const uint64_t h = bytes ? sip_of_bytes(bytes, offsets[v], offsets[v + 1], ch.k0[c], ch.k1[c])
                               : sip_of_int((int64_t)ids[v], ch.k0[c], ch.k1[c]);
      const uint64_t bucket = mod_barrett(h, num_bins, recip);
      const int64_t row = v * out_row_stride + out_chunk0 + c;   // position in [n, C]
      if (buckets) buckets[row] = (int64_t)bucket;
      src = reinterpret_cast<const f4 *>(ch.table[c]) + bucket * PER_ROW;
      dst = reinterpret_cast<f4 *>(out) + row * PER_ROW;
    }
    s_src[wave][lane] = src;
    s_dst[wave][lane] = dst;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int it = 0; it < PER_ROW; ++it) {   // = 64 / kRowsPerIter iterations
      const int r = it * kRowsPerIter + sub;
      const f4 *sp = s_src[wave][r];
      f4 *dp = s_dst[wave][r];
      if (sp != nullptr) {
        const f4 x = __builtin_nontemporal_load(sp + piece);
        __builtin_nontemporal_store(x, dp + piece);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- every dense feature of a UnifiedEmbedding layer in ONE launch -----------------------------------
// The per-feature kernel above is launched once per feature: 26 launches of ~22 us at the DCN-v2 shapes
// (26 features x 65536 ids), launch-bound.  Here a "unit" is one (feature, chunk): lookup p = value * U + unit
// hashes ids[unit][value] under the unit's salt and copies the table row into the unit's column block of ITS
// feature's output (every feature keeps its own [n, chunks * D] tensor: the backward then receives one
// gradient per feature, no slicing of a wide buffer).
constexpr int kUnifiedMaxUnits = 64;    // 64 x 48 bytes of descriptors < 4 KiB of kernel arguments

struct UnifiedUnits {
  const void *ids[kUnifiedMaxUnits];
  const float *table[kUnifiedMaxUnits];
  float *out[kUnifiedMaxUnits];          // the unit's feature output [n, chunks * D]
  uint64_t k0[kUnifiedMaxUnits], k1[kUnifiedMaxUnits];
  int32_t chunks[kUnifiedMaxUnits];      // chunks of the unit's feature (row stride of `out` in units of D)
  int32_t chunk[kUnifiedMaxUnits];       // the unit's chunk number inside its feature
};

template <typename IdT, int PER_ROW>
__global__ void __launch_bounds__(256) unified_lookup_multi_kernel(
    const UnifiedUnits un, int n_units, int64_t n_values, uint64_t num_bins, uint64_t recip,
    int64_t *__restrict__ buckets, int64_t bucket_stride, int64_t bucket_col0) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int kRowsPerIter = 64 / PER_ROW;
  __shared__ const f4 *s_src[4][64];
  __shared__ f4 *s_dst[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = n_values * n_units;
  const int64_t wave_stride = (int64_t)gridDim.x * 4 * 64;
  const int piece = lane % PER_ROW, sub = lane / PER_ROW;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < total; base += wave_stride) {
    const int64_t p = base + lane;
    const f4 *src = nullptr;
    f4 *dst = nullptr;
    if (p < total) {
      const int64_t v = p / n_units;
      const int u = (int)(p - v * n_units);
      const int64_t id = (int64_t)static_cast<const IdT *>(un.ids[u])[v];
      const uint64_t bucket = mod_barrett(sip_of_int(id, un.k0[u], un.k1[u]), num_bins, recip);
      if (buckets) buckets[v * bucket_stride + bucket_col0 + u] = (int64_t)bucket;
      src = reinterpret_cast<const f4 *>(un.table[u]) + bucket * PER_ROW;
      dst = reinterpret_cast<f4 *>(un.out[u]) + (v * un.chunks[u] + un.chunk[u]) * PER_ROW;
    }
    s_src[wave][lane] = src;
    s_dst[wave][lane] = dst;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int it = 0; it < PER_ROW; ++it) {
      const int r = it * kRowsPerIter + sub;
      const f4 *sp = s_src[wave][r];
      f4 *dp = s_dst[wave][r];
      if (sp != nullptr) {
        const f4 x = __builtin_nontemporal_load(sp + piece);
        __builtin_nontemporal_store(x, dp + piece);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace tfrs

using namespace tfrs;

static inline uint64_t barrett_recip(int64_t num_bins) { return ~0ull / (uint64_t)num_bins; }

static inline unsigned hash_grid(int64_t n) {
  const int64_t blocks = (n + 255) / 256;
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, 256 * 32));
}

extern "C" int tfrs_hash_bucket_strong_ids(const void *ids, int ids_are_i64, int64_t n,
                                           int64_t num_bins, uint64_t salt0, uint64_t salt1,
                                           int64_t *out, void *stream) {
  TFRS_CHECK_ARG(n >= 0, "hash_bucket_strong: bad count");
  TFRS_CHECK_ARG(num_bins >= 1, "hash_bucket_strong: num_bins must be positive");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(ids && out, "hash_bucket_strong: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (ids_are_i64)
    hipLaunchKernelGGL((hash_ids_kernel<int64_t>), dim3(hash_grid(n)), dim3(256), 0, s,
                       (const int64_t *)ids, n, (uint64_t)num_bins, barrett_recip(num_bins), salt0, salt1, out);
  else
    hipLaunchKernelGGL((hash_ids_kernel<int32_t>), dim3(hash_grid(n)), dim3(256), 0, s,
                       (const int32_t *)ids, n, (uint64_t)num_bins, barrett_recip(num_bins), salt0, salt1, out);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_hash_bucket_strong_bytes(const unsigned char *bytes, const int64_t *offsets,
                                             int64_t n, int64_t num_bins, uint64_t salt0,
                                             uint64_t salt1, int64_t *out, void *stream) {
  TFRS_CHECK_ARG(n >= 0, "hash_bucket_strong: bad count");
  TFRS_CHECK_ARG(num_bins >= 1, "hash_bucket_strong: num_bins must be positive");
  if (n == 0) return TFRS_OK;
  TFRS_CHECK_ARG(offsets && out, "hash_bucket_strong: NULL pointer");
  hipLaunchKernelGGL(hash_bytes_kernel, dim3(hash_grid(n)), dim3(256), 0, (hipStream_t)stream,
                     bytes, offsets, n, (uint64_t)num_bins, barrett_recip(num_bins), salt0, salt1, out);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}

extern "C" int tfrs_unified_embedding_fwd(const void *ids, int ids_are_i64,
                                          const unsigned char *bytes, const int64_t *offsets,
                                          int64_t n_values, int n_chunks,
                                          const float *const *tables, const uint64_t *salt0,
                                          const uint64_t *salt1, int64_t num_bins, int d,
                                          float *out, int64_t *buckets, void *stream) {
  TFRS_CHECK_ARG(n_values >= 0 && n_chunks >= 1 && d >= 4, "unified_embedding_fwd: bad shape");
  TFRS_CHECK_ARG(num_bins >= 1, "unified_embedding_fwd: num_bins must be positive");
  const int per_row = d / 4;
  if (d % 4 != 0 || per_row > 64 || (per_row & (per_row - 1)) != 0) {
    set_error("unified_embedding_fwd: dim_per_table must be 4 * 2^k <= 256 (got %d); use the "
              "unfused Hashing + lookup path", d);
    return TFRS_ENOTIMPL;
  }
  if (n_values == 0) return TFRS_OK;
  TFRS_CHECK_ARG((ids || (bytes && offsets)) && tables && salt0 && salt1 && out,
                 "unified_embedding_fwd: NULL pointer");
  TFRS_CHECK_ARG(((uintptr_t)out) % 16 == 0, "unified_embedding_fwd: out must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // more chunks than one launch's descriptor holds: several launches, each writing its own
  // column block of the same [n, C * D] output
  for (int c0 = 0; c0 < n_chunks; c0 += kUnifiedMaxChunks) {
    const int cl = std::min(kUnifiedMaxChunks, n_chunks - c0);
    UnifiedChunks ch;
    for (int c = 0; c < kUnifiedMaxChunks; ++c) {
      const int cc = c < cl ? c0 + c : c0;
      TFRS_CHECK_ARG(tables[cc] && ((uintptr_t)tables[cc]) % 16 == 0,
                     "unified_embedding_fwd: tables must be non-NULL and 16-byte aligned");
      ch.table[c] = tables[cc];
      ch.k0[c] = salt0[cc];
      ch.k1[c] = salt1[cc];
    }
    const int64_t total = n_values * cl;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 256 * 16)));
#define TFRS_UE_LAUNCH(IDT, PR)                                                              \
  hipLaunchKernelGGL((unified_lookup_kernel<IDT, PR>), grid, dim3(256), 0, s, (const IDT *)ids, \
                     bytes, offsets, n_values, cl, ch, (uint64_t)num_bins, barrett_recip(num_bins), out,              \
                     (int64_t)n_chunks, (int64_t)c0, buckets)
#define TFRS_UE_DISPATCH(IDT)                                                                \
  switch (per_row) {                                                                         \
    case 1: TFRS_UE_LAUNCH(IDT, 1); break;                                                   \
    case 2: TFRS_UE_LAUNCH(IDT, 2); break;                                                   \
    case 4: TFRS_UE_LAUNCH(IDT, 4); break;                                                   \
    case 8: TFRS_UE_LAUNCH(IDT, 8); break;                                                   \
    case 16: TFRS_UE_LAUNCH(IDT, 16); break;                                                 \
    case 32: TFRS_UE_LAUNCH(IDT, 32); break;                                                 \
    default: TFRS_UE_LAUNCH(IDT, 64); break;                                                 \
  }
    if (ids_are_i64 || bytes) {
      TFRS_UE_DISPATCH(int64_t)
    } else {
      TFRS_UE_DISPATCH(int32_t)
    }
#undef TFRS_UE_DISPATCH
#undef TFRS_UE_LAUNCH
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}

extern "C" int tfrs_unified_embedding_fwd_multi(int n_units, const void *const *ids, int ids_are_i64,
                                                int64_t n_values, const float *const *tables,
                                                const uint64_t *salt0, const uint64_t *salt1,
                                                int64_t num_bins, int d, float *const *outs,
                                                const int32_t *feature_chunks, const int32_t *chunk_index,
                                                int64_t *buckets, void *stream) {
  TFRS_CHECK_ARG(n_units >= 1 && n_values >= 0 && d >= 4, "unified_embedding_fwd_multi: bad shape");
  TFRS_CHECK_ARG(num_bins >= 1, "unified_embedding_fwd_multi: num_bins must be positive");
  const int per_row = d / 4;
  if (d % 4 != 0 || per_row > 64 || (per_row & (per_row - 1)) != 0) {
    set_error("unified_embedding_fwd_multi: dim_per_table must be 4 * 2^k <= 256 (got %d); use the "
              "unfused Hashing + lookup path", d);
    return TFRS_ENOTIMPL;
  }
  if (n_values == 0) return TFRS_OK;
  TFRS_CHECK_ARG(ids && tables && salt0 && salt1 && outs && feature_chunks && chunk_index,
                 "unified_embedding_fwd_multi: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  for (int u0 = 0; u0 < n_units; u0 += kUnifiedMaxUnits) {
    const int ul = std::min(kUnifiedMaxUnits, n_units - u0);
    UnifiedUnits un;
    for (int u = 0; u < kUnifiedMaxUnits; ++u) {
      const int uu = u < ul ? u0 + u : u0;
      TFRS_CHECK_ARG(ids[uu] && tables[uu] && outs[uu] && ((uintptr_t)tables[uu]) % 16 == 0 &&
                         ((uintptr_t)outs[uu]) % 16 == 0 && feature_chunks[uu] >= 1 &&
                         chunk_index[uu] >= 0 && chunk_index[uu] < feature_chunks[uu],
                     "unified_embedding_fwd_multi: unit %d: NULL / misaligned pointer or bad chunk index", uu);
      un.ids[u] = ids[uu];
      un.table[u] = tables[uu];
      un.out[u] = outs[uu];
      un.k0[u] = salt0[uu];
      un.k1[u] = salt1[uu];
      un.chunks[u] = feature_chunks[uu];
      un.chunk[u] = chunk_index[uu];
    }
    const int64_t total = n_values * ul;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 256 * 16)));
#define TFRS_UM_LAUNCH(IDT, PR)                                                                       \
  hipLaunchKernelGGL((unified_lookup_multi_kernel<IDT, PR>), grid, dim3(256), 0, s, un, ul, n_values, \
                     (uint64_t)num_bins, barrett_recip(num_bins), buckets, (int64_t)n_units, (int64_t)u0)
#define TFRS_UM_DISPATCH(IDT)                                                                         \
  switch (per_row) {                                                                                  \
    case 1: TFRS_UM_LAUNCH(IDT, 1); break;                                                            \
    case 2: TFRS_UM_LAUNCH(IDT, 2); break;                                                            \
    case 4: TFRS_UM_LAUNCH(IDT, 4); break;                                                            \
    case 8: TFRS_UM_LAUNCH(IDT, 8); break;                                                            \
    case 16: TFRS_UM_LAUNCH(IDT, 16); break;                                                          \
    case 32: TFRS_UM_LAUNCH(IDT, 32); break;                                                          \
    default: TFRS_UM_LAUNCH(IDT, 64); break;                                                          \
  }
    if (ids_are_i64) {
      TFRS_UM_DISPATCH(int64_t)
    } else {
      TFRS_UM_DISPATCH(int32_t)
    }
#undef TFRS_UM_DISPATCH
#undef TFRS_UM_LAUNCH
    TFRS_LAUNCH_CHECK();
  }
  return TFRS_OK;
}
