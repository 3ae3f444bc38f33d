// shard_route.hip -- owner bucketing of a row-sharded embedding lookup (SURVEY.md section 8e, row 3:
// "ncclAllToAll ids -> local gather -> ncclAllToAll rows back"; BASELINE configs[4]: 100 tables x
// 10M rows x dim 32 row-sharded over the 8 GPUs of a node).
//
// Rank r owns table rows [r * rows_per_rank, (r + 1) * rows_per_rank).  Before the first exchange
// every rank must (a) count how many of its lookups each owner serves, (b) lay the shard-local row
// numbers out owner by owner (the send buffer of the all-to-all) and (c) remember where every
// lookup went, so that the rows that come back can be written to their final positions and the
// gradient rows can be sent the same way.  Round 2 did this with torch.argsort over the ids,
// torch.bincount(...).tolist() and an index_put (three passes over 13M ids and two host
// synchronisations per lookup); here it is three short kernels, no sort:
//
//   shard_count_kernel : per 4096-id tile, ids per owner                      -> tile_counts
//   shard_scan_kernel  : per owner, exclusive scan over the tiles + the total  -> tile_base, counts
//   shard_place_kernel : STABLE placement -- lookup i of owner o goes to send position
//                        start[o] + #{i' < i owned by o} -- so the order of the gradient rows an
//                        owner receives (hence the summation order of duplicate ids in its
//                        update) does not depend on scheduling: bit-reproducible
//
// Ids outside [0, input_dim) read as a zero row and take no gradient (like the plain gather):
// they are routed to owner 0 as row -1, consistently on every rank, so split sizes never diverge.
// Integer work, HBM-bound: n * (2 * id bytes read + 8 + 4 + 4 written); 13M ids in a few tens of
// microseconds against ~1.7 GB of embedding rows per lookup at configs[4].
#include <algorithm>

#include "common.h"

namespace tfrs {

constexpr int kRouteTile = 4096;     // ids per workgroup
constexpr int kRouteMaxWorld = 64;

// (takes the id VALUE: the callers load all their ids first -- the 64-bit division below expands into
// branches, and a load that sits behind them is awaited before the next one is issued)
__device__ __forceinline__ int owner_of(int64_t id, int64_t input_dim, int64_t rows_per_rank, int64_t *local) {
  if (id < 0 || id >= input_dim) {
    *local = -1;
    return 0;
  }
  const int64_t o = id / rows_per_rank;
  *local = id - o * rows_per_rank;
  return (int)o;
}

template <typename IdT>
__global__ void __launch_bounds__(256) shard_count_kernel(const void *ids, int64_t n, int64_t input_dim,
                                                          int64_t rows_per_rank, int world,
                                                          uint32_t *tile_counts) {
  __shared__ uint32_t hist[kRouteMaxWorld];
  if (threadIdx.x < kRouteMaxWorld) hist[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRouteTile;
  // every thread's 16 id loads are independent: one memory round trip per tile
  int own[kRouteTile / 256];
  int64_t idv[kRouteTile / 256];
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u) {   // unconditional, at a clamped index: 16 loads in flight
    const int64_t i = base + u * 256 + threadIdx.x;
    idv[u] = (int64_t) reinterpret_cast<const IdT *>(ids)[i < n ? i : n - 1];
  }
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u) {
    const int64_t i = base + u * 256 + threadIdx.x;
    int64_t local;
    const int o = owner_of(idv[u], input_dim, rows_per_rank, &local);
    own[u] = i < n ? o : -1;
  }
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u)
    if (own[u] >= 0) atomicAdd(&hist[own[u]], 1u);
  __syncthreads();
  if ((int)threadIdx.x < world) tile_counts[(int64_t)blockIdx.x * world + threadIdx.x] = hist[threadIdx.x];
}

// One wave per owner: lane l sums the counts of its contiguous range of tiles (independent loads),
// a wave scan of the 64 range sums gives every range its base, the lane then rewrites its range as
// running positions WITHIN the owner's bucket (the place kernel adds the bucket starts from counts).
__global__ void __launch_bounds__(64) shard_scan_kernel(const uint32_t *tile_counts, int64_t ntiles, int world,
                                                        int64_t *tile_base, int64_t *counts) {
  const int o = blockIdx.x, lane = threadIdx.x;
  const int64_t per = (ntiles + 63) / 64;
  const int64_t t0 = lane * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  int64_t sum = 0;
#pragma unroll 8
  for (int64_t t = t0; t < t1; ++t) sum += tile_counts[t * world + o];
  int64_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  int64_t run = incl - sum;
  uint32_t c[8];
  for (int64_t tb = t0; tb < t1; tb += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = tb + u < t1 ? tile_counts[(tb + u) * world + o] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (tb + u < t1) tile_base[(tb + u) * world + o] = run;
      run += c[u];
    }
  }
  if (lane == 63) counts[o] = incl;
}

// Stable placement.  The tile is 64 segments of 64 consecutive ids; segment s is handled by wave
// s % 4 in iteration s / 4.  Pass 1 counts ids per (segment, owner) with one ballot per owner, a
// scan over the segments gives each segment its base, pass 2 places every id at
// tile_base[owner] + segment base + (ids of the same owner before it in its segment).
template <typename IdT>
__global__ void __launch_bounds__(256) shard_place_kernel(const void *ids, int64_t n, int64_t input_dim,
                                                          int64_t rows_per_rank, int world,
                                                          const int64_t *tile_base, const int64_t *counts,
                                                          int64_t *send_ids, int32_t *perm, int32_t *order) {
  __shared__ uint32_t seg_cnt[64][kRouteMaxWorld];   // becomes the exclusive scan over the segments
  __shared__ int64_t start_s[kRouteMaxWorld];        // first send position of every owner's bucket
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int o = 0; o < world; ++o) {
      start_s[o] = run;
      run += counts[o];
    }
  }
  const int64_t base = (int64_t)blockIdx.x * kRouteTile;
  int own[kRouteTile / 256];
  int64_t loc[kRouteTile / 256];
  int64_t idv[kRouteTile / 256];
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u) {
    const int64_t i = base + (u * 4 + wave) * 64 + lane;
    idv[u] = (int64_t) reinterpret_cast<const IdT *>(ids)[i < n ? i : n - 1];
  }
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u) {
    const int64_t i = base + (u * 4 + wave) * 64 + lane;
    const int o = owner_of(idv[u], input_dim, rows_per_rank, &loc[u]);
    own[u] = i < n ? o : -1;
    if (!(i < n)) loc[u] = -1;
  }
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u) {
    const int seg = u * 4 + wave;
    for (int o = 0; o < world; ++o) {
      const uint64_t m = __ballot(own[u] == o);
      if (lane == 0) seg_cnt[seg][o] = (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < world) {
    uint32_t run = 0;
    for (int s = 0; s < 64; ++s) {
      const uint32_t c = seg_cnt[s][threadIdx.x];
      seg_cnt[s][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kRouteTile / 256; ++u) {
    const int seg = u * 4 + wave;
    const int64_t i = base + seg * 64 + lane;
    uint32_t rank = 0;
    for (int o = 0; o < world; ++o) {
      const uint64_t m = __ballot(own[u] == o);
      if (own[u] == o) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    if (own[u] >= 0) {
      const int64_t pos = start_s[own[u]] + tile_base[(int64_t)blockIdx.x * world + own[u]] +
                          seg_cnt[seg][own[u]] + rank;
      send_ids[pos] = loc[u];
      perm[i] = (int32_t)pos;       // lookup i travels in send slot pos (and its row comes back there)
      order[pos] = (int32_t)i;      // send slot pos carries lookup i
    }
  }
}

}  // namespace tfrs

extern "C" size_t tfrs_shard_route_workspace_bytes(int64_t n, int world) {
  if (n <= 0 || world <= 0) return 256;
  const int64_t ntiles = (n + tfrs::kRouteTile - 1) / tfrs::kRouteTile;
  return (size_t)ntiles * world * (sizeof(uint32_t) + sizeof(int64_t)) + 256;
}

extern "C" int tfrs_shard_route_ids(const void *ids, int ids_are_i64, int64_t n, int64_t input_dim,
                                    int64_t rows_per_rank, int world, int64_t *send_ids, int32_t *perm,
                                    int32_t *order, int64_t *counts, void *workspace, size_t workspace_bytes,
                                    void *stream) {
  using namespace tfrs;
  TFRS_CHECK_ARG(n >= 0 && input_dim >= 1 && rows_per_rank >= 1 && world >= 1, "shard_route_ids: bad shape");
  TFRS_CHECK_ARG(world <= kRouteMaxWorld, "shard_route_ids: world=%d > %d", world, kRouteMaxWorld);
  TFRS_CHECK_ARG((input_dim + rows_per_rank - 1) / rows_per_rank <= world,
                 "shard_route_ids: %lld rows per rank x %d ranks do not cover %lld rows",
                 (long long)rows_per_rank, world, (long long)input_dim);
  TFRS_CHECK_ARG(n <= 0x7FFFFFFFll, "shard_route_ids: more than 2^31 lookups in one call");
  TFRS_CHECK_ARG(counts, "shard_route_ids: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    TFRS_HIP(hipMemsetAsync(counts, 0, (size_t)world * sizeof(int64_t), s));
    return TFRS_OK;
  }
  TFRS_CHECK_ARG(ids && send_ids && perm && order && workspace, "shard_route_ids: NULL pointer");
  const size_t need = tfrs_shard_route_workspace_bytes(n, world);
  if (workspace_bytes < need) {
    set_error("shard_route_ids: workspace %zu < required %zu", workspace_bytes, need);
    return TFRS_ENOMEM;
  }
  const int64_t ntiles = (n + kRouteTile - 1) / kRouteTile;
  int64_t *tile_base = reinterpret_cast<int64_t *>(workspace);
  uint32_t *tile_counts = reinterpret_cast<uint32_t *>(tile_base + ntiles * world);
  const dim3 grid((unsigned)ntiles), block(256);
  if (ids_are_i64)
    hipLaunchKernelGGL((shard_count_kernel<int64_t>), grid, block, 0, s, ids, n, input_dim, rows_per_rank, world, tile_counts);
  else
    hipLaunchKernelGGL((shard_count_kernel<int32_t>), grid, block, 0, s, ids, n, input_dim, rows_per_rank, world, tile_counts);
  TFRS_LAUNCH_CHECK();
  hipLaunchKernelGGL(shard_scan_kernel, dim3((unsigned)world), dim3(64), 0, s, tile_counts, ntiles, world, tile_base,
                     counts);
  TFRS_LAUNCH_CHECK();
  if (ids_are_i64)
    hipLaunchKernelGGL((shard_place_kernel<int64_t>), grid, block, 0, s, ids, n, input_dim, rows_per_rank, world,
                       tile_base, counts, send_ids, perm, order);
  else
    hipLaunchKernelGGL((shard_place_kernel<int32_t>), grid, block, 0, s, ids, n, input_dim, rows_per_rank, world,
                       tile_base, counts, send_ids, perm, order);
  TFRS_LAUNCH_CHECK();
  return TFRS_OK;
}
