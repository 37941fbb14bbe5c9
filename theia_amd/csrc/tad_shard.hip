// tad_shard.hip — row-sharded ingest: bucket a rank's rows by the owner of their key,
// owner = key_id mod world, local key = key_id / world, so that ONE all-to-all(v) (RCCL over xGMI) can ship every row —
// or every partial point pre-aggregated with tad_aggregate — to the GPU that owns its key.  The reference has no such
// step (Spark's shuffle plays this role, anomaly_detection.py:664-684: groupby(key)).
//
// One pass over the rows per kernel, no per-row global atomics: a workgroup histograms its tile by destination in LDS,
// reserves one contiguous range per destination with a single global atomic each, and places its rows there (ranks
// from the LDS counters).  Rows of one destination end up in workgroup-sized runs; their order is not deterministic,
// which is immaterial: Stage 0 aggregates with commutative integer operators.
#include "tad_internal.h"

namespace tad {

static constexpr int kShardBlock = 256;
static constexpr int kShardRows = 8;     // rows per thread per tile
static constexpr uint32_t kShardMaxWorld = 1024;

__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t world, uint64_t magic) {
  if (world == 1) return 0;                                 // (2^64 / 1 does not fit the magic)
  if ((key >> 32) == 0) {                                   // dense dictionary codes: one multiply-high
    const uint32_t q = (uint32_t)__umul64hi(key, magic);    // exact for key < 2^32 (magic = ceil(2^64 / world))
    return (uint32_t)key - q * world;
  }
  return (uint32_t)(key % world);
}

// counts[d] += rows of this tile going to d (TAD_KEY_SKIP rows are dropped)
__global__ __launch_bounds__(kShardBlock) void k_shard_count(const uint64_t *__restrict__ key, uint64_t n, uint32_t world,
                                                            uint64_t magic, unsigned long long *__restrict__ counts) {
  __shared__ uint32_t h[kShardMaxWorld];
  for (uint32_t d = threadIdx.x; d < world; d += kShardBlock) h[d] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * (kShardBlock * kShardRows);
#pragma unroll
  for (int j = 0; j < kShardRows; ++j) {
    const uint64_t i = base + (uint64_t)j * kShardBlock + threadIdx.x;
    if (i < n) {
      const uint64_t k = key[i];
      if (k != TAD_KEY_SKIP) atomicAdd(&h[owner_of(k, world, magic)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < world; d += kShardBlock)
    if (h[d]) atomicAdd(&counts[d], (unsigned long long)h[d]);
}

// cursor[d] starts at the destination's offset; every workgroup reserves its run with one atomic per destination
__global__ __launch_bounds__(kShardBlock) void k_shard_scatter(const uint64_t *__restrict__ key, const int64_t *__restrict__ t_end,
                                                              const uint64_t *__restrict__ value, uint64_t n, uint32_t world,
                                                              uint64_t magic, unsigned long long *__restrict__ cursor,
                                                              uint64_t *__restrict__ out_key, int64_t *__restrict__ out_t,
                                                              uint64_t *__restrict__ out_val) {
  __shared__ uint32_t h[kShardMaxWorld];
  __shared__ unsigned long long start[kShardMaxWorld];
  for (uint32_t d = threadIdx.x; d < world; d += kShardBlock) h[d] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * (kShardBlock * kShardRows);
  uint64_t k[kShardRows];
  uint32_t dst[kShardRows], rank[kShardRows];
#pragma unroll
  for (int j = 0; j < kShardRows; ++j) {
    const uint64_t i = base + (uint64_t)j * kShardBlock + threadIdx.x;
    k[j] = i < n ? key[i] : TAD_KEY_SKIP;
    dst[j] = 0; rank[j] = 0;
    if (k[j] != TAD_KEY_SKIP) {
      dst[j] = owner_of(k[j], world, magic);
      rank[j] = atomicAdd(&h[dst[j]], 1u);
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < world; d += kShardBlock)
    if (h[d]) start[d] = atomicAdd(&cursor[d], (unsigned long long)h[d]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kShardRows; ++j) {
    if (k[j] == TAD_KEY_SKIP) continue;
    const uint64_t i = base + (uint64_t)j * kShardBlock + threadIdx.x;
    const unsigned long long at = start[dst[j]] + rank[j];
    out_key[at] = k[j] / world;
    out_t[at] = t_end[i];
    out_val[at] = value[i];
  }
}

static uint64_t shard_magic(uint32_t world) { return world <= 1 ? 0 : UINT64_MAX / world + 1; }

bool shard_world_ok(uint32_t world) { return world >= 1 && world <= kShardMaxWorld; }

// counts: device array of `world` zeroed u64
void launch_shard_count(hipStream_t s, const uint64_t *key, uint64_t n, uint32_t world, unsigned long long *counts) {
  if (n == 0) return;
  const uint64_t blocks = (n + kShardBlock * kShardRows - 1) / (kShardBlock * kShardRows);
  hipLaunchKernelGGL(k_shard_count, dim3((unsigned)blocks), dim3(kShardBlock), 0, s, key, n, world, shard_magic(world), counts);
}

// cursor: device array of `world` u64 holding each destination's first output slot
void launch_shard_scatter(hipStream_t s, const uint64_t *key, const int64_t *t_end, const uint64_t *value, uint64_t n, uint32_t world,
                          unsigned long long *cursor, uint64_t *out_key, int64_t *out_t, uint64_t *out_val) {
  if (n == 0) return;
  const uint64_t blocks = (n + kShardBlock * kShardRows - 1) / (kShardBlock * kShardRows);
  hipLaunchKernelGGL(k_shard_scatter, dim3((unsigned)blocks), dim3(kShardBlock), 0, s, key, t_end, value, n, world, shard_magic(world),
                     cursor, out_key, out_t, out_val);
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_shard() { return reinterpret_cast<const void *>(&k_shard_count); }

}  // namespace tad
