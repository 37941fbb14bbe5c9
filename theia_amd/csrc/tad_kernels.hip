// tad_kernels.hip — Stage 0 (GROUP BY key, flowEndSeconds), per-key stddev_samp, EWMA detector,
// compaction of anomalous points.  gfx950 only; built with -ffp-contract=off so that every FP64
// expression below rounds exactly like the reference's Python floats.
//
// Reference semantics (plugins/anomaly-detection/anomaly_detection.py):
//   Stage 0  :507-614  GROUP BY <key cols>, flowEndSeconds with max()/sum() over UInt64
//   Stage 1  :664-684  per-key series (ascending flowEndSeconds) + stddev_samp
//   EWMA     :146-165  e_t = (1-a) e_{t-1} + a float(x_t), e_-1 = 0
//   verdict  :168-212  |float(x_t) - e_t| > stddev  (strict; stddev null -> False)
//   filter   :352-421  keep anomaly == True
#include <cstdlib>

#include "tad_internal.h"

namespace tad {

static constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t gcd_u64(uint64_t a, uint64_t b) {
  if (a == 0) return b;
  if (b == 0) return a;
  if (((a | b) >> 32) == 0) {
    uint32_t x = (uint32_t)a, y = (uint32_t)b;
    while (y) { uint32_t r = x % y; x = y; y = r; }
    return x;
  }
  while (b) { uint64_t r = a % b; a = b; b = r; }
  return a;
}

__device__ __forceinline__ uint64_t absdiff_i64(int64_t a, int64_t b) {
  return a >= b ? (uint64_t)a - (uint64_t)b : (uint64_t)b - (uint64_t)a;
}

struct MetaAcc {
  int64_t tmin, tmax, tref;
  uint64_t g, used;
};

__device__ __forceinline__ MetaAcc meta_merge(MetaAcc a, const MetaAcc &b) {
  if (b.used == 0) return a;
  if (a.used == 0) return b;
  a.tmin = b.tmin < a.tmin ? b.tmin : a.tmin;
  a.tmax = b.tmax > a.tmax ? b.tmax : a.tmax;
  a.g = gcd_u64(gcd_u64(a.g, b.g), absdiff_i64(a.tref, b.tref));
  a.used += b.used;
  return a;
}

__device__ __forceinline__ MetaAcc meta_shfl_down(const MetaAcc &a, int d) {
  MetaAcc r;
  r.tmin = __shfl_down((long long)a.tmin, d);
  r.tmax = __shfl_down((long long)a.tmax, d);
  r.tref = __shfl_down((long long)a.tref, d);
  r.g = __shfl_down((unsigned long long)a.g, d);
  r.used = __shfl_down((unsigned long long)a.used, d);
  return r;
}

__device__ __forceinline__ bool row_kept(int64_t te, const int64_t *t_start, uint64_t i, RowFilter f) {
  if (f.end_time != 0 && !(te < f.end_time)) return false;          // :584-586
  if (f.start_time != 0 && t_start != nullptr && !(t_start[i] >= f.start_time)) return false;  // :581-583
  return true;
}

// ------------------------------------------------------------------------------------------------
// k_meta — derive the flowEndSeconds lattice (min, max, gcd of differences) of the rows that pass
// the filters.  One read of the time column (+ key columns for TAD_KEY_SKIP).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_meta(const uint64_t *__restrict__ key,
                                                 const uint64_t *__restrict__ key2,
                                                 const int64_t *__restrict__ t_end,
                                                 const int64_t *__restrict__ t_start, uint64_t n,
                                                 RowFilter f, MetaPartial *__restrict__ partials) {
  MetaAcc acc{0, 0, 0, 0, 0};
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t te = t_end[i];
    bool live = key[i] != TAD_KEY_SKIP;
    if (key2 != nullptr) live = live || key2[i] != TAD_KEY_SKIP;
    if (!live || !row_kept(te, t_start, i, f)) continue;
    if (acc.used == 0) {
      acc.tmin = acc.tmax = acc.tref = te;
      acc.g = 0;
    } else {
      acc.tmin = te < acc.tmin ? te : acc.tmin;
      acc.tmax = te > acc.tmax ? te : acc.tmax;
      const uint64_t d = absdiff_i64(te, acc.tref);
      if (d != 0 && (acc.g == 0 || d % acc.g != 0)) acc.g = gcd_u64(acc.g, d);
    }
    acc.used++;
  }
  for (int d = 32; d >= 1; d >>= 1) acc = meta_merge(acc, meta_shfl_down(acc, d));
  __shared__ MetaAcc s_acc[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_acc[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    MetaAcc a = s_acc[0];
    for (int w = 1; w < kBlock / 64; ++w) a = meta_merge(a, s_acc[w]);
    MetaPartial p;
    p.tmin = a.tmin; p.tmax = a.tmax; p.tref = a.tref; p.g = a.g; p.used = a.used;
    partials[blockIdx.x] = p;
  }
}

int launch_meta(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                const int64_t *t_start, uint64_t n, RowFilter f, MetaPartial *partials, int n_blocks) {
  hipLaunchKernelGGL(k_meta, dim3(n_blocks), dim3(kBlock), 0, s, key, key2, t_end, t_start, n, f, partials);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// k_scatter — Stage 0 v1: every row updates its (bucket, key) cell with one agent-scope u64 atomic
// (add wraps mod 2^64 = ClickHouse sum(UInt64); max is unsigned) and marks the cell present.
// Integer atomics are associative and commutative: the result is bit-exact whatever the order.
// Measured on MI355X (tools/ubench_scatter.hip): the 24 B/row column stream alone runs at 5.6 TB/s;
// the random 8-byte read-modify-write is bound at ~23.5e9 L2-miss line transactions/s.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cell_index(const Lattice &L, int64_t te, uint64_t &bucket) {
  const uint64_t d = (uint64_t)te - (uint64_t)L.t0;  // te < t0 wraps to a huge value -> rejected below
  uint64_t b;
  if (L.mode == 0) {
    b = d;
  } else if (L.mode == 1) {
    if (d >> 32) return false;
    b = __umul64hi(d, L.magic);
  } else {
    b = d / (uint64_t)L.step;
  }
  if (b >= L.nb || b * (uint64_t)L.step != d) return false;
  bucket = b;
  return true;
}

template <bool OPMAX>
__device__ __forceinline__ void cell_update(const Grid &g, uint64_t bucket, uint64_t key, uint64_t v) {
  const uint64_t c = bucket * g.K + key;
  if (OPMAX)
    __hip_atomic_fetch_max(g.val + c, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_fetch_add(g.val + c, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  g.flag[c] = FLAG_PRESENT;  // same value from every writer: a plain byte store is enough
}

template <bool OPMAX>
__device__ __forceinline__ void scatter_row(uint64_t k1, uint64_t k2, bool has2, int64_t te, uint64_t v,
                                            const Lattice &L, const Grid &g, uint32_t &err, uint32_t &used) {
  uint64_t bucket;
  const bool live1 = k1 != TAD_KEY_SKIP, live2 = has2 && k2 != TAD_KEY_SKIP;
  if (!live1 && !live2) return;
  if (!cell_index(L, te, bucket)) { err |= DEV_ERR_OFF_LATTICE; return; }
  if (live1) {
    if (k1 >= g.K) err |= DEV_ERR_KEY_RANGE;
    else { cell_update<OPMAX>(g, bucket, k1, v); used++; }
  }
  if (live2) {
    if (k2 >= g.K) err |= DEV_ERR_KEY_RANGE;
    else { cell_update<OPMAX>(g, bucket, k2, v); used++; }
  }
}

template <bool OPMAX, bool VEC2>
__global__ __launch_bounds__(kBlock) void k_scatter(const uint64_t *__restrict__ key,
                                                    const uint64_t *__restrict__ key2,
                                                    const int64_t *__restrict__ t_end,
                                                    const int64_t *__restrict__ t_start,
                                                    const uint64_t *__restrict__ value, uint64_t n,
                                                    RowFilter f, Lattice L, Grid g, DevCounters *ctr) {
  uint32_t err = 0, used = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool has2 = key2 != nullptr;
  if (VEC2) {
    // 16-byte loads: two rows per lane per column, 1 KiB per wave-instruction.
    const uint64_t n2 = n >> 1;
    const ulonglong2 *key_v = reinterpret_cast<const ulonglong2 *>(key);
    const ulonglong2 *key2_v = reinterpret_cast<const ulonglong2 *>(key2);
    const longlong2 *te_v = reinterpret_cast<const longlong2 *>(t_end);
    const ulonglong2 *val_v = reinterpret_cast<const ulonglong2 *>(value);
    for (uint64_t i = tid; i < n2; i += stride) {
      const ulonglong2 k = key_v[i];
      const longlong2 te = te_v[i];
      const ulonglong2 v = val_v[i];
      ulonglong2 k2 = make_ulonglong2(TAD_KEY_SKIP, TAD_KEY_SKIP);
      if (has2) k2 = key2_v[i];
      if (row_kept(te.x, t_start, 2 * i, f)) scatter_row<OPMAX>(k.x, k2.x, has2, te.x, v.x, L, g, err, used);
      if (row_kept(te.y, t_start, 2 * i + 1, f)) scatter_row<OPMAX>(k.y, k2.y, has2, te.y, v.y, L, g, err, used);
    }
    if ((n & 1) && tid == 0) {
      const uint64_t i = n - 1;
      if (row_kept(t_end[i], t_start, i, f))
        scatter_row<OPMAX>(key[i], has2 ? key2[i] : TAD_KEY_SKIP, has2, t_end[i], value[i], L, g, err, used);
    }
  } else {
    for (uint64_t i = tid; i < n; i += stride) {
      if (row_kept(t_end[i], t_start, i, f))
        scatter_row<OPMAX>(key[i], has2 ? key2[i] : TAD_KEY_SKIP, has2, t_end[i], value[i], L, g, err, used);
    }
  }
  // one counter update per wave
  unsigned long long u = used;
  for (int d = 32; d >= 1; d >>= 1) {
    u += __shfl_down(u, d);
    err |= __shfl_down(err, d);
  }
  if ((threadIdx.x & 63) == 0) {
    if (u) atomicAdd(&ctr->rows_used, u);
    if (err) atomicOr(&ctr->err, err);
  }
}

void launch_scatter(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                    const int64_t *t_start, const uint64_t *value, uint64_t n, RowFilter f,
                    Lattice lat, Grid g, bool op_max, DevCounters *ctr) {
  if (n == 0) return;
  auto aligned16 = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = aligned16(key) && aligned16(key2) && aligned16(t_end) && aligned16(value) && n >= 2;
  const uint64_t work = vec ? (n >> 1) : n;
  int blocks = (int)((work + kBlock - 1) / kBlock);
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride the rest (guide: ~8-16 blocks per CU)
  if (blocks < 1) blocks = 1;
#define TAD_LAUNCH_SCATTER(OPMAX, VEC) \
  hipLaunchKernelGGL((k_scatter<OPMAX, VEC>), dim3(blocks), dim3(kBlock), 0, s, key, key2, t_end, t_start, value, n, f, lat, g, ctr)
  if (op_max) { if (vec) TAD_LAUNCH_SCATTER(true, true); else TAD_LAUNCH_SCATTER(true, false); }
  else        { if (vec) TAD_LAUNCH_SCATTER(false, true); else TAD_LAUNCH_SCATTER(false, false); }
#undef TAD_LAUNCH_SCATTER
}

// ------------------------------------------------------------------------------------------------
// k_key_sigma — one lane = one key, walking its series in ascending time over the time-major grid.
// stddev_samp exactly as Spark's CentralMomentAgg streams it:
//   n += 1; d = x - avg; dn = d / n; avg += dn; m2 += d * (d - dn);  sigma = sqrt(m2 / (n - 1))
// sequential per key, so the bits do not depend on how the GPU is partitioned.  n < 2 -> no sigma
// (Spark >= 3.1 returns null; anomaly_detection.py:198-201 then yields False for every point).
// EWMA_COUNT additionally runs the EWMA recurrence and counts the key's anomalous points.
// ------------------------------------------------------------------------------------------------
static constexpr int kUnroll = 16;  // independent loads in flight per lane (k_emit_points)

// RCP_LDS: the reciprocal table (T + 1 doubles) is copied to LDS first.  A table lookup from global memory inside the
// step would be a vector-memory load YOUNGER than the prefetched chunk of the walk, and vmcnt retires in order: waiting
// for it waits for the whole prefetch, which serialises the walk into one memory round trip per chunk (127 -> 101 us).
// COOP: one WAVEFRONT per key (walk_series_coop: long series on few keys); every lane runs the same recurrence, lane 0 writes.
template <bool EWMA_COUNT, bool RCP_LDS, bool COOP = false>
__global__ __launch_bounds__(kBlock) void k_key_sigma(Grid g, double alpha, const double *__restrict__ rcp_g, double *__restrict__ sigma,
                                                      uint32_t *__restrict__ n_pts,
                                                      uint32_t *__restrict__ n_anom, DevCounters *ctr,
                                                      double *__restrict__ key_mean, double *__restrict__ key_m2) {
  const uint64_t gtid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t k = COOP ? gtid >> 6 : gtid;
  const bool writer = !COOP || (threadIdx.x & 63) == 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rcp[];
  const double *rcp = rcp_g;
  if (RCP_LDS) {
    double *l = reinterpret_cast<double *>(smem_rcp);
    for (uint64_t i = threadIdx.x; i <= g.T; i += kBlock) l[i] = rcp_g[i];
    __syncthreads();
    rcp = l;
  }
  unsigned long long my_pts = 0;
  unsigned my_key = 0;
  if (k < g.K) {
    double cnt = 0.0, avg = 0.0, m2 = 0.0;
    uint32_t n = 0;
    auto sigma_step = [&](uint64_t, uint8_t fl, unsigned long long raw) {
      if (fl & FLAG_PRESENT) {
        const double x = (double)raw;
        cnt = cnt + 1.0;
        n++;
        const double d = x - avg;
        // == d / cnt, bit for bit.  Without the table in LDS (series of more than 4095 buckets) the lookup would be a global load on the
        // step's dependency chain, younger than the walk's prefetched blocks (vmcnt retires in order): the IEEE division is cheaper
        const double dn = RCP_LDS ? div_by_count(d, cnt, rcp[n]) : d / cnt;
        avg = avg + dn;
        m2 = m2 + d * (d - dn);
      }
    };
    if (COOP) {
      // the reciprocals of a block's 64 possible counts, one IEEE division per lane and block (= RN(1 / count), what div_by_count
      // needs), picked up by readlane: the step itself is 5 FMAs instead of a division on the one wavefront's issue-bound chain
      double rl = 0.0;
      walk_series_coop(g, k,
                       [&](uint64_t) { rl = 1.0 / (double)(n + 1u + (threadIdx.x & 63u)); },
                       [&](uint64_t, uint8_t, double x, int ord) {
                         cnt = cnt + 1.0;
                         n++;
                         const double d = x - avg;
                         const double dn = div_by_count(d, cnt, readlane_f64(rl, ord));
                         avg = avg + dn;
                         m2 = m2 + d * (d - dn);
                       });
    } else {
      walk_series(g, k, sigma_step);
    }
    const bool has_sigma = n >= 2;
    const double sg = has_sigma ? sqrt(m2 / (cnt - 1.0)) : 0.0;
    if (writer) {
      sigma[k] = sg;
      n_pts[k] = n;
      if (key_mean != nullptr) { key_mean[k] = avg; key_m2[k] = m2; }
      my_pts = n;
      my_key = n > 0;
    }
    if (EWMA_COUNT) {
      uint32_t a = 0;
      if (has_sigma) {
        const double one_minus = 1.0 - alpha;
        double e = 0.0;
        auto ewma_step = [&](uint64_t, uint8_t fl, unsigned long long raw) {
          if (fl & FLAG_PRESENT) {
            const double x = (double)raw;
            e = one_minus * e + alpha * x;
            a += fabs(x - e) > sg ? 1u : 0u;
          }
        };
        if (COOP) walk_series_coop(g, k, [](uint64_t) {}, [&](uint64_t, uint8_t, double x, int) {
                    e = one_minus * e + alpha * x;
                    a += fabs(x - e) > sg ? 1u : 0u;
                  });
        else walk_series(g, k, ewma_step);
      }
      if (writer) n_anom[k] = a;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    my_pts += __shfl_down(my_pts, d);
    my_key += __shfl_down(my_key, d);
  }
  if ((threadIdx.x & 63) == 0 && my_key) {
    atomicAdd(&ctr->n_points, my_pts);
    atomicAdd(&ctr->n_keys, (unsigned long long)my_key);
  }
}

void launch_key_sigma(hipStream_t s, Grid g, double alpha, bool ewma_count, const double *rcp, double *sigma,
                      uint32_t *n_pts, uint32_t *n_anom, DevCounters *ctr, double *key_mean, double *key_m2) {
  if (g.K == 0) return;
  const size_t lds = (size_t)(g.T + 1) * 8;
  const bool in_lds = lds <= 32768;
  if (coop_shape(g)) {   // long series on few keys: a wavefront per key
    const int cblocks = (int)((g.K * 64 + kBlock - 1) / kBlock);
#define TAD_KSC(EC, RL) hipLaunchKernelGGL((k_key_sigma<EC, RL, true>), dim3(cblocks), dim3(kBlock), RL ? lds : 0, s, g, alpha, rcp, sigma, n_pts, n_anom, ctr, key_mean, key_m2)
    if (ewma_count) { if (in_lds) TAD_KSC(true, true); else TAD_KSC(true, false); }
    else { if (in_lds) TAD_KSC(false, true); else TAD_KSC(false, false); }
#undef TAD_KSC
    return;
  }
  const int blocks = (int)((g.K + kBlock - 1) / kBlock);
#define TAD_KS(EC, RL) hipLaunchKernelGGL((k_key_sigma<EC, RL>), dim3(blocks), dim3(kBlock), RL ? lds : 0, s, g, alpha, rcp, sigma, n_pts, n_anom, ctr, key_mean, key_m2)
  if (ewma_count) { if (in_lds) TAD_KS(true, true); else TAD_KS(true, false); }
  else { if (in_lds) TAD_KS(false, true); else TAD_KS(false, false); }
#undef TAD_KS
}

// ------------------------------------------------------------------------------------------------
// k_moments — Chan et al. pairwise merge of the per-key (n, mean, M2) triples in a FIXED order
// (strided per thread, then a fixed shuffle tree), so a shard's moments do not depend on scheduling.
// The host merges the kMomentBlocks partials, and the multi-GPU host merges shards the same way.
// ------------------------------------------------------------------------------------------------

struct MomentsArgs {
  uint64_t K;
  const uint32_t *n_pts;
  const double *key_mean, *key_m2;
  Moments *partials;
  DevCounters *ctr;
};

// block `mb` of kMomentBlocks (a launch of its own, or the trailing blocks of the scan's first launch: launch_scan_moments)
__device__ __forceinline__ void moments_block(uint32_t mb, uint64_t K, const uint32_t *__restrict__ n_pts,
                                              const double *__restrict__ key_mean,
                                              const double *__restrict__ key_m2,
                                              Moments *__restrict__ partials, DevCounters *ctr) {
  Moments acc{0.0, 0.0, 0.0};
  unsigned long long pts = 0, keys = 0;
  const uint64_t stride = (uint64_t)kMomentBlocks * kBlock;
  // The merge order is fixed (strided per thread); the loads are not part of that chain: eight keys' triples are fetched
  // together, then merged in order — at 1e6 keys a thread has ~30 keys and the loop was 30 dependent memory round trips
  // (C4: detect + emit 0.50 -> 0.48 ms).
  constexpr int kMU = 8;
  for (uint64_t k = (uint64_t)mb * kBlock + threadIdx.x; k < K; k += stride * kMU) {
    uint32_t n[kMU];
    double mean[kMU], m2[kMU];
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
      const uint64_t ku = k + (uint64_t)u * stride;
      const bool in = ku < K;
      n[u] = in ? n_pts[ku] : 0u;
      mean[u] = in ? key_mean[ku] : 0.0;
      m2[u] = in ? key_m2[ku] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
      acc = chan_merge(acc, Moments{(double)n[u], mean[u], m2[u]});   // n == 0 (also: past the end) leaves acc unchanged
      pts += n[u];
      keys += n[u] > 0;
    }
  }
  if (ctr != nullptr) {
    for (int d = 32; d >= 1; d >>= 1) { pts += __shfl_down(pts, d); keys += __shfl_down(keys, d); }
    if ((threadIdx.x & 63) == 0 && keys) { atomicAdd(&ctr->n_points, pts); atomicAdd(&ctr->n_keys, keys); }
  }
  for (int d = 1; d < 64; d <<= 1) {
    Moments o{__shfl_xor(acc.n, d), __shfl_xor(acc.mean, d), __shfl_xor(acc.m2, d)};
    // both partners must compute the same value: order the pair by lane
    acc = (threadIdx.x & d) ? chan_merge(o, acc) : chan_merge(acc, o);
  }
  __shared__ Moments s_m[kBlock / 64];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    Moments a = s_m[0];
    for (int w = 1; w < kBlock / 64; ++w) a = chan_merge(a, s_m[w]);
    partials[mb] = a;
  }
}

__global__ __launch_bounds__(kBlock) void k_moments(MomentsArgs M) {
  moments_block(blockIdx.x, M.K, M.n_pts, M.key_mean, M.key_m2, M.partials, M.ctr);
}

void launch_moments(hipStream_t s, uint64_t K, const uint32_t *n_pts, const double *key_mean,
                    const double *key_m2, Moments *partials, DevCounters *ctr) {
  hipLaunchKernelGGL(k_moments, dim3(kMomentBlocks), dim3(kBlock), 0, s, MomentsArgs{K, n_pts, key_mean, key_m2, partials, ctr});
}

// per-key count of points flagged by a detector kernel (DBSCAN / ARIMA), or of all points
__global__ __launch_bounds__(kBlock) void k_count_flags(Grid g, bool all_points, uint32_t *__restrict__ n_anom) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= g.K) return;
  const uint8_t want = all_points ? FLAG_PRESENT : (uint8_t)(FLAG_PRESENT | FLAG_ANOMALY);
  uint32_t a = 0;
  for (uint64_t t = 0; t < g.T; ++t) a += (g.flag[t * g.K + k] & want) == want ? 1u : 0u;
  n_anom[k] = a;
}

// the same with a wavefront per key (long series on few keys): the lanes take the buckets in strides of 64
__global__ __launch_bounds__(kBlock) void k_count_flags_coop(Grid g, bool all_points, uint32_t *__restrict__ n_anom) {
  const uint64_t k = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (k >= g.K) return;   // wavefront-uniform
  const uint8_t want = all_points ? FLAG_PRESENT : (uint8_t)(FLAG_PRESENT | FLAG_ANOMALY);
  uint32_t a = 0;
  for (uint64_t t = threadIdx.x & 63; t < g.T; t += 64) a += (g.flag[t * g.K + k] & want) == want ? 1u : 0u;
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_down(a, d);
  if ((threadIdx.x & 63) == 0) n_anom[k] = a;
}

void launch_count_flags(hipStream_t s, Grid g, bool all_points, uint32_t *n_anom) {
  if (g.K == 0) return;
  if (coop_shape(g)) {
    hipLaunchKernelGGL(k_count_flags_coop, dim3((unsigned)((g.K * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, g, all_points, n_anom);
    return;
  }
  const int blocks = (int)((g.K + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_count_flags, dim3(blocks), dim3(kBlock), 0, s, g, all_points, n_anom);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of per-key counts -> row offsets (deterministic output order: key, then time)
// ------------------------------------------------------------------------------------------------
static constexpr int kScanItems = 8;
static constexpr int kScanTile = kBlock * kScanItems;

__device__ __forceinline__ unsigned long long block_exclusive_scan(unsigned long long x, unsigned long long *total) {
  // returns the exclusive prefix of x over the block's threads; *total (all threads) = block sum
  __shared__ unsigned long long s_wave[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = x;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) base += s_wave[w];
    tot += s_wave[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - x;
}

// MOMENTS: the blocks behind the scan's nb blocks are the kMomentBlocks blocks of the moments merge (same inputs' producer, one launch)
template <bool MOMENTS>
__global__ __launch_bounds__(kBlock) void k_scan_reduce(const uint32_t *__restrict__ cnt, uint64_t K,
                                                        unsigned long long *__restrict__ bsum, uint32_t nb, MomentsArgs M) {
  if (MOMENTS && blockIdx.x >= nb) {   // workgroup-uniform
    moments_block(blockIdx.x - nb, M.K, M.n_pts, M.key_mean, M.key_m2, M.partials, M.ctr);
    return;
  }
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  unsigned long long s = 0;
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < K) s += cnt[base + j];
  unsigned long long tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void k_scan_top(unsigned long long *bsum, uint64_t nb,
                                                     unsigned long long *total_out, unsigned long long *total_copy) {
  unsigned long long carry = 0;
  for (uint64_t base = 0; base < nb; base += kBlock) {
    const uint64_t i = base + threadIdx.x;
    const unsigned long long x = i < nb ? bsum[i] : 0ull;
    unsigned long long tot;
    const unsigned long long ex = block_exclusive_scan(x, &tot);
    if (i < nb) bsum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    *total_out = carry;
    if (total_copy != nullptr) *total_copy = carry;   // the job tail (one device-to-host copy per job)
  }
}

// OWN_BASE: bsum holds the raw block sums (no k_scan_top launch): every block adds up the sums before it itself — at most
// kScanOwnBaseBlocks values — and block 0 writes the total (off[K] and the job tail)
static constexpr uint64_t kScanOwnBaseBlocks = 4096;
template <bool OWN_BASE>
__global__ __launch_bounds__(kBlock) void k_scan_apply(const uint32_t *__restrict__ cnt, uint64_t K,
                                                       const unsigned long long *__restrict__ bsum,
                                                       unsigned long long *__restrict__ off, uint32_t nb, unsigned long long *total_copy) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint32_t c[kScanItems];
  unsigned long long s = 0;
  for (int j = 0; j < kScanItems; ++j) {
    c[j] = base + j < K ? cnt[base + j] : 0u;
    s += c[j];
  }
  unsigned long long bbase;
  if (OWN_BASE) {
    unsigned long long before = 0, all = 0;
    for (uint32_t i = threadIdx.x; i < nb; i += kBlock) {
      const unsigned long long v = bsum[i];
      all += v;
      if (i < blockIdx.x) before += v;
    }
    unsigned long long t1, t2;
    block_exclusive_scan(before, &t1);
    block_exclusive_scan(all, &t2);
    bbase = t1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      off[K] = t2;
      if (total_copy != nullptr) *total_copy = t2;
    }
  } else {
    bbase = bsum[blockIdx.x];
  }
  unsigned long long tot;
  unsigned long long ex = block_exclusive_scan(s, &tot) + bbase;
  for (int j = 0; j < kScanItems; ++j) {
    if (base + j < K) off[base + j] = ex;
    ex += c[j];
  }
}

size_t scan_scratch_elems(uint64_t K) { return (size_t)((K + kScanTile - 1) / kScanTile) + 1; }

static void scan_launches(hipStream_t s, const uint32_t *cnt, unsigned long long *off, uint64_t K, unsigned long long *scratch,
                          unsigned long long *total_copy, const MomentsArgs *M) {
  const uint64_t nb = (K + kScanTile - 1) / kScanTile;
  const MomentsArgs none{0, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (M != nullptr) hipLaunchKernelGGL(k_scan_reduce<true>, dim3((unsigned)nb + kMomentBlocks), dim3(kBlock), 0, s, cnt, K, scratch, (uint32_t)nb, *M);
  else hipLaunchKernelGGL(k_scan_reduce<false>, dim3((unsigned)nb), dim3(kBlock), 0, s, cnt, K, scratch, (uint32_t)nb, none);
  if (nb <= kScanOwnBaseBlocks) {   // two launches: the blocks of the second one derive their own base from the raw block sums
    hipLaunchKernelGGL(k_scan_apply<true>, dim3((unsigned)nb), dim3(kBlock), 0, s, cnt, K, scratch, off, (uint32_t)nb, total_copy);
    return;
  }
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, s, scratch, nb, off + K, total_copy);
  hipLaunchKernelGGL(k_scan_apply<false>, dim3((unsigned)nb), dim3(kBlock), 0, s, cnt, K, scratch, off, (uint32_t)nb, total_copy);
}

void launch_scan(hipStream_t s, const uint32_t *cnt, unsigned long long *off, uint64_t K,
                 unsigned long long *scratch, unsigned long long *total_copy) {
  if (K == 0) {
    hipMemsetAsync(off, 0, sizeof(unsigned long long), s);
    if (total_copy != nullptr) hipMemsetAsync(total_copy, 0, sizeof(unsigned long long), s);
    return;
  }
  scan_launches(s, cnt, off, K, scratch, total_copy, nullptr);
}

// the scan of the per-key row counts and the moments merge of the same per-key arrays in the scan's launches (one launch fewer)
void launch_scan_moments(hipStream_t s, const uint32_t *cnt, unsigned long long *off, uint64_t K, unsigned long long *scratch,
                         unsigned long long *total_copy, const uint32_t *n_pts, const double *key_mean, const double *key_m2, Moments *partials,
                         DevCounters *ctr) {
  if (K == 0) {
    launch_scan(s, cnt, off, K, scratch, total_copy);
    launch_moments(s, K, n_pts, key_mean, key_m2, partials, ctr);
    return;
  }
  const MomentsArgs M{K, n_pts, key_mean, key_m2, partials, ctr};
  scan_launches(s, cnt, off, K, scratch, total_copy, &M);
}

// ------------------------------------------------------------------------------------------------
// k_emit — write the anomalous points (anomaly_detection.py:352-394) in (key, time) order.
// KIND 0: EWMA, recomputed on the fly (cheaper than storing e_t for every point);
// KIND 1: verdict bits + calc[] written by a detector kernel (ARIMA);
// KIND 2: verdict bits, algoCalc = 0.0 (DBSCAN placeholder, :312-322).
// ------------------------------------------------------------------------------------------------
// COOP: one wavefront per key (walk_series_coop); all lanes run the recurrences, lane 0 stores the rows.
template <int KIND, bool ALL, bool COOP = false>
__global__ __launch_bounds__(kBlock) void k_emit(Grid g, Lattice L, double alpha,
                                                 const double *__restrict__ sigma,
                                                 const uint32_t *__restrict__ n_pts,
                                                 const double *__restrict__ calc,
                                                 const unsigned long long *__restrict__ off, OutRows out) {
  const uint64_t gtid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t k = COOP ? gtid >> 6 : gtid;
  const bool writer = !COOP || (threadIdx.x & 63) == 0;
  if (k >= g.K) return;
  unsigned long long pos = off[k];
  const unsigned long long end = off[k + 1];
  if (pos == end) return;
  constexpr bool LAZY = KIND == 4;   // stddev_samp is not in sigma[]: streamed here, written after the walk
  const unsigned long long first = pos;
  double sg = LAZY ? 0.0 : sigma[k];
  double s_cnt = 0.0, s_avg = 0.0, s_m2 = 0.0;
  const bool has_sigma = n_pts[k] >= 2;
  const double one_minus = 1.0 - alpha;
  double e = 0.0;
  // Every scattered 8-byte store is its own write transaction on this memory system (k_emit wrote 280 MB for 81 MB of
  // rows), so rows are buffered four at a time per lane and stored as aligned 16-byte pairs.
  long long bt[4];
  double bx[4], ba[4];
  uint8_t bv[4] = {0, 0, 0, 0};
  int nb = 0;
  auto write_row = [&](unsigned long long at, long long ts, double x, double a, bool verdict) {
    if (!writer) return;
    out.key_id[at] = k;
    out.flow_end_s[at] = ts;
    out.throughput[at] = x;
    out.algo_calc[at] = a;
    if (!LAZY) out.stddev[at] = sg;
    if (ALL) out.anomaly[at] = verdict ? 1 : 0;
  };
  auto point = [&](uint64_t t, uint8_t fl, double x) {
    if (LAZY) {   // Spark CentralMomentAgg, as k_key_sigma (an IEEE division = the bits of div_by_count)
      s_cnt = s_cnt + 1.0;
      const double d = x - s_avg;
      const double dn = d / s_cnt;
      s_avg = s_avg + dn;
      s_m2 = s_m2 + d * (d - dn);
    }
    double a;
    bool verdict;
    if (KIND == 0) {
      e = one_minus * e + alpha * x;
      a = e;
      verdict = has_sigma && fabs(x - e) > sg;
    } else {
      a = KIND == 1 ? calc[t * g.K + k] : (KIND == 3 ? calc[k] : 0.0);
      verdict = (fl & FLAG_ANOMALY) != 0;
    }
    if ((ALL || verdict) && pos + nb < end) {
      const long long ts = g.times != nullptr ? g.times[t * g.K + k] : (long long)(L.t0 + (int64_t)t * L.step);
      if (nb == 0 && (pos & 3ull) != 0) {  // head of the key's segment: single rows up to the next 4-row boundary
        write_row(pos, ts, x, a, verdict);
        pos++;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (nb == i) { bt[i] = ts; bx[i] = x; ba[i] = a; if (ALL) bv[i] = verdict ? 1 : 0; }
        if (++nb == 4 && !writer) { pos += 4; nb = 0; }
        else if (nb == 4) {  // four consecutive rows of every column = 32 aligned bytes: two 16-byte stores instead of four 8-byte ones
          const ulonglong2 kk = make_ulonglong2(k, k);
          reinterpret_cast<ulonglong2 *>(out.key_id + pos)[0] = kk;
          reinterpret_cast<ulonglong2 *>(out.key_id + pos)[1] = kk;
          reinterpret_cast<longlong2 *>(out.flow_end_s + pos)[0] = make_longlong2(bt[0], bt[1]);
          reinterpret_cast<longlong2 *>(out.flow_end_s + pos)[1] = make_longlong2(bt[2], bt[3]);
          reinterpret_cast<double2 *>(out.throughput + pos)[0] = make_double2(bx[0], bx[1]);
          reinterpret_cast<double2 *>(out.throughput + pos)[1] = make_double2(bx[2], bx[3]);
          reinterpret_cast<double2 *>(out.algo_calc + pos)[0] = make_double2(ba[0], ba[1]);
          reinterpret_cast<double2 *>(out.algo_calc + pos)[1] = make_double2(ba[2], ba[3]);
          if (!LAZY) {
            const double2 ss = make_double2(sg, sg);
            reinterpret_cast<double2 *>(out.stddev + pos)[0] = ss;
            reinterpret_cast<double2 *>(out.stddev + pos)[1] = ss;
          }
          if (ALL) *reinterpret_cast<uint32_t *>(out.anomaly + pos) = (uint32_t)bv[0] | ((uint32_t)bv[1] << 8) | ((uint32_t)bv[2] << 16) | ((uint32_t)bv[3] << 24);
          pos += 4;
          nb = 0;
        }
      }
    }
  };
  if (COOP) walk_series_coop(g, k, [](uint64_t) {}, [&](uint64_t t, uint8_t fl, double x, int) { point(t, fl, x); });
  else walk_series(g, k, [&](uint64_t t, uint8_t fl, unsigned long long raw) { if (fl & FLAG_PRESENT) point(t, fl, (double)raw); });
#pragma unroll
  for (int i = 0; i < 3; ++i)   // tail of the segment: fewer than four buffered rows
    if (i < nb) write_row(pos + i, bt[i], bx[i], ba[i], ALL && bv[i] != 0);
  if (LAZY) {
    sg = has_sigma ? sqrt(s_m2 / (s_cnt - 1.0)) : 0.0;
    if (COOP) { for (unsigned long long at = first + (threadIdx.x & 63); at < end; at += 64) out.stddev[at] = sg; }
    else { for (unsigned long long at = first; at < end; ++at) out.stddev[at] = sg; }
  }
}

// k_emit_staged — the EWMA job's emit (KIND 0, anomalous rows only) with COALESCED row stores.
// k_emit's lanes each store their own key's rows: ~5 lanes of a wavefront per time step, every one a separate partial-line
// write (and, on gfx9, stores share vmcnt with the walk's prefetched loads).  Four lanes per key did not help (§3 table):
// the kernel is bound by those write transactions.  Here one wavefront = 64 consecutive keys, whose rows are ONE contiguous
// range [off[k0], off[k0 + 64]) of every output column: during the walk a lane parks (e_t, lane, t) of each anomalous point
// in LDS at its row's position inside that range (12 B per row), and afterwards the wavefront writes the range row by row —
// five fully coalesced stores per 64 rows; throughput is re-read from the grid cell the marker names (an L2 / MALL hit: the
// wavefront has just walked it) and sigma comes from a 64-entry LDS table.  Rows past the LDS capacity (a wavefront with
// far more anomalies than usual) are stored directly, as k_emit does, so any capacity >= 0 is correct.
static constexpr int kStageMarkTBits = 26;   // marker = lane << 26 | t
__global__ __launch_bounds__(64) void k_emit_staged(Grid g, Lattice L, double alpha, const double *__restrict__ sigma,
                                                    const uint32_t *__restrict__ n_pts,
                                                    const unsigned long long *__restrict__ off, OutRows out, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_stage[];
  double *s_e = reinterpret_cast<double *>(smem_stage);                       // [cap]
  double *s_sg = s_e + cap;                                                   // [64]
  uint32_t *s_m = reinterpret_cast<uint32_t *>(s_sg + 64);                    // [cap]
  const uint32_t lane = threadIdx.x;
  const uint64_t k0 = (uint64_t)blockIdx.x * 64;
  const uint64_t k = k0 + lane;
  const uint64_t kend = k0 + 64 < g.K ? k0 + 64 : g.K;
  const unsigned long long base = off[k0];
  const unsigned long long total = off[kend] - base;
  if (total == 0) return;   // uniform over the wavefront
  const bool live = k < g.K;
  unsigned long long pos = live ? off[k] - base : 0;
  const unsigned long long end = live ? off[k + 1] - base : 0;
  const double sg = live ? sigma[k] : 0.0;
  s_sg[lane] = sg;
  if (pos != end) {   // n_anom > 0 implies a defined sigma (k_key_sigma counts nothing otherwise)
    const double one_minus = 1.0 - alpha;
    double e = 0.0;
    walk_series(g, k, [&](uint64_t t, uint8_t fl, unsigned long long raw) {
      if (!(fl & FLAG_PRESENT)) return;
      const double x = (double)raw;
      e = one_minus * e + alpha * x;
      if (fabs(x - e) > sg && pos < end) {
        if (pos < cap) {
          s_e[pos] = e;
          s_m[pos] = (lane << kStageMarkTBits) | (uint32_t)t;
        } else {
          const unsigned long long at = base + pos;
          out.key_id[at] = k;
          out.flow_end_s[at] = g.times != nullptr ? g.times[t * g.K + k] : (long long)(L.t0 + (int64_t)t * L.step);
          out.throughput[at] = x;
          out.algo_calc[at] = e;
          out.stddev[at] = sg;
        }
        pos++;
      }
    });
  }
  __syncthreads();
  const unsigned long long staged = total < cap ? total : cap;
  for (unsigned long long r = lane; r < staged; r += 64) {
    const uint32_t m = s_m[r];
    const uint32_t ln = m >> kStageMarkTBits;
    const uint64_t t = m & ((1u << kStageMarkTBits) - 1u);
    const uint64_t cell = t * g.K + (k0 + ln);
    const unsigned long long at = base + r;
    out.key_id[at] = k0 + ln;
    out.flow_end_s[at] = g.times != nullptr ? g.times[cell] : (long long)(L.t0 + (int64_t)t * L.step);
    out.throughput[at] = (double)g.val[cell];
    out.algo_calc[at] = s_e[r];
    out.stddev[at] = s_sg[ln];
  }
}

// LDS rows per wavefront.  The mean row count of a 64-key range plus 1/8 plus 128 rows (C2: 1290 -> 1600 rows, 19.3 KB:
// eight wavefronts per CU, so that its 1563 wavefronts are resident together; 2048 rows = six per CU put 27 of them into a
// second round: 254 vs 223 us for detect + emit).  tad_plan: ewma_emit = 1 -> k_emit; ewma_emit_rows pins the capacity (tests).
static uint32_t emit_stage_rows(uint64_t K, uint64_t rows_hint, int ewma_emit, uint32_t ewma_emit_rows) {
  if (ewma_emit == 1) return 0;
  uint64_t cap = 1536;
  if (rows_hint != 0) {
    const uint64_t mean = (rows_hint * 64 + K - 1) / K;
    cap = mean + mean / 8 + 128;
  }
  if (ewma_emit_rows != 0) cap = ewma_emit_rows;
  cap = (cap + 63) & ~63ull;
  if (cap < 64) cap = 64;
  if (cap > 4096) cap = 4096;
  return (uint32_t)cap;
}

void launch_emit(hipStream_t s, Grid g, Lattice lat, int kind, bool all_points, double alpha,
                 const double *sigma, const uint32_t *n_pts, const double *calc,
                 const unsigned long long *off, OutRows out, uint64_t rows_hint, int ewma_emit, uint32_t ewma_emit_rows) {
  if (g.K == 0) return;
  if (coop_shape(g)) {   // long series on few keys: a wavefront per key (no staged variant: the rows of a key are written by one lane in time order)
    const int cblocks = (int)((g.K * 64 + kBlock - 1) / kBlock);
#define TAD_LAUNCH_EMIT_C(KIND, ALL) \
  hipLaunchKernelGGL((k_emit<KIND, ALL, true>), dim3(cblocks), dim3(kBlock), 0, s, g, lat, alpha, sigma, n_pts, calc, off, out)
    if (all_points) {
      if (kind == 0) TAD_LAUNCH_EMIT_C(0, true); else if (kind == 1) TAD_LAUNCH_EMIT_C(1, true); else if (kind == 3) TAD_LAUNCH_EMIT_C(3, true); else TAD_LAUNCH_EMIT_C(2, true);
    } else {
      if (kind == 0) TAD_LAUNCH_EMIT_C(0, false); else if (kind == 1) TAD_LAUNCH_EMIT_C(1, false); else if (kind == 3) TAD_LAUNCH_EMIT_C(3, false);
      else if (kind == 4) TAD_LAUNCH_EMIT_C(4, false); else TAD_LAUNCH_EMIT_C(2, false);
    }
#undef TAD_LAUNCH_EMIT_C
    return;
  }
  if (kind == 0 && !all_points && g.T < (1ull << kStageMarkTBits)) {
    if (const uint32_t cap = emit_stage_rows(g.K, rows_hint, ewma_emit, ewma_emit_rows)) {
      const unsigned blocks64 = (unsigned)((g.K + 63) / 64);
      hipLaunchKernelGGL(k_emit_staged, dim3(blocks64), dim3(64), (size_t)cap * 12 + 64 * 8, s, g, lat, alpha, sigma, n_pts, off, out, cap);
      return;
    }
  }
  const int blocks = (int)((g.K + kBlock - 1) / kBlock);
#define TAD_LAUNCH_EMIT(KIND, ALL) \
  hipLaunchKernelGGL((k_emit<KIND, ALL>), dim3(blocks), dim3(kBlock), 0, s, g, lat, alpha, sigma, n_pts, calc, off, out)
  if (all_points) {
    if (kind == 0) TAD_LAUNCH_EMIT(0, true); else if (kind == 1) TAD_LAUNCH_EMIT(1, true); else if (kind == 3) TAD_LAUNCH_EMIT(3, true); else TAD_LAUNCH_EMIT(2, true);
  } else {
    if (kind == 0) TAD_LAUNCH_EMIT(0, false); else if (kind == 1) TAD_LAUNCH_EMIT(1, false); else if (kind == 3) TAD_LAUNCH_EMIT(3, false);
    else if (kind == 4) TAD_LAUNCH_EMIT(4, false); else TAD_LAUNCH_EMIT(2, false);
  }
#undef TAD_LAUNCH_EMIT
}

// every present point with its raw UInt64 aggregate, in (key, time) order (tad_aggregate)
__global__ __launch_bounds__(kBlock) void k_emit_points(Grid g, Lattice L, const unsigned long long *__restrict__ off,
                                                        unsigned long long *__restrict__ out_key, long long *__restrict__ out_t,
                                                        unsigned long long *__restrict__ out_val) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= g.K) return;
  unsigned long long pos = off[k];
  const unsigned long long end = off[k + 1];
  auto step = [&](uint64_t t, uint8_t fl, unsigned long long raw) {
    if ((fl & FLAG_PRESENT) && pos < end) {
      out_key[pos] = k;
      out_t[pos] = g.times != nullptr ? g.times[t * g.K + k] : (long long)(L.t0 + (int64_t)t * L.step);
      out_val[pos] = raw;
      pos++;
    }
  };
  uint64_t t = 0;
  for (; t + kUnroll <= g.T && pos < end; t += kUnroll) {
    uint8_t fl[kUnroll];
    unsigned long long v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) { fl[u] = g.flag[(t + u) * g.K + k]; v[u] = g.val[(t + u) * g.K + k]; }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) step(t + u, fl[u], v[u]);
  }
  for (; t < g.T && pos < end; ++t) step(t, g.flag[t * g.K + k], g.val[t * g.K + k]);
}

void launch_emit_points(hipStream_t s, Grid g, Lattice lat, const unsigned long long *off, unsigned long long *out_key,
                        long long *out_t, unsigned long long *out_val) {
  if (g.K == 0) return;
  const int blocks = (int)((g.K + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_emit_points, dim3(blocks), dim3(kBlock), 0, s, g, lat, off, out_key, out_t, out_val);
}

// ------------------------------------------------------------------------------------------------
// k_stream — streaming EWMA (include/tad.h: tad_run_stream).  One lane = one key: continue Spark's moment update and the
// EWMA recurrence from the key's stored state over its new points; verdict against the RUNNING stddev_samp.
// EMIT = false counts (and produces the next state), EMIT = true replays from the old state and writes the rows.
// The division stays a division here (n grows without bound, no reciprocal table); it rounds like div_by_count.
// ------------------------------------------------------------------------------------------------
template <bool EMIT, bool ALL>
__global__ __launch_bounds__(kBlock) void k_stream(Grid g, Lattice L, double alpha, StreamState cur, StreamState next,
                                                   uint32_t *__restrict__ n_anom, const unsigned long long *__restrict__ off,
                                                   OutRows out, DevCounters *ctr) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned long long my_pts = 0;
  unsigned my_key = 0;
  uint32_t err = 0;
  if (k < g.K) {
    uint32_t n = cur.n[k];
    double cnt = (double)n, avg = cur.avg[k], m2 = cur.m2[k], e = cur.ewma[k];
    long long last_t = cur.last_t[k];
    bool seen = cur.seen[k] != 0;
    const double one_minus = 1.0 - alpha;
    unsigned long long pos = EMIT ? off[k] : 0ull;
    uint32_t a = 0, fresh = 0;
    walk_series(g, k, [&](uint64_t t, uint8_t fl, unsigned long long raw) {
      if (!(fl & FLAG_PRESENT)) return;
      const long long ts = (long long)(L.t0 + (int64_t)t * L.step);
      if (seen && ts <= last_t) { err |= DEV_ERR_LATE_ROW; return; }
      const double x = (double)raw;
      cnt = cnt + 1.0;
      n++;
      const double d = x - avg;
      const double dn = d / cnt;
      avg = avg + dn;
      m2 = m2 + d * (d - dn);
      e = one_minus * e + alpha * x;
      const bool has_sigma = n >= 2;
      const double sg = has_sigma ? sqrt(m2 / (cnt - 1.0)) : 0.0;
      const bool verdict = has_sigma && fabs(x - e) > sg;
      last_t = ts;
      seen = true;
      fresh++;
      if (ALL || verdict) {
        if (EMIT) {
          out.key_id[pos] = k;
          out.flow_end_s[pos] = ts;
          out.throughput[pos] = x;
          out.algo_calc[pos] = e;
          out.stddev[pos] = sg;
          if (ALL) out.anomaly[pos] = verdict ? 1 : 0;
          pos++;
        }
        a++;
      }
    });
    if (!EMIT) {
      next.n[k] = n; next.avg[k] = avg; next.m2[k] = m2; next.ewma[k] = e; next.last_t[k] = last_t; next.seen[k] = seen ? 1 : 0;
      n_anom[k] = a;
      my_pts = fresh;
      my_key = fresh > 0;
    }
  }
  if (!EMIT) {
    for (int d = 32; d >= 1; d >>= 1) {
      my_pts += __shfl_down(my_pts, d);
      my_key += __shfl_down(my_key, d);
      err |= __shfl_down(err, d);
    }
    if ((threadIdx.x & 63) == 0) {
      if (my_key) { atomicAdd(&ctr->n_points, my_pts); atomicAdd(&ctr->n_keys, (unsigned long long)my_key); }
      if (err) atomicOr(&ctr->err, err);
    }
  }
}

void launch_stream(hipStream_t s, Grid g, Lattice lat, double alpha, bool all_points, bool emit, StreamState cur, StreamState next,
                   uint32_t *n_anom, const unsigned long long *off, OutRows out, DevCounters *ctr) {
  if (g.K == 0) return;
  const int blocks = (int)((g.K + kBlock - 1) / kBlock);
#define TAD_STREAM(E, A) hipLaunchKernelGGL((k_stream<E, A>), dim3(blocks), dim3(kBlock), 0, s, g, lat, alpha, cur, next, n_anom, off, out, ctr)
  if (emit) { if (all_points) TAD_STREAM(true, true); else TAD_STREAM(true, false); }
  else { if (all_points) TAD_STREAM(false, true); else TAD_STREAM(false, false); }
#undef TAD_STREAM
}

// EWMA value of every present point (tad_series_ewma = calculate_ewma, :146-165)
__global__ __launch_bounds__(kBlock) void k_ewma_values(Grid g, double alpha, double *__restrict__ calc) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= g.K) return;
  const double one_minus = 1.0 - alpha;
  double e = 0.0;
  for (uint64_t t = 0; t < g.T; ++t) {
    const uint64_t c = t * g.K + k;
    if (g.flag[c] & FLAG_PRESENT) {
      e = one_minus * e + alpha * (double)g.val[c];
      calc[c] = e;
    }
  }
}

void launch_ewma_values(hipStream_t s, Grid g, double alpha, double *calc) {
  if (g.K == 0) return;
  const int blocks = (int)((g.K + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_ewma_values, dim3(blocks), dim3(kBlock), 0, s, g, alpha, calc);
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_kernels() { return reinterpret_cast<const void *>(&k_meta); }

}  // namespace tad
