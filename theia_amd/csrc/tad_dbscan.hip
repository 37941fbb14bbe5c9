// tad_dbscan.hip — DBSCAN verdicts per key (anomaly_detection.py:325-349):
//   DBSCAN(min_samples=4, eps=250000000).fit_predict(x.reshape(-1, 1)) == -1
// on the 1-D values of ONE key's series.  Cluster ids are discarded by the reference (:344-348), so
// only the noise predicate is needed (checked against sklearn in the oracle tests):
//   core(i)  <=>  #{ j : |x_i - x_j| <= eps } >= min_samples        (self counted, inclusive <=)
//   noise(i) <=>  !core(i)  and  no core j with |x_i - x_j| <= eps
// Distances are FP64 on correctly rounded uint64 -> double values: pure comparisons, so verdicts
// are exact.
//
// Two kernels: k_dbscan_scan (one lane per key, one coalesced walk over the time-major grid) settles every key whose
// values all lie within eps of each other and lists the rest; the exact pair tests for the listed keys run with one
// wavefront per key (k_dbscan_list_wave, series of up to 256 buckets held in registers) or one workgroup per key
// (k_dbscan_list: points compacted into LDS, lanes = points i, LDS-broadcast x_j stream) for longer series.
#include <cstdlib>

#include "tad_internal.h"

namespace tad {

static constexpr int kDbBlock = 256;
static constexpr int kDbWaves = kDbBlock / 64;

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63; }

// ------------------------------------------------------------------------------------------------
// k_dbscan_scan — one lane = one key, one coalesced walk over its column of the time-major grid: point count, min / max
// and shifted moments of the values.  If max - min <= eps every pair of the key's points is within eps (|x_i - x_j| <=
// max - min, and FP subtraction is monotone): with >= min_samples points all are core (no noise), which settles the key
// here — the common case by far (a flow's throughput rarely spreads over more than eps = 250 MB/s).  Every other key
// (wide spread, or fewer than min_samples points) goes to a work list for the exact pair tests (k_dbscan_list).
// st (optional): per-key n_pts, n_anom (0 here; k_dbscan_list overwrites its keys) and (mean, M2) for the job telemetry
// (M2 from sums shifted by the key's first value; only the exact stddev_samp column of emitted rows follows Spark's
// streaming order, k_emit<4>).
// ------------------------------------------------------------------------------------------------
// REDO_ONLY: Stage 0's pass C ran in settle mode and has done this kernel's work for every key it could see whole (SettleArgs,
// tad_stage0_part.hip); only keys it marked kSettleRedo (split partitions, overflow-list records) are walked here.
template <bool REDO_ONLY>
__global__ __launch_bounds__(kDbBlock) void k_dbscan_scan(Grid g, double eps, int min_samples, DbscanStats st,
                                                         uint32_t *__restrict__ list, unsigned int *__restrict__ count) {
  const uint64_t k = (uint64_t)blockIdx.x * kDbBlock + threadIdx.x;
  bool slow = false;
  if (k < g.K && (!REDO_ONLY || st.n_pts[k] == kSettleRedo)) {
    uint32_t n = 0;
    double mn = 0.0, mx = 0.0, x0 = 0.0, s1 = 0.0, s2 = 0.0;
    walk_series(g, k, [&](uint64_t, uint8_t fl, unsigned long long raw) {
      if (fl & FLAG_PRESENT) {
        const double x = (double)raw;
        if (n == 0) { mn = x; mx = x; x0 = x; }
        mn = fmin(mn, x);
        mx = fmax(mx, x);
        const double d = x - x0;
        s1 += d;
        s2 += d * d;
        n++;
      }
    });
    slow = n > 0 && (!(mx - mn <= eps) || n < (uint32_t)min_samples);
    if (st.n_pts != nullptr) {
      st.n_pts[k] = n;
      st.n_anom[k] = 0;
      const double dn = (double)(n ? n : 1);
      st.key_mean[k] = n ? x0 + s1 / dn : 0.0;
      st.key_m2[k] = n ? fmax(s2 - s1 * (s1 / dn), 0.0) : 0.0;
    }
  }
  const unsigned long long m = __ballot(slow);
  if (m) {
    const unsigned lane = lane_id();
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(m));
    base = __shfl(base, 0);
    if (slow) list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)k;
  }
}

// ------------------------------------------------------------------------------------------------
// k_dbscan_list — the exact noise predicate for the listed keys: one workgroup per key (grid-stride over the list), the
// key's present points compacted into LDS, pair tests with lanes = points i and an LDS-broadcast x_j stream.
// LDS carve: xs[T] f64 | ct[T] u32 | core[T] u8
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kDbBlock) void k_dbscan_list(Grid g, double eps, int min_samples, const uint32_t *__restrict__ list,
                                                         const unsigned int *__restrict__ count, uint32_t *__restrict__ n_anom) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t T = g.T;
  double *xs = reinterpret_cast<double *>(smem);
  uint32_t *ct = reinterpret_cast<uint32_t *>(smem + T * 8);
  uint8_t *core = reinterpret_cast<uint8_t *>(ct + T);
  __shared__ uint32_t s_wave[kDbWaves];
  __shared__ uint32_t s_n, s_noise;
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const unsigned total = *count;
  for (unsigned e = blockIdx.x; e < total; e += gridDim.x) {
    const uint64_t k = list[e];
    if (threadIdx.x == 0) { s_n = 0; s_noise = 0; }
    __syncthreads();
    for (uint64_t c0 = 0; c0 < T; c0 += kDbBlock) {
      const uint64_t t = c0 + threadIdx.x;
      const bool p = t < T && (g.flag[t * g.K + k] & FLAG_PRESENT);
      const unsigned long long m = __ballot(p);
      if (lane == 0) s_wave[wave] = __popcll(m);
      __syncthreads();
      uint32_t base = s_n;
      for (int w = 0; w < wave; ++w) base += s_wave[w];
      if (p) {
        const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
        xs[pos] = (double)g.val[t * g.K + k];
        ct[pos] = (uint32_t)t;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kDbWaves; ++w) tot += s_wave[w];
        s_n += tot;
      }
      __syncthreads();
    }
    const uint32_t n = s_n;
    for (uint32_t i = threadIdx.x; i < n; i += kDbBlock) {
      const double xi = xs[i];
      int cnt = 0;
#pragma unroll 4
      for (uint32_t j = 0; j < n; ++j) cnt += fabs(xi - xs[j]) <= eps ? 1 : 0;
      core[i] = cnt >= min_samples ? 1 : 0;
    }
    __syncthreads();
    uint32_t noise = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kDbBlock) {
      if (core[i]) continue;
      const double xi = xs[i];
      bool reach = false;
      for (uint32_t j = 0; j < n && !reach; ++j) reach = core[j] && fabs(xi - xs[j]) <= eps;
      if (!reach) { g.flag[(uint64_t)ct[i] * g.K + k] = FLAG_PRESENT | FLAG_ANOMALY; noise++; }
    }
    if (noise) atomicAdd(&s_noise, noise);
    __syncthreads();
    if (threadIdx.x == 0 && n_anom != nullptr) n_anom[k] = s_noise;
    __syncthreads();
  }
}

// k_dbscan_list_wave (series of up to 256 buckets; round 3: C4 detect + emit 0.48 -> 0.41 ms against k_dbscan_list) — the same
// predicate with ONE WAVEFRONT per listed key and no LDS, no workgroup barrier: lane l holds the key's buckets l, l + 64, ... (PPL of them: T <= 64 * PPL);
// x_j reaches all lanes by a readlane with a wavefront-uniform index, absent buckets are skipped through the ballot masks.
// k_dbscan_list spends a workgroup and ~8 barriers on a key of ~100 points (C4: ~7 listed keys per workgroup, 97 us);
// here four keys are in flight per workgroup and nothing waits for anything.
template <int PPL>
__global__ __launch_bounds__(kDbBlock) void k_dbscan_list_wave(Grid g, double eps, int min_samples, const uint32_t *__restrict__ list,
                                                              const unsigned int *__restrict__ count, uint32_t *__restrict__ n_anom,
                                                              double *__restrict__ sg_e, unsigned long long *__restrict__ am_e) {
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x >> 6;
  const unsigned total = *count;
  for (unsigned e = blockIdx.x * kDbWaves + wave; e < total; e += gridDim.x * kDbWaves) {   // wavefront-uniform
    const uint64_t k = list[e];
    double x[PPL];
    bool p[PPL];
    unsigned long long pm[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const uint64_t t = lane + 64u * (unsigned)j;
      p[j] = t < g.T && (g.flag[t * g.K + k] & FLAG_PRESENT);
      x[j] = p[j] ? (double)g.val[t * g.K + k] : 0.0;
      pm[j] = __ballot(p[j]);
    }
    int cnt[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) cnt[i] = 0;
#pragma unroll
    for (int jj = 0; jj < PPL; ++jj)
      for (unsigned long long m = pm[jj]; m; m &= m - 1) {
        const double xj = __shfl(x[jj], __ffsll((long long)m) - 1);
#pragma unroll
        for (int i = 0; i < PPL; ++i) cnt[i] += fabs(x[i] - xj) <= eps ? 1 : 0;
      }
    bool core[PPL], reach[PPL];
    unsigned long long cm[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      core[i] = p[i] && cnt[i] >= min_samples;
      reach[i] = false;
      cm[i] = __ballot(core[i]);
    }
#pragma unroll
    for (int jj = 0; jj < PPL; ++jj)
      for (unsigned long long m = cm[jj]; m; m &= m - 1) {
        const double xj = __shfl(x[jj], __ffsll((long long)m) - 1);
#pragma unroll
        for (int i = 0; i < PPL; ++i) reach[i] = reach[i] || fabs(x[i] - xj) <= eps;
      }
    uint32_t noise = 0;
    unsigned long long am[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const bool nz = p[i] && !core[i] && !reach[i];
      if (nz) g.flag[(uint64_t)(lane + 64u * (unsigned)i) * g.K + k] = FLAG_PRESENT | FLAG_ANOMALY;
      am[i] = __ballot(nz);
      noise += (uint32_t)__popcll(am[i]);
    }
    if (lane == 0 && n_anom != nullptr) n_anom[k] = noise;
    // What the job's emit needs of a key with noise points, left next to the list entry: its noise masks and its stddev_samp
    // (Spark CentralMomentAgg in time order: an IEEE division = the bits of div_by_count; every lane computes the same
    // recurrence over the present points by readlane) — the series is in registers HERE; k_emit_dbscan_wave used to fetch the
    // key's whole column (one 64-byte sector per bucket for 8 useful bytes: 238 MB at C4) a second time for these two things.
    if (sg_e != nullptr && noise) {   // wavefront-uniform
      double cn = 0.0, avg = 0.0, m2 = 0.0;
#pragma unroll
      for (int jj = 0; jj < PPL; ++jj)
        for (unsigned long long m = pm[jj]; m; m &= m - 1) {
          const double xv = __shfl(x[jj], __ffsll((long long)m) - 1);
          cn = cn + 1.0;
          const double d = xv - avg;
          const double dn = d / cn;
          avg = avg + dn;
          m2 = m2 + d * (d - dn);
        }
      if (lane == 0) {
        sg_e[e] = cn >= 2.0 ? sqrt(m2 / (cn - 1.0)) : 0.0;
#pragma unroll
        for (int i = 0; i < PPL; ++i) am_e[(size_t)e * PPL + i] = am[i];
      }
    }
  }
}

// k_emit_dbscan_wave — the DBSCAN job's emit from the detector's WORK LIST instead of a walk over all keys: only listed keys can
// have noise points, and there are few of them (C4: 1.8 % of the keys, 14 488 rows).  One wavefront per listed key with noise:
// the noise masks and the key's stddev_samp come from the list entry (k_dbscan_list_wave), the lanes of the noise points fetch
// their value from the grid; the rows of the key go to off[k] + rank in time order.
template <int PPL>
__global__ __launch_bounds__(kDbBlock) void k_emit_dbscan_wave(Grid g, Lattice L, const uint32_t *__restrict__ list,
                                                              const unsigned int *__restrict__ count, const uint32_t *__restrict__ n_anom,
                                                              const double *__restrict__ sg_e, const unsigned long long *__restrict__ am_e,
                                                              const unsigned long long *__restrict__ off, OutRows out) {
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x >> 6;
  const unsigned total = *count;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (unsigned e = blockIdx.x * kDbWaves + wave; e < total; e += gridDim.x * kDbWaves) {   // wavefront-uniform
    const uint64_t k = list[e];
    if (n_anom[k] == 0) continue;
    const double sg = sg_e[e];              // k_dbscan_list_wave left both for every listed key with noise points
    unsigned long long at = off[k];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const unsigned long long am = am_e[(size_t)e * PPL + j];
      if ((am >> lane) & 1ull) {
        const uint64_t t = lane + 64u * (unsigned)j;
        const unsigned long long r = at + (unsigned long long)__popcll(am & lt_mask);
        out.key_id[r] = k;
        out.flow_end_s[r] = g.times != nullptr ? g.times[t * g.K + k] : (long long)(L.t0 + (int64_t)t * L.step);
        out.throughput[r] = (double)g.val[t * g.K + k];
        out.algo_calc[r] = 0.0;
        out.stddev[r] = sg;
      }
      at += (unsigned long long)__popcll(am);
    }
  }
}

// Fallback for series too long for an LDS row: one workgroup per key, points compacted into a global
// scratch row, x_j streamed from L1/L2 (every lane reads the same address -> one fetch per wave).
__global__ __launch_bounds__(kDbBlock) void k_dbscan_long(Grid g, double eps, int min_samples,
                                                          double *__restrict__ scratch_x,
                                                          uint32_t *__restrict__ scratch_t,
                                                          uint8_t *__restrict__ scratch_core) {
  const uint64_t k = blockIdx.x;
  const uint64_t T = g.T;
  double *xs = scratch_x + k * T;
  uint32_t *ct = scratch_t + k * T;
  uint8_t *core = scratch_core + k * T;
  __shared__ uint32_t s_wave[kDbWaves];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  for (uint64_t c0 = 0; c0 < T; c0 += kDbBlock) {
    const uint64_t t = c0 + threadIdx.x;
    const bool p = t < T && (g.flag[t * g.K + k] & FLAG_PRESENT);
    const unsigned long long m = __ballot(p);
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    uint32_t base = s_n;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    if (p) {
      const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
      xs[pos] = (double)g.val[t * g.K + k];
      ct[pos] = (uint32_t)t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (int w = 0; w < kDbWaves; ++w) tot += s_wave[w];
      s_n += tot;
    }
    __syncthreads();
  }
  const uint32_t n = s_n;
  __threadfence_block();
  for (uint32_t i = threadIdx.x; i < n; i += kDbBlock) {
    const double xi = xs[i];
    int cnt = 0;
    for (uint32_t j = 0; j < n; ++j) cnt += fabs(xi - xs[j]) <= eps ? 1 : 0;
    core[i] = cnt >= min_samples ? 1 : 0;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += kDbBlock) {
    if (core[i]) continue;
    const double xi = xs[i];
    bool reach = false;
    for (uint32_t j = 0; j < n && !reach; ++j) reach = core[j] && fabs(xi - xs[j]) <= eps;
    if (!reach) g.flag[(uint64_t)ct[i] * g.K + k] = FLAG_PRESENT | FLAG_ANOMALY;
  }
}

static bool list_fits_lds(uint64_t T) { return T * 13 + 64 <= 150 * 1024; }

// bytes of device scratch the DBSCAN launch needs: the work list (4 B per key + a counter) when a series fits an LDS row,
// else the global rows of the long-series kernel
size_t dbscan_scratch_bytes(Grid g) {
  if (g.T <= 256) return 64 + (((size_t)g.K * 4 + 63) & ~(size_t)63) + (size_t)g.K * (8 + 8 * 4);   // + per list entry: stddev, noise masks (wave list)
  if (list_fits_lds(g.T)) return (size_t)g.K * 4 + 64;
  return (size_t)g.K * g.T * (8 + 4 + 1);
}

// scratch of the wave list (T <= 256): count | list[K] u32 | sg[K] f64 | am[K * 4] u64
static double *wave_list_sg(const void *scratch, Grid g) {
  return reinterpret_cast<double *>(const_cast<unsigned char *>(static_cast<const unsigned char *>(scratch)) + 64 + (((size_t)g.K * 4 + 63) & ~(size_t)63));
}
static unsigned long long *wave_list_am(const void *scratch, Grid g) { return reinterpret_cast<unsigned long long *>(wave_list_sg(scratch, g) + g.K); }

bool dbscan_uses_list(Grid g) { return list_fits_lds(g.T); }

int launch_dbscan(hipStream_t s, Grid g, double eps, int min_samples, void *scratch, DbscanStats st, bool settled_by_stage0) {
  if (g.K == 0 || g.T == 0) return 0;
  if (!list_fits_lds(g.T)) return -1;
  unsigned int *count = static_cast<unsigned int *>(scratch);
  uint32_t *list = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(scratch) + 64);
  if (settled_by_stage0) {   // the list was started by pass C (its counter zeroed before Stage 0)
    hipLaunchKernelGGL(k_dbscan_scan<true>, dim3((unsigned)((g.K + kDbBlock - 1) / kDbBlock)), dim3(kDbBlock), 0, s, g, eps, min_samples, st, list, count);
  } else {
    hipMemsetAsync(count, 0, sizeof(unsigned int), s);
    hipLaunchKernelGGL(k_dbscan_scan<false>, dim3((unsigned)((g.K + kDbBlock - 1) / kDbBlock)), dim3(kDbBlock), 0, s, g, eps, min_samples, st, list, count);
  }
  // grid-stride over the (device-side) list length: 8192 workgroups of four wavefronts give every one of C4's ~2e4 listed keys its
  // own wavefront (2048: 2-3 keys per wavefront one after the other; detect + emit 0.179 -> 0.174 ms, 32768 the same:
  // profiles/r3_v9_c4_list_blocks_ab.log)
  uint64_t blocks = g.K < 8192 ? g.K : 8192;
  if (g.T <= 256) {   // a wavefront's registers hold the whole series
#define TAD_DBW(PPL) hipLaunchKernelGGL((k_dbscan_list_wave<PPL>), dim3((unsigned)blocks), dim3(kDbBlock), 0, s, g, eps, min_samples, list, count, st.n_anom, wave_list_sg(scratch, g), wave_list_am(scratch, g))
    if (g.T <= 64) TAD_DBW(1); else if (g.T <= 128) TAD_DBW(2); else if (g.T <= 192) TAD_DBW(3); else TAD_DBW(4);
#undef TAD_DBW
    return 0;
  }
  const size_t lds = (size_t)((g.T * 13 + 15) & ~(uint64_t)15);
  allow_big_lds(reinterpret_cast<const void *>(k_dbscan_list), 152 * 1024);
  hipLaunchKernelGGL(k_dbscan_list, dim3((unsigned)blocks), dim3(kDbBlock), lds, s, g, eps, min_samples, list, count, st.n_anom);
  return 0;
}

// the DBSCAN job's emit from the work list launch_dbscan left in `scratch` (false: shape not supported, use launch_emit kind 4)
bool launch_emit_dbscan_list(hipStream_t s, Grid g, Lattice lat, const void *scratch, const uint32_t *n_anom, const unsigned long long *off,
                             OutRows out) {
  if (g.K == 0 || g.T == 0 || g.T > 256 || !list_fits_lds(g.T)) return false;
  const unsigned int *count = static_cast<const unsigned int *>(scratch);
  const uint32_t *list = reinterpret_cast<const uint32_t *>(static_cast<const unsigned char *>(scratch) + 64);
  const uint64_t blocks = g.K < 2048 ? g.K : 2048;
#define TAD_DBE(PPL) hipLaunchKernelGGL((k_emit_dbscan_wave<PPL>), dim3((unsigned)blocks), dim3(kDbBlock), 0, s, g, lat, list, count, n_anom, wave_list_sg(scratch, g), wave_list_am(scratch, g), off, out)
  if (g.T <= 64) TAD_DBE(1); else if (g.T <= 128) TAD_DBE(2); else if (g.T <= 192) TAD_DBE(3); else TAD_DBE(4);
#undef TAD_DBE
  return true;
}

int launch_dbscan_long(hipStream_t s, Grid g, double eps, int min_samples, void *scratch) {
  unsigned char *p = static_cast<unsigned char *>(scratch);
  double *sx = reinterpret_cast<double *>(p);
  uint32_t *st = reinterpret_cast<uint32_t *>(p + (size_t)g.K * g.T * 8);
  uint8_t *sc = p + (size_t)g.K * g.T * 12;
  hipLaunchKernelGGL(k_dbscan_long, dim3((unsigned)g.K), dim3(kDbBlock), 0, s, g, eps, min_samples, sx, st, sc);
  return 0;
}

}  // namespace tad
