// tad_dbscan.hip — DBSCAN verdicts per key (anomaly_detection.py:325-349):
//   DBSCAN(min_samples=4, eps=250000000).fit_predict(x.reshape(-1, 1)) == -1
// on the 1-D values of ONE key's series.  Cluster ids are discarded by the reference (:344-348), so
// only the noise predicate is needed (checked against sklearn in the oracle tests):
//   core(i)  <=>  #{ j : |x_i - x_j| <= eps } >= min_samples        (self counted, inclusive <=)
//   noise(i) <=>  !core(i)  and  no core j with |x_i - x_j| <= eps
// Distances are FP64 on correctly rounded uint64 -> double values: pure comparisons, so verdicts
// are exact.
//
// Two kernels: k_dbscan_scan (one lane per key, one coalesced walk over the time-major grid; a wavefront per key for long series on
// few keys) settles every key whose values all lie within eps of each other and lists the rest; the exact predicate for the listed
// keys runs with one wavefront per key (k_dbscan_list_wave: series of up to 256 buckets held in registers, pair tests by readlane)
// or, for longer series, one workgroup per key on SORTED values (k_dbscan_sorted: in one dimension the eps-neighbourhood of a point
// is a window of the sorted series, so core / noise are window counts — O(n log n), no pair tests).
#include <type_traits>
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
// (wide spread, or fewer than min_samples points) goes to a work list for the exact predicate (k_dbscan_list_wave / k_dbscan_sorted).
// st (optional): per-key n_pts, n_anom (0 here; the list kernels overwrite their keys) and (mean, M2) for the job telemetry
// (M2 from sums shifted by the key's first value; only the exact stddev_samp column of emitted rows follows Spark's
// streaming order, k_emit<4>).
// ------------------------------------------------------------------------------------------------
// REDO_ONLY: Stage 0's pass C ran in settle mode and has done this kernel's work for every key it could see whole (SettleArgs,
// tad_stage0_part.hip); only keys it marked kSettleRedo (split partitions, overflow-list records) are walked here.
// partial statistics of a set of points, sums shifted by x0 (a value of the set)
struct ScanPart { uint32_t n; double mn, mx, x0, s1, s2; };
// a (its shift is kept) + b: b's values are (its d) + sh relative to a's shift — exact algebra, no division
__device__ __forceinline__ ScanPart scan_merge(const ScanPart &a, const ScanPart &b) {
  if (a.n == 0) return b;
  if (b.n == 0) return a;
  const double sh = b.x0 - a.x0, dnb = (double)b.n;
  ScanPart r;
  r.n = a.n + b.n; r.mn = fmin(a.mn, b.mn); r.mx = fmax(a.mx, b.mx); r.x0 = a.x0;
  r.s1 = a.s1 + (b.s1 + dnb * sh);
  r.s2 = a.s2 + (b.s2 + 2.0 * sh * b.s1 + dnb * (sh * sh));
  return r;
}

// COOP: one wavefront per key (long series on few keys, tad_internal.h:coop_shape): every statistic of the scan is associative, so
// the lanes take the buckets in strides of 64 and merge their partials in a fixed xor tree (lower lane first).
template <bool REDO_ONLY, bool COOP = false>
__global__ __launch_bounds__(kDbBlock) void k_dbscan_scan(Grid g, double eps, int min_samples, DbscanStats st,
                                                         uint32_t *__restrict__ list, unsigned int *__restrict__ count, uint8_t *__restrict__ cs_has,
                                                         uint32_t cs_cap) {
  const uint64_t gtid = (uint64_t)blockIdx.x * kDbBlock + threadIdx.x;
  const uint64_t k = COOP ? gtid >> 6 : gtid;
  bool slow = false;
  if (k < g.K && (!REDO_ONLY || st.n_pts[k] == kSettleRedo)) {
    uint32_t n = 0;
    double mn = 0.0, mx = 0.0, x0 = 0.0, s1 = 0.0, s2 = 0.0;
    auto point = [&](uint8_t fl, unsigned long long raw) {
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
    };
    if (COOP) {
      for (uint64_t t = lane_id(); t < g.T; t += 64) point(g.flag[t * g.K + k], g.val[t * g.K + k]);
      ScanPart a{n, mn, mx, x0, s1, s2};
      for (int d = 1; d < 64; d <<= 1) {
        ScanPart o{__shfl_xor(a.n, d), __shfl_xor(a.mn, d), __shfl_xor(a.mx, d), __shfl_xor(a.x0, d), __shfl_xor(a.s1, d), __shfl_xor(a.s2, d)};
        a = (lane_id() & (unsigned)d) ? scan_merge(o, a) : scan_merge(a, o);   // both partners compute the same value
      }
      n = a.n; mn = a.mn; mx = a.mx; x0 = a.x0; s1 = a.s1; s2 = a.s2;
    } else {
      walk_series(g, k, [&](uint64_t, uint8_t fl, unsigned long long raw) { point(fl, raw); });
    }
    slow = n > 0 && (!(mx - mn <= eps) || n < (uint32_t)min_samples);
    if (st.n_pts != nullptr && (!COOP || lane_id() == 0)) {
      st.n_pts[k] = n;
      st.n_anom[k] = 0;
      const double dn = (double)(n ? n : 1);
      st.key_mean[k] = n ? x0 + s1 / dn : 0.0;
      st.key_m2[k] = n ? fmax(s2 - s1 * (s1 / dn), 0.0) : 0.0;
    }
    if (COOP && lane_id() != 0) slow = false;   // one list entry per key
  }
  const unsigned long long m = __ballot(slow);
  if (m) {
    const unsigned lane = lane_id();
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(m));
    base = __shfl(base, 0);
    if (slow) {
      const unsigned e = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
      list[e] = (uint32_t)k;
      if (cs_has != nullptr && e < cs_cap) cs_has[e] = 0;     // no contiguous series for this entry: the list kernel gathers from the grid
    }
  }
}

// k_dbscan_scan_redo — k_dbscan_scan for the keys of the REDO LIST only (settle mode: the keys the tile pass could not decide), a wavefront
// per key: the lanes take the buckets in strides of 64, partial statistics merged in a fixed xor tree (the COOP form of k_dbscan_scan).
__global__ __launch_bounds__(kDbBlock) void k_dbscan_scan_redo(Grid g, double eps, int min_samples, DbscanStats st, const uint32_t *__restrict__ redo,
                                                              const unsigned int *__restrict__ redo_count, uint32_t *__restrict__ list,
                                                              unsigned int *__restrict__ count, uint8_t *__restrict__ cs_has, uint32_t cs_cap,
                                                              const unsigned long long *__restrict__ rs_val, const uint8_t *__restrict__ rs_flag,
                                                              const uint8_t *__restrict__ rs_has, uint32_t rs_cap, uint32_t *__restrict__ cs_src) {
  // The keys that turn out "slow" join the work list.  Most redo keys do (a value beyond 2^32 next to ordinary ones is a wide spread), and one
  // atomic per key on the list counter is a serial queue of ~12 ns each (144 us for C4's 1e4 redo keys): the workgroup collects its slow keys
  // in LDS and reserves their list slots with ONE atomic.
  constexpr uint32_t kSlowCap = 256;
  __shared__ uint32_t s_slow[kSlowCap], s_slow_e[kSlowCap];
  __shared__ uint32_t s_nslow, s_base;
  if (threadIdx.x == 0) s_nslow = 0;
  __syncthreads();
  const unsigned lane = lane_id();
  const unsigned total = *redo_count;
  for (unsigned e = blockIdx.x * kDbWaves + (threadIdx.x >> 6); e < total; e += gridDim.x * kDbWaves) {   // wavefront-uniform
    const uint64_t k = redo[e];
    ScanPart a{0, 0.0, 0.0, 0.0, 0.0, 0.0};
    // the tile pass left the key's series contiguous behind the redo list (rs_has 1): read it coalesced and fetch from the grid — where the
    // fold has put the real aggregates — only the cells flagged as "on the overflow list"; otherwise gather the whole column
    const bool compact = rs_has != nullptr && e < rs_cap && rs_has[e] == 1;   // wavefront-uniform
    for (uint64_t t = lane; t < g.T; t += 64) {
      uint8_t fl;
      unsigned long long raw;
      if (compact) {
        fl = rs_flag[(size_t)e * g.T + t];
        raw = (fl & 2) ? g.val[t * g.K + k] : rs_val[(size_t)e * g.T + t];
      } else {
        fl = g.flag[t * g.K + k];
        raw = (fl & FLAG_PRESENT) ? g.val[t * g.K + k] : 0ull;
      }
      if (fl & FLAG_PRESENT) {
        const double x = (double)raw;
        if (a.n == 0) { a.mn = x; a.mx = x; a.x0 = x; }
        a.mn = fmin(a.mn, x);
        a.mx = fmax(a.mx, x);
        const double d = x - a.x0;
        a.s1 += d;
        a.s2 += d * d;
        a.n++;
      }
    }
    for (int d = 1; d < 64; d <<= 1) {
      ScanPart o{__shfl_xor(a.n, d), __shfl_xor(a.mn, d), __shfl_xor(a.mx, d), __shfl_xor(a.x0, d), __shfl_xor(a.s1, d), __shfl_xor(a.s2, d)};
      a = (lane & (unsigned)d) ? scan_merge(o, a) : scan_merge(a, o);
    }
    if (lane == 0) {
      const bool slow = a.n > 0 && (!(a.mx - a.mn <= eps) || a.n < (uint32_t)min_samples);
      st.n_pts[k] = a.n;
      st.n_anom[k] = 0;
      const double dn = (double)(a.n ? a.n : 1);
      st.key_mean[k] = a.n ? a.x0 + a.s1 / dn : 0.0;
      st.key_m2[k] = a.n ? fmax(a.s2 - a.s1 * (a.s1 / dn), 0.0) : 0.0;
      if (slow) {
        const uint32_t i = atomicAdd(&s_nslow, 1u);
        if (i < kSlowCap) { s_slow[i] = (uint32_t)k; s_slow_e[i] = compact ? e : 0xFFFFFFFFu; }
        else {   // (more slow keys in one workgroup than the LDS list holds: straight to the list)
          const unsigned at = atomicAdd(count, 1u);
          list[at] = (uint32_t)k;
          if (cs_has != nullptr && at < cs_cap) cs_has[at] = 0;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t ns = s_nslow < kSlowCap ? s_nslow : kSlowCap;
  if (threadIdx.x == 0 && ns) s_base = atomicAdd(count, ns);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ns; i += kDbBlock) {
    list[s_base + i] = s_slow[i];
    if (cs_has != nullptr && s_base + i < cs_cap) {   // the key's series lies contiguous behind the REDO list: the list kernel reads it there (2)
      cs_has[s_base + i] = s_slow_e[i] != 0xFFFFFFFFu ? 2 : 0;
      cs_src[s_base + i] = s_slow_e[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_dbscan_sorted — the exact noise predicate for listed keys with series of more than 256 buckets, on SORTED values.
// |x_i - x_j| <= eps is evaluated as the FP64 difference of the two correctly rounded values (what sklearn's distance computes);
// IEEE subtraction is monotone in each operand, so for a fixed x_i the predicate is true exactly on a WINDOW [lo_i, hi_i] of the
// series sorted by value.  core(i) <=> hi_i - lo_i + 1 >= min_samples (self counted); noise(i) <=> not core and no core point in
// the window (a prefix sum of the core flags over the sorted order).  The window ends are found by binary search with the very
// predicate of the pair test (fl(x_i - x_j) <= eps on the sorted pair), so the verdicts are those of the O(n^2) pair tests, exactly
// — including chains of points exactly eps apart (tests/test_gpu_parity.py, tests/test_gpu_sparse.py).
// One workgroup per listed key (grid-stride over the list): the present points are compacted in time order, sorted by value with a
// bitonic network (padded to a power of two with +inf), windows, core prefix, verdicts scattered back through the points' buckets.
// Working set per key: x f64 | bucket u32 | lo u32 | hi u32 | core prefix u32 = 24 B per point — in LDS for up to kSortLdsPoints
// points, else in a per-workgroup row of global scratch (L2-resident: 32 768 points = 768 KB).
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t kSortLdsPoints = 4096;

struct SortRows {   // per-workgroup rows of global scratch (stride = cap points), or all NULL: LDS
  double *x;
  uint32_t *bk, *lo, *hi, *cp;
  uint32_t cap;
};

__global__ __launch_bounds__(kDbBlock) void k_dbscan_sorted(Grid g, double eps, int min_samples, const uint32_t *__restrict__ list,
                                                           const unsigned int *__restrict__ count, uint32_t *__restrict__ n_anom, SortRows rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t T = g.T;
  double *xs;
  uint32_t *bk, *lo, *hi, *cp;
  uint32_t cap;
  if (rows.x != nullptr) {
    cap = rows.cap;
    xs = rows.x + (size_t)blockIdx.x * cap;
    bk = rows.bk + (size_t)blockIdx.x * cap; lo = rows.lo + (size_t)blockIdx.x * cap;
    hi = rows.hi + (size_t)blockIdx.x * cap; cp = rows.cp + (size_t)blockIdx.x * cap;
  } else {
    cap = kSortLdsPoints;
    xs = reinterpret_cast<double *>(smem);
    bk = reinterpret_cast<uint32_t *>(xs + cap); lo = bk + cap; hi = lo + cap; cp = hi + cap;
  }
  __shared__ uint32_t s_wave[kDbWaves];
  __shared__ uint32_t s_n, s_noise, s_carry;
  const unsigned lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const unsigned total = *count;
  for (unsigned e = blockIdx.x; e < total; e += gridDim.x) {
    const uint64_t k = list[e];
    if (threadIdx.x == 0) { s_n = 0; s_noise = 0; s_carry = 0; }
    __syncthreads();
    // ---- compaction of the present points, in time order ----
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
        bk[pos] = (uint32_t)t;
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
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (uint32_t i = n + threadIdx.x; i < np2; i += kDbBlock) { xs[i] = __builtin_huge_val(); bk[i] = 0xFFFFFFFFu; }
    __syncthreads();
    // ---- bitonic sort by value (ties in any order: the predicates only see values) ----
    for (uint32_t kk = 2; kk <= np2; kk <<= 1)
      for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < np2; i += kDbBlock) {
          const uint32_t l = i ^ j;
          if (l > i) {
            const double a = xs[i], b = xs[l];
            const bool up = (i & kk) == 0;
            if (up ? a > b : a < b) {
              xs[i] = b; xs[l] = a;
              const uint32_t ta = bk[i]; bk[i] = bk[l]; bk[l] = ta;
            }
          }
        }
        __syncthreads();
      }
    // ---- windows: lo = first j <= i with x_i - x_j <= eps, hi = last j >= i with x_j - x_i <= eps; core flag ----
    for (uint32_t i = threadIdx.x; i < n; i += kDbBlock) {
      const double xi = xs[i];
      uint32_t a = 0, b = i;                 // the predicate holds at i and is monotone: false ... false true ... true on [0, i]
      while (a < b) { const uint32_t mid = (a + b) >> 1; if (fabs(xi - xs[mid]) <= eps) b = mid; else a = mid + 1; }
      const uint32_t l = a;
      a = i; b = n - 1;                      // true ... true false ... false on [i, n)
      while (a < b) { const uint32_t mid = (a + b + 1) >> 1; if (fabs(xi - xs[mid]) <= eps) a = mid; else b = mid - 1; }
      lo[i] = l; hi[i] = a;
      cp[i] = (a - l + 1 >= (uint32_t)min_samples) ? 1u : 0u;
    }
    __syncthreads();
    // ---- inclusive prefix of the core flags over the sorted order (chunks of the workgroup, running carry) ----
    for (uint32_t c0 = 0; c0 < n; c0 += kDbBlock) {
      const uint32_t i = c0 + threadIdx.x;
      const uint32_t v = i < n ? cp[i] : 0u;
      uint32_t incl = v;
      for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += y; }
      if (lane == 63) s_wave[wave] = incl;
      __syncthreads();
      uint32_t base = s_carry;
      for (int w = 0; w < wave; ++w) base += s_wave[w];
      if (i < n) cp[i] = base + incl;
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kDbWaves; ++w) tot += s_wave[w];
        s_carry += tot;
      }
      __syncthreads();
    }
    // ---- noise: not core and no core point in the window ----
    uint32_t noise = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kDbBlock) {
      const uint32_t before = i ? cp[i - 1] : 0u;
      const bool core = cp[i] != before;
      if (core) continue;
      const uint32_t l = lo[i], h = hi[i];
      const uint32_t in_window = cp[h] - (l ? cp[l - 1] : 0u);
      if (in_window == 0) { g.flag[(uint64_t)bk[i] * g.K + k] = FLAG_PRESENT | FLAG_ANOMALY; noise++; }
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
                                                              double *__restrict__ sg_e, unsigned long long *__restrict__ am_e,
                                                              const unsigned long long *__restrict__ cs_val, const uint8_t *__restrict__ cs_flag,
                                                              const uint8_t *__restrict__ cs_has, uint32_t cs_cap, const uint32_t *__restrict__ cs_src,
                                                              const unsigned long long *__restrict__ rs_val, const uint8_t *__restrict__ rs_flag) {
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x >> 6;
  const unsigned total = *count;
  for (unsigned e = blockIdx.x * kDbWaves + wave; e < total; e += gridDim.x * kDbWaves) {   // wavefront-uniform
    const uint64_t k = list[e];
    double x[PPL];
    bool p[PPL];
    unsigned long long pm[PPL];
    // wavefront-uniform: 1 = the series lies contiguous behind this list, 2 = behind the redo list (entry cs_src[e]; cells flagged 2 — their
    // aggregate came from the overflow list — are read from the grid), 0 = gather the key's column from the grid
    const uint8_t has = (cs_has != nullptr && e < cs_cap) ? cs_has[e] : (uint8_t)0;
    const size_t src = has == 2 ? (size_t)cs_src[e] * g.T : (size_t)e * g.T;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const uint64_t t = lane + 64u * (unsigned)j;
      if (has == 1) {
        p[j] = t < g.T && (cs_flag[src + t] & FLAG_PRESENT);
        x[j] = p[j] ? (double)cs_val[src + t] : 0.0;
      } else if (has == 2) {
        const uint8_t fl = t < g.T ? rs_flag[src + t] : (uint8_t)0;
        p[j] = (fl & FLAG_PRESENT) != 0;
        x[j] = p[j] ? (double)((fl & 2) ? g.val[t * g.K + k] : rs_val[src + t]) : 0.0;
      } else {
        p[j] = t < g.T && (g.flag[t * g.K + k] & FLAG_PRESENT);
        x[j] = p[j] ? (double)g.val[t * g.K + k] : 0.0;
      }
      pm[j] = __ballot(p[j]);
    }
    int cnt[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) cnt[i] = 0;
#pragma unroll
    for (int jj = 0; jj < PPL; ++jj)
      for (unsigned long long m = pm[jj]; m; m &= m - 1) {
        const double xj = readlane_f64(x[jj], __ffsll((long long)m) - 1);
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
    // reach matters for the present points that are NOT core, and a listed key typically has a handful of those next to ~T core points:
    // walk whichever set is smaller (wavefront-uniform).  Walking the non-core points asks "is a core point within eps of x_j" of every
    // lane at once — |x_i - x_j| and |x_j - x_i| are the same double, so it is the same predicate on the same pairs.
    unsigned long long nm[PPL];
    int n_core = 0, n_rest = 0;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      nm[i] = pm[i] & ~cm[i];
      n_core += __popcll(cm[i]);
      n_rest += __popcll(nm[i]);
    }
    if (n_rest < n_core) {
#pragma unroll
      for (int jj = 0; jj < PPL; ++jj)
        for (unsigned long long m = nm[jj]; m; m &= m - 1) {
          const int l = __ffsll((long long)m) - 1;
          const double xj = readlane_f64(x[jj], l);
          bool near = false;
#pragma unroll
          for (int i = 0; i < PPL; ++i) near = near || (core[i] && fabs(x[i] - xj) <= eps);
          if (__ballot(near) != 0ull && (int)lane == l) reach[jj] = true;
        }
    } else {
#pragma unroll
      for (int jj = 0; jj < PPL; ++jj)
        for (unsigned long long m = cm[jj]; m; m &= m - 1) {
          const double xj = readlane_f64(x[jj], __ffsll((long long)m) - 1);
#pragma unroll
          for (int i = 0; i < PPL; ++i) reach[i] = reach[i] || fabs(x[i] - xj) <= eps;
        }
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
    // What the job's emit needs of a key with noise points, left next to the list entry: its noise masks (the series is in registers
    // HERE; k_emit_dbscan_wave used to fetch the key's whole column a second time for them) — and its stddev_samp, which
    // k_dbscan_list_sigma computes with one LANE per entry afterwards.
    if (am_e != nullptr && noise && lane == 0) {
#pragma unroll
      for (int i = 0; i < PPL; ++i) am_e[(size_t)e * PPL + i] = am[i];
    }
  }
}

// k_dbscan_list_sigma — stddev_samp of every listed key that has noise points (an output column of its rows), one LANE per list entry.
// Spark's CentralMomentAgg over the present points in time order (an IEEE division = the bits of div_by_count with RN(1 / count) from
// an LDS table, as k_key_sigma).  Until round 5 k_dbscan_list_wave ran this recurrence itself, every lane of the key's wavefront
// computing the same serial chain from readlanes: ~14 instructions x T per KEY, more than the key's pair tests — 60 of that kernel's
// 86 us at C4, where 70 % of the listed keys have noise.  A lane per entry issues the same chain once per 64 entries: 30 us, of which
// (probe builds, profiles/r5_d5_* ... r5_d9_*) 5 are the launch over a grid sized from K, 9 the loads — every lane walks its own cache
// lines — and 14 the chain itself: 7 dependent FP64 operations per present point on one wavefront per SIMD.  What did NOT change it:
// the division instead of the table (+1.6 us), the batches 1 or 3 ahead, the u64 -> f64 conversions and the table lookup taken out of
// the step, m2's three operations moved into the next step's stalls.  On a second stream beside the count scan it saved 5 us
// (profiles/r5_d10_*: two cross-stream dependencies cost what the overlap gives), so it runs on the job's stream.
static constexpr int kSigmaBlock = 64;   // one wavefront per workgroup: the ~300 wavefronts of C4's list spread over all CUs (256 threads: 54 us)
__global__ __launch_bounds__(kSigmaBlock) void k_dbscan_list_sigma(Grid g, const uint32_t *__restrict__ list, const unsigned int *__restrict__ count,
                                                                  const uint32_t *__restrict__ n_anom, double *__restrict__ sg_e,
                                                                  const unsigned long long *__restrict__ cs_val, const uint8_t *__restrict__ cs_flag,
                                                                  const uint8_t *__restrict__ cs_has, uint32_t cs_cap, const uint32_t *__restrict__ cs_src,
                                                                  const unsigned long long *__restrict__ rs_val, const uint8_t *__restrict__ rs_flag) {
  __shared__ double s_rcp[257];   // (the list_wave path: T <= 256)
  const uint32_t T = (uint32_t)g.T;
  const unsigned total = *count;
  if (blockIdx.x * kSigmaBlock >= total) return;   // (the grid is sized from K, the list is ~2 % of it)
  for (uint32_t i = threadIdx.x; i <= T; i += kSigmaBlock) s_rcp[i] = 1.0 / (double)i;
  __syncthreads();
  for (unsigned e = blockIdx.x * kSigmaBlock + threadIdx.x; e < total; e += gridDim.x * kSigmaBlock) {
    const uint64_t k = list[e];
    if (n_anom[k] == 0) continue;
    // 1 = the series lies contiguous behind this list, 2 = behind the redo list (entry cs_src[e]; cells flagged 2 — their aggregate came
    // from the overflow list — are read from the grid), 0 = the key's column of the grid                 (as k_dbscan_list_wave)
    const uint8_t has = (cs_has != nullptr && e < cs_cap) ? cs_has[e] : (uint8_t)0;
    const size_t src = has == 2 ? (size_t)cs_src[e] * T : (size_t)e * T;
    const unsigned long long *pv = has == 1 ? cs_val + src : (has == 2 ? rs_val + src : g.val + k);
    const uint8_t *pf = has == 1 ? cs_flag + src : (has == 2 ? rs_flag + src : g.flag + k);
    double cn = 0.0, avg = 0.0, m2 = 0.0;
    uint32_t n = 0;
    // (OVF: a series from the redo list — a cell flagged 2 has its aggregate in the grid.  Only that instantiation has a global load
    // inside the step; in the others nothing is younger than the prefetched batch, so the compiler waits with vmcnt(N > 0))
    auto point = [&](auto ovf, uint32_t t, uint8_t fl, unsigned long long raw) {
      if (fl & FLAG_PRESENT) {
        if (decltype(ovf)::value && (fl & 2)) raw = g.val[(uint64_t)t * g.K + k];
        const double xv = (double)raw;
        n++;
        cn = cn + 1.0;
        const double d = xv - avg;
        const double dn = div_by_count(d, cn, s_rcp[n]);     // == d / cn, bit for bit
        avg = avg + dn;
        m2 = m2 + d * (d - dn);
      }
    };
    // the walk in batches of eight buckets, the next batch requested before the current one is worked on; once with a compile-time stride
    // of 1 (contiguous series: the eight loads of a batch are adjacent, so they merge into 16-byte ones) and once with the grid's stride
    auto walk = [&](auto contiguous, auto ovf) {
      const size_t st = decltype(contiguous)::value ? (size_t)1 : (size_t)g.K;
      uint8_t fa[8], fb[8];
      unsigned long long va[8], vb[8];
      const uint32_t full = T & ~7u;
      if (full) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = pf[(size_t)j * st]; va[j] = pv[(size_t)j * st]; }
      }
      for (uint32_t t0 = 0; t0 < full; t0 += 8) {
        const uint32_t tn = t0 + 8 < full ? t0 + 8 : t0;   // (the last batch loads itself again: no branch around the loads)
#pragma unroll
        for (int j = 0; j < 8; ++j) { fb[j] = pf[(size_t)(tn + j) * st]; vb[j] = pv[(size_t)(tn + j) * st]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) point(ovf, t0 + (uint32_t)j, fa[j], va[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = fb[j]; va[j] = vb[j]; }
      }
      for (uint32_t t = full; t < T; ++t) point(ovf, t, pf[(size_t)t * st], pv[(size_t)t * st]);
    };
    if (has == 1) walk(std::true_type{}, std::false_type{});
    else if (has == 2) walk(std::true_type{}, std::true_type{});
    else walk(std::false_type{}, std::false_type{});
    sg_e[e] = cn >= 2.0 ? sqrt(m2 / (cn - 1.0)) : 0.0;
  }
}
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

// Scratch of one DBSCAN launch: counters (64 B: work-list length, redo-list length) | list[K] u32 | redo list[K] u32 | then, by series length,
//   T <= 256                : sg[K] f64 | am[K * 4] u64          (what k_dbscan_list_wave leaves for the emit)
//   T <= kSortLdsPoints     : nothing (k_dbscan_sorted works in LDS)
//   longer                  : per-workgroup rows of k_dbscan_sorted, 24 B per point, cap = T rounded up to a power of two
static size_t one_list_bytes(Grid g) { return ((size_t)g.K * 4 + 63) & ~(size_t)63; }
static size_t list_bytes(Grid g) { return 64 + 2 * one_list_bytes(g); }     // counters | work list | redo list
uint32_t *dbscan_redo_list(Grid g, void *scratch) { return reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(scratch) + 64 + one_list_bytes(g)); }
static uint32_t sort_cap(uint64_t T) { uint32_t c = 1; while (c < T) c <<= 1; return c; }
static uint32_t sort_blocks(Grid g) {   // workgroups of the long-series form: bounded scratch (256 MB), at least one
  const uint64_t per = (uint64_t)sort_cap(g.T) * 24;
  uint64_t b = (256ull << 20) / per;
  if (b > 1024) b = 1024;
  if (b > g.K) b = g.K;
  return (uint32_t)(b ? b : 1);
}

// contiguous series of the listed keys (T <= 256): cs_cap entries of T values + T flags, + one byte per entry
static uint32_t compact_cap(Grid g) { const uint64_t c = g.K / 8 > 4096 ? g.K / 8 : 4096; return (uint32_t)(c < g.K ? c : g.K); }
static size_t wave_list_bytes(Grid g) { return list_bytes(g) + (((size_t)g.K * (8 + 8 * 4) + 63) & ~(size_t)63); }
static size_t compact_bytes(Grid g) {   // values | flags | has | src (the redo entry a work-list entry's series lives in: cs_has == 2)
  const size_t c = compact_cap(g);
  return ((c * g.T * 8 + 63) & ~(size_t)63) + ((c * g.T + 63) & ~(size_t)63) + ((c + 63) & ~(size_t)63) + ((c * 4 + 63) & ~(size_t)63);
}
static uint32_t *compact_src(Grid g, void *scratch) {
  const size_t c = compact_cap(g);
  return reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(scratch) + wave_list_bytes(g) + ((c * g.T * 8 + 63) & ~(size_t)63) + ((c * g.T + 63) & ~(size_t)63) +
                                      ((c + 63) & ~(size_t)63));
}

void dbscan_compact_series(Grid g, void *scratch, unsigned long long **cs_val, uint8_t **cs_flag, uint8_t **cs_has, uint32_t *cs_cap) {
  *cs_val = nullptr; *cs_flag = nullptr; *cs_has = nullptr; *cs_cap = 0;
  if (g.T == 0 || g.T > 256 || g.K == 0) return;
  const size_t c = compact_cap(g);
  unsigned char *p = static_cast<unsigned char *>(scratch) + wave_list_bytes(g);
  *cs_val = reinterpret_cast<unsigned long long *>(p); p += (c * g.T * 8 + 63) & ~(size_t)63;
  *cs_flag = p; p += (c * g.T + 63) & ~(size_t)63;
  *cs_has = p;
  *cs_cap = (uint32_t)c;
}

void dbscan_redo_series(Grid g, void *scratch, unsigned long long **rs_val, uint8_t **rs_flag, uint8_t **rs_has, uint32_t *rs_cap) {
  *rs_val = nullptr; *rs_flag = nullptr; *rs_has = nullptr; *rs_cap = 0;
  if (g.T == 0 || g.T > 256 || g.K == 0) return;
  const size_t c = compact_cap(g);
  unsigned char *p = static_cast<unsigned char *>(scratch) + wave_list_bytes(g) + compact_bytes(g);
  *rs_val = reinterpret_cast<unsigned long long *>(p); p += (c * g.T * 8 + 63) & ~(size_t)63;
  *rs_flag = p; p += (c * g.T + 63) & ~(size_t)63;
  *rs_has = p;
  *rs_cap = (uint32_t)c;
}

size_t dbscan_scratch_bytes(Grid g) {
  if (g.T <= 256) return wave_list_bytes(g) + 2 * compact_bytes(g);
  if (g.T <= kSortLdsPoints) return list_bytes(g);
  return list_bytes(g) + (size_t)sort_blocks(g) * sort_cap(g.T) * 24 + 64;
}

// scratch of the wave list (T <= 256): count | list[K] u32 | sg[K] f64 | am[K * 4] u64
static double *wave_list_sg(const void *scratch, Grid g) {
  return reinterpret_cast<double *>(const_cast<unsigned char *>(static_cast<const unsigned char *>(scratch)) + list_bytes(g));
}
static unsigned long long *wave_list_am(const void *scratch, Grid g) { return reinterpret_cast<unsigned long long *>(wave_list_sg(scratch, g) + g.K); }

bool dbscan_uses_list(Grid g) { return g.T < (1ull << 31); }   // every series length goes through scan + work list since round 4

int launch_dbscan(hipStream_t s, Grid g, double eps, int min_samples, void *scratch, DbscanStats st, bool settled_by_stage0) {
  if (g.K == 0 || g.T == 0) return 0;
  if (!dbscan_uses_list(g)) return -1;
  unsigned int *count = static_cast<unsigned int *>(scratch);
  uint32_t *list = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(scratch) + 64);
  const unsigned lane_blocks = (unsigned)((g.K + kDbBlock - 1) / kDbBlock);
  unsigned long long *cs_val; uint8_t *cs_flag, *cs_has; uint32_t cs_cap;
  dbscan_compact_series(g, scratch, &cs_val, &cs_flag, &cs_has, &cs_cap);
  unsigned long long *rs_val = nullptr; uint8_t *rs_flag = nullptr, *rs_has = nullptr; uint32_t rs_cap = 0;
  if (settled_by_stage0) {   // the list was started by pass C (its counters zeroed before Stage 0), with the listed keys' series contiguous behind it;
    // the keys it could not decide are on the redo list
#ifndef TAD_REDO_BLOCKS   // (measurement knob.  C4, ~1e4 redo keys: 512 workgroups 33.6 us, 1024 30.5, 2048 36.1, 4096 42.4 — more workgroups, more atomics on the list counter)
#define TAD_REDO_BLOCKS 1024
#endif
    const uint64_t rb = g.K < TAD_REDO_BLOCKS ? g.K : TAD_REDO_BLOCKS;     // (grid-stride over the device-side redo count)
    dbscan_redo_series(g, scratch, &rs_val, &rs_flag, &rs_has, &rs_cap);
    hipLaunchKernelGGL(k_dbscan_scan_redo, dim3((unsigned)rb), dim3(kDbBlock), 0, s, g, eps, min_samples, st, dbscan_redo_list(g, scratch), count + 1, list, count,
                       cs_has, cs_cap, rs_val, rs_flag, rs_has, rs_cap, compact_src(g, scratch));
  } else {
    cs_has = nullptr;        // nobody wrote contiguous series
    hipMemsetAsync(count, 0, sizeof(unsigned int), s);
    if (coop_shape(g))   // long series on few keys: a wavefront per key
      hipLaunchKernelGGL((k_dbscan_scan<false, true>), dim3((unsigned)((g.K * 64 + kDbBlock - 1) / kDbBlock)), dim3(kDbBlock), 0, s, g, eps, min_samples, st, list, count, cs_has, cs_cap);
    else
      hipLaunchKernelGGL((k_dbscan_scan<false, false>), dim3(lane_blocks), dim3(kDbBlock), 0, s, g, eps, min_samples, st, list, count, cs_has, cs_cap);
  }
  // grid-stride over the (device-side) list length: 8192 workgroups of four wavefronts give every one of C4's ~2e4 listed keys its
  // own wavefront (2048: 2-3 keys per wavefront one after the other; detect + emit 0.179 -> 0.174 ms, 32768 the same:
  // profiles/r3_v9_c4_list_blocks_ab.log)
  uint64_t blocks = g.K < 8192 ? g.K : 8192;
  if (g.T <= 256) {   // a wavefront's registers hold the whole series
#define TAD_DBW(PPL) hipLaunchKernelGGL((k_dbscan_list_wave<PPL>), dim3((unsigned)blocks), dim3(kDbBlock), 0, s, g, eps, min_samples, list, count, st.n_anom, wave_list_sg(scratch, g), wave_list_am(scratch, g), cs_val, cs_flag, cs_has, cs_cap, compact_src(g, scratch), rs_val, rs_flag)
    if (g.T <= 64) TAD_DBW(1); else if (g.T <= 128) TAD_DBW(2); else if (g.T <= 192) TAD_DBW(3); else TAD_DBW(4);
#undef TAD_DBW
    if (st.n_anom != nullptr) {   // (the list emit is the only reader of the entries' stddev_samp, and it runs for jobs that count here)
      const uint64_t sblocks = (g.K + kSigmaBlock - 1) / kSigmaBlock;
      // (on the job's stream: forked onto a second stream beside the count scan it saved 5 us of its 30 — the two cross-stream dependencies
      // cost what the overlap gives, profiles/r5_d10_*)
      hipLaunchKernelGGL(k_dbscan_list_sigma, dim3((unsigned)(sblocks < 1024 ? sblocks : 1024)), dim3(kSigmaBlock), 0, s, g, list, count, st.n_anom,
                         wave_list_sg(scratch, g), cs_val, cs_flag, cs_has, cs_cap, compact_src(g, scratch), rs_val, rs_flag);
    }
    return 0;
  }
  SortRows rows{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  size_t lds = (size_t)kSortLdsPoints * 24;
  if (g.T > kSortLdsPoints) {
    const uint32_t cap = sort_cap(g.T), nb = sort_blocks(g);
    unsigned char *base = static_cast<unsigned char *>(scratch) + list_bytes(g);
    rows.x = reinterpret_cast<double *>(base);
    rows.bk = reinterpret_cast<uint32_t *>(rows.x + (size_t)nb * cap);
    rows.lo = rows.bk + (size_t)nb * cap;
    rows.hi = rows.lo + (size_t)nb * cap;
    rows.cp = rows.hi + (size_t)nb * cap;
    rows.cap = cap;
    blocks = nb;
    lds = 0;
  } else if (blocks > 2048) {
    blocks = 2048;
  }
  allow_big_lds(reinterpret_cast<const void *>(k_dbscan_sorted), 152 * 1024);
  hipLaunchKernelGGL(k_dbscan_sorted, dim3((unsigned)blocks), dim3(kDbBlock), lds, s, g, eps, min_samples, list, count, st.n_anom, rows);
  return 0;
}

// the DBSCAN job's emit from the work list launch_dbscan left in `scratch` (false: shape not supported, use launch_emit kind 4)
bool launch_emit_dbscan_list(hipStream_t s, Grid g, Lattice lat, const void *scratch, const uint32_t *n_anom, const unsigned long long *off,
                             OutRows out) {
  if (g.K == 0 || g.T == 0 || g.T > 256) return false;
  const unsigned int *count = static_cast<const unsigned int *>(scratch);
  const uint32_t *list = reinterpret_cast<const uint32_t *>(static_cast<const unsigned char *>(scratch) + 64);
  const uint64_t blocks = g.K < 2048 ? g.K : 2048;
#define TAD_DBE(PPL) hipLaunchKernelGGL((k_emit_dbscan_wave<PPL>), dim3((unsigned)blocks), dim3(kDbBlock), 0, s, g, lat, list, count, n_anom, wave_list_sg(scratch, g), wave_list_am(scratch, g), off, out)
  if (g.T <= 64) TAD_DBE(1); else if (g.T <= 128) TAD_DBE(2); else if (g.T <= 192) TAD_DBE(3); else TAD_DBE(4);
#undef TAD_DBE
  return true;
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_dbscan() { return reinterpret_cast<const void *>(&k_dbscan_scan_redo); }

}  // namespace tad
