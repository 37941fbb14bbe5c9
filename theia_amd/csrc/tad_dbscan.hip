// tad_dbscan.hip — DBSCAN verdicts per key (anomaly_detection.py:325-349):
//   DBSCAN(min_samples=4, eps=250000000).fit_predict(x.reshape(-1, 1)) == -1
// on the 1-D values of ONE key's series.  Cluster ids are discarded by the reference (:344-348), so
// only the noise predicate is needed (checked against sklearn in the oracle tests):
//   core(i)  <=>  #{ j : |x_i - x_j| <= eps } >= min_samples        (self counted, inclusive <=)
//   noise(i) <=>  !core(i)  and  no core j with |x_i - x_j| <= eps
// Distances are FP64 on correctly rounded uint64 -> double values: pure comparisons, so verdicts
// are exact.
//
// Layout: a workgroup stages a tile of KT consecutive keys x all T buckets of the time-major grid
// into LDS (coalesced: KT*8 contiguous bytes per bucket), transposed to one row per key.  Each
// wavefront then owns one key at a time: it compacts the present points of the row (ballot +
// popcount prefix), and runs the pair tests with lanes = points i and an LDS-broadcast x_j stream
// — the O(n^2) compares never leave LDS/registers.
#include <cstdlib>

#include "tad_internal.h"

namespace tad {

static constexpr int kDbBlock = 256;
static constexpr int kDbWaves = kDbBlock / 64;

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63; }

// LDS carve: xs[KT][Tp] f64 | ct[kDbWaves][T] u32 | fl[KT][T] u8
__global__ __launch_bounds__(kDbBlock) void k_dbscan_tile(Grid g, double eps, int min_samples, int KT, int Tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t T = g.T;
  double *xs = reinterpret_cast<double *>(smem);
  uint32_t *ct = reinterpret_cast<uint32_t *>(smem + (size_t)KT * Tp * sizeof(double));
  uint8_t *fl = reinterpret_cast<uint8_t *>(ct + (size_t)kDbWaves * T);

  const uint64_t k0 = (uint64_t)blockIdx.x * KT;
  const int kt = (int)((g.K - k0) < (uint64_t)KT ? (g.K - k0) : (uint64_t)KT);

  // ---- stage the tile: global (time-major, coalesced over keys) -> LDS (key-major rows) ----
  const uint64_t total = (uint64_t)KT * T;
  const int kt_shift = __builtin_ctz((unsigned)KT);  // KT is a power of two
  for (uint64_t e = threadIdx.x; e < total; e += kDbBlock) {
    const uint64_t t = e >> kt_shift;
    const int kk = (int)(e & (uint64_t)(KT - 1));
    uint8_t f = 0;
    double x = 0.0;
    if (kk < kt) {
      const uint64_t c = t * g.K + k0 + kk;
      f = g.flag[c];
      x = (double)g.val[c];
    }
    xs[(size_t)kk * Tp + t] = x;
    fl[(size_t)kk * T + t] = f;
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6;
  const unsigned lane = lane_id();
  uint32_t *my_ct = ct + (size_t)wave * T;

  for (int kk = wave; kk < kt; kk += kDbWaves) {
    double *row = xs + (size_t)kk * Tp;
    uint8_t *frow = fl + (size_t)kk * T;
    // ---- compact the present points to the front of the row (in place), remember their bucket ----
    uint32_t n = 0;
    for (uint64_t c0 = 0; c0 < T; c0 += 64) {
      const uint64_t t = c0 + lane;
      const bool p = t < T && (frow[t] & FLAG_PRESENT);
      const double x = t < T ? row[t] : 0.0;
      const unsigned long long m = __ballot(p);
      const uint32_t pos = n + __popcll(m & ((1ull << lane) - 1ull));
      if (p) {
        row[pos] = x;
        my_ct[pos] = (uint32_t)t;
      }
      n += __popcll(m);
    }
    if (n == 0) continue;
    // ---- pass 1: core points ----
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
      const uint32_t i = i0 + lane;
      const double xi = i < n ? row[i] : 0.0;
      int cnt = 0;
#pragma unroll 4
      for (uint32_t j = 0; j < n; ++j) cnt += fabs(xi - row[j]) <= eps ? 1 : 0;
      if (i < n) frow[i] = cnt >= min_samples ? 1 : 0;  // frow is now: core flag per compacted point
    }
    // ---- pass 2: noise = not core and no core point within eps ----
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool valid = i < n;
      const bool core = valid && frow[i];
      const bool need = valid && !core;
      bool reach = false;
      if (__any(need)) {
        const double xi = valid ? row[i] : 0.0;
#pragma unroll 4
        for (uint32_t j = 0; j < n; ++j) reach = reach || (frow[j] && fabs(xi - row[j]) <= eps);
      }
      if (need && !reach) {
        const uint64_t c = (uint64_t)my_ct[i] * g.K + k0 + kk;
        g.flag[c] = FLAG_PRESENT | FLAG_ANOMALY;
      }
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

static bool pick_tile(uint64_t T, int *KT, int *Tp, size_t *bytes) {
  const int tp = (int)(T | 1);  // odd row stride (in doubles): conflict-free transposed LDS writes
  auto need = [&](int kt) { return (size_t)kt * tp * 8 + (size_t)kDbWaves * T * 4 + (size_t)kt * T; };
  int kt = 0;
  // occupancy first: the pair loops are latency-bound (LDS broadcast reads), so aim at >= 4 workgroups (16 wavefronts)
  // per CU; 16 keys still give 128-byte coalesced segments when the tile is staged from the time-major grid
  const char *kt_env = getenv("TAD_DB_KT");  // tuning knob
  const int kt_force = kt_env ? atoi(kt_env) : 0;
  if (kt_force > 0 && (kt_force & (kt_force - 1)) == 0 && need(kt_force) <= 150 * 1024) kt = kt_force;
  else if (need(64) <= 20 * 1024) kt = 64;   // measured at C4 (T = 100): 64 keys 1.96 ms, 32 keys 1.1 ms, 16 / 8 keys 0.95 ms
  else if (need(32) <= 20 * 1024) kt = 32;
  else if (need(16) <= 40 * 1024) kt = 16;
  else if (need(8) <= 40 * 1024) kt = 8;
  else {
    for (int c = 64; c >= 1; c >>= 1)
      if (need(c) <= 150 * 1024) { kt = c; break; }
  }
  if (kt == 0) return false;
  *KT = kt;
  *Tp = tp;
  *bytes = (need(kt) + 15) & ~(size_t)15;
  return true;
}

size_t dbscan_long_scratch_bytes(Grid g) {
  int kt, tp; size_t b;
  if (pick_tile(g.T, &kt, &tp, &b)) return 0;
  return (size_t)g.K * g.T * (8 + 4 + 1);
}

int launch_dbscan(hipStream_t s, Grid g, double eps, int min_samples) {
  if (g.K == 0 || g.T == 0) return 0;
  int KT, Tp;
  size_t bytes;
  if (!pick_tile(g.T, &KT, &Tp, &bytes)) return -1;
  allow_big_lds(reinterpret_cast<const void *>(k_dbscan_tile), 152 * 1024);
  const uint64_t blocks = (g.K + KT - 1) / KT;
  hipLaunchKernelGGL(k_dbscan_tile, dim3((unsigned)blocks), dim3(kDbBlock), bytes, s, g, eps, min_samples, KT, Tp);
  return 0;
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
