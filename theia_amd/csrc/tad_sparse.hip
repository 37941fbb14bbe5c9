// tad_sparse.hip — Stage 0 for SPARSE tables: GROUP BY (key, flowEndSeconds) in time proportional to the rows, not to
// keys x time-lattice.
//
// The dense path lays the aggregated points on a K x T grid over the flowEndSeconds lattice.  That is the right layout
// for the benchmarked tables (minute-resolution lattices, every key active most of the time), but the reference accepts
// any flowEndSeconds and, in mode None, keys per connection (anomaly_detection.py:52-61, 109-116: the key contains
// flowStartSeconds): second-resolution timestamps with gcd 1 over a day give T = 86 400 for a handful of points per
// key.  Here the rows are sorted by (key, time) instead (rocPRIM radix sort on key << 32 | (t - t0)), equal (key, time)
// runs are reduced with the job's operator (wrapping u64 add / unsigned max — the same associative integer operators,
// so the aggregates are bit-identical to the dense path's), and the points of every key are laid out by RANK in time
// order: a K x Tmax grid, Tmax = the longest series, plus a parallel grid of the points' timestamps.  Every per-key
// kernel downstream (stddev_samp, EWMA, DBSCAN, ARIMA, emit) only needs a key's points in time order, so they run
// unchanged on the rank grid; emit takes the timestamps from the parallel grid instead of the lattice.
#include <hipcub/hipcub.hpp>

#include "tad_internal.h"

namespace tad {

static constexpr int kSpBlock = 256;
static constexpr unsigned long long kInvalid = ~0ull;

struct SumOp { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; } };
struct MaxOp { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a > b ? a : b; } };

// composite sort key of every (row, key) slot: key << 32 | (t - t0); rows that are filtered out get kInvalid (sorts last)
__global__ __launch_bounds__(kSpBlock) void k_sparse_keys(const uint64_t *__restrict__ key, const uint64_t *__restrict__ key2,
                                                         const int64_t *__restrict__ t_end, const int64_t *__restrict__ t_start,
                                                         const uint64_t *__restrict__ value, uint64_t n, uint64_t K, RowFilter f, int64_t t0,
                                                         unsigned long long *__restrict__ comp, unsigned long long *__restrict__ vals,
                                                         DevCounters *ctr) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  uint32_t err = 0, used = 0;
  if (i < n) {
    const int64_t te = t_end[i];
    bool kept = true;
    if (f.end_time != 0 && !(te < f.end_time)) kept = false;                                       // anomaly_detection.py:584-586
    if (f.start_time != 0 && t_start != nullptr && !(t_start[i] >= f.start_time)) kept = false;    // :581-583
    const uint64_t dt = (uint64_t)te - (uint64_t)t0;
    const uint64_t v = value[i];
    const int nk = key2 != nullptr ? 2 : 1;
    for (int h = 0; h < nk; ++h) {
      const uint64_t k = h == 0 ? key[i] : key2[i];
      unsigned long long c = kInvalid;
      if (kept && k != TAD_KEY_SKIP) {
        if (k >= K) err |= DEV_ERR_KEY_RANGE;
        else if ((dt >> 32) != 0) err |= DEV_ERR_OFF_LATTICE;     // te < t0 or a span of more than 2^32 s: the caller's lattice hint was wrong
        else { c = ((unsigned long long)k << 32) | dt; used++; }
      }
      comp[i * nk + h] = c;
      vals[i * nk + h] = v;
    }
  }
  unsigned long long u = used;
  for (int d = 32; d >= 1; d >>= 1) { u += __shfl_down(u, d); err |= __shfl_down(err, d); }
  if ((threadIdx.x & 63) == 0) {
    if (u) atomicAdd(&ctr->rows_used, u);
    if (err) atomicOr(&ctr->err, err);
  }
}

// first[k] = index of the key's first point in the sorted unique list
__global__ __launch_bounds__(kSpBlock) void k_sparse_first(const unsigned long long *__restrict__ ucomp, uint64_t P, uint32_t *__restrict__ first) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (i >= P) return;
  const uint32_t k = (uint32_t)(ucomp[i] >> 32);
  if (i == 0 || (uint32_t)(ucomp[i - 1] >> 32) != k) first[k] = (uint32_t)i;
}

// longest series
__global__ __launch_bounds__(kSpBlock) void k_sparse_tmax(const unsigned long long *__restrict__ ucomp, uint64_t P, const uint32_t *__restrict__ first,
                                                         unsigned int *__restrict__ tmax) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  unsigned int n = 0;
  if (i < P) {
    const uint32_t k = (uint32_t)(ucomp[i] >> 32);
    if (i + 1 == P || (uint32_t)(ucomp[i + 1] >> 32) != k) n = (unsigned int)(i - first[k] + 1);
  }
  for (int d = 32; d >= 1; d >>= 1) { const unsigned int o = __shfl_down(n, d); n = o > n ? o : n; }
  if ((threadIdx.x & 63) == 0 && n) atomicMax(tmax, n);
}

// point i -> cell (rank in its key's series, key) of the rank grid
__global__ __launch_bounds__(kSpBlock) void k_sparse_place(const unsigned long long *__restrict__ ucomp, const unsigned long long *__restrict__ uval,
                                                          uint64_t P, const uint32_t *__restrict__ first, int64_t t0, Grid g,
                                                          long long *__restrict__ times) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (i >= P) return;
  const unsigned long long c = ucomp[i];
  const uint32_t k = (uint32_t)(c >> 32);
  const uint64_t cell = (uint64_t)(i - first[k]) * g.K + k;
  g.val[cell] = uval[i];
  g.flag[cell] = FLAG_PRESENT;
  times[cell] = (long long)(t0 + (int64_t)(c & 0xffffffffull));
}

size_t sparse_sort_temp_bytes(uint64_t slots) {
  size_t a = 0, b = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                     (const unsigned long long *)nullptr, (unsigned long long *)nullptr, slots);
  hipcub::DeviceReduce::ReduceByKey(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (unsigned long long *)nullptr, SumOp(),
                                    slots);
  return (((a > b ? a : b) + 255) & ~(size_t)255) + 256;   // 256-byte multiple: the caller places two counters right behind it
}

// rows -> sorted unique (key, time) points with aggregated values: ucomp / uval (device), *num_runs (device)
int launch_sparse_group(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end, const int64_t *t_start,
                        const uint64_t *value, uint64_t n, uint64_t K, RowFilter f, int64_t t0, bool op_max, unsigned long long *comp_a,
                        unsigned long long *val_a, unsigned long long *comp_b, unsigned long long *val_b, void *temp, size_t temp_bytes,
                        unsigned long long *num_runs, DevCounters *ctr) {
  const uint64_t slots = n * (key2 != nullptr ? 2 : 1);
  hipLaunchKernelGGL(k_sparse_keys, dim3((unsigned)((n + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, key, key2, t_end, t_start, value, n, K,
                     f, t0, comp_a, val_a, ctr);
  size_t tb = temp_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(temp, tb, comp_a, comp_b, val_a, val_b, slots, 0, 64, s) != hipSuccess) return -1;
  tb = temp_bytes;
  hipError_t r;
  if (op_max) r = hipcub::DeviceReduce::ReduceByKey(temp, tb, comp_b, comp_a, val_b, val_a, num_runs, MaxOp(), slots, s);
  else r = hipcub::DeviceReduce::ReduceByKey(temp, tb, comp_b, comp_a, val_b, val_a, num_runs, SumOp(), slots, s);
  return r == hipSuccess ? 0 : -1;
}

void launch_sparse_tmax(hipStream_t s, const unsigned long long *ucomp, uint64_t P, uint32_t *first, unsigned int *tmax) {
  if (P == 0) return;
  const unsigned blocks = (unsigned)((P + kSpBlock - 1) / kSpBlock);
  hipLaunchKernelGGL(k_sparse_first, dim3(blocks), dim3(kSpBlock), 0, s, ucomp, P, first);
  hipLaunchKernelGGL(k_sparse_tmax, dim3(blocks), dim3(kSpBlock), 0, s, ucomp, P, first, tmax);
}

void launch_sparse_place(hipStream_t s, const unsigned long long *ucomp, const unsigned long long *uval, uint64_t P, const uint32_t *first,
                         int64_t t0, Grid g, long long *times) {
  if (P == 0) return;
  hipLaunchKernelGGL(k_sparse_place, dim3((unsigned)((P + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, ucomp, uval, P, first, t0, g, times);
}

}  // namespace tad
