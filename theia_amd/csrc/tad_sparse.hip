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


// ------------------------------------------------------------------------------------------------
// Length classes (skewed sparse tables).  The rank grid is K x Tmax cells: one key with a day of second-resolution points
// next to a million short-lived keys would need 86 400 cells for every key.  When that does not fit the workspace the keys
// are split into classes by series length (<= 16, <= 64, <= 256, ... points), every class becomes a points table of its own
// (key ids renumbered densely, order kept) and runs as a job of its own — its rank grid holds at most 4x its points (16
// cells per key in the first class) — and the row sets are merged back in key order (tad_capi.cpp:run_sparse_classes).
// Plain kernels: one thread per key / point / row, no shared state beyond integer atomics.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t sparse_class_of(uint32_t len) {   // 1..16 -> 0, 17..64 -> 1, 65..256 -> 2, ...
  uint32_t c = 0;
  for (uint64_t bound = 16; len > bound; bound <<= 2) ++c;
  return c;
}

// len[k] = points of key k (len zeroed by the caller: keys without points are not visited)
__global__ __launch_bounds__(kSpBlock) void k_sparse_len(const unsigned long long *__restrict__ ucomp, uint64_t P, const uint32_t *__restrict__ first,
                                                        uint32_t *__restrict__ len) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (i >= P) return;
  const uint32_t k = (uint32_t)(ucomp[i] >> 32);
  if (i + 1 == P || (uint32_t)(ucomp[i + 1] >> 32) != k) len[k] = (uint32_t)(i - first[k] + 1);
}

// member[k] = 1 and pts[k] = len[k] for the keys of class c, 0 otherwise (inputs of the two exclusive scans)
__global__ __launch_bounds__(kSpBlock) void k_sparse_class_counts(const uint32_t *__restrict__ len, uint64_t K, uint32_t c,
                                                                 uint32_t *__restrict__ member, uint32_t *__restrict__ pts) {
  const uint64_t k = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (k >= K) return;
  const uint32_t n = len[k];
  const bool in = n != 0 && sparse_class_of(n) == c;
  member[k] = in ? 1u : 0u;
  pts[k] = in ? n : 0u;
}

// the points of class c as a table of their own: (renumbered key, flowEndSeconds, aggregated value), (key, time) order kept
__global__ __launch_bounds__(kSpBlock) void k_sparse_class_columns(const unsigned long long *__restrict__ ucomp, const unsigned long long *__restrict__ uval,
                                                                  uint64_t P, const uint32_t *__restrict__ first, const uint32_t *__restrict__ len,
                                                                  uint32_t c, const unsigned long long *__restrict__ key_off,
                                                                  const unsigned long long *__restrict__ pt_off, int64_t t0,
                                                                  unsigned long long *__restrict__ out_key, long long *__restrict__ out_t,
                                                                  unsigned long long *__restrict__ out_val, uint32_t *__restrict__ keymap) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (i >= P) return;
  const unsigned long long comp = ucomp[i];
  const uint32_t k = (uint32_t)(comp >> 32);
  if (sparse_class_of(len[k]) != c) return;
  const uint64_t rank = i - first[k];
  const unsigned long long nk = key_off[k];
  const unsigned long long at = pt_off[k] + rank;
  out_key[at] = nk;
  out_t[at] = (long long)(t0 + (int64_t)(comp & 0xffffffffull));
  out_val[at] = uval[i];
  if (rank == 0) keymap[nk] = k;
}

// rows of one class result (ordered by its renumbered keys): per ORIGINAL key the row count and the first row
__global__ __launch_bounds__(kSpBlock) void k_class_count_rows(const unsigned long long *__restrict__ row_key, uint64_t R, const uint32_t *__restrict__ keymap,
                                                              uint32_t *__restrict__ cnt, unsigned long long *__restrict__ first_row) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (i >= R) return;
  const unsigned long long nk = row_key[i];
  const uint32_t k = keymap[nk];
  atomicAdd(&cnt[k], 1u);
  if (i == 0 || row_key[i - 1] != nk) first_row[k] = i;
}

// ... and the rows moved to their place in the merged result (ordered by original key, then time)
__global__ __launch_bounds__(kSpBlock) void k_class_gather(OutRows src, uint64_t R, const uint32_t *__restrict__ keymap,
                                                          const unsigned long long *__restrict__ off, const unsigned long long *__restrict__ first_row,
                                                          OutRows dst) {
  const uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x;
  if (i >= R) return;
  const uint32_t k = keymap[src.key_id[i]];
  const unsigned long long at = off[k] + (i - first_row[k]);
  dst.key_id[at] = k;
  dst.flow_end_s[at] = src.flow_end_s[i];
  dst.throughput[at] = src.throughput[i];
  dst.algo_calc[at] = src.algo_calc[i];
  dst.stddev[at] = src.stddev[i];
  if (src.anomaly != nullptr) dst.anomaly[at] = src.anomaly[i];
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

// Stage 0 alone (tad_aggregate) needs no grid at all: the sorted unique list IS the result.  Columns out, plus the job
// counters (keys = key changes in the sorted list) and the points' (n, mean, M2) partials in k_moments' fixed order
// (strided per thread, shuffle tree, per-block partials; every point enters as (1, value, 0)).
__global__ __launch_bounds__(kSpBlock) void k_sparse_points_out(const unsigned long long *__restrict__ ucomp, const unsigned long long *__restrict__ uval,
                                                               uint64_t P, int64_t t0, unsigned long long *__restrict__ out_key,
                                                               long long *__restrict__ out_t, unsigned long long *__restrict__ out_val,
                                                               Moments *__restrict__ partials, DevCounters *ctr) {
  Moments acc{0.0, 0.0, 0.0};
  unsigned long long keys = 0, pts = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x; i < P; i += (uint64_t)gridDim.x * kSpBlock) {
    const unsigned long long c = ucomp[i], v = uval[i];
    out_key[i] = c >> 32;
    out_t[i] = (long long)(t0 + (int64_t)(c & 0xffffffffull));
    out_val[i] = v;
    acc = chan_merge(acc, Moments{1.0, (double)v, 0.0});
    pts++;
    keys += (i == 0 || (ucomp[i - 1] >> 32) != (c >> 32)) ? 1u : 0u;
  }
  for (int d = 32; d >= 1; d >>= 1) { pts += __shfl_down(pts, d); keys += __shfl_down(keys, d); }
  if ((threadIdx.x & 63) == 0 && pts) { atomicAdd(&ctr->n_points, pts); atomicAdd(&ctr->n_keys, keys); }
  for (int d = 1; d < 64; d <<= 1) {
    Moments o{__shfl_xor(acc.n, d), __shfl_xor(acc.mean, d), __shfl_xor(acc.m2, d)};
    acc = (threadIdx.x & d) ? chan_merge(o, acc) : chan_merge(acc, o);
  }
  __shared__ Moments s_m[kSpBlock / 64];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    Moments a = s_m[0];
    for (int w = 1; w < kSpBlock / 64; ++w) a = chan_merge(a, s_m[w]);
    partials[blockIdx.x] = a;
  }
}

void launch_sparse_points_out(hipStream_t s, const unsigned long long *ucomp, const unsigned long long *uval, uint64_t P, int64_t t0,
                              unsigned long long *out_key, long long *out_t, unsigned long long *out_val, Moments *partials, DevCounters *ctr) {
  hipLaunchKernelGGL(k_sparse_points_out, dim3(kMomentBlocks), dim3(kSpBlock), 0, s, ucomp, uval, P, t0, out_key, out_t, out_val, partials, ctr);
}

uint32_t sparse_class_count(uint32_t tmax) { return tmax ? sparse_class_of(tmax) + 1 : 0; }

void launch_sparse_len(hipStream_t s, const unsigned long long *ucomp, uint64_t P, const uint32_t *first, uint32_t *len) {
  if (P == 0) return;
  hipLaunchKernelGGL(k_sparse_len, dim3((unsigned)((P + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, ucomp, P, first, len);
}

void launch_sparse_class_counts(hipStream_t s, const uint32_t *len, uint64_t K, uint32_t c, uint32_t *member, uint32_t *pts) {
  if (K == 0) return;
  hipLaunchKernelGGL(k_sparse_class_counts, dim3((unsigned)((K + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, len, K, c, member, pts);
}

void launch_sparse_class_columns(hipStream_t s, const unsigned long long *ucomp, const unsigned long long *uval, uint64_t P, const uint32_t *first,
                                 const uint32_t *len, uint32_t c, const unsigned long long *key_off, const unsigned long long *pt_off, int64_t t0,
                                 unsigned long long *out_key, long long *out_t, unsigned long long *out_val, uint32_t *keymap) {
  if (P == 0) return;
  hipLaunchKernelGGL(k_sparse_class_columns, dim3((unsigned)((P + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, ucomp, uval, P, first, len, c, key_off,
                     pt_off, t0, out_key, out_t, out_val, keymap);
}

void launch_class_count_rows(hipStream_t s, const unsigned long long *row_key, uint64_t R, const uint32_t *keymap, uint32_t *cnt,
                             unsigned long long *first_row) {
  if (R == 0) return;
  hipLaunchKernelGGL(k_class_count_rows, dim3((unsigned)((R + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, row_key, R, keymap, cnt, first_row);
}

void launch_class_gather(hipStream_t s, OutRows src, uint64_t R, const uint32_t *keymap, const unsigned long long *off,
                         const unsigned long long *first_row, OutRows dst) {
  if (R == 0) return;
  hipLaunchKernelGGL(k_class_gather, dim3((unsigned)((R + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, s, src, R, keymap, off, first_row, dst);
}

}  // namespace tad
