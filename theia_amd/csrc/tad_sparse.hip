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
#include "tad_internal.h"

namespace tad {

static constexpr int kSpBlock = 256;

// composite sort key of every (row, key) slot: key << 32 | (t - t0); slots that are filtered out get the key field K (one past the
// last valid key: they sort behind every point and the reduction drops them)
__global__ __launch_bounds__(kSpBlock) void k_sparse_keys(const uint64_t *__restrict__ key, const uint64_t *__restrict__ key2,
                                                         const int64_t *__restrict__ t_end, const int64_t *__restrict__ t_start,
                                                         const uint64_t *__restrict__ value, uint64_t n, uint64_t K, RowFilter f, int64_t t0,
                                                         uint64_t span, unsigned long long *__restrict__ comp, unsigned long long *__restrict__ vals,
                                                         DevCounters *ctr) {
  uint32_t err = 0;
  unsigned long long used = 0;
  const int nk = key2 != nullptr ? 2 : 1;
  // grid-stride: the job counter gets ONE atomic per workgroup — one per wavefront of a 1e8-row table is 1.5e6 atomics on one address,
  // which serialise at ~12 ns each (the kernel took 18.8 ms for 4 GB of traffic, profiles/r4_v2_sparse_scale_kernel_stats.csv)
  for (uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kSpBlock) {
    const int64_t te = t_end[i];
    bool kept = true;
    if (f.end_time != 0 && !(te < f.end_time)) kept = false;                                       // anomaly_detection.py:584-586
    if (f.start_time != 0 && t_start != nullptr && !(t_start[i] >= f.start_time)) kept = false;    // :581-583
    const uint64_t dt = (uint64_t)te - (uint64_t)t0;
    const uint64_t v = value[i];
    for (int h = 0; h < nk; ++h) {
      const uint64_t k = h == 0 ? key[i] : key2[i];
      unsigned long long c = (unsigned long long)K << 32;
      if (kept && k != TAD_KEY_SKIP) {
        if (k >= K) err |= DEV_ERR_KEY_RANGE;
        else if (dt > span) err |= DEV_ERR_OFF_LATTICE;     // te < t0 or beyond the lattice's last bucket (the sort covers bit_width(span) time bits): the lattice (hint / sample) was wrong
        else { c = ((unsigned long long)k << 32) | dt; used++; }
      }
      comp[i * nk + h] = c;
      vals[i * nk + h] = v;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) { used += __shfl_down(used, d); err |= __shfl_down(err, d); }
  __shared__ unsigned long long s_used[kSpBlock / 64];
  __shared__ uint32_t s_err[kSpBlock / 64];
  if ((threadIdx.x & 63) == 0) { s_used[threadIdx.x >> 6] = used; s_err[threadIdx.x >> 6] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kSpBlock / 64; ++w) { used += s_used[w]; err |= s_err[w]; }
    if (used) atomicAdd(&ctr->rows_used, used);
    if (err) atomicOr(&ctr->err, err);
  }
}

// first[k] = index of the key's first point in the sorted unique list
// (P_dev: the number of points, still on the device, so that the host fetches the point count and the longest series with ONE
// round trip; grid-stride loops over a bounded grid)
__global__ __launch_bounds__(kSpBlock) void k_sparse_first(const unsigned long long *__restrict__ ucomp, const unsigned long long *__restrict__ P_dev,
                                                          uint32_t *__restrict__ first) {
  const uint64_t P = *P_dev;
  for (uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x; i < P; i += (uint64_t)gridDim.x * kSpBlock) {
    const uint32_t k = (uint32_t)(ucomp[i] >> 32);
    if (i == 0 || (uint32_t)(ucomp[i - 1] >> 32) != k) first[k] = (uint32_t)i;
  }
}

// longest series (one atomic per workgroup: one per wavefront made this kernel 6 ms at 3.3e7 points)
__global__ __launch_bounds__(kSpBlock) void k_sparse_tmax(const unsigned long long *__restrict__ ucomp, const unsigned long long *__restrict__ P_dev,
                                                         const uint32_t *__restrict__ first, unsigned int *__restrict__ tmax) {
  const uint64_t P = *P_dev;
  unsigned int n = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kSpBlock + threadIdx.x; i < P; i += (uint64_t)gridDim.x * kSpBlock) {
    const uint32_t k = (uint32_t)(ucomp[i] >> 32);
    if (i + 1 == P || (uint32_t)(ucomp[i + 1] >> 32) != k) {
      const unsigned int len = (unsigned int)(i - first[k] + 1);
      n = len > n ? len : n;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) { const unsigned int o = __shfl_down(n, d); n = o > n ? o : n; }
  __shared__ unsigned int s_n[kSpBlock / 64];
  if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kSpBlock / 64; ++w) n = s_n[w] > n ? s_n[w] : n;
    if (n) atomicMax(tmax, n);
  }
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

// ------------------------------------------------------------------------------------------------
// The sort: least-significant-digit radix sort of the (comp, value) pairs, hand-written for gfx950.
//
// Only the bits that can differ are sorted: the virtual key of a slot is (key << tb) | (t - t0) with tb = bit_width(span) time
// bits and kb = bit_width(K) key bits (K itself is the key field of the filtered-out slots), cut into ceil((kb + tb) / 8) digits of
// <= 8 bits — 1e6 connections over a day of seconds: 37 bits, five passes, where a 64-bit library sort runs eight.
// One pass = three steps, all of them plain streaming kernels without cross-workgroup waiting:
//   k_rs_hist     per tile of 4096 slots: digit histogram in LDS (8 B/slot read) -> counts[digit][tile]
//   launch_scan   exclusive scan of counts in digit-major order = the global position of every (digit, tile) run
//   k_rs_scatter  per tile: every wavefront owns 512 consecutive slots and ranks them item by item — the lanes holding the same
//                 digit find each other with one ballot per digit bit, a per-wavefront digit counter in LDS carries the rank from
//                 item to item — then the tile is laid out by digit in LDS and copied out run by run with consecutive lanes on
//                 consecutive slots (a digit's run of a tile is ~16 records: 128-byte pieces per column).  Ranks follow the slot
//                 order (lane order inside an item, items in order, wavefronts in order, tiles in order), so every pass is
//                 stable, which is what makes the digits compose.
// 40 B per slot and pass (8 histogram + 16 in + 16 out).  Then the reduction of equal (key, time) runs:
//   k_rs_heads    per tile: number of run heads among the valid slots -> launch_scan -> the output position of every run
//   k_rs_zero     the (at most one per tile) output values that are assembled from several tiles start at the operator's identity
//   k_rs_reduce   per tile in LDS: a head folds its run; a run that lies inside one tile is stored, the pieces of a run that
//                 crosses tile boundaries meet in the output with 64-bit integer atomics (add wraps mod 2^64 / unsigned max: commutative,
//                 so the aggregates are bit-identical to the dense path's and to ClickHouse's)
// ------------------------------------------------------------------------------------------------
static constexpr int kRsThreads = 512;                       // 8 wavefronts
static constexpr int kRsWaves = kRsThreads / 64;
static constexpr int kRsItems = 8;                           // slots per thread
static constexpr uint32_t kRsTile = kRsThreads * kRsItems;   // 4096 slots per workgroup
static constexpr uint32_t kRsRadix = 256;

__device__ __forceinline__ uint32_t rs_digit(unsigned long long c, int tb, int shift, uint32_t mask) {
  // (tb == 32: the composite key is the virtual key already; a 64-bit shift by 32 of the high half would be fine too, but keep it branch-free)
  const unsigned long long v = ((c >> 32) << tb) | (c & 0xffffffffull);
  return (uint32_t)(v >> shift) & mask;
}

__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const unsigned long long *__restrict__ comp, uint64_t N, int tb, int shift, uint32_t mask,
                                                         uint32_t *__restrict__ counts, uint32_t NB) {
  __shared__ uint32_t hist[kRsRadix];
  if (threadIdx.x < kRsRadix) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
  unsigned long long c[kRsItems];
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint64_t j = base + (uint64_t)i * kRsThreads + threadIdx.x;
    c[i] = j < N ? comp[j] : 0ull;
  }
#pragma unroll
  for (int i = 0; i < kRsItems; ++i)
    if (base + (uint64_t)i * kRsThreads + threadIdx.x < N) atomicAdd(&hist[rs_digit(c[i], tb, shift, mask)], 1u);
  __syncthreads();
  if (threadIdx.x <= mask) counts[(size_t)threadIdx.x * NB + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan of one value per thread over the workgroup (kRsThreads threads); s_w: kRsWaves + 1 words of LDS
__device__ __forceinline__ uint32_t rs_block_excl_scan(uint32_t x, uint32_t *s_w, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = x;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < kRsWaves; ++w) {
    if (w < wave) base += s_w[w];
    tot += s_w[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - x;
}

__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const unsigned long long *__restrict__ comp_in, const unsigned long long *__restrict__ val_in,
                                                            unsigned long long *__restrict__ comp_out, unsigned long long *__restrict__ val_out,
                                                            uint64_t N, int tb, int shift, int width, const unsigned long long *__restrict__ off, uint32_t NB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  unsigned long long *s_comp = reinterpret_cast<unsigned long long *>(rs_smem);
  unsigned long long *s_val = s_comp + kRsTile;
  unsigned long long *s_gofs = s_val + kRsTile;                       // [256] global slot of the digit's run of this tile, minus its tile position
  uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_gofs + kRsRadix); // [waves][256]: per-wavefront digit counters, then their exclusive prefix over the wavefronts
  uint32_t *s_base = s_hist + kRsWaves * kRsRadix;                    // [256] first tile position of the digit
  uint32_t *s_w = s_base + kRsRadix;                                  // scan scratch
  const uint32_t mask = (1u << width) - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < kRsWaves * kRsRadix; i += kRsThreads) s_hist[i] = 0;
  const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
  const uint32_t n_tile = base + kRsTile <= N ? kRsTile : (uint32_t)(N - base);
  unsigned long long c[kRsItems], v[kRsItems];
  uint32_t rr[kRsItems];
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {   // wavefront w owns the slots [w * 512, w * 512 + 512) of the tile, item i its i-th 64
    const uint32_t j = (uint32_t)wave * (64 * kRsItems) + (uint32_t)i * 64 + (uint32_t)lane;
    const bool in = j < n_tile;
    c[i] = in ? comp_in[base + j] : 0ull;
    v[i] = in ? val_in[base + j] : 0ull;
  }
  __syncthreads();
  uint32_t *my_hist = s_hist + wave * kRsRadix;
  const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint32_t j = (uint32_t)wave * (64 * kRsItems) + (uint32_t)i * 64 + (uint32_t)lane;
    const bool in = j < n_tile;
    const uint32_t d = rs_digit(c[i], tb, shift, mask);
    unsigned long long peers = __ballot(in);
    for (int b = 0; b < width; ++b) {          // the lanes of this item that hold the same digit
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t old = in ? my_hist[d] : 0u;            // the same word for all peers (LDS broadcast)
    __builtin_amdgcn_wave_barrier();                      // every peer has read before the first of them adds the item's count
    if (in && (peers & lt_mask) == 0ull) my_hist[d] = old + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rr[i] = old + (uint32_t)__popcll(peers & lt_mask);
  }
  __syncthreads();
  // per digit: counts of the wavefronts -> exclusive prefix over the wavefronts, tile total -> exclusive scan over the digits
  uint32_t tot_d = 0;
  if (threadIdx.x < kRsRadix) {
    for (int w = 0; w < kRsWaves; ++w) {
      const uint32_t t = s_hist[w * kRsRadix + threadIdx.x];
      s_hist[w * kRsRadix + threadIdx.x] = tot_d;
      tot_d += t;
    }
  }
  uint32_t tile_total;
  const uint32_t ex = rs_block_excl_scan(tot_d, s_w, &tile_total);
  if (threadIdx.x < kRsRadix) {
    s_base[threadIdx.x] = ex;
    s_gofs[threadIdx.x] = threadIdx.x <= mask ? off[(size_t)threadIdx.x * NB + blockIdx.x] - ex : 0ull;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint32_t j = (uint32_t)wave * (64 * kRsItems) + (uint32_t)i * 64 + (uint32_t)lane;
    if (j < n_tile) {
      const uint32_t d = rs_digit(c[i], tb, shift, mask);
      const uint32_t pos = s_base[d] + my_hist[d] + rr[i];
      s_comp[pos] = c[i];
      s_val[pos] = v[i];
    }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < n_tile; j += kRsThreads) {   // consecutive lanes -> consecutive slots of a digit's run
    const unsigned long long cc = s_comp[j];
    const unsigned long long dst = s_gofs[rs_digit(cc, tb, shift, mask)] + j;
    comp_out[dst] = cc;
    val_out[dst] = s_val[j];
  }
}

// run heads among the valid slots of each tile (valid: key field < K; the filtered-out slots carry K and sort last)
__global__ __launch_bounds__(kRsThreads) void k_rs_heads(const unsigned long long *__restrict__ comp, uint64_t N, uint64_t K, uint32_t *__restrict__ cnt) {
  __shared__ uint32_t s_w[kRsWaves];
  const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
  uint32_t h = 0;
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint64_t j = base + (uint64_t)i * kRsThreads + threadIdx.x;
    if (j < N) {
      const unsigned long long c = comp[j];
      if ((c >> 32) < K && (j == 0 || comp[j - 1] != c)) ++h;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) h += __shfl_down(h, d);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kRsWaves; ++w) t += s_w[w];
    cnt[blockIdx.x] = t;
  }
}

// a run that continues into the next tile is assembled in the output by atomics: its value starts at 0 (the identity of wrapping add
// and of unsigned max).  off[b + 1] - 1 = the run the last slot of tile b belongs to.
__global__ __launch_bounds__(256) void k_rs_zero(const unsigned long long *__restrict__ comp, uint64_t N, uint64_t K, const unsigned long long *__restrict__ off,
                                                  uint32_t NB, unsigned long long *__restrict__ out_val, unsigned long long *__restrict__ num_runs) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b == 0) *num_runs = off[NB];
  if (b >= NB) return;
  const uint64_t end = (uint64_t)(b + 1) * kRsTile;
  if (end >= N) return;
  const unsigned long long c = comp[end - 1];
  if ((c >> 32) < K && comp[end] == c) out_val[off[b + 1] - 1] = 0ull;
}

template <bool OPMAX>
__global__ __launch_bounds__(kRsThreads) void k_rs_reduce(const unsigned long long *__restrict__ comp, const unsigned long long *__restrict__ val, uint64_t N,
                                                           uint64_t K, const unsigned long long *__restrict__ off, unsigned long long *__restrict__ out_comp,
                                                           unsigned long long *__restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  unsigned long long *s_comp = reinterpret_cast<unsigned long long *>(rs_smem);
  unsigned long long *s_val = s_comp + kRsTile;
  __shared__ uint32_t s_w[kRsWaves + 1];
  const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
  const uint32_t n_tile = base + kRsTile <= N ? kRsTile : (uint32_t)(N - base);
  for (uint32_t j = threadIdx.x; j < n_tile; j += kRsThreads) { s_comp[j] = comp[base + j]; s_val[j] = val[base + j]; }
  const unsigned long long prev = base ? comp[base - 1] : 0ull;
  const bool has_next = base + n_tile < N;
  const unsigned long long next = has_next ? comp[base + n_tile] : 0ull;
  __syncthreads();
  auto is_head = [&](uint32_t j) -> bool {
    const unsigned long long c = s_comp[j];
    if ((c >> 32) >= K) return false;
    return j ? s_comp[j - 1] != c : (base == 0 || prev != c);
  };
  // thread t owns the slots [t * 8, t * 8 + 8): heads before them in the tile
  const uint32_t j0 = threadIdx.x * kRsItems;
  uint32_t mine = 0;
#pragma unroll
  for (int i = 0; i < kRsItems; ++i)
    if (j0 + i < n_tile && is_head(j0 + i)) ++mine;
  uint32_t total;
  uint32_t before = rs_block_excl_scan(mine, s_w, &total);
  const unsigned long long o0 = off[blockIdx.x];
  auto fold = [&](uint32_t j, uint32_t *end) -> unsigned long long {   // the run piece that starts at slot j of this tile
    const unsigned long long c = s_comp[j];
    unsigned long long acc = s_val[j];
    uint32_t k = j + 1;
    for (; k < n_tile && s_comp[k] == c; ++k) acc = OPMAX ? (s_val[k] > acc ? s_val[k] : acc) : acc + s_val[k];
    *end = k;
    return acc;
  };
  auto combine = [&](unsigned long long idx, unsigned long long acc) {
    if (OPMAX) __hip_atomic_fetch_max(out_val + idx, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(out_val + idx, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint32_t j = j0 + i;
    if (j < n_tile && is_head(j)) {
      uint32_t end;
      const unsigned long long acc = fold(j, &end);
      const unsigned long long idx = o0 + before;
      ++before;
      out_comp[idx] = s_comp[j];
      if (end == n_tile && has_next && next == s_comp[j]) combine(idx, acc);   // continues in the next tile (k_rs_zero prepared the word)
      else out_val[idx] = acc;
    }
  }
  // the tile starts inside a run that began in an earlier tile: its piece joins that run's output word
  if (threadIdx.x == 0 && base != 0 && n_tile != 0 && (s_comp[0] >> 32) < K && prev == s_comp[0]) {
    uint32_t end;
    const unsigned long long acc = fold(0, &end);
    combine(o0 - 1, acc);
  }
}

static inline int bit_width64(uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

struct RsPlan {
  int tb, np;
  int shift[8], width[8];
  uint32_t NB;
};

static RsPlan rs_plan(uint64_t slots, uint64_t K, uint64_t span) {
  RsPlan p{};
  p.tb = bit_width64(span);
  if (p.tb > 32) p.tb = 32;
  const int bits = p.tb + bit_width64(K);   // K itself must sort (the filtered-out slots)
  p.np = (bits + 7) / 8;
  if (p.np < 1) p.np = 1;
  int at = 0, left = bits;
  for (int i = 0; i < p.np; ++i) {
    int w = (left + (p.np - i) - 1) / (p.np - i);   // balanced digit widths
    if (w < 1) w = 1;
    p.shift[i] = at; p.width[i] = w;
    at += w; left -= w;
  }
  p.NB = (uint32_t)((slots + kRsTile - 1) / kRsTile);
  return p;
}

// temp: counts u32 [256 * NB] | off u64 [256 * NB + 1] | scan scratch | head counts u32 [NB] | head offsets u64 [NB + 1]
struct RsTemp { size_t counts, off, scratch, hcnt, hoff, total; };
static RsTemp rs_temp_layout(uint64_t slots) {
  const size_t NB = (size_t)((slots + kRsTile - 1) / kRsTile);
  const size_t m = (size_t)kRsRadix * NB;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  RsTemp t;
  size_t at = 0;
  t.counts = at; at = up(at + m * 4);
  t.off = at; at = up(at + (m + 1) * 8);
  t.scratch = at; at = up(at + scan_scratch_elems(m ? m : 1) * 8);
  t.hcnt = at; at = up(at + (NB ? NB : 1) * 4);
  t.hoff = at; at = up(at + (NB + 1) * 8);
  t.total = at;
  return t;
}

size_t sparse_sort_temp_bytes(uint64_t slots) { return rs_temp_layout(slots).total + 256; }   // 256-byte multiple: the caller places two counters right behind it

// rows -> sorted unique (key, time) points with aggregated values in comp_a / val_a (device), their number in *num_runs (device).
// span = the largest t - t0 the lattice allows ((n_buckets - 1) * step): rows beyond it raise DEV_ERR_OFF_LATTICE.
int launch_sparse_group(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end, const int64_t *t_start,
                        const uint64_t *value, uint64_t n, uint64_t K, RowFilter f, int64_t t0, uint64_t span, bool op_max, unsigned long long *comp_a,
                        unsigned long long *val_a, unsigned long long *comp_b, unsigned long long *val_b, void *temp, size_t temp_bytes,
                        unsigned long long *num_runs, DevCounters *ctr) {
  const uint64_t slots = n * (key2 != nullptr ? 2 : 1);
  if (slots == 0 || slots >= (1ull << 32) || K > 0xFFFFFFFFull) return -1;
  if (span > 0xFFFFFFFFull) span = 0xFFFFFFFFull;
  const RsPlan pl = rs_plan(slots, K, span);
  const RsTemp tl = rs_temp_layout(slots);
  if (temp_bytes < tl.total) return -1;
  unsigned char *tp = static_cast<unsigned char *>(temp);
  uint32_t *counts = reinterpret_cast<uint32_t *>(tp + tl.counts);
  unsigned long long *off = reinterpret_cast<unsigned long long *>(tp + tl.off);
  unsigned long long *scratch = reinterpret_cast<unsigned long long *>(tp + tl.scratch);
  uint32_t *hcnt = reinterpret_cast<uint32_t *>(tp + tl.hcnt);
  unsigned long long *hoff = reinterpret_cast<unsigned long long *>(tp + tl.hoff);
  // an odd number of moves (the passes + the reduction) ends in the a buffers: start in a when the number of passes is odd
  unsigned long long *ca = (pl.np & 1) ? comp_a : comp_b, *va = (pl.np & 1) ? val_a : val_b;
  unsigned long long *cb = (pl.np & 1) ? comp_b : comp_a, *vb = (pl.np & 1) ? val_b : val_a;
  const uint64_t kb = (n + kSpBlock - 1) / kSpBlock;
  hipLaunchKernelGGL(k_sparse_keys, dim3((unsigned)(kb < 8192 ? kb : 8192)), dim3(kSpBlock), 0, s, key, key2, t_end, t_start, value, n, K,
                     f, t0, span, ca, va, ctr);
  const size_t lds = (size_t)kRsTile * 16 + kRsRadix * 8 + (size_t)kRsWaves * kRsRadix * 4 + kRsRadix * 4 + 64;
  allow_big_lds(reinterpret_cast<const void *>(k_rs_scatter), lds);
  for (int p = 0; p < pl.np; ++p) {
    const uint32_t mask = (1u << pl.width[p]) - 1u;
    hipLaunchKernelGGL(k_rs_hist, dim3(pl.NB), dim3(kRsThreads), 0, s, ca, slots, pl.tb, pl.shift[p], mask, counts, pl.NB);
    launch_scan(s, counts, off, (uint64_t)(mask + 1) * pl.NB, scratch);
    hipLaunchKernelGGL(k_rs_scatter, dim3(pl.NB), dim3(kRsThreads), lds, s, ca, va, cb, vb, slots, pl.tb, pl.shift[p], pl.width[p], off, pl.NB);
    unsigned long long *t = ca; ca = cb; cb = t;
    t = va; va = vb; vb = t;
  }
  hipLaunchKernelGGL(k_rs_heads, dim3(pl.NB), dim3(kRsThreads), 0, s, ca, slots, K, hcnt);
  launch_scan(s, hcnt, hoff, pl.NB, scratch);
  hipLaunchKernelGGL(k_rs_zero, dim3((pl.NB + 255) / 256), dim3(256), 0, s, ca, slots, K, hoff, pl.NB, vb, num_runs);
  const size_t rlds = (size_t)kRsTile * 16;
  if (op_max) {
    allow_big_lds(reinterpret_cast<const void *>(k_rs_reduce<true>), rlds);
    hipLaunchKernelGGL(k_rs_reduce<true>, dim3(pl.NB), dim3(kRsThreads), rlds, s, ca, va, slots, K, hoff, cb, vb);
  } else {
    allow_big_lds(reinterpret_cast<const void *>(k_rs_reduce<false>), rlds);
    hipLaunchKernelGGL(k_rs_reduce<false>, dim3(pl.NB), dim3(kRsThreads), rlds, s, ca, va, slots, K, hoff, cb, vb);
  }
  return (cb == comp_a && hipGetLastError() == hipSuccess) ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// Big sparse tables: partition by key block, then sort in LDS (round 4).  The LSD sort above moves every 16-byte (composite, value) pair
// through HBM once per digit — five passes for 1e6 connections over a day of seconds, ~290 B per row.  Here the rows go through the dense
// path's passes A and B first (exact histogram per key bin, write-combining partition pass: 8-byte records value << cell_bits |
// (bucket * KP + key-in-block), tad_stage0_part.hip), which leaves every key block's records contiguous, and then
//   k_ss_plan   a wavefront per key block: the block's bin totals (pass A's histogram summed over its workgroups) are grouped greedily
//               into ROUNDS of <= kSsCap records — whole bins, so that every key's records are in exactly one round
//   k_ss_split  a workgroup per key block: one more stream over the block's records moves each to its round's place in a second
//               record buffer (the rounds' sizes are known: exact positions, one LDS cursor per round, the lanes of a wavefront that
//               hold the same round take their places together).  The first form had every round's workgroup filter the whole block
//               instead: eight reads of the block, of which L2 caught too few (k_ss_sort 2.5 ms, profiles/r4_v24_*)
//   k_ss_sort   a workgroup per (block, round): loads its records into LDS as value << cell_bits | (key-in-round << tbits | bucket), sorts them
//               there by that key — least-significant-digit passes with the ballot ranking of k_rs_scatter, the items in registers between
//               the passes —, folds the runs of equal (key, bucket) with the job's operator (wrapping add / unsigned max) and writes the unique
//               points of the round, in order, to a staging area at (block start + records of the earlier bins): coalesced, through LDS
//   scan + k_ss_compact   the rounds' point counts -> positions; every round's stage is copied to its place in the final sorted list.
// The columns are read once (24 B/row), the records written and read twice more through HBM (2 x (8 + 8) B/row).
// A bin that alone exceeds a round (a heavy key), a value that does not fit the record, or a shape the plan refuses: the LSD sort runs.
// ------------------------------------------------------------------------------------------------
#if !defined(TAD_SS_THREADS)
#define TAD_SS_THREADS 1024
#endif
static constexpr int kSsThreads = TAD_SS_THREADS;             // of k_ss_sort (a measurement build may halve the workgroup: two per CU)
static constexpr int kSplitThreads = 1024;                     // of k_ss_split
static constexpr int kSsWaves = kSsThreads / 64;
static constexpr int kSsItems = 14;
static constexpr uint32_t kSsCap = kSsThreads * kSsItems;       // 14336 records (112 KB of LDS)
static constexpr uint32_t kSsSlice = 64 * kSsItems;             // slots a wavefront owns in a sorting pass

struct SsRound { uint32_t key0, key1, n, stage; };             // key-in-block range, records (from the histogram), staging offset

struct SsArgs {
  const unsigned long long *recs;
  const unsigned long long *part_start;
  const uint32_t *binhist;      // [G][nbins]
  uint32_t nbins, bins_per_part, nparts;
  int G, shift_bin, shift_part, cell_bits, tbits;
  uint64_t K;
  unsigned long long step;
  SsRound *rounds;              // [nparts][bins_per_part]
  uint32_t *n_rounds;           // [nparts]
  uint8_t *bin_round;           // [nparts][bins_per_part] round of every bin (k_ss_plan; needs <= 256 rounds per block: checked)
  uint32_t *round_fill;         // [nparts * bins_per_part] records k_ss_split placed (the histogram also counts rows off the lattice)
  unsigned long long *recs2;    // the records by round: round (p, r) at [rounds[p][r].stage, + fill)
  uint32_t *stage_rank;         // place of every staged point in its key's series
  uint32_t *seg_count;          // [nparts * bins_per_part] unique points of the round (0 = no such round)
  unsigned long long *stage_comp, *stage_val;
  unsigned long long *num_runs;  // += the round's unique points
  unsigned int *tmax;            // max= the longest series of the round (all of a key's points are in one round)
  DevCounters *ctr;
};

__global__ __launch_bounds__(256) void k_ss_plan(SsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ss_plan_smem[];
  uint32_t *ss_tot = reinterpret_cast<uint32_t *>(ss_plan_smem);   // [4][bins_per_part]
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 4u + wave;
  if (p >= A.nparts) return;                                   // (wavefront-uniform; no workgroup barrier below)
  uint32_t *tot = ss_tot + (size_t)wave * A.bins_per_part;
  const uint32_t b0 = p * A.bins_per_part;
  for (uint32_t b = lane; b < A.bins_per_part; b += 64) {
    uint32_t c = 0;
    if (b0 + b < A.nbins) {      // batches of independent loads (one at a time: 256 memory round trips per lane, 98 us)
      int g = 0;
      for (; g + 8 <= A.G; g += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = A.binhist[(size_t)(g + u) * A.nbins + b0 + b];
#pragma unroll
        for (int u = 0; u < 8; ++u) c += v[u];
      }
      for (; g < A.G; ++g) c += A.binhist[(size_t)g * A.nbins + b0 + b];
    }
    tot[b] = c;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  if (lane != 0) return;
  SsRound *out = A.rounds + (size_t)p * A.bins_per_part;
  uint8_t *br = A.bin_round + (size_t)p * A.bins_per_part;
  const uint32_t base = (uint32_t)A.part_start[p];
  uint32_t r = 0, acc = 0, first = 0, before = 0, err = 0;
  for (uint32_t b = 0; b < A.bins_per_part; ++b) {
    const uint32_t c = tot[b];
    if (c > kSsCap) err = DEV_ERR_SPARSE_ROUND;
    if (acc != 0 && acc + c > kSsCap) {
      out[r++] = SsRound{first << A.shift_bin, b << A.shift_bin, acc, base + before};
      before += acc; acc = 0;
    }
    if (acc == 0) first = b;
    acc += c;
    br[b] = (uint8_t)r;
    if (r > 255u) err = DEV_ERR_SPARSE_ROUND;
  }
  if (acc != 0) out[r++] = SsRound{first << A.shift_bin, A.bins_per_part << A.shift_bin, acc, base + before};
  A.n_rounds[p] = r;
  if (err) atomicOr(&A.ctr->err, err);
}

#if defined(TAD_SS_PROF)
// profiling build (tools/build_variants.py ssprof:TAD_SS_PROF; never the shipped library): shader-clock cycles of thread 0 per phase of k_ss_sort
__device__ unsigned long long g_ss_prof[8];
#define SS_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define SS_ADD(slot, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_ss_prof[slot], (b) - (a)); } while (0)
#else
#define SS_T(v)
#define SS_ADD(slot, a, b)
#endif

__global__ __launch_bounds__(kSplitThreads) void k_ss_split(SsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ss_split_smem[];
  const uint32_t R = A.bins_per_part;
  uint32_t *s_cur = reinterpret_cast<uint32_t *>(ss_split_smem);     // [R] records placed per round
  uint32_t *s_at = s_cur + R;                                          // [R] the round's place in recs2
  uint8_t *s_tab = reinterpret_cast<uint8_t *>(s_at + R);              // [R] bin -> round
  const uint32_t p = blockIdx.x;
  const uint32_t nr = A.n_rounds[p];
  for (uint32_t i = threadIdx.x; i < R; i += kSplitThreads) {
    s_cur[i] = 0;
    s_at[i] = i < nr ? A.rounds[(size_t)p * R + i].stage : 0u;
    s_tab[i] = A.bin_round[(size_t)p * R + i];
  }
  __syncthreads();
  int rbits = 0;
  while (nr > 1 && ((nr - 1u) >> rbits) != 0) ++rbits;
  const int lane = threadIdx.x & 63;
  const uint32_t cell_mask = (1u << A.cell_bits) - 1u, kp_mask = (1u << A.shift_part) - 1u;
  const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
  const uint64_t lo = A.part_start[p], hi = A.part_start[p + 1];
  constexpr int kL = 4;
  constexpr uint32_t kSet = (uint32_t)kSplitThreads * 2u * kL;
  const uint32_t sets = (uint32_t)((hi - lo + kSet - 1) / kSet);
  auto load_set = [&](ulonglong2 (&x)[kL], uint32_t set) {
#pragma unroll
    for (int u = 0; u < kL; ++u) {
      const uint64_t i = lo + (uint64_t)set * kSet + (uint64_t)u * (2u * kSplitThreads) + 2u * threadIdx.x;
      x[u] = (set < sets && i < hi) ? *reinterpret_cast<const ulonglong2 *>(A.recs + i) : ulonglong2{~0ull, ~0ull};
    }
  };
  auto process = [&](const ulonglong2 (&x)[kL]) {
#pragma unroll
    for (int u = 0; u < 2 * kL; ++u) {
      const unsigned long long rc = (u & 1) ? x[u >> 1].y : x[u >> 1].x;
      const uint32_t cell = (uint32_t)rc & cell_mask;
      const bool take = rc != ~0ull && cell != cell_mask;
      const uint32_t r = take ? s_tab[(cell & kp_mask) >> A.shift_bin] : 0u;
      unsigned long long peers = __ballot(take);
      if (peers == 0ull) continue;                            // wavefront-uniform
      for (int b = 0; b < rbits; ++b) {                        // the lanes that hold the same round
        const bool bit = (r >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      const int leader = __ffsll((long long)peers) - 1;        // (lanes that do not take: their own `peers` is never used)
      uint32_t at = 0;
      if (take && lane == leader) at = atomicAdd(&s_cur[r], (uint32_t)__popcll(peers));
      at = __shfl(at, take ? leader : lane);
      if (take) A.recs2[(size_t)s_at[r] + at + (uint32_t)__popcll(peers & lt_mask)] = rc;
    }
  };
  ulonglong2 xa[kL], xb[kL];
  load_set(xa, 0);
  for (uint32_t set = 0; set < sets; set += 2) {
    load_set(xb, set + 1);
    process(xa);
    load_set(xa, set + 2);
    if (set + 1 < sets) process(xb);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nr; i += kSplitThreads) A.round_fill[(size_t)p * R + i] = s_cur[i];
}

// LDS barrier: LDS traffic only (the kernel's global stores are never read back by the workgroup; __syncthreads() would wait for them:
// the three store phases of the output each cost a memory round trip, profiles/r4_v26_ss_sort_phases.log)
__device__ __forceinline__ void ss_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

static constexpr uint32_t kSsMsdBits = 13;                       // 8192 buckets of ~1.8 records in a full round (12 bits: the ranks inside the buckets cost 45k of a round's 90k cycles)
static constexpr uint32_t kSsMaxBucket = 32;                     // a larger bucket (many points of one key inside one time window): the stable LSD passes sort the round

template <bool OPMAX>
__global__ __launch_bounds__(kSsThreads) void k_ss_sort(SsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
  unsigned long long *rec = reinterpret_cast<unsigned long long *>(ss_smem);             // [kSsCap]
  uint32_t *s_hist = reinterpret_cast<uint32_t *>(rec + kSsCap);                         // [waves][256] = [4096]
  uint32_t *s_base = s_hist + (1u << kSsMsdBits);                                         // [256]
  __shared__ uint32_t s_w[kSsWaves + 1];
  __shared__ uint32_t s_best, s_maxb;
  // the rounds of a key block side by side (workgroups are dealt round-robin to the 8 XCDs)
  const uint32_t R = A.bins_per_part;
  const uint32_t slot = blockIdx.x >> 3;
  const uint32_t p = (slot / R) * 8u + (blockIdx.x & 7u), r = slot % R;
  if (p >= A.nparts || r >= A.n_rounds[p]) return;
  const SsRound rd = A.rounds[(size_t)p * R + r];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_best = 0; s_maxb = 0; }
  SS_T(t_0);
  // ---- the round's records (k_ss_split brought them together): LDS as value << cell_bits | (key-in-round << tbits | bucket) ----
  const uint32_t cell_mask = (1u << A.cell_bits) - 1u, kp_mask = (1u << A.shift_part) - 1u;
  const unsigned long long sk_mask = (1ull << A.cell_bits) - 1ull;
  const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
  const uint32_t n = A.round_fill[(size_t)p * R + r] < kSsCap ? A.round_fill[(size_t)p * R + r] : kSsCap;   // (<= the histogram's count <= kSsCap by the plan)
  int kbits = 0;
  while (((rd.key1 - rd.key0 - 1u) >> kbits) != 0) ++kbits;
  const int bits = A.tbits + kbits;
  const int mbits = bits < (int)kSsMsdBits ? bits : (int)kSsMsdBits;
  const int mshift = bits - mbits;
  unsigned long long c[kSsItems];
  for (uint32_t i = threadIdx.x; i < (1u << kSsMsdBits); i += kSsThreads) s_hist[i] = 0;
  {
    const unsigned long long *src = A.recs2 + rd.stage;
#pragma unroll
    for (int i = 0; i < kSsItems; ++i) {
      const uint32_t j = (uint32_t)i * kSsThreads + threadIdx.x;
      c[i] = j < n ? src[j] : 0ull;
    }
    ss_barrier();                                               // the bucket counters are zero
#pragma unroll
    for (int i = 0; i < kSsItems; ++i) {
      const uint32_t j = (uint32_t)i * kSsThreads + threadIdx.x;
      if (j < n) {
        const uint32_t cell = (uint32_t)c[i] & cell_mask;
        const unsigned long long x = ((c[i] >> A.cell_bits) << A.cell_bits) | ((unsigned long long)((cell & kp_mask) - rd.key0) << A.tbits) | (cell >> A.shift_part);
        rec[j] = x;                                             // (kept in LDS, not in registers, over the scan: 128 VGPRs at 1024 threads)
        atomicAdd(&s_hist[(uint32_t)((x & sk_mask) >> mshift)], 1u);
      }
    }
  }
  ss_barrier();
  SS_T(t_1);
  SS_ADD(0, t_0, t_1);
  // ---- sort by (key-in-round, bucket).  Most significant 12 bits first: a counting sort with LDS atomics (no order inside a bucket), then
  // every bucket — ~3.5 records — is put in order by counting (below).  The stable LSD passes with the ballot ranking of
  // k_rs_scatter cost 36 of a round's 57 us (240 VALU instructions per record); they remain for rounds with a crowded bucket. ----
  {
    // exclusive scan of the 4096 bucket counts, 4 per thread; the largest bucket
    constexpr int kBk = (1 << kSsMsdBits) / kSsThreads;
    uint32_t b4[kBk], sum = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < kBk; ++q) { b4[q] = s_hist[threadIdx.x * kBk + q]; sum += b4[q]; mx = b4[q] > mx ? b4[q] : mx; }
    uint32_t incl = sum;
    for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
    for (int dd = 32; dd >= 1; dd >>= 1) { const uint32_t y = __shfl_down(mx, dd); mx = y > mx ? y : mx; }
    if (lane == 63) s_w[wave] = incl;
    if (lane == 0 && mx) atomicMax(&s_maxb, mx);
    ss_barrier();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int q = 0; q < kBk; ++q) { s_hist[threadIdx.x * kBk + q] = run; run += b4[q]; }
    ss_barrier();
  }
  SS_T(t_1a);
  SS_ADD(5, t_1, t_1a);
  if (s_maxb <= kSsMaxBucket || mshift == 0) {                  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < kSsItems; ++i) {
      const uint32_t j = (uint32_t)i * kSsThreads + threadIdx.x;
      c[i] = j < n ? rec[j] : 0ull;
    }
    ss_barrier();
#pragma unroll
    for (int i = 0; i < kSsItems; ++i) {
      const uint32_t j = (uint32_t)i * kSsThreads + threadIdx.x;
      if (j < n) rec[atomicAdd(&s_hist[(uint32_t)((c[i] & sk_mask) >> mshift)], 1u)] = c[i];     // (the counter of bucket d ends at the start of bucket d + 1)
    }
    ss_barrier();
    SS_T(t_1b);
    SS_ADD(6, t_1a, t_1b);
    if (mshift != 0) {
      // order inside the buckets by COUNTING: a record's place = the bucket's records with a smaller key (or the same key further left).
      // One record per thread and step: the lanes of a wavefront walk buckets of similar sizes (an insertion sort per bucket, one thread per
      // bucket, paid the square of the largest bucket of every 64: 143k cycles per round against the LSD passes' 76k)
      uint32_t pos[kSsItems];
      const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(rec);
      const uint32_t skm = (uint32_t)sk_mask;
#pragma unroll
      for (int i = 0; i < kSsItems; ++i) {
        const uint32_t j = (uint32_t)i * kSsThreads + threadIdx.x;
        pos[i] = j;
        if (j < n) {
          c[i] = rec[j];
          const uint32_t key = (uint32_t)c[i] & skm;
          const uint32_t d = key >> mshift;
          const uint32_t b0 = d ? s_hist[d - 1] : 0u, b1 = s_hist[d];
          const unsigned long long me = ((unsigned long long)key << 32) | j;
          uint32_t rank = 0;
          for (uint32_t k = b0; k < b1; k += 4) {            // four independent LDS reads per trip (one at a time: a round trip each)
            uint32_t y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] = rec32[2 * (k + q < b1 ? k + q : b1 - 1)] & skm;
            // (key, slot) pairs compared as one 64-bit number; a slot past the bucket's end compares as the largest
#pragma unroll
            for (int q = 0; q < 4; ++q) rank += ((k + q < b1 ? (((unsigned long long)y[q] << 32) | (k + q)) : ~0ull) < me) ? 1u : 0u;
          }
          pos[i] = b0 + rank;
        }
      }
      ss_barrier();
      SS_T(t_1c);
      SS_ADD(7, t_1b, t_1c);
#pragma unroll
      for (int i = 0; i < kSsItems; ++i)
        if ((uint32_t)i * kSsThreads + threadIdx.x < n) rec[pos[i]] = c[i];
      ss_barrier();
    }
  } else {
    // stable LSD passes over the bits in use (digits of <= 8 bits), the items in registers between the passes
    const int np = (bits + 7) / 8;
    int shift = 0, left = bits;
    uint32_t rr[kSsItems];
    uint32_t *my_hist = s_hist + wave * kRsRadix;
    for (int ps = 0; ps < np; ++ps) {
      const int width = (left + (np - ps) - 1) / (np - ps);
      const uint32_t mask = (1u << width) - 1u;
      for (uint32_t i = threadIdx.x; i < kSsWaves * kRsRadix; i += kSsThreads) s_hist[i] = 0;
#pragma unroll
      for (int i = 0; i < kSsItems; ++i) {
        const uint32_t j = (uint32_t)wave * kSsSlice + (uint32_t)i * 64 + (uint32_t)lane;
        c[i] = j < n ? rec[j] : 0ull;
      }
      ss_barrier();
#pragma unroll
      for (int i = 0; i < kSsItems; ++i) {
        const uint32_t j = (uint32_t)wave * kSsSlice + (uint32_t)i * 64 + (uint32_t)lane;
        const bool in = j < n;
        const uint32_t d = (uint32_t)(c[i] >> shift) & mask;
        unsigned long long peers = __ballot(in);
        for (int b = 0; b < width; ++b) {
          const bool bit = (d >> b) & 1u;
          const unsigned long long m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const uint32_t old = in ? my_hist[d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (in && (peers & lt_mask) == 0ull) my_hist[d] = old + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rr[i] = old + (uint32_t)__popcll(peers & lt_mask);
      }
      ss_barrier();
      uint32_t tot_d = 0;
      if (threadIdx.x < kRsRadix) {
        for (int w = 0; w < kSsWaves; ++w) {
          const uint32_t t = s_hist[w * kRsRadix + threadIdx.x];
          s_hist[w * kRsRadix + threadIdx.x] = tot_d;
          tot_d += t;
        }
      }
      {   // exclusive scan of the digit totals (the first 256 threads = 4 wavefronts hold one each)
        uint32_t incl = tot_d;
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
        if (lane == 63) s_w[wave] = incl;
        ss_barrier();
        uint32_t before = 0;
        for (int w = 0; w < 4; ++w) if (w < wave) before += s_w[w];
        if (threadIdx.x < kRsRadix) s_base[threadIdx.x] = before + incl - tot_d;
      }
      ss_barrier();
#pragma unroll
      for (int i = 0; i < kSsItems; ++i) {
        const uint32_t j = (uint32_t)wave * kSsSlice + (uint32_t)i * 64 + (uint32_t)lane;
        if (j < n) {
          const uint32_t d = (uint32_t)(c[i] >> shift) & mask;
          rec[s_base[d] + my_hist[d] + rr[i]] = c[i];
        }
      }
      ss_barrier();
      shift += width; left -= width;
    }
  }
  SS_T(t_2);
  SS_ADD(1, t_1, t_2);
  // ---- fold the runs of equal (key, bucket).  Thread t owns the sorted slots [t * kSsItems, ...): one pass over them in registers; the
  // piece of a run that began in an earlier thread's slots (`lead`) is handed over through LDS and absorbed by the run's head ----
  const uint32_t j0 = threadIdx.x * kSsItems;
  uint32_t head_bits = 0, start_bits = 0;      // heads of (key, bucket) runs; those that are their key's first point
  unsigned long long hv[kSsItems];
  uint32_t hk[kSsItems];
  {
    unsigned long long *s_lead = reinterpret_cast<unsigned long long *>(s_hist);          // [threads] (the bucket counters are done)
    uint32_t *s_lflag = reinterpret_cast<uint32_t *>(s_lead + kSsThreads);                  // [threads] bit 0: a lead exists, bit 1: no head among the slots
#pragma unroll
    for (int i = 0; i < kSsItems; ++i) c[i] = j0 + i < n ? rec[j0 + i] : 0ull;
    const unsigned long long prev = (j0 != 0 && j0 < n) ? rec[j0 - 1] : 0ull;
    // heads: a slot whose key differs from its predecessor's
    {
      uint32_t before_sk = (uint32_t)(prev & sk_mask);
      bool have_prev = j0 != 0 && j0 < n;
#pragma unroll
      for (int i = 0; i < kSsItems; ++i) {
        hv[i] = 0; hk[i] = 0;
        if (j0 + i < n) {
          const uint32_t sk = (uint32_t)(c[i] & sk_mask);
          if (!have_prev || sk != before_sk) {
            head_bits |= 1u << i; hk[i] = sk;
            if (!have_prev || (sk >> A.tbits) != (before_sk >> A.tbits)) start_bits |= 1u << i;
          }
          before_sk = sk; have_prev = true;
        }
      }
    }
    // backwards: the fold of every run's piece inside my slots lands on the piece's first slot (a head, or slot 0 = the lead)
    unsigned long long racc = 0ull;
#pragma unroll
    for (int i = kSsItems - 1; i >= 0; --i) {
      if (j0 + i < n) {
        const unsigned long long v = c[i] >> A.cell_bits;
        const bool last_of_piece = i == kSsItems - 1 || j0 + i + 1 >= n || (head_bits & (2u << i)) != 0;
        racc = last_of_piece ? v : (OPMAX ? (v > racc ? v : racc) : v + racc);
        if (head_bits & (1u << i)) hv[i] = racc;
      }
    }
    const bool lead_has = j0 < n && (head_bits & 1u) == 0;
    s_lead[threadIdx.x] = racc;                          // (slot 0's piece)
    s_lflag[threadIdx.x] = (lead_has ? 1u : 0u) | (head_bits == 0 ? 2u : 0u);
    ss_barrier();
    if (head_bits != 0) {                                // my last head's run may go on in the following threads' slots
      unsigned long long more = 0ull;
      bool any = false;
      for (uint32_t tt = threadIdx.x + 1; tt < kSsThreads && tt * kSsItems < n; ++tt) {
        const uint32_t fl = s_lflag[tt];
        if (fl & 1u) { const unsigned long long v = s_lead[tt]; more = any ? (OPMAX ? (v > more ? v : more) : more + v) : v; any = true; }
        if (!(fl & 2u) || !(fl & 1u)) break;
      }
      if (any) {
        const int h = 31 - __clz((int)head_bits);
#pragma unroll
        for (int q = 0; q < kSsItems; ++q)
          if (q == h) hv[q] = OPMAX ? (more > hv[q] ? more : hv[q]) : hv[q] + more;
      }
    }
  }
  // positions of my points in the round's list (sum scan) and the list index + 1 of the last key start before them (max scan)
  uint32_t mine = (uint32_t)__popc(head_bits), incl = mine;
  for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
  if (lane == 63) s_w[wave] = incl;
  ss_barrier();                          // every thread has read its slots and the leads: the record area and the counter area are free
  uint32_t before = incl - mine, U = 0;
  for (int w = 0; w < kSsWaves; ++w) { if (w < wave) before += s_w[w]; U += s_w[w]; }
  uint32_t carry;
  {
    uint32_t last1 = 0, u = before;
#pragma unroll
    for (int i = 0; i < kSsItems; ++i)
      if (head_bits & (1u << i)) { if (start_bits & (1u << i)) last1 = u + 1; ++u; }
    uint32_t sc = last1;
    for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(sc, dd); if (lane >= dd && y > sc) sc = y; }
    carry = __shfl_up(sc, 1);
    if (lane == 0) carry = 0;
    if (lane == 63) s_hist[wave] = sc;
    ss_barrier();
    for (int w = 0; w < wave; ++w) { const uint32_t y = s_hist[w]; if (y > carry) carry = y; }
  }
  SS_T(t_3);
  SS_ADD(2, t_2, t_3);
  // ---- the round's unique points, in order, to the stage: composite, value, place in the key's series (= list index - index of the key's first
  // point); through LDS so that consecutive lanes store consecutive words.  All three at once when they fit the record area (20 B per point) ----
  const unsigned long long key_base = ((unsigned long long)p << A.shift_part) + rd.key0;
  const uint32_t t_mask = (1u << A.tbits) - 1u;
  const bool one_pass = (size_t)U * 20 <= (size_t)kSsCap * 8;        // workgroup-uniform
  unsigned long long *l_val = rec + (one_pass ? U : 0u);
  uint32_t *l_rank = reinterpret_cast<uint32_t *>(rec + (one_pass ? 2u * U : 0u));
  uint32_t best = 0;
  {
    uint32_t u = before;
#pragma unroll
    for (int i = 0; i < kSsItems; ++i)
      if (head_bits & (1u << i)) {
        if (start_bits & (1u << i)) carry = u + 1;
        const uint32_t rank = u + 1 - carry;
        best = rank + 1 > best ? rank + 1 : best;
        rec[u] = ((key_base + (hk[i] >> A.tbits)) << 32) | ((unsigned long long)(hk[i] & t_mask) * A.step);
        if (one_pass) { l_val[u] = hv[i]; l_rank[u] = rank; }
        hk[i] = rank;
        ++u;
      }
    for (int dd = 32; dd >= 1; dd >>= 1) { const uint32_t y = __shfl_down(best, dd); best = y > best ? y : best; }
    if (lane == 0 && best) atomicMax(&s_best, best);
  }
  ss_barrier();
  for (uint32_t j = threadIdx.x; j < U; j += kSsThreads) A.stage_comp[(size_t)rd.stage + j] = rec[j];
  if (!one_pass) {
    ss_barrier();
    uint32_t u = before;
#pragma unroll
    for (int i = 0; i < kSsItems; ++i)
      if (head_bits & (1u << i)) l_val[u++] = hv[i];
    ss_barrier();
  }
  for (uint32_t j = threadIdx.x; j < U; j += kSsThreads) A.stage_val[(size_t)rd.stage + j] = l_val[j];
  if (!one_pass) {
    ss_barrier();
    uint32_t u = before;
#pragma unroll
    for (int i = 0; i < kSsItems; ++i)
      if (head_bits & (1u << i)) l_rank[u++] = hk[i];
    ss_barrier();
  }
  for (uint32_t j = threadIdx.x; j < U; j += kSsThreads) A.stage_rank[(size_t)rd.stage + j] = l_rank[j];
  SS_T(t_4);
  SS_ADD(3, t_3, t_4);
  if (threadIdx.x == 0) {
#if defined(TAD_SS_PROF)
    atomicAdd(&g_ss_prof[4], 1ull);
#endif
    A.seg_count[(size_t)p * R + r] = U;
    if (U) { atomicAdd(A.num_runs, (unsigned long long)U); atomicMax(A.tmax, s_best); }
  }
}


// The rank grid straight from the stages (no sorted list in between): cell(rank, key) = rank * K + key like k_sparse_place, the rank staged by
// k_ss_sort.  The stage is key-major, the grid rank-major: written point by point every lane of a store hits its own 64-byte sector, and the
// address unit handles those one at a time — 0.97 ms for 3.3e7 points (k_sparse_place pays the same 0.54 ms for the LSD path).  So the
// transposition goes through LDS: a workgroup per round loads the stage in chunks of kSpChunk points (coalesced), notes per key where its
// points start in the chunk (and at which rank: a long series spans chunks), then walks (rank, key) with the KEY fastest: consecutive lanes
// store consecutive keys of one rank row.
static constexpr int kSpThreads = 1024;
static constexpr uint32_t kSpChunk = 4096;

__global__ __launch_bounds__(kSpThreads) void k_ss_place(SsArgs A, int64_t t0, Grid g, long long *__restrict__ times) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  const uint32_t seg = blockIdx.x;
  const uint32_t cnt = A.seg_count[seg];
  if (cnt == 0) return;
  const SsRound rd = A.rounds[seg];
  const uint32_t range = rd.key1 - rd.key0;
  unsigned long long *l_val = reinterpret_cast<unsigned long long *>(sp_smem);        // [kSpChunk]
  uint32_t *l_dt = reinterpret_cast<uint32_t *>(l_val + kSpChunk);                   // [kSpChunk]
  uint32_t *l_start = l_dt + kSpChunk;                                               // [range] first point of the key in the chunk
  uint32_t *l_rank0 = l_start + range;                                               // [range] its rank
  uint32_t *l_len = l_rank0 + range;                                                 // [range] points of the key in the chunk
  __shared__ uint32_t s_maxlen;
  const unsigned long long *comp = A.stage_comp + rd.stage, *val = A.stage_val + rd.stage;
  const uint32_t *rank = A.stage_rank + rd.stage;
  const unsigned long long key_first = ((unsigned long long)(seg / A.bins_per_part) << A.shift_part) + rd.key0;
  for (uint32_t c0 = 0; c0 < cnt; c0 += kSpChunk) {                                  // workgroup-uniform
    const uint32_t m = cnt - c0 < kSpChunk ? cnt - c0 : kSpChunk;
    for (uint32_t k = threadIdx.x; k < range; k += kSpThreads) l_len[k] = 0;
    if (threadIdx.x == 0) s_maxlen = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < m; j += kSpThreads) {
      const unsigned long long c = comp[c0 + j];
      const uint32_t rk = rank[c0 + j];
      const uint32_t kl = (uint32_t)((c >> 32) - key_first);
      l_val[j] = val[c0 + j];
      l_dt[j] = (uint32_t)c;
      if (j == 0 || rk == 0) { l_start[kl] = j; l_rank0[kl] = rk; }
      if (j + 1 == m || rank[c0 + j + 1] == 0) {              // the key's last point in the chunk (ranks count up inside a key)
        const uint32_t first = rk < j ? rk : j;               // points of the key before this one inside the chunk
        l_len[kl] = first + 1;
        atomicMax(&s_maxlen, first + 1);
      }
    }
    __syncthreads();
    const uint32_t maxlen = s_maxlen;
    // (rank r, key k) = cell q of the maxlen x range rectangle, q = threadIdx.x + i * kSpThreads: advanced without a division per step
    const uint32_t dr = kSpThreads / range, dk = kSpThreads % range;
    uint32_t r = threadIdx.x / range, k = threadIdx.x % range;
    for (; r < maxlen; r += dr, k += dk) {
      if (k >= range) { k -= range; ++r; if (r >= maxlen) break; }
      if (r < l_len[k]) {
        const uint32_t j = l_start[k] + r;
        const uint64_t cell = (uint64_t)(l_rank0[k] + r) * g.K + key_first + k;
        g.val[cell] = l_val[j];
        g.flag[cell] = FLAG_PRESENT;
        times[cell] = (long long)(t0 + (int64_t)l_dt[j]);
      }
    }
    __syncthreads();
  }
}

// every round's stage to its place in the sorted list (off = exclusive scan of seg_count in (block, round) order)
__global__ __launch_bounds__(256) void k_ss_compact(SsArgs A, const unsigned long long *__restrict__ off, unsigned long long *__restrict__ comp_out,
                                                     unsigned long long *__restrict__ val_out) {
  const uint32_t seg = blockIdx.x;
  const uint32_t cnt = A.seg_count[seg];
  if (cnt == 0) return;
  const size_t src = A.rounds[seg].stage;
  const unsigned long long dst = off[seg];
  for (uint32_t j = threadIdx.x; j < cnt; j += 256) {
    comp_out[dst + j] = A.stage_comp[src + j];
    val_out[dst + j] = A.stage_val[src + j];
  }
}

// temp: rounds [nparts * bpp] | n_rounds [nparts] | seg_count [nparts * bpp] | off u64 [nparts * bpp + 1] | scan scratch
struct SsTemp { size_t rounds, n_rounds, bin_round, round_fill, seg_count, off, scratch, total; };
static SsTemp ss_temp_layout(const PartPlan &pl) {
  const size_t m = (size_t)pl.nparts * pl.bins_per_part;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  SsTemp t;
  size_t at = 0;
  t.rounds = at; at = up(at + m * sizeof(SsRound));
  t.n_rounds = at; at = up(at + (size_t)pl.nparts * 4);
  t.bin_round = at; at = up(at + m);
  t.round_fill = at; at = up(at + m * 4);
  t.seg_count = at; at = up(at + m * 4);
  t.off = at; at = up(at + (m + 1) * 8);
  t.scratch = at; at = up(at + scan_scratch_elems(m ? m : 1) * 8);
  t.total = at;
  return t;
}

size_t sparse_part_temp_bytes(const PartPlan &pl) { return ss_temp_layout(pl).total + 256; }

static SsArgs ss_args(const PartPlan &pl, void *temp, unsigned long long *stage_comp, unsigned long long *stage_val) {
  const SsTemp tl = ss_temp_layout(pl);
  unsigned char *tp = static_cast<unsigned char *>(temp);
  SsArgs A{};
  A.nbins = pl.nbins; A.bins_per_part = pl.bins_per_part; A.nparts = pl.nparts;
  A.G = pl.G; A.shift_bin = pl.shift_bin; A.shift_part = pl.shift_part; A.cell_bits = pl.cell_bits; A.tbits = pl.sp_tbits;
  A.rounds = reinterpret_cast<SsRound *>(tp + tl.rounds);
  A.n_rounds = reinterpret_cast<uint32_t *>(tp + tl.n_rounds);
  A.seg_count = reinterpret_cast<uint32_t *>(tp + tl.seg_count);
  A.bin_round = reinterpret_cast<uint8_t *>(tp + tl.bin_round);
  A.round_fill = reinterpret_cast<uint32_t *>(tp + tl.round_fill);
  A.stage_comp = stage_comp; A.stage_val = stage_val;
  return A;
}

// num_runs[0] = unique points, num_runs[1] (as unsigned int) = the longest series; both zeroed by the caller
void launch_sparse_sort(hipStream_t s, const void *recs, const unsigned long long *part_start, const uint32_t *binhist, const PartPlan &pl, uint64_t K,
                        int64_t step, bool op_max, unsigned long long *recs2, unsigned long long *stage_comp, unsigned long long *stage_val,
                        uint32_t *stage_rank, void *temp, unsigned long long *num_runs, DevCounters *ctr) {
  SsArgs A = ss_args(pl, temp, stage_comp, stage_val);
  A.recs = static_cast<const unsigned long long *>(recs); A.part_start = part_start; A.binhist = binhist;
  A.K = K; A.step = (unsigned long long)step; A.ctr = ctr;
  A.recs2 = recs2; A.stage_rank = stage_rank;
  A.num_runs = num_runs; A.tmax = reinterpret_cast<unsigned int *>(num_runs + 1);
  const size_t m = (size_t)pl.nparts * pl.bins_per_part;
  hipMemsetAsync(A.seg_count, 0, m * 4, s);
  hipLaunchKernelGGL(k_ss_plan, dim3((pl.nparts + 3) / 4), dim3(256), (size_t)4 * pl.bins_per_part * 4, s, A);
  hipLaunchKernelGGL(k_ss_split, dim3(pl.nparts), dim3(kSplitThreads), (((size_t)pl.bins_per_part * 9 + 15) & ~(size_t)15), s, A);
  const size_t lds = (size_t)kSsCap * 8 + ((size_t)1 << kSsMsdBits) * 4 + kRsRadix * 4;
  const unsigned blocks = (unsigned)(((pl.nparts + 7u) / 8u) * 8u * pl.bins_per_part);
  if (op_max) {
    allow_big_lds(reinterpret_cast<const void *>(k_ss_sort<true>), lds);
    hipLaunchKernelGGL(k_ss_sort<true>, dim3(blocks), dim3(kSsThreads), lds, s, A);
  } else {
    allow_big_lds(reinterpret_cast<const void *>(k_ss_sort<false>), lds);
    hipLaunchKernelGGL(k_ss_sort<false>, dim3(blocks), dim3(kSsThreads), lds, s, A);
  }
#if defined(TAD_SS_PROF)
  {
    hipStreamSynchronize(s);
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ss_prof), sizeof h);
    const double w = h[4] ? (double)h[4] : 1.0;
    fprintf(stderr, "ss prof: %llu rounds; shader-clock cycles per round: load + count %.0f, sort %.0f (bucket scan %.0f, scatter %.0f, ranks in the buckets %.0f), fold %.0f, output %.0f\n", h[4], h[0] / w, h[1] / w, h[5] / w, h[6] / w, h[7] / w, h[2] / w, h[3] / w);
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_ss_prof), z, sizeof z);
  }
#endif
}

// the stages -> the rank grid (the job's normal way on)
void launch_sparse_place_staged(hipStream_t s, const PartPlan &pl, void *temp, const unsigned long long *stage_comp, const unsigned long long *stage_val,
                                const uint32_t *stage_rank, int64_t t0, Grid g, long long *times) {
  SsArgs A = ss_args(pl, temp, const_cast<unsigned long long *>(stage_comp), const_cast<unsigned long long *>(stage_val));
  A.stage_rank = const_cast<uint32_t *>(stage_rank);
  const size_t lds = (size_t)kSpChunk * 12 + (size_t)pl.KP * 12;
  allow_big_lds(reinterpret_cast<const void *>(k_ss_place), lds);
  hipLaunchKernelGGL(k_ss_place, dim3((unsigned)((size_t)pl.nparts * pl.bins_per_part)), dim3(kSpThreads), lds, s, A, t0, g, times);
}

// the stages -> the sorted unique list comp_out / val_out (length classes, tad_aggregate: they read the list itself)
void launch_sparse_compact(hipStream_t s, const PartPlan &pl, void *temp, const unsigned long long *stage_comp, const unsigned long long *stage_val,
                           unsigned long long *comp_out, unsigned long long *val_out) {
  SsArgs A = ss_args(pl, temp, const_cast<unsigned long long *>(stage_comp), const_cast<unsigned long long *>(stage_val));
  const SsTemp tl = ss_temp_layout(pl);
  unsigned char *tp = static_cast<unsigned char *>(temp);
  unsigned long long *off = reinterpret_cast<unsigned long long *>(tp + tl.off);
  unsigned long long *scratch = reinterpret_cast<unsigned long long *>(tp + tl.scratch);
  const size_t m = (size_t)pl.nparts * pl.bins_per_part;
  launch_scan(s, A.seg_count, off, m, scratch);
  hipLaunchKernelGGL(k_ss_compact, dim3((unsigned)m), dim3(256), 0, s, A, off, comp_out, val_out);
}

void launch_sparse_tmax(hipStream_t s, const unsigned long long *ucomp, uint64_t slots, const unsigned long long *P_dev, uint32_t *first, unsigned int *tmax) {
  if (slots == 0) return;
  const uint64_t need = (slots + kSpBlock - 1) / kSpBlock;
  const unsigned blocks = (unsigned)(need < 8192 ? need : 8192);
  hipLaunchKernelGGL(k_sparse_first, dim3(blocks), dim3(kSpBlock), 0, s, ucomp, P_dev, first);
  hipLaunchKernelGGL(k_sparse_tmax, dim3(blocks), dim3(kSpBlock), 0, s, ucomp, P_dev, first, tmax);
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

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_sparse() { return reinterpret_cast<const void *>(&k_sparse_first); }

}  // namespace tad
