// tad_internal.h — shared between the C-ABI host code (tad_engine.cpp, tad_capi*.cpp) and the gfx950 kernels.
// Product code.  Nothing here may include or call anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tad.h"

namespace tad {

// flowEndSeconds lattice: t = t0 + step * bucket, bucket < nb.
struct Lattice {
  int64_t t0;
  int64_t step;
  uint64_t nb;
  uint64_t magic;  // ceil(2^64 / step): exact quotient for dividends < 2^32 (mode 1)
  int mode;        // 0: step == 1, 1: multiply-high path ((tmax - t0) < 2^32), 2: 64-bit division
};

// The aggregated point grid, TIME-MAJOR: cell(bucket b, key k) = b * K + k.
// Time-major so that "one lane = one key, walk the series sequentially" is a fully coalesced
// access at every step (64 consecutive keys = 512 contiguous bytes).
struct Grid {
  unsigned long long *val;  // aggregated UInt64 value of the point
  uint8_t *flag;            // bit0: point present, bit1: anomaly verdict (DBSCAN / ARIMA detectors)
  uint64_t K;               // keys
  uint64_t T;               // buckets
  const long long *times;   // NULL: cell (b, k) is at t0 + b * step (lattice); else the sparse path's RANK grid (tad_sparse.hip):
                            // row b holds every key's b-th point in time order and times[b * K + k] its flowEndSeconds
};

enum : uint32_t { DEV_ERR_KEY_RANGE = 1u, DEV_ERR_OFF_LATTICE = 2u, DEV_ERR_OVERFLOW_LIST = 4u, DEV_ERR_LATE_ROW = 8u,
                  DEV_ERR_REGION_FULL = 16u,   // Stage 0 v2 with a SAMPLED histogram: a (workgroup, partition) region was sized too small
                  // (32u was DEV_ERR_SPEC of the one-synchronisation job, ABI 8-11: removed in round 6, see docs/HISTORY.md)
                  DEV_ERR_SPARSE_ROUND = 64u };  // sparse Stage 0 through the partition pass: one key bin holds more records than a workgroup sorts in LDS (the LSD sort takes over)
enum : uint8_t { FLAG_PRESENT = 1, FLAG_ANOMALY = 2 };

// Per-block partial of the lattice-derivation pass.
struct MetaPartial {
  int64_t tmin, tmax, tref;
  uint64_t g;     // gcd of |t - tref| over the rows this block kept
  uint64_t used;  // rows kept
  uint64_t seen;  // rows examined (k_meta_hist with a sampled histogram: the histogram's sampling ratio is seen / chunk rows)
};

// Device-side counters of one run (one 64-byte block, zeroed per run).
struct DevCounters {
  unsigned long long rows_used;
  unsigned long long n_keys;
  unsigned long long n_points;
  unsigned long long keys_no_result;
  unsigned long long kalman_steps;
  unsigned long long arima_fits;
  unsigned long long arima_nan_fits;   // fits whose prediction is not finite (the optimiser walked into a non-finite likelihood)
  uint32_t err;
  uint32_t pad;
};

struct RowFilter {
  int64_t start_time;  // 0 = unset
  int64_t end_time;    // 0 = unset
};

struct OutRows {
  unsigned long long *key_id;
  long long *flow_end_s;
  double *throughput;
  double *algo_calc;
  double *stddev;
  uint8_t *anomaly;  // only with TAD_FLAG_EMIT_ALL_POINTS
};

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute is per (function, device) in the HIP runtime and a
// process may hold one engine per GPU (the Go controller does), so it is set — a cheap host call — before every
// launch for the current device instead of being cached in a process-wide flag.
inline void allow_big_lds(const void *kernel, size_t bytes) {
  hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ---- launchers (tad_kernels.hip / tad_dbscan.hip / tad_arima.hip / tad_synth.hip) ----
int launch_meta(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                const int64_t *t_start, uint64_t n, RowFilter f, MetaPartial *partials, int n_blocks);

void launch_scatter(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                    const int64_t *t_start, const uint64_t *value, uint64_t n, RowFilter f,
                    Lattice lat, Grid g, bool op_max, DevCounters *ctr);

// a / b for an integer-valued b >= 1, given y = RN(1 / b): Markstein's correction steps on the FMA unit.
// q0 = RN(a y); r = a - b q (exact in one FMA); q' = RN(q + r y) is the correctly rounded quotient once q is a
// faithful approximation — two steps guarantee that (checked against IEEE division on 4e8 operand pairs,
// and by the bit-exact sigma parity tests).  5 dependent FMA-class ops instead of the ~12-op v_div sequence:
// the per-key stddev_samp recurrence (Spark's d / n) is a pure dependency chain.
__device__ __forceinline__ double div_by_count(double a, double b, double y) {
  double q = a * y;
  double r = fma(-b, q, a);
  q = fma(r, y, q);
  r = fma(-b, q, a);
  return fma(r, y, q);
}

struct Moments { double n, mean, m2; };   // per-key / per-block (n, mean, M2) of the points

#if defined(__HIPCC__)
// (the per-key walks are HBM-latency bound: one lane = one key, 64 consecutive keys = one coalesced 512-byte access)
// Walk one key's column of the time-major grid in time order, calling step(t, flag, raw_value) for every bucket.
// The walk is a dependency chain per lane fed by HBM: it is software-pipelined — the loads of chunk c+1 are
// issued before chunk c is consumed, in fixed numbers (the last chunk re-loads itself) so that hipcc can wait with
// vmcnt(N > 0) instead of draining the pipe.  kWalkChunk buckets = 2 loads each; two chunks stay in flight.
static constexpr int kWalkChunk = 8;

template <typename Step>
__device__ __forceinline__ void walk_series(const Grid &g, uint64_t k, Step step) {
  const uint64_t T = g.T;
  const uint64_t nfull = T / kWalkChunk;
  uint8_t fa[kWalkChunk], fb[kWalkChunk];
  unsigned long long va[kWalkChunk], vb[kWalkChunk];
  auto load = [&](uint64_t c, uint8_t *f, unsigned long long *v) {
#pragma unroll
    for (int u = 0; u < kWalkChunk; ++u) {
      const uint64_t cell = (c * kWalkChunk + u) * g.K + k;
      f[u] = g.flag[cell];
      v[u] = g.val[cell];
    }
  };
  if (nfull) {
    load(0, fa, va);
    uint64_t c = 0;
    for (; c + 2 <= nfull; c += 2) {
      load(c + 1, fb, vb);
#pragma unroll
      for (int u = 0; u < kWalkChunk; ++u) step(c * kWalkChunk + u, fa[u], va[u]);
      load(c + 2 < nfull ? c + 2 : c + 1, fa, va);   // past the end: a redundant in-bounds reload keeps the count fixed
#pragma unroll
      for (int u = 0; u < kWalkChunk; ++u) step((c + 1) * kWalkChunk + u, fb[u], vb[u]);
    }
    if (c < nfull) {
#pragma unroll
      for (int u = 0; u < kWalkChunk; ++u) step(c * kWalkChunk + u, fa[u], va[u]);
    }
  }
  for (uint64_t t = nfull * kWalkChunk; t < T; ++t) step(t, g.flag[t * g.K + k], g.val[t * g.K + k]);
}

// Long series on few keys (the rank grid of a sparse table's longest class: two keys of 20 000 points): one lane per key leaves the
// chip empty and every 8-bucket chunk of the walk is one exposed memory round trip (~2.5 us: 320 ns per step measured).  Here a whole
// WAVEFRONT walks ONE key: lane l fetches bucket 64 c + l of block c (one wave-wide load covers 64 time steps, the next two blocks are in
// flight while one is consumed), then every lane runs the step for the 64 buckets in time order on values broadcast by readlane — the same
// sequential recurrence in all 64 lanes, identical bits, lane 0 (or any lane) owns the outputs.  step(t, flag, raw) as walk_series.
static constexpr uint64_t kCoopMinT = 512, kCoopMaxK = 8192;   // used when T >= kCoopMinT and K <= kCoopMaxK (launchers: coop_shape)
inline bool coop_shape(const Grid &g) { return g.T >= kCoopMinT && g.K <= kCoopMaxK; }

// (v_readlane with a wavefront-uniform lane: __shfl would go through the LDS crossbar — ds_bpermute, ~100 clocks on the step's chain)
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned long long r = ((unsigned long long)readlane_u32((uint32_t)(b >> 32), l) << 32) | readlane_u32((uint32_t)b, l);
  return __longlong_as_double((long long)r);
}

// block(c): once per 64-bucket block, before its steps (wavefront-uniform; per-lane values the steps then pick up by readlane_*);
// step(t, flag, x, ord): x = the value as double (converted by the lane that loaded it), ord = ordinal of the point among the
// present points of its block.  Only present buckets get a step call.
template <typename Block, typename Step>
__device__ __forceinline__ void walk_series_coop(const Grid &g, uint64_t k, Block block, Step step) {
  const uint64_t T = g.T;
  const unsigned lane = threadIdx.x & 63u;
  const uint64_t nblk = (T + 63) / 64;
  auto load = [&](uint64_t c, uint32_t &f, unsigned long long &v) {
    const uint64_t t = c * 64 + lane;
    const bool in = c < nblk && t < T;
    f = in ? (uint32_t)g.flag[t * g.K + k] : 0u;
    v = in ? g.val[t * g.K + k] : 0ull;
  };
  uint32_t f0, f1, f2;
  unsigned long long v0, v1, v2;
  load(0, f0, v0);
  load(1, f1, v1);
  for (uint64_t c = 0; c < nblk; ++c) {
    load(c + 2, f2, v2);
    const uint64_t t0 = c * 64;
    const double x0 = (double)v0;                                            // every lane converts its own bucket
    const unsigned long long present = __ballot((f0 & FLAG_PRESENT) != 0);   // absent buckets are skipped without a step call
    block(c);
    int ord = 0;
    for (unsigned long long m = present; m; m &= m - 1, ++ord) {
      const int u = __ffsll((long long)m) - 1;                 // wavefront-uniform (scalar)
      step(t0 + (uint64_t)u, (uint8_t)readlane_u32(f0, u), readlane_f64(x0, u), ord);
    }
    f0 = f1; v0 = v1;
    f1 = f2; v1 = v2;
  }
}

// Chan et al. pairwise merge of (n, mean, M2)
__device__ __forceinline__ Moments chan_merge(Moments a, Moments b) {
  if (b.n == 0.0) return a;
  if (a.n == 0.0) return b;
  Moments r;
  r.n = a.n + b.n;
  const double d = b.mean - a.mean;
  r.mean = a.mean + d * (b.n / r.n);
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
  return r;
}

#endif

// per-key n / sigma (+ EWMA anomaly count when ewma != 0).  rcp[n] = RN(1/n) for n = 0..T (rcp[0] unused).
void launch_key_sigma(hipStream_t s, Grid g, double alpha, bool ewma_count, const double *rcp, double *sigma,
                      uint32_t *n_pts, uint32_t *n_anom, DevCounters *ctr, double *key_mean, double *key_m2);
// deterministic Chan merge of the per-key (n, mean, M2) into kMomentBlocks partials
static constexpr int kMomentBlocks = 128;
// ctr != NULL: also adds the job counters n_keys / n_points (for paths whose per-key kernel does not count them itself)
void launch_moments(hipStream_t s, uint64_t K, const uint32_t *n_pts, const double *key_mean,
                    const double *key_m2, Moments *partials, DevCounters *ctr = nullptr);
void launch_count_flags(hipStream_t s, Grid g, bool all_points, uint32_t *n_anom);
// exclusive scan of cnt[K] into off[K], total in off[K] (and in *total_copy, if given)
void launch_scan(hipStream_t s, const uint32_t *cnt, unsigned long long *off, uint64_t K,
                 unsigned long long *scratch, unsigned long long *total_copy = nullptr);
size_t scan_scratch_elems(uint64_t K);
// launch_scan + launch_moments with the moments merge riding in the scan's first launch
void launch_scan_moments(hipStream_t s, const uint32_t *cnt, unsigned long long *off, uint64_t K, unsigned long long *scratch,
                         unsigned long long *total_copy, const uint32_t *n_pts, const double *key_mean, const double *key_m2, Moments *partials,
                         DevCounters *ctr);
// kind: 0 EWMA (recompute), 1 flags + calc array, 2 flags with calc = 0, 3 flags with calc = per-key value calc[k],
// 4 = 2 with the stddev column computed here (Spark's streaming update over the key's series) for the keys that have rows
void launch_emit(hipStream_t s, Grid g, Lattice lat, int kind, bool all_points, double alpha,
                 const double *sigma, const uint32_t *n_pts, const double *calc,
                 const unsigned long long *off, OutRows out, uint64_t rows_hint = 0,   // rows_hint: off[K] if the caller knows it
                 int ewma_emit = 0, uint32_t ewma_emit_rows = 0);  // tad_plan: 1 = lane-per-key k_emit for the EWMA job; LDS rows per wavefront of the staged emit
void launch_emit_points(hipStream_t s, Grid g, Lattice lat, const unsigned long long *off, unsigned long long *out_key,
                        long long *out_t, unsigned long long *out_val);
// streaming EWMA: per-key running state (tad_state); k_stream continues the recurrences over the new grid
struct StreamState {
  uint32_t *n;
  double *avg, *m2, *ewma;
  long long *last_t;
  unsigned char *seen;  // 0 until the key has seen a point
};
// emit == false: next = updated state, n_anom[k] = anomalies among the new points (or all new points), late rows flagged;
// emit == true: rows written from the OLD state `cur` at off[] (next is not touched)
void launch_stream(hipStream_t s, Grid g, Lattice lat, double alpha, bool all_points, bool emit, StreamState cur, StreamState next,
                   uint32_t *n_anom, const unsigned long long *off, OutRows out, DevCounters *ctr);
// EWMA value for every present point into calc[T][K] (series entry points)
void launch_ewma_values(hipStream_t s, Grid g, double alpha, double *calc);

// DBSCAN: sets FLAG_ANOMALY on noise points.  scratch = dbscan_scratch_bytes(g) bytes of device memory.
// dbscan_uses_list: launch_dbscan (scan + work list) handles the shape — every series length since round 4 (sorted windows).
// st (all pointers NULL = not wanted): per-key point / anomaly counts and (mean, M2) moments
struct DbscanStats {
  uint32_t *n_pts, *n_anom;
  double *key_mean, *key_m2;
};
// DBSCAN job, Stage 0 pass C in SETTLE mode (round 3): a partition's rounds split it by KEY sub-range (kt keys x ALL buckets per LDS
// tile) instead of by bucket range, so that every tile holds whole series; the tile pass then does k_dbscan_scan's work itself —
// per-key count / min / max / shifted moments in the same sequential order — and writes the grid columns of UNSETTLED keys only
// (a key is settled iff it has no points, or >= min_samples points all within eps of each other: no noise, nothing reads its
// column again).  Unsettled keys go to the detector's work list.  Tiles that cannot decide (a split partition merges several
// slices; overflow-list records are folded in later) write every column and leave n_pts[k] = kSettleRedo for k_dbscan_scan.
struct SettleArgs {
  DbscanStats st;            // all four arrays, K entries
  uint32_t *list;            // work list of the detector (dbscan scratch + 64)
  unsigned int *count;       // its length (dbscan scratch), zeroed before pass C
  double eps;
  int32_t min_samples;
  uint32_t on;               // 0: plain pass C
  // The series of the listed keys, CONTIGUOUS per list entry (entry e < cs_cap: cs_val[e * T + b], cs_flag[e * T + b], cs_has[e] = 1):
  // the tile has them in LDS when it lists the key; the list kernel then reads 100 consecutive values instead of gathering the key's
  // column from the time-major grid, one 64-byte sector per bucket (C4: 235 MB fetched for 18 MB of series).  NULL = not kept.
  unsigned long long *cs_val;
  uint8_t *cs_flag;
  uint8_t *cs_has;
  uint32_t cs_cap;
  uint32_t *redo_list;       // keys the tile pass could not decide (split partition, a value on the overflow list): walked by k_dbscan_scan_redo,
  unsigned int *redo_count;  //   a wavefront per listed key (a lane-per-key pass over ALL keys to find 1 % of them cost a whole scan: 122 us at C4)
  // the redo keys' series, contiguous per redo entry like cs_* (rs_has: 0 = not kept, 1 = kept, cells flagged 2 must be read from the grid)
  unsigned long long *rs_val;
  uint8_t *rs_flag;
  uint8_t *rs_has;
  uint32_t rs_cap;
  const uint32_t *ovf_keys;  // bitmap of the keys with a value on the overflow list (pass B): only THOSE keys are left to the scan; NULL: any overflow record sends every key there
};
// where launch_dbscan keeps the contiguous series inside its scratch (series of <= 256 buckets; cs_cap entries)
void dbscan_compact_series(Grid g, void *scratch, unsigned long long **cs_val, uint8_t **cs_flag, uint8_t **cs_has, uint32_t *cs_cap);
// the redo list inside launch_dbscan's scratch: its counter is the second word of the scratch (zeroed with the list counter: 8 bytes)
uint32_t *dbscan_redo_list(Grid g, void *scratch);
void dbscan_redo_series(Grid g, void *scratch, unsigned long long **rs_val, uint8_t **rs_flag, uint8_t **rs_has, uint32_t *rs_cap);
static constexpr uint32_t kSettleRedo = 0xFFFFFFFFu;
size_t dbscan_scratch_bytes(Grid g);
bool dbscan_uses_list(Grid g);
// settled_by_stage0: pass C ran in settle mode (SettleArgs): the scan only walks keys marked kSettleRedo, the list is already started
int launch_dbscan(hipStream_t s, Grid g, double eps, int min_samples, void *scratch,
                  DbscanStats st = DbscanStats{nullptr, nullptr, nullptr, nullptr}, bool settled_by_stage0 = false);
// DBSCAN job (statistics from the scan, sigma computed at emit): the rows from the work list launch_dbscan left in `scratch`,
// one wavefront per listed key.  false: not applicable (series longer than a wavefront's registers hold) -> launch_emit(kind 4)
bool launch_emit_dbscan_list(hipStream_t s, Grid g, Lattice lat, const void *scratch, const uint32_t *n_anom, const unsigned long long *off,
                             OutRows out);

// drop detector (tad_drop.hip): sigma / n_pts / key_mean / key_m2 / counters + FLAG_ANOMALY; ws = K * T doubles
void launch_drop(hipStream_t s, Grid g, double n_sigma, int min_samples, double *ws, double *sigma, uint32_t *n_pts,
                 double *key_mean, double *key_m2, DevCounters *ctr);

// ARIMA(1,1,1) walk-forward on Box-Cox data: calc[T][K] + FLAG_ANOMALY.
// pause (NULL = never yield): a word in device memory; while it is non-zero the wavefronts of k_arima_fit SUSPEND their fits at the end
// of the running optimiser cycle (state saved per wavefront in the workspace) and retire — other jobs' whole-CU workgroups cannot be placed
// beside them.  *yielded (device memory inside the workspace) is non-zero afterwards when that happened: the caller waits for `pause` to clear
// and calls launch_arima_fit until *yielded stays zero — every wavefront takes its own lanes back, the per-position cursors carry on.
// grace: optimiser cycles during which a (re)launched wavefront ignores the word (progress under a steady stream of short jobs).
int launch_arima(hipStream_t s, Grid g, const double *sigma, const uint32_t *n_pts, int maxiter,
                 double *calc, DevCounters *ctr, void *workspace, size_t workspace_bytes, const int *pause = nullptr, const unsigned int **yielded = nullptr);
int launch_arima_fit(hipStream_t s, Grid g, const double *sigma, const uint32_t *n_pts, int maxiter, double *calc, DevCounters *ctr, void *workspace,
                     const int *pause, const unsigned int **yielded, uint32_t grace);
size_t arima_workspace_bytes(Grid g);

// ---- Stage 0 v2: partition rows by key range, aggregate tiles in LDS (tad_stage0_part.hip) ----
struct OverflowRec {  // a row whose value needs more than 49 bits: applied to the grid after the tile pass
  unsigned long long val;
  unsigned long long gcell;  // bucket * K + key
};
constexpr uint32_t kMaxBins = 16384;   // pass-A histogram bins (key >> shift_bin); also the rule k_fz_lookup_hist applies on the device
struct PartPlan {
  int shift_bin;       // pass-A histogram bin = key >> shift_bin
  uint32_t nbins;
  int G;               // workgroups of pass A and pass B (identical row chunking)
  uint64_t chunk;      // rows per workgroup
  int shift_part;      // partition = key >> shift_part ; KP = 1 << shift_part keys per tile
  uint32_t KP, nparts, bins_per_part;
  size_t agg_lds, part_lds;
  int rpt;             // rows per thread per tile in pass B
  int cell_bits;       // record = value << cell_bits | partition-local cell
  uint32_t tb, n_chunks;  // pass C: buckets per LDS round, rounds per partition
  uint32_t settle_kt;     // pass C in settle mode: keys per tile (0 = bucket rounds); then n_chunks = ceil(KP / settle_kt) key rounds, tb = T
  bool narrow;            // settle mode with 32-bit tile cells (value + 1; `max` only): values >= 2^32 - 2 take the overflow list, a sentinel record marks their cell
  uint32_t wc_cap;        // write-combining pass B: queue slots per partition (0 = use the sort-by-tile pass B)
  uint32_t wc_sec, wc_rpt;  // wc: records per emitted piece (8 or 16), rows per thread per tile (2 or 4)
  uint64_t pad_slots;     // wc: upper bound of the filler slots (regions rounded up to whole 64-byte sectors)
  int sp_tbits;           // sparse tables through the partition pass (part_plan_sparse): bit_width(T); cell_bits = shift_part + sp_tbits
};
// decide whether pass B runs as the write-combining variant (sets wc_cap / pad_slots; needs 16-byte aligned columns)
// partition_pass: tad_plan.partition_pass (0 = decide from the shape, 1 = sort-by-tile, 2 = write-combining whenever it fits)
void part_plan_wc(uint64_t slots, bool aligned, bool has2, int partition_pass, PartPlan *pl);
bool columns_aligned16(const void *key, const void *key2, const void *t_end, const void *value);
bool part_plan_bins(uint64_t n, uint64_t K, bool has2, PartPlan *pl);
bool part_plan_tiles(uint64_t K, uint64_t T, bool has2, PartPlan *pl);
// sparse tables (tad_sparse.hip, launch_sparse_sort): key blocks for the partition pass alone — no LDS tile has to hold a block's buckets;
// false = the shape does not fit the 8-byte records (too many keys / too long a lattice): the LSD sort runs
bool part_plan_sparse(uint64_t K, uint64_t T, bool has2, PartPlan *pl);
// sample_hist: histogram only the rows whose time is sampled too (one iteration in eight + the chunk ends): pass A then
// reads 1/8 of the key column; the regions of pass B are SIZED from the estimate (launch_part_offsets) instead of counted.
// Returns whether the histogram is sampled (only without a time-window filter and with 16-byte aligned columns).
bool launch_meta_hist(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                      const int64_t *t_start, uint64_t n, uint64_t K, RowFilter f, const PartPlan &pl,
                      MetaPartial *partials, uint32_t *binhist, DevCounters *ctr, bool sample_hist);
// offs32[G][nparts] (exclusive per-workgroup prefix inside each partition), total[nparts], part_start[nparts + 1]
// sampled: the histogram is a sample -> region capacities (estimate + 6 sigma + margin); partials carry the sampling ratios
// TWO launches (k_part_offsets, k_part_tail): also build the slice table of pass C in slice_mem (slice_table_bytes(slots, pl)) and zero
// the grid tile of every partition that will be split into several slices.
void launch_part_offsets(hipStream_t s, const uint32_t *binhist, const PartPlan &pl, uint32_t *offs32, uint32_t *total,
                         unsigned long long *part_start, bool sampled, const MetaPartial *partials, uint64_t n, uint64_t slots, void *slice_mem,
                         Grid g, DevCounters *ctr = nullptr);   // ctr (sampled): DEV_ERR_REGION_FULL when a partition sits in a few large regions
// upper bound of the record slots pass B may be given when the regions are sized from a sampled histogram
uint64_t sampled_slots_bound(uint64_t slots, const PartPlan &pl);
// fin != NULL (sampled regions): no fillers; fin[(g * nparts + p) * 2 + {0, 1}] = end of the records written upwards /
// start of the spilled records written downwards in region (g, p)
void launch_partition(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                      const int64_t *t_start, const uint64_t *value, uint64_t n, uint64_t K, RowFilter f,
                      Lattice L, const PartPlan &pl, const uint32_t *offs32, const unsigned long long *part_start,
                      void *recs, OverflowRec *ovf, unsigned long long *ovf_count, uint32_t ovf_cap, DevCounters *ctr,
                      uint32_t *fin = nullptr, uint32_t *ovf_keys = nullptr);   // ovf_keys: bitmap (K bits, zeroed) of the keys with a value on the overflow list
// slots = record slots of the run (rows x keys per row); slice_mem = slice_table_bytes(slots, pl) bytes of device scratch
size_t slice_table_bytes(uint64_t slots, const PartPlan &pl);
void launch_tile_aggregate(hipStream_t s, const void *recs, const unsigned long long *part_start, const PartPlan &pl,
                           uint64_t slots, void *slice_mem, Grid g, bool op_max, const OverflowRec *ovf,
                           const unsigned long long *ovf_count, uint32_t ovf_cap, const uint32_t *offs32, const uint32_t *fin, SettleArgs settle);
// whether pass C can run in settle mode for this plan (whole series of >= 8 keys fit an LDS tile; same number of rounds or fewer than 2x)
bool part_plan_settle(uint64_t T, PartPlan *pl, bool narrow = false);

// ---- Stage 0 for sparse tables: sort by (key, time), reduce, rank grid (tad_sparse.hip) ----
size_t sparse_sort_temp_bytes(uint64_t slots);
// The sparse Stage 0 of big tables: pass B's records (key blocks, launch_partition) -> split by round into recs2 (a round = a key sub-range of a
// block whose records fit LDS) -> one workgroup per round sorts its records in LDS by (key, bucket), folds equal runs and STAGES the unique
// points (stage_comp / stage_val / stage_rank, recs2: slots + pad words each); num_runs[0] += unique points, (unsigned int &)num_runs[1] max=
// the longest series (both zeroed by the caller).  From the stages either the rank grid directly (launch_sparse_place_staged) or, for the
// consumers of the sorted list itself (length classes, tad_aggregate), the compacted list (launch_sparse_compact: comp_out / val_out as
// launch_sparse_group leaves them).
size_t sparse_part_temp_bytes(const PartPlan &pl);
void launch_sparse_sort(hipStream_t s, const void *recs, const unsigned long long *part_start, const uint32_t *binhist, const PartPlan &pl, uint64_t K,
                        int64_t step, bool op_max, unsigned long long *recs2, unsigned long long *stage_comp, unsigned long long *stage_val,
                        uint32_t *stage_rank, void *temp, unsigned long long *num_runs, DevCounters *ctr);
void launch_sparse_place_staged(hipStream_t s, const PartPlan &pl, void *temp, const unsigned long long *stage_comp, const unsigned long long *stage_val,
                                const uint32_t *stage_rank, int64_t t0, Grid g, long long *times);
void launch_sparse_compact(hipStream_t s, const PartPlan &pl, void *temp, const unsigned long long *stage_comp, const unsigned long long *stage_val,
                           unsigned long long *comp_out, unsigned long long *val_out);
int launch_sparse_group(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end, const int64_t *t_start,
                        const uint64_t *value, uint64_t n, uint64_t K, RowFilter f, int64_t t0, uint64_t span, bool op_max, unsigned long long *comp_a,
                        unsigned long long *val_a, unsigned long long *comp_b, unsigned long long *val_b, void *temp, size_t temp_bytes,
                        unsigned long long *num_runs, DevCounters *ctr);
// slots: upper bound of the points (grid size); the point count itself is read on the device (*P_dev)
void launch_sparse_tmax(hipStream_t s, const unsigned long long *ucomp, uint64_t slots, const unsigned long long *P_dev, uint32_t *first, unsigned int *tmax);
void launch_sparse_place(hipStream_t s, const unsigned long long *ucomp, const unsigned long long *uval, uint64_t P, const uint32_t *first,
                         int64_t t0, Grid g, long long *times);
// length classes for skewed sparse tables (tad_sparse.hip; orchestration: tad_capi.cpp:run_sparse_classes)
uint32_t sparse_class_count(uint32_t tmax);   // classes 0 .. count-1: series of <= 16, <= 64, <= 256, ... points
void launch_sparse_len(hipStream_t s, const unsigned long long *ucomp, uint64_t P, const uint32_t *first, uint32_t *len);
void launch_sparse_class_counts(hipStream_t s, const uint32_t *len, uint64_t K, uint32_t c, uint32_t *member, uint32_t *pts);
void launch_sparse_class_columns(hipStream_t s, const unsigned long long *ucomp, const unsigned long long *uval, uint64_t P, const uint32_t *first,
                                 const uint32_t *len, uint32_t c, const unsigned long long *key_off, const unsigned long long *pt_off, int64_t t0,
                                 unsigned long long *out_key, long long *out_t, unsigned long long *out_val, uint32_t *keymap);
// Stage 0 alone on a sparse table: the sorted unique points as columns + counters + kMomentBlocks partials of (n, mean, M2)
void launch_sparse_points_out(hipStream_t s, const unsigned long long *ucomp, const unsigned long long *uval, uint64_t P, int64_t t0,
                              unsigned long long *out_key, long long *out_t, unsigned long long *out_val, Moments *partials, DevCounters *ctr);
void launch_class_count_rows(hipStream_t s, const unsigned long long *row_key, uint64_t R, const uint32_t *keymap, uint32_t *cnt,
                             unsigned long long *first_row);
void launch_class_gather(hipStream_t s, OutRows src, uint64_t R, const uint32_t *keymap, const unsigned long long *off,
                         const unsigned long long *first_row, OutRows dst);


// ---- row-sharded ingest: bucket rows by owner = key mod world (tad_shard.hip) ----
bool shard_world_ok(uint32_t world);
void launch_shard_count(hipStream_t s, const uint64_t *key, uint64_t n, uint32_t world, unsigned long long *counts);
void launch_shard_scatter(hipStream_t s, const uint64_t *key, const int64_t *t_end, const uint64_t *value, uint64_t n, uint32_t world,
                          unsigned long long *cursor, uint64_t *out_key, int64_t *out_t, uint64_t *out_val);

// ---- ingest: key tuples -> dense ids in order of first appearance (tad_factorize.hip) ----
static constexpr int kFzMaxCols = 8;
uint64_t factorize_table_slots(uint64_t virtual_rows);                  // the full size (2 n slots)
uint64_t factorize_first_slots(uint64_t virtual_rows);                  // the first attempt's table: min(full, 2^20)
uint64_t factorize_next_slots(uint64_t virtual_rows, uint64_t slots);   // 2^20 -> 2^24 -> full
size_t factorize_temp_bytes(uint64_t virtual_rows, uint64_t slots);
void launch_factorize(hipStream_t s, const long long *const *cols_a, const uint8_t *keep_a, const long long *const *cols_b, const uint8_t *keep_b, uint64_t n,
                      int n_cols, uint64_t slots, void *temp, uint64_t *key_a, uint64_t *key_b, uint64_t *first_row, uint64_t first_row_cap,
                      unsigned long long *num_keys_dev, uint32_t **flags_dev_out, uint32_t *hist_bins = nullptr, int hist_workgroups = 0, uint64_t hist_chunk = 0);
// Arrow string column -> dictionary codes in order of first appearance (tad_factorize.hip, ABI 10); same table sizes and temp layout
void launch_encode_strings(hipStream_t s, const void *offsets, int off64, const uint8_t *data, uint64_t data_bytes, const uint8_t *valid, uint64_t valid_off,
                           uint64_t n, uint64_t slots, void *temp, long long *codes, uint64_t *first_row, uint64_t first_row_cap,
                           unsigned long long *num_values_dev, uint32_t **flags_dev_out);

void launch_synth(hipStream_t s, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                  uint64_t num_keys, uint64_t n_buckets, uint64_t *key_id, int64_t *flow_end_s,
                  uint64_t *value);

// ---- columnar ingest (tad_ingest.hip) ----
constexpr int kMaskMaxTerms = 8;
void launch_widen(hipStream_t s, const void *src, int bits, bool is_signed, uint64_t n, const long long *table, uint64_t table_len, long long *dst, unsigned int *err);
void launch_mask_rows(hipStream_t s, uint64_t n, int n_terms, const long long *const *codes, const uint8_t *const *masks, const uint64_t *mask_len, bool combine,
                      uint8_t *keep, unsigned int *err);

// ---- code-object preload (tad_engine.cpp:preload_code_objects) ----
// HIP loads a translation unit's code object on the first use of one of its kernels (~0.4 ms each, inside the first job otherwise).
const void *code_anchor_arima();
const void *code_anchor_dbscan();
const void *code_anchor_drop();
const void *code_anchor_factorize();
const void *code_anchor_ingest();
const void *code_anchor_kernels();
const void *code_anchor_shard();
const void *code_anchor_sparse();
const void *code_anchor_stage0_part();
const void *code_anchor_synth();

}  // namespace tad
