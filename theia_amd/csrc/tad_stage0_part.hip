// tad_stage0_part.hip — Stage 0 v2: GROUP BY (key, flowEndSeconds) without random HBM atomics.
//
// Why: on MI355X the direct scatter (tad_kernels.hip:k_scatter) is bound by L2-miss read-modify-write
// transactions (~20e9/s measured, tools/ubench_scatter.hip) — 5.5 ms for 1e8 rows while the 24 B/row
// column stream alone takes 0.4 ms.  Here every random access lands in LDS instead:
//
//   pass A  k_meta_hist     one streaming read of the key column (8 B/row, 16-byte loads) and of a SAMPLE of the time
//                           column: min / max of flowEndSeconds and gcd of its differences over the sample (verified
//                           on every row by pass B, see below) AND a
//                           per-workgroup histogram of rows per key bin (bin = key >> shift_bin, LDS counters).
//   (tiny)  k_part_rows / k_part_colscan / k_part_scan1
//                           bins -> partitions of KP = 2^shift_part consecutive keys whose KP x T tile of the
//                           point grid fits in LDS; exclusive offsets per (workgroup, partition): the partition
//                           pass needs no global atomics and its output order is fixed.
//   pass B  k_partition_wc  streams the rows once more (24 B/row, two register sets of prefetched rows), turns each into
//                           an 8-byte record (value << cell_bits | tile-local cell) and appends it to its partition's
//                           queue in LDS; after every tile each queue that holds a whole 64-byte sector (or 128-byte line,
//                           when there are few enough partitions) emits it with ONE store instruction of 8 / 16
//                           consecutive lanes.  Nothing but whole aligned sectors ever goes to HBM: the memory system's
//                           price for short unaligned runs is what bounded the earlier sort-by-tile pass
//                           (k_partition: LDS histogram + scan per 10240-row tile, runs of ~7 records copied out; still
//                           used when the queues do not fit LDS or the runs are long anyway; tools/ubench_runs.hip).
//   pass C  k_tile_aggregate  one workgroup per partition: LDS u64 atomics (add wraps mod 2^64 / unsigned max)
//                           over its records, then writes its KP x T tile of the time-major grid (values +
//                           presence flags) with coalesced stores.
//
// Integer add/max are associative and commutative, so the aggregates are bit-identical to v1 and to
// ClickHouse's sum()/max() over UInt64 whatever the record order.
//
// Sampled lattice: every thread of pass A feeds only its first kGcdSamples kept rows into the gcd (a 64-bit
// modulo per row would make the pass ALU-bound) and, when no time-window filter is set, reads the time column
// for one iteration in sixteen plus both ends of its chunk.  The sampled step is a multiple of the true one and the
// sampled [min, max] lies inside the true range.  Pass B checks EVERY row against the lattice; a row off it
// (between lattice points, or outside the range) raises DEV_ERR_OFF_LATTICE and the host reruns with the exact
// derivation over all rows (tad_kernels.hip:k_meta).  Results are therefore never computed on a wrong lattice.
#include <cstdlib>
#include <cstring>

#include "tad_internal.h"

namespace tad {

static constexpr int kPartThreads = 1024;

// A Stage-0 record is ONE 64-bit word: value << cell_bits | partition-local cell (all ones = no cell).  Measured on MI355X
// the partition pass is bound by the 64-byte write sectors its short per-partition runs touch, so bytes per record
// are what matters: 8-byte records beat {u64 value, u16 cell} arrays (10 B), packed 12-byte and the original 16-byte
// records.  A value >= 2^(64 - cell_bits) does not fit: it goes, with its GLOBAL cell, to a small overflow list (global atomic
// append) that k_apply_overflow folds into the grid after the tile pass — same associative integer operator, so the
// aggregates stay bit-exact for the full UInt64 range.  If the list overflows the host falls back to Stage 0 v1.
// The cell field is cell_bits wide (15 .. kMaxCellBits, chosen by the plan from KP x T); the all-ones cell means
// "no cell" (row off the lattice, or its value went to the overflow list).
static constexpr int kMinCellBits = 15, kMaxCellBits = 24;
static constexpr uint32_t kTileCells = 17744;   // cells of one LDS tile of pass C: 9 B per cell within kLdsBudget (C4: 512 keys x 34 buckets, 3 rounds)
static constexpr uint32_t kMaxParts = 2048;     // partitions of pass B: 12 B of LDS each, and >= ~5 records per run per tile
static constexpr size_t kLdsBudget = 156 * 1024;  // dynamic LDS per workgroup; the rest of the CU's 160 KiB is for static __shared__
static constexpr int kGcdSamples = 4;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier and makes
// hipcc drain vmcnt(0) first — that would wait for the prefetched rows of the next tile and for the record stores
// of the previous one at every barrier.  Threads of pass B exchange data through LDS alone (global stores go to
// disjoint addresses nobody reads inside the kernel), so waiting for lgkmcnt (LDS) before s_barrier is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// shared helpers (duplicated from tad_kernels.hip on purpose: separate translation units)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t p_gcd_u64(uint64_t a, uint64_t b) {
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

__device__ __forceinline__ uint64_t p_absdiff(int64_t a, int64_t b) {
  return a >= b ? (uint64_t)a - (uint64_t)b : (uint64_t)b - (uint64_t)a;
}

struct PMeta {
  int64_t tmin, tmax, tref;
  uint64_t g, used;
};

__device__ __forceinline__ PMeta pmeta_merge(PMeta a, const PMeta &b) {
  if (b.used == 0) return a;
  if (a.used == 0) return b;
  a.tmin = b.tmin < a.tmin ? b.tmin : a.tmin;
  a.tmax = b.tmax > a.tmax ? b.tmax : a.tmax;
  a.g = p_gcd_u64(p_gcd_u64(a.g, b.g), p_absdiff(a.tref, b.tref));
  a.used += b.used;
  return a;
}

__device__ __forceinline__ bool p_time_kept(int64_t te, int64_t ts, bool has_ts, RowFilter f) {
  if (f.end_time != 0 && !(te < f.end_time)) return false;                  // anomaly_detection.py:584-586
  if (f.start_time != 0 && has_ts && !(ts >= f.start_time)) return false;   // :581-583
  return true;
}

__device__ __forceinline__ bool p_bucket(const Lattice &L, int64_t te, uint32_t &bucket) {
  const uint64_t d = (uint64_t)te - (uint64_t)L.t0;
  uint64_t b;
  if (L.mode == 0) b = d;
  else if (L.mode == 1) { if (d >> 32) return false; b = __umul64hi(d, L.magic); }
  else b = d / (uint64_t)L.step;
  if (b >= L.nb || b * (uint64_t)L.step != d) return false;
  bucket = (uint32_t)b;
  return true;
}

// ------------------------------------------------------------------------------------------------
// pass A — lattice partial + per-workgroup key-bin histogram.  Workgroup b owns rows
// [b * chunk, (b+1) * chunk): the SAME chunking as pass B, so the histogram row b is exactly what
// workgroup b of pass B will emit.  chunk is even and the columns are 16-byte aligned when VEC.
// ------------------------------------------------------------------------------------------------
struct MetaAcc {
  PMeta m;
  int nsample;
  uint32_t err;
};

#ifndef TAD_WAVE_AGG             // measurement builds: -DTAD_WAVE_AGG=0 takes the wavefront aggregation out of pass A and pass B
#define TAD_WAVE_AGG 1
#endif
#ifndef TAD_ADAPTIVE_QUEUES       // measurement builds: -DTAD_ADAPTIVE_QUEUES=0 gives every partition's queue the same depth
#define TAD_ADAPTIVE_QUEUES 1
#endif
#ifndef TAD_WAVE_AGG_MIN
#define TAD_WAVE_AGG_MIN 8
#endif
// lanes of a wavefront on one partition / one histogram bin from which they are handled together: 8 records = one 64-byte sector (hashed keys
// put 8 of 64 lanes on one of ~800 partitions with probability < 1e-9; a table of very few keys takes this path all the time, and may)
static constexpr int kWaveAggMin = TAD_WAVE_AGG_MIN;
static constexpr size_t kWcFixedBytes = 18;   // LDS of k_partition_wc per partition beside its queue: cnt, gcur, gend, nsp u32, jobs u16 (+ 4 bytes: cnt has F + 1 words)

// one count for a key bin.  Sorted rows put a whole wavefront on one bin — 64 atomics on one LDS word; the lanes that share the first
// active lane's bin add their number at once when they are many (hashed keys never are).
// weight: the rows this one stands for (1, or the sampling interval for a row of a sampled stretch: a sampled histogram holds ESTIMATED rows)
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, uint32_t weight) {
#if TAD_WAVE_AGG
  const uint32_t b0 = __builtin_amdgcn_readfirstlane(bin);
  const unsigned long long grp = __ballot(bin == b0);
  if (__popcll(grp) >= kWaveAggMin && bin == b0) {
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(grp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)grp, 0u)) == 0) atomicAdd(&hist[b0], (uint32_t)__popcll(grp) * weight);
    return;
  }
#endif
  atomicAdd(&hist[bin], weight);
}

__device__ __forceinline__ void meta_row(MetaAcc &a, uint32_t *hist, uint64_t k1, uint64_t k2, int64_t te, bool kept,
                                         uint64_t K, int shift_bin, uint32_t weight = 1u) {
  if (!kept) return;
  bool counted = false;
  if (k1 != TAD_KEY_SKIP) {
    if (k1 < K) { hist_add(hist, (uint32_t)(k1 >> shift_bin), weight); counted = true; }
    else a.err |= DEV_ERR_KEY_RANGE;
  }
  if (k2 != TAD_KEY_SKIP) {
    if (k2 < K) { hist_add(hist, (uint32_t)(k2 >> shift_bin), weight); counted = true; }
    else a.err |= DEV_ERR_KEY_RANGE;
  }
  if (!counted) return;
  if (a.m.used == 0) {
    a.m.tmin = a.m.tmax = a.m.tref = te;
    a.m.g = 0;
  } else {
    a.m.tmin = te < a.m.tmin ? te : a.m.tmin;
    a.m.tmax = te > a.m.tmax ? te : a.m.tmax;
    if (a.nsample < kGcdSamples) {
      const uint64_t d = p_absdiff(te, a.m.tref);
      if (d != 0) { a.m.g = p_gcd_u64(a.m.g, d); a.nsample++; }
    }
  }
  a.m.used++;
}

// a row whose time is not sampled: histogram only
__device__ __forceinline__ void hist_row(MetaAcc &a, uint32_t *hist, uint64_t k1, uint64_t k2, uint64_t K, int shift_bin) {
  if (k1 != TAD_KEY_SKIP) {
    if (k1 < K) hist_add(hist, (uint32_t)(k1 >> shift_bin), 1u);
    else a.err |= DEV_ERR_KEY_RANGE;
  }
  if (k2 != TAD_KEY_SKIP) {
    if (k2 < K) hist_add(hist, (uint32_t)(k2 >> shift_bin), 1u);
    else a.err |= DEV_ERR_KEY_RANGE;
  }
}

// Pass A reads one iteration (8192 rows of a workgroup's chunk) in kSampleMask + 1, plus both ends of the chunk.  One in eight until
// late in round 3; one in sixteen: pass A 0.106 -> 0.083 ms at C2, regions 2.4x instead of 2x their records (6 sigma of a noisier
// estimate: address space, never touched), pass B and pass C unchanged once a slice is long enough to keep a partition whole
// (profiles/r3_v9_passA_sample16_ab.log; with slices of 2x the records every partition split in two and merged through atomics: +0.13 ms).
#ifndef TAD_SAMPLE_MASK          // measurement builds (tools/build_variants.py): -DTAD_SAMPLE_MASK=3 samples one iteration in four
#define TAD_SAMPLE_MASK 15
#endif
#ifndef TAD_REGION_UR            // records per lane and batch of pass C's walk over sampled regions (x 64 lanes)
#define TAD_REGION_UR 12
#endif
static constexpr uint32_t kSampleMask = TAD_SAMPLE_MASK;

template <bool VEC, bool HAS2, bool SAMPLE_H>
__global__ __launch_bounds__(kPartThreads) void k_meta_hist(const uint64_t *__restrict__ key,
                                                            const uint64_t *__restrict__ key2,
                                                            const int64_t *__restrict__ t_end,
                                                            const int64_t *__restrict__ t_start,
                                                            uint64_t n, uint64_t chunk, uint64_t K, RowFilter f,
                                                            int shift_bin, uint32_t nbins,
                                                            MetaPartial *__restrict__ partials,
                                                            uint32_t *__restrict__ binhist, DevCounters *ctr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
  for (uint32_t i = threadIdx.x; i < nbins; i += kPartThreads) hist[i] = 0;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * chunk;
  const uint64_t hi = lo + chunk < n ? lo + chunk : n;
  MetaAcc acc{{0, 0, 0, 0, 0}, 0, 0};
  uint32_t seen = 0;   // rows this thread examined (= histogrammed or rejected): the sampling ratio of the histogram
  const bool has_ts = t_start != nullptr && f.start_time != 0;
  if (VEC) {
    // two rows per lane per column: 1 KiB per wave-instruction
    const uint64_t npair = hi > lo ? (hi - lo) >> 1 : 0;
    const ulonglong2 *kv = reinterpret_cast<const ulonglong2 *>(key + lo);
    const ulonglong2 *k2v = reinterpret_cast<const ulonglong2 *>(key2 + (HAS2 ? lo : 0));
    const longlong2 *tv = reinterpret_cast<const longlong2 *>(t_end + lo);
    constexpr int U = 4;
    // Without a time-window filter the time column is only needed for (min, max, sampled gcd): read it for one
    // iteration in sixteen plus both ends of the chunk (time-ordered tables have their extremes there) and let the
    // partition pass, which checks every row against the lattice, catch a missed extreme (-> exact re-derivation).
    // (a job with a time window tests its rows here too: with a sampled histogram the sampled rows — a window narrower than the sample's
    //  reach shows as `no live row` or as a region found full, and the job is redone exactly; with an exact histogram every row, as it must)
    const bool sample_t = SAMPLE_H || (f.end_time == 0 && !has_ts);
    uint64_t i = threadIdx.x;
    uint32_t it = 0;
    for (; i + (U - 1) * kPartThreads < npair; i += U * kPartThreads, ++it) {
      // wavefront-uniform.  The sample of an iteration is ONE of its sixteen wavefronts' rows (4 x 128 rows of the 8192), a different one every
      // iteration — 1/16 of the bytes in 1 KB pieces, no stretch of 2048 rows unseen.  (Until round 6 it was every sixteenth iteration whole:
      // rows sorted by key then hid entire partitions between two samples and every such job paid the exact retry.)
      // The chunk's last two iterations (and its ragged tail) are read WHOLE — a time-ordered table has its extremes there.  Their rows count
      // once; a row of the sampled stretches stands for kSampleMask + 1 rows: the histogram holds estimated rows that add up to the chunk's.
      // (Until late in round 6 every row counted once and the estimate was count x rows / seen: the whole-read end pulled that scale
      // down to ~9, and a partition whose rows all sit in the sampled stretch — keys that leave while the chunk is being read, in any
      // time-ordered table — was under-estimated by almost half: its region overflowed and the job was redone with the exact histogram.)
      const bool zone = (i - threadIdx.x) + 2 * U * kPartThreads >= npair;                                  // workgroup-uniform
      const bool with_t = !sample_t || zone || (it & kSampleMask) == ((threadIdx.x >> 6) & kSampleMask);
      if (SAMPLE_H && !with_t) continue;   // sampled histogram: the unsampled rows are not read at all
      const uint32_t weight = SAMPLE_H && !zone ? kSampleMask + 1u : 1u;
      seen += 2 * U;
      ulonglong2 k[U], k2[U];
      longlong2 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        k[u] = kv[i + u * kPartThreads];
        k2[u] = HAS2 ? k2v[i + u * kPartThreads] : make_ulonglong2(TAD_KEY_SKIP, TAD_KEY_SKIP);
      }
      if (with_t) {
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = tv[i + u * kPartThreads];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint64_t r = lo + 2 * (i + u * kPartThreads);
          const int64_t ts0 = has_ts ? t_start[r] : 0, ts1 = has_ts ? t_start[r + 1] : 0;
          meta_row(acc, hist, k[u].x, k2[u].x, t[u].x, p_time_kept(t[u].x, ts0, has_ts, f), K, shift_bin, weight);
          meta_row(acc, hist, k[u].y, k2[u].y, t[u].y, p_time_kept(t[u].y, ts1, has_ts, f), K, shift_bin, weight);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          hist_row(acc, hist, k[u].x, k2[u].x, K, shift_bin);
          hist_row(acc, hist, k[u].y, k2[u].y, K, shift_bin);
        }
      }
    }
    for (; i < npair; i += kPartThreads) {
      seen += 2;
      const ulonglong2 k = kv[i];
      const longlong2 t = tv[i];
      const ulonglong2 k2 = HAS2 ? k2v[i] : make_ulonglong2(TAD_KEY_SKIP, TAD_KEY_SKIP);
      const uint64_t r = lo + 2 * i;
      const int64_t ts0 = has_ts ? t_start[r] : 0, ts1 = has_ts ? t_start[r + 1] : 0;
      meta_row(acc, hist, k.x, k2.x, t.x, p_time_kept(t.x, ts0, has_ts, f), K, shift_bin);
      meta_row(acc, hist, k.y, k2.y, t.y, p_time_kept(t.y, ts1, has_ts, f), K, shift_bin);
    }
    if (((hi - lo) & 1) && threadIdx.x == 0 && hi > lo) {
      seen += 1;
      const uint64_t r = hi - 1;
      const int64_t te = t_end[r];
      meta_row(acc, hist, key[r], HAS2 ? key2[r] : TAD_KEY_SKIP, te, p_time_kept(te, has_ts ? t_start[r] : 0, has_ts, f), K, shift_bin);
    }
  } else {
    for (uint64_t r = lo + threadIdx.x; r < hi; r += kPartThreads) {
      seen += 1;
      const int64_t te = t_end[r];
      meta_row(acc, hist, key[r], HAS2 ? key2[r] : TAD_KEY_SKIP, te, p_time_kept(te, has_ts ? t_start[r] : 0, has_ts, f), K, shift_bin);
    }
  }
  PMeta m = acc.m;
  uint32_t err = acc.err;
  unsigned long long seen_w = seen;
  for (int d = 32; d >= 1; d >>= 1) seen_w += __shfl_down(seen_w, d);
  for (int d = 32; d >= 1; d >>= 1) {
    PMeta o;
    o.tmin = __shfl_down((long long)m.tmin, d);
    o.tmax = __shfl_down((long long)m.tmax, d);
    o.tref = __shfl_down((long long)m.tref, d);
    o.g = __shfl_down((unsigned long long)m.g, d);
    o.used = __shfl_down((unsigned long long)m.used, d);
    m = pmeta_merge(m, o);
    err |= __shfl_down(err, d);
  }
  __shared__ PMeta s_acc[kPartThreads / 64];
  __shared__ uint32_t s_err[kPartThreads / 64];
  __shared__ unsigned long long s_seen[kPartThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_acc[wave] = m; s_err[wave] = err; s_seen[wave] = seen_w; }
  __syncthreads();  // also: every histogram update of this workgroup is done
  if (threadIdx.x == 0) {
    PMeta a = s_acc[0];
    uint32_t e = s_err[0];
    unsigned long long sn = s_seen[0];
    for (int w = 1; w < kPartThreads / 64; ++w) { a = pmeta_merge(a, s_acc[w]); e |= s_err[w]; sn += s_seen[w]; }
    MetaPartial p;
    p.tmin = a.tmin; p.tmax = a.tmax; p.tref = a.tref; p.g = a.g; p.used = a.used; p.seen = sn;
    partials[blockIdx.x] = p;
    if (e) atomicOr(&ctr->err, e);
  }
  uint32_t *out = binhist + (size_t)blockIdx.x * nbins;
  for (uint32_t i = threadIdx.x; i < nbins; i += kPartThreads) out[i] = hist[i];
}

// ------------------------------------------------------------------------------------------------
// offsets: cnt[g][p] (row reduce of the bins) -> column-wise exclusive prefix over g -> part_start[p]
// ------------------------------------------------------------------------------------------------
// Capacity of a (workgroup, partition) region from pass A's ESTIMATED record count est = 16 S + E (S rows of the partition in the sampled
// stretches, each standing for 16; E rows in the whole-read end, counted once): est + 5.5 sqrt(16 est) + 192.  Too small -> pass B raises
// DEV_ERR_REGION_FULL and the host reruns with the exact histogram.  The danger is the LOW tail of S (~Poisson(n / 16)): a low S makes the
// estimate and the slack computed from it small together, so the constant decides, not the multiple of sigma — with 5.5 sigma + 96 a C2
// region (S ~ 29) overflowed when S <= 7: 5e-7 per region, one job in ten (and deterministically the C2 job with a start_time + end_time
// window); with + 192, S <= 2 or less: < 1e-3 per job for regions of 60 .. 2000 records (binomial tails, profiles/r6_s28_*).  Too large is
// not free either: pass C's walk follows the address span of a partition's regions (k_tile_aggregate 302-309 us at C2 for these ~1180-slot
// regions against 295 for the ~980 of the biased estimate before, profiles/r6_s24_*) — the price of an estimate that holds for keys that
// come and go.
static constexpr double kCapSigmas = 5.5, kCapIntervals = 12.0;    // (sampled_slots_bound on the host sums these capacities: keep the two in step)
__device__ __forceinline__ uint32_t sampled_capacity(uint32_t est) {
  constexpr double scale = (double)(kSampleMask + 1);
  return (uint32_t)((double)est + kCapSigmas * sqrt((double)est * scale) + kCapIntervals * scale);
}
// Pass C walks sampled regions one wavefront per region.  A partition whose records sit in a FEW large regions (rows sorted by key: two
// regions of 1e5 records) would be walked by one or two wavefronts — 1.4 ms of pass C at C2.  Such a table is sent to the exact histogram
// (contiguous partitions, all sixteen wavefronts streaming) before pass B starts: k_part_offsets raises DEV_ERR_REGION_FULL, pass B returns
// at once, pass C finds no slices (k_part_tail), the host redoes the job as it does for a region that overflowed.
static constexpr uint32_t kBigSampledRegion = 8192;      // records; and more than 8x the partition's mean region

// Work unit of pass C = a SLICE of at most slice_len record slots of one partition (k_tile_aggregate).
struct SliceTable {
  uint32_t *slice_part;          // [max_slices] partition of each slice
  uint32_t *slice_first;         // [nparts] index of the partition's first slice
  uint32_t *n_slices;            // [1]
};

struct OffsetsArgs {
  const uint32_t *binhist;       // [G][nbins] pass A's per-workgroup histogram of rows per key bin
  uint32_t nbins, bins_per_part, nparts, round_mask;
  int G;
  const MetaPartial *partials;   // sampled histogram: the sampling ratios (seen rows per workgroup); NULL = exact histogram
  uint64_t n, chunk;
  uint32_t *offs32;              // out [G][nparts]: exclusive prefix of workgroup g inside partition p
  uint32_t *total;               // out [nparts]
  unsigned long long *part_start;  // out [nparts + 1]
  SliceTable st;                 // out
  uint32_t slice_len;
  Grid g;                        // the grid tile of a partition that will be split into several slices is zeroed here
  int shift_part;
  DevCounters *ctr;              // sampled regions: DEV_ERR_REGION_FULL for a table the sampled layout does not suit (NULL: no check)
};

// TWO launches for everything between pass A and pass B (round 4; five before: k_part_rows, k_part_colscan — 256 dependent steps per
// thread —, k_part_scan1, k_build_slices and the pre-zero launch of pass C).  k_part_offsets, a wavefront per partition p: the G
// per-workgroup counts of p (bins reduced, capacities from a sampled histogram, regions rounded to whole sectors) are scanned across the
// lanes (lane l holds the workgroups G/64 * l ...), offs32[g][p] written, the total kept; a partition that will not fit one slice gets
// its grid tile zeroed (its slices merge with atomics).  k_part_tail, one workgroup: totals -> part_start, slice table.
// (A ticket that let the last workgroup of the first kernel do the tail was measured: the device-scope fences and the agent-scope
// loads it needs made the one kernel slower — 40 us — than the two.)
__global__ __launch_bounds__(256) void k_part_offsets(OffsetsArgs A) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 4u + wave;
  const uint32_t per = (uint32_t)(A.G + 63) / 64u;                 // workgroups per lane (4 at G = 256)
  if (p < A.nparts) {   // wavefront-uniform
    const uint32_t b0 = p * A.bins_per_part;
    const uint32_t b1 = b0 + A.bins_per_part < A.nbins ? b0 + A.bins_per_part : A.nbins;
    auto count_of = [&](uint32_t gi) -> uint32_t {   // record slots of workgroup gi in partition p
      const uint32_t *row = A.binhist + (size_t)gi * A.nbins;
      uint32_t c = 0;
      for (uint32_t b = b0; b < b1; ++b) c += row[b];
      if (A.partials != nullptr) c = sampled_capacity(c);
      return (c + A.round_mask) & ~A.round_mask;
    };
    constexpr uint32_t kKeep = 4;                      // G <= 256: the lane's counts stay in registers between the two loops
    uint32_t kept[kKeep] = {0, 0, 0, 0};
    uint32_t sum = 0;
    if (per <= kKeep) {
#pragma unroll
      for (uint32_t j = 0; j < kKeep; ++j) {
        const uint32_t gi = lane * per + j;
        if (j < per && gi < (uint32_t)A.G) { kept[j] = count_of(gi); sum += kept[j]; }
      }
    } else {
      for (uint32_t j = 0; j < per; ++j) {
        const uint32_t gi = lane * per + j;
        if (gi < (uint32_t)A.G) sum += count_of(gi);
      }
    }
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += y; }
    uint32_t run = incl - sum;
    const uint32_t tot = __shfl(incl, 63);
    if (per <= kKeep) {
#pragma unroll
      for (uint32_t j = 0; j < kKeep; ++j) {
        const uint32_t gi = lane * per + j;
        if (j < per && gi < (uint32_t)A.G) { A.offs32[(size_t)gi * A.nparts + p] = run; run += kept[j]; }
      }
    } else {
      for (uint32_t j = 0; j < per; ++j) {
        const uint32_t gi = lane * per + j;
        if (gi < (uint32_t)A.G) { A.offs32[(size_t)gi * A.nparts + p] = run; run += count_of(gi); }
      }
    }
    if (lane == 0) A.total[p] = tot;   // v2 requires n_rows * 2 < 2^32 (checked on the host)
    if (A.partials != nullptr && A.ctr != nullptr) {   // sampled regions: is this partition concentrated in a few large ones?
      bool big = false;
      if (per <= kKeep) {
#pragma unroll
        for (uint32_t j = 0; j < kKeep; ++j) big |= kept[j] > kBigSampledRegion && (unsigned long long)kept[j] * (unsigned)A.G > 8ull * tot;
      } else {
        for (uint32_t j = 0; j < per; ++j) {
          const uint32_t gi = lane * per + j;
          if (gi < (uint32_t)A.G) { const uint32_t c = count_of(gi); big |= c > kBigSampledRegion && (unsigned long long)c * (unsigned)A.G > 8ull * tot; }
        }
      }
      if (__any(big) && lane == 0) atomicOr(&A.ctr->err, DEV_ERR_REGION_FULL);
    }
    if (tot > A.slice_len) {           // split partition: its slices merge into the pre-zeroed tile
      const uint32_t KP = 1u << A.shift_part;
      const uint64_t k0 = (uint64_t)p << A.shift_part, cells_all = (uint64_t)KP * A.g.T;
      for (uint64_t c = lane; c < cells_all; c += 64) {
        const uint64_t b = c >> A.shift_part, k = k0 + (c & (KP - 1));
        if (k < A.g.K) { A.g.val[b * A.g.K + k] = 0ull; A.g.flag[b * A.g.K + k] = 0; }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_part_tail(OffsetsArgs A) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __shared__ uint32_t s_wave[4];
  const uint32_t F = A.nparts;
  constexpr uint32_t kEach = 8;                                   // kMaxParts / 256
  const uint32_t f0 = threadIdx.x * kEach;
  uint32_t tot[kEach];
#pragma unroll
  for (uint32_t j = 0; j < kEach; ++j) tot[j] = f0 + j < F ? A.total[f0 + j] : 0u;
  auto nsl = [&](uint32_t t) -> uint32_t { const uint32_t k = (uint32_t)(((unsigned long long)t + A.slice_len - 1) / A.slice_len); return k ? k : 1u; };
  unsigned long long rsum = 0;
  uint32_t ssum = 0;
#pragma unroll
  for (uint32_t j = 0; j < kEach; ++j)
    if (f0 + j < F) { rsum += tot[j]; ssum += nsl(tot[j]); }
  unsigned long long rincl = rsum;
  uint32_t sincl = ssum;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long y = __shfl_up(rincl, d);
    const uint32_t z = __shfl_up(sincl, d);
    if (lane >= (uint32_t)d) { rincl += y; sincl += z; }
  }
  __shared__ unsigned long long s_r[4];
  if (lane == 63) { s_r[wave] = rincl; s_wave[wave] = sincl; }
  __syncthreads();
  unsigned long long rbase = 0, rtot = 0;
  uint32_t sbase = 0, stot = 0;
  for (uint32_t w = 0; w < 4; ++w) {
    if (w < wave) { rbase += s_r[w]; sbase += s_wave[w]; }
    rtot += s_r[w]; stot += s_wave[w];
  }
  unsigned long long rrun = rbase + rincl - rsum;
  uint32_t srun = sbase + sincl - ssum;
#pragma unroll
  for (uint32_t j = 0; j < kEach; ++j) {
    const uint32_t q = f0 + j;
    if (q < F) {
      A.part_start[q] = rrun;
      rrun += tot[j];
      const uint32_t k = nsl(tot[j]);
      A.st.slice_first[q] = srun;
      for (uint32_t i = 0; i < k; ++i) A.st.slice_part[srun + i] = q;
      srun += k;
    }
  }
  // (a table k_part_offsets sends to the exact histogram: no slices — pass C's workgroups return where they look their slice up, pass B returns
  //  on the flag itself; a test of its own at the top of pass C was an exposed memory round trip per workgroup, + 0.01-0.03 ms at C2)
  if (threadIdx.x == 0) { A.part_start[F] = rtot; *A.st.n_slices = (A.ctr != nullptr && (A.ctr->err & DEV_ERR_REGION_FULL)) ? 0u : stot; }
}

// rows_used / error bits of a workgroup: ONE atomic per workgroup.  One per wavefront was 4096 atomics on one address at the very
// end of pass B, when all workgroups finish together — contended atomics serialise at ~10 ns each.
__device__ __forceinline__ void block_count_rows(uint32_t used, uint32_t err, DevCounters *ctr) {
  __shared__ unsigned long long s_u[kPartThreads / 64];
  __shared__ uint32_t s_e[kPartThreads / 64];
  unsigned long long u = used;
  for (int d = 32; d >= 1; d >>= 1) { u += __shfl_down(u, d); err |= __shfl_down(err, d); }
  if ((threadIdx.x & 63) == 0) { s_u[threadIdx.x >> 6] = u; s_e[threadIdx.x >> 6] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kPartThreads / 64; ++w) { u += s_u[w]; err |= s_e[w]; }
    if (u) atomicAdd(&ctr->rows_used, u);
    if (err) atomicOr(&ctr->err, err);
  }
}

// ------------------------------------------------------------------------------------------------
// pass B — group records by partition through LDS.
// LDS carve (dynamic): rec[S] u64 | part[S] u16 | off[F+1] u32 | gcur[F] u32 | delta[F] u32
// (record indices fit 32 bits: the host requires rows * keys-per-row < 2^32 for this path)
// ------------------------------------------------------------------------------------------------
struct PartArgs {
  const uint64_t *key, *key2;
  const int64_t *t_end, *t_start;
  const uint64_t *value;
  uint64_t n, chunk, K;
  RowFilter f;
  Lattice L;
  int shift_part;     // partition = key >> shift_part
  uint32_t kp_mask;   // KP - 1
  int cell_bits;      // record = value << cell_bits | cell
  uint32_t nparts;
  const uint32_t *offs32;                  // [G][nparts] exclusive row prefix of this workgroup inside each partition
  const unsigned long long *part_start;    // [nparts + 1]
  unsigned long long *recs;                // out: value << cell_bits | tile-local cell (bucket * KP + key-in-tile)
  OverflowRec *ovf;                        // out: records whose value needs more than 64 - cell_bits bits
  unsigned long long *ovf_count;
  uint32_t ovf_cap;
  DevCounters *ctr;
  uint32_t *fin;      // sampled regions (NULL = exact regions): [(g * nparts + p) * 2] = {end of the upward records, start of the spilled ones}
  int G;
  unsigned long long value_limit;   // values >= this do not fit a record (2^(64 - cell_bits)): overflow list
  unsigned long long narrow_limit;  // pass C with 32-bit tile cells (0 = off): values >= this (2^32 - 2) go to the overflow list too, but
                                    // their record stays in the stream with THIS value in the value field: the tile cell becomes the all-ones
                                    // word, which tells the tile pass that the cell's aggregate is on the overflow list
  uint32_t *ovf_keys;               // bitmap of the keys with a value on the overflow list (NULL = not kept)
};

// in-place exclusive scan of a[0..n) held in LDS by the whole workgroup; a[n] = total
__device__ __forceinline__ void lds_exclusive_scan(uint32_t *a, uint32_t n, uint32_t *s_wave) {
  const uint32_t per = (n + kPartThreads - 1) / kPartThreads;
  const uint32_t b0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; ++j)
    if (b0 + j < n) sum += a[b0 + j];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  if (lane == 63) s_wave[wave] = incl;
  lds_barrier();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < kPartThreads / 64; ++w) {
    if (w < wave) base += s_wave[w];
    tot += s_wave[w];
  }
  uint32_t run = base + incl - sum;
  for (uint32_t j = 0; j < per; ++j)
    if (b0 + j < n) { const uint32_t c = a[b0 + j]; a[b0 + j] = run; run += c; }
  if (threadIdx.x == 0) a[n] = tot;
  lds_barrier();
}

// RPT rows per thread per tile.  VEC: RPT even, columns 16-byte aligned, chunk even -> 16-byte loads
// (thread owns the row pairs 2*tid, 2*tid+1 of every 2048-row slab).
// GENERIC = false is the fast path: no time-window filter and a lattice whose bucket is one multiply-high
// (step == 1 or the 2^32 span case); GENERIC = true evaluates both at run time (64-bit division, t_start loads).
//
// Register plan (128 VGPRs at 16 wavefronts per CU): key + time of the NEXT tile are loaded right after
// phase 1 has consumed the current ones; the value column is only needed in phase 3, so the next tile's values
// are loaded after phase 3 into the registers phase 3 has just drained.  Every load therefore has two to four
// LDS phases to land, and nothing is live twice.
template <int RPT, bool HAS2, bool VEC, bool GENERIC>
__global__ __launch_bounds__(kPartThreads) void k_partition(PartArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NSLOT = RPT * (HAS2 ? 2 : 1);
  constexpr uint32_t S = (uint32_t)NSLOT * kPartThreads;  // record slots per tile
  constexpr uint32_t TILE = (uint32_t)RPT * kPartThreads;  // rows per tile
  const uint32_t F = A.nparts;
  unsigned long long *rec = reinterpret_cast<unsigned long long *>(smem);
  uint16_t *part = reinterpret_cast<uint16_t *>(smem + (size_t)S * 8);
  uint32_t *off = reinterpret_cast<uint32_t *>(smem + (size_t)S * 10);  // F + 1 entries
  uint32_t *gcur = off + (F + 1);  // next global record slot of (this workgroup, partition)
  uint32_t *delta = gcur + F;
  uint32_t *gend = delta + F;      // end of the region: capacity, checked (exact counts of THIS batch never reach it; a caller's stale tad_key_hist can)
  __shared__ uint32_t s_wave[kPartThreads / 64];
  __shared__ uint32_t s_full;

  // no global load may sit inside the tile loop: vmcnt retires in order, so waiting for one fresh load would
  // also wait for every prefetched row behind it
  if (A.fin != nullptr && (A.ctr->err & DEV_ERR_REGION_FULL)) return;   // k_part_offsets: this table is redone with the exact histogram
  const uint32_t *my_offs = A.offs32 + (size_t)blockIdx.x * F;
  {
    const uint32_t *nx = A.offs32 + (size_t)(blockIdx.x + 1) * F;
    const bool last = (int)blockIdx.x + 1 == A.G;
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) {
      const uint32_t ps = (uint32_t)A.part_start[p];
      gcur[p] = ps + my_offs[p];
      gend[p] = last ? (uint32_t)A.part_start[p + 1] : ps + nx[p];
      off[p] = 0;
    }
    if (threadIdx.x == 0) s_full = 0;
  }

  const uint64_t lo = (uint64_t)blockIdx.x * A.chunk;
  const uint64_t hi = lo + A.chunk < A.n ? lo + A.chunk : A.n;
  const bool has_ts = GENERIC && A.t_start != nullptr && A.f.start_time != 0;
  const uint32_t KP = A.kp_mask + 1u;
  const uint32_t cell_none = (1u << A.cell_bits) - 1u;
  const unsigned long long value_limit = A.value_limit;
  uint32_t err = 0, used = 0;

  uint64_t pk[RPT], pk2[HAS2 ? RPT : 1], pv[RPT];
  int64_t pt[RPT];
  auto row_index = [&](uint64_t base, int j) -> uint64_t {
    return VEC ? base + (uint64_t)(j >> 1) * (2 * kPartThreads) + 2 * threadIdx.x + (j & 1)
               : base + (uint64_t)j * kPartThreads + threadIdx.x;
  };
  // full tiles: unconditional loads (no per-lane branch anywhere near a load, so hipcc keeps them in flight);
  // the one ragged tile at the end of the chunk takes the guarded scalar loads.
  auto load_keys_full = [&](uint64_t base) {
#pragma unroll
    for (int j = 0; j < RPT; j += (VEC ? 2 : 1)) {
      const uint64_t i = row_index(base, j);
      if (VEC) {
        const ulonglong2 k = *reinterpret_cast<const ulonglong2 *>(A.key + i);
        const longlong2 t = *reinterpret_cast<const longlong2 *>(A.t_end + i);
        pk[j] = k.x; pk[j + 1] = k.y; pt[j] = t.x; pt[j + 1] = t.y;
        if (HAS2) { const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(A.key2 + i); pk2[j] = k2.x; pk2[j + 1] = k2.y; }
      } else {
        pk[j] = A.key[i]; pt[j] = A.t_end[i];
        if (HAS2) pk2[j] = A.key2[i];
      }
    }
  };
  auto load_values_full = [&](uint64_t base) {
#pragma unroll
    for (int j = 0; j < RPT; j += (VEC ? 2 : 1)) {
      const uint64_t i = row_index(base, j);
      if (VEC) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(A.value + i);
        pv[j] = v.x; pv[j + 1] = v.y;
      } else {
        pv[j] = A.value[i];
      }
    }
  };
  auto load_keys_tail = [&](uint64_t base) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const uint64_t i = row_index(base, j);
      const bool in = i < hi;
      pk[j] = in ? A.key[i] : TAD_KEY_SKIP;
      pt[j] = in ? A.t_end[i] : 0;
      if (HAS2) pk2[j] = in ? A.key2[i] : TAD_KEY_SKIP;
    }
  };
  auto load_values_tail = [&](uint64_t base) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const uint64_t i = row_index(base, j);
      pv[j] = i < hi ? A.value[i] : 0;
    }
  };
  const uint64_t nfull = hi > lo ? (hi - lo) / TILE : 0;
  const uint64_t ntiles = hi > lo ? (hi - lo + TILE - 1) / TILE : 0;
  if (ntiles) {
    if (nfull) { load_keys_full(lo); load_values_full(lo); }
    else { load_keys_tail(lo); load_values_tail(lo); }
  }
  lds_barrier();

  for (uint64_t tile = 0; tile < ntiles; ++tile) {
    const uint64_t base = lo + tile * TILE;
    const int next_kind = tile + 1 < nfull ? 1 : (tile + 1 < ntiles ? 2 : 0);  // workgroup-uniform
    // ---- phase 1: partition + tile-local cell of every row, ranked inside its partition (LDS atomic) ----
    uint32_t r_cell[NSLOT], r_pr[NSLOT];  // cell; partition << 16 | rank (rank < S <= 2^14... stored in 16 bits)
    uint32_t r_big = 0;                   // slots whose record carries the sentinel value (32-bit tile cells)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int64_t te = pt[j];
      bool kept = true;
      if (GENERIC && (A.f.end_time != 0 || has_ts)) {
        const uint64_t i = row_index(base, j);
        const int64_t ts = (has_ts && i < hi) ? A.t_start[i] : 0;
        kept = p_time_kept(te, ts, has_ts, A.f);
      }
      uint32_t bucket = 0;
      bool on_lattice;
      if (GENERIC) {
        on_lattice = p_bucket(A.L, te, bucket);
      } else {  // mode 0 or 1: one multiply-high
        const uint64_t d = (uint64_t)te - (uint64_t)A.L.t0;
        const uint64_t bq = A.L.mode == 0 ? d : __umul64hi(d, A.L.magic);
        on_lattice = (d >> 32) == 0 && bq < A.L.nb && bq * (uint64_t)A.L.step == d;
        bucket = (uint32_t)bq;
      }
#pragma unroll
      for (int h = 0; h < (HAS2 ? 2 : 1); ++h) {
        const int slot = j * (HAS2 ? 2 : 1) + h;
        const uint64_t k = h == 0 ? pk[j] : pk2[HAS2 ? j : 0];
        r_cell[slot] = cell_none;
        r_pr[slot] = 0xFFFFFFFFu;
        if (kept && k != TAD_KEY_SKIP && k < A.K) {  // same predicate as pass A: the slot is reserved
          uint32_t cell = cell_none;
          if (on_lattice) {
            used++;
            const bool big = pv[j] >= value_limit || (A.narrow_limit != 0 && pv[j] >= A.narrow_limit);
            if (!big || A.narrow_limit != 0) cell = bucket * KP + ((uint32_t)k & A.kp_mask);
            if (big) {  // rare: the value does not fit the record (or the 32-bit tile cell) -> overflow list
              const unsigned long long o = atomicAdd(A.ovf_count, 1ull);
              if (o < A.ovf_cap) { A.ovf[o].val = pv[j]; A.ovf[o].gcell = (unsigned long long)bucket * A.K + k; }
              else err |= DEV_ERR_OVERFLOW_LIST;
              if (A.ovf_keys != nullptr) atomicOr(A.ovf_keys + (k >> 5), 1u << (k & 31u));
              if (A.narrow_limit != 0) r_big |= 1u << slot;      // the record stays, with the sentinel value
            }
          } else {
            err |= DEV_ERR_OFF_LATTICE;  // wrong lattice hint, or the sampled gcd missed a residue: host re-derives
          }
          const uint32_t p = (uint32_t)(k >> A.shift_part);
          r_cell[slot] = cell;
          r_pr[slot] = (p << 16) | atomicAdd(&off[p], 1u);
        } else if (kept && k != TAD_KEY_SKIP) {
          err |= DEV_ERR_KEY_RANGE;   // pass A reports it too, but only for the rows it reads: with a sampled histogram that is one row in sixteen
        }
      }
    }
    if (next_kind == 1) load_keys_full(base + TILE);  // lands during phases 2-4
    else if (next_kind == 2) load_keys_tail(base + TILE);
    lds_barrier();
    // ---- phase 2: exclusive scan of the tile's partition histogram (in place) ----
    lds_exclusive_scan(off, F, s_wave);
    // ---- phase 3: place the records in partition order; per-partition global destination ----
#pragma unroll
    for (int slot = 0; slot < NSLOT; ++slot) {
      if (r_pr[slot] != 0xFFFFFFFFu) {
        const uint32_t pos = off[r_pr[slot] >> 16] + (r_pr[slot] & 0xFFFFu);
        rec[pos] = (((r_big >> slot) & 1u ? A.narrow_limit : pv[slot / (HAS2 ? 2 : 1)]) << A.cell_bits) | r_cell[slot];
        part[pos] = (uint16_t)(r_pr[slot] >> 16);
      }
    }
    if (next_kind == 1) load_values_full(base + TILE);  // needed again only in the next tile's phase 3
    else if (next_kind == 2) load_values_tail(base + TILE);
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) {
      const uint32_t o = off[p], c = off[p + 1] - o;
      const uint32_t gc = gcur[p];
      const bool full = c > gend[p] - gc;   // regions sized from a sampled histogram, or from counts that are not this batch's
      delta[p] = full ? 0xFFFFFFFFu : gc - o;  // global slot of the partition's first record of this tile, minus its tile position
      if (!full) gcur[p] = gc + c;
      else s_full = 1;
    }
    lds_barrier();
    // ---- phase 4: copy the runs out (consecutive lanes -> consecutive records of one partition) ----
    const uint32_t total = off[F];
    for (uint32_t idx = threadIdx.x; idx < total; idx += kPartThreads) {
      const uint32_t dl = delta[part[idx]];
      if (dl != 0xFFFFFFFFu) A.recs[dl + idx] = rec[idx];
    }
    lds_barrier();
    for (uint32_t p = threadIdx.x; p <= F; p += kPartThreads) off[p] = 0;
    lds_barrier();
  }
  if (A.fin != nullptr) {
    uint32_t *fo = A.fin + (size_t)blockIdx.x * F * 2;
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) { fo[2 * p] = gcur[p]; fo[2 * p + 1] = gend[p]; }
  } else {
    // exact regions are read whole by pass C: counts of this batch fill them to the last slot; counts that are not this batch's (a caller's
    // tad_key_hist) may leave slots over, which become `no cell` fillers like the line padding of the write-combining pass
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads)
      for (uint32_t g = gcur[p]; g < gend[p]; ++g) A.recs[g] = ~0ull;
  }
  // a region that was too small: sampled histogram — or exact counts that are not this batch's (a caller's tad_key_hist): the records were
  // left out, never written past the region; the host redoes the job with its own count
  if (s_full) err |= DEV_ERR_REGION_FULL;
  block_count_rows(used, err, A.ctr);
}

// ------------------------------------------------------------------------------------------------
// pass B, write-combining variant.  Measured (tools/ubench_runs.hip): what k_partition pays for is not the sort but
// the memory system's handling of ~7-record runs at arbitrary alignment (1.05-1.11 ms for the bare address pattern),
// while runs of 8 records on 64-byte boundaries cost 0.82 ms and whole 128-byte lines run at streaming speed.  So this
// variant never writes a partial sector: every (workgroup, partition) region starts on a 64-byte boundary (k_part_rows
// rounds the counts up to 8), records are appended to a per-partition queue of `cap` (9..12) slots in LDS, and after
// each tile every queue holding >= 8 records emits exactly one aligned 8-record sector (8 consecutive lanes, one
// store instruction) and slides its remainder down.  A record that finds its queue full goes straight to the END of
// the region (downward cursor, rare); at the end of the kernel the leftovers (< 8 per partition) and `no cell`
// fillers close the gap, so pass C reads the same contiguous partitions as before and skips the fillers.
// LDS: q[F * cap] u64 | cnt[F + 1] | gcur[F] | gend[F] | nsp[F] u32 | jobs[F] u16.  Tiles are small (RPT rows per thread, 2 * RPT
// with a second key), two register sets alternate so that every load has more than a tile to land.
// ------------------------------------------------------------------------------------------------

// SEC = records per emitted piece: 8 (one 64-byte sector) when LDS leaves only 9..12 queue slots per partition, 16 (one
// whole 128-byte line — streaming speed in the micro-benchmark) when there are few enough partitions for >= 22 slots.
// TS (with GENERIC): the job has a start_time and a 16-byte aligned flowStartSeconds column — it is loaded with the tile's other columns, one
// more unconditional 16-byte load per row pair.  (Loaded in the append phase, as the generic variant without TS still does for an unaligned
// column, the load sits between the LDS appends and is waited for together with the prefetched rows of the next tile: pass B 0.80 against
// 0.66 ms.  A first form prefetched it behind a run-time test inside load_tile: the compiler then cannot count the loads in flight and waits
// for all of them everywhere — slower even for jobs without a start_time, profiles/r6_s14_*.)
template <int RPT, int SEC, bool HAS2, bool GENERIC, bool TS = false>
__global__ __launch_bounds__(kPartThreads) void k_partition_wc(PartArgs A, uint32_t cap, int G) {
  static_assert(!TS || GENERIC, "a start_time filter is a generic job");
  static_assert(RPT == 2 || RPT == 4, "rows per thread: one or two 16-byte loads per column");
  static_assert(SEC == 8 || SEC == 16, "emit one 64-byte sector or one 128-byte line");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t TILE = (uint32_t)RPT * kPartThreads;
  const uint32_t F = A.nparts;
  unsigned long long *q = reinterpret_cast<unsigned long long *>(smem);
  // cnt[p]: records in partition p's queue (low half; it runs past the queue's depth for the records that spilled, < 2^16 within a tile) under
  // the queue's first slot in q (high half, constant): the append's one atomic returns both.  F + 1 words: a queue's depth is the next
  // word's offset minus its own.
  uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + (size_t)F * cap * 8);
  uint32_t *gcur = cnt + F + 1;   // next sector of (this workgroup, partition), upward
  uint32_t *gend = gcur + F;  // end of the region (constant)
  uint32_t *nsp = gend + F;   // records spilled to the top of the region, downward from gend (a counter: it cannot wrap)
  uint16_t *jobs = reinterpret_cast<uint16_t *>(nsp + F);
#if TAD_ADAPTIVE_QUEUES
  __shared__ uint32_t s_red[3 * (kPartThreads / 64)];
  __shared__ float s_redf[kPartThreads / 64];
#endif
  // (the two counters below are written in a tile's APPEND phase and read in its emit phase: one pair per tile parity, so that clearing the
  //  pair of tile t — after the barrier that ends its emit phase — cannot meet an append of tile t + 1, which starts behind that same barrier)
  __shared__ uint32_t s_njobs2[2];
  // Records that find their queue full are PARKED here during the append phase and stored in the emit phase (round 4).  They used to be
  // stored straight away — a global store between the LDS appends, for which the compiler drains the prefetched rows of the next tiles
  // (vmcnt counts stores too on gfx9): with ~1 % of the records spilling three quarters of all wavefront-tiles took that drain, which is
  // why whole 128-byte lines at 18 queue slots had measured 0.99 against 0.80 ms (profiles/r3_v8_c4_line18_ab.log).
  constexpr uint32_t kSpillSlots = 288;
  __shared__ unsigned long long s_spill_rec[kSpillSlots];
  __shared__ uint32_t s_spill_at[kSpillSlots];
  __shared__ uint32_t s_nspill2[2];

  if (A.fin != nullptr && (A.ctr->err & DEV_ERR_REGION_FULL)) return;   // k_part_offsets: this table is redone with the exact histogram
  {
    const uint32_t *my = A.offs32 + (size_t)blockIdx.x * F;
    const uint32_t *nx = A.offs32 + (size_t)(blockIdx.x + 1) * F;
    const bool last = (int)blockIdx.x + 1 == G;
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) {
      const uint32_t ps = (uint32_t)A.part_start[p];
      nsp[p] = 0;
      gcur[p] = ps + my[p];
      gend[p] = last ? (uint32_t)A.part_start[p + 1] : ps + nx[p];
    }
    if (threadIdx.x == 0) { s_njobs2[0] = s_njobs2[1] = 0; s_nspill2[0] = s_nspill2[1] = 0; }
  }
  // Queue depths follow the workgroup's histogram row.  A queue keeps SEC - 1 slots for its leftovers; behind them it needs room for a
  // tile's arrivals, ~Poisson(lambda_p) with lambda_p = the tile's record slots x the partition's share of this workgroup's records
  // (region size = pass A's count, or its capacity estimate).  The pool is shared out as lambda_p + z sqrt(lambda_p) with ONE z for all
  // partitions — equal overflow probability everywhere, which is what minimises the records that find their queue full.  Hashed keys:
  // every queue as deep as the others, as before.  Keys that come and go with time, ids in order of first appearance, a hot key: the
  // partitions a workgroup really feeds get the LDS of those it never touches, and their records leave as whole lines instead of one
  // by one over the region's top.  Depths only steer WHERE a record waits: the records written, and everything downstream, do not change.
  {
#if TAD_ADAPTIVE_QUEUES
    constexpr float kTileSlots = (float)(TILE * (HAS2 ? 2u : 1u));
    const uint32_t pool = F * cap;
    uint32_t *tmp = reinterpret_cast<uint32_t *>(q);     // (the queues are empty until the first append)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t rsum = 0, ract = 0, rmax = 0;      // (a thread reads the regions it wrote itself: same stride as the loop above)
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) { const uint32_t r = gend[p] - gcur[p]; rsum += r; ract += r != 0u; rmax = r > rmax ? r : rmax; }
    for (int d = 32; d >= 1; d >>= 1) { rsum += __shfl_down(rsum, d); ract += __shfl_down(ract, d); const uint32_t o = __shfl_down(rmax, d); rmax = o > rmax ? o : rmax; }
    if (lane == 0) { s_red[wave] = rsum; s_red[kPartThreads / 64 + wave] = ract; s_red[2 * (kPartThreads / 64) + wave] = rmax; }
    lds_barrier();
    rsum = 0; ract = 0; rmax = 0;
    for (int w = 0; w < kPartThreads / 64; ++w) {
      rsum += s_red[w]; ract += s_red[kPartThreads / 64 + w];
      rmax = s_red[2 * (kPartThreads / 64) + w] > rmax ? s_red[2 * (kPartThreads / 64) + w] : rmax;
    }
    // ... when there is something to follow: a workgroup that feeds every partition about alike (no region above 1.5 x the mean, at least
    // three quarters of them in use) keeps the equal depths — rounding lambda + z sqrt(lambda) to whole slots would give half of C4's
    // queues 17 slots and the other half 18 for counts that differ by their Poisson noise (measured: pass B + 2.7 %)
    const bool skewed = (unsigned long long)rmax * F * 2u > (unsigned long long)rsum * 3u || ract * 4u < F * 3u;
    const float per_rec = rsum ? kTileSlots / (float)rsum : 0.0f;
    float sq = 0.0f;
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) sq += sqrtf((float)(gend[p] - gcur[p]) * per_rec);
    for (int d = 32; d >= 1; d >>= 1) sq += __shfl_down(sq, d);
    if (lane == 0) s_redf[wave] = sq;
    lds_barrier();
    sq = 0.0f;
    for (int w = 0; w < kPartThreads / 64; ++w) sq += s_redf[w];      // (fixed order: every thread gets the same bits)
    // what is left after the leftovers' slots; half a slot per queue for the rounding below, two slots for the float sums
    const float room = (float)pool - ((float)(SEC - 1) + 0.5f) * (float)ract - 2.0f;
    const float z = room > kTileSlots && sq > 0.0f ? (room - kTileSlots) / sq : 0.0f;
    const float shrink = room > kTileSlots ? 1.0f : (room > 0.0f ? room / kTileSlots : 0.0f);
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) {
      const uint32_t r = gend[p] - gcur[p];
      const float lam = (float)r * per_rec;
      tmp[p] = r ? (uint32_t)(SEC - 1) + (uint32_t)(lam * shrink + z * sqrtf(lam) + 0.5f) : 0u;
    }
    lds_barrier();
    lds_exclusive_scan(tmp, F, s_red);
    const bool ok = skewed && tmp[F] <= pool;                          // (the sum fits by the arithmetic above: a safety net, not a code path)
    for (uint32_t p = threadIdx.x; p <= F; p += kPartThreads) cnt[p] = (ok ? tmp[p] : p * cap) << 16;   // (cnt lies behind the queues: no overlap with tmp)
#else
    for (uint32_t p = threadIdx.x; p <= F; p += kPartThreads) cnt[p] = (p * cap) << 16;
#endif
  }

  const uint64_t lo = (uint64_t)blockIdx.x * A.chunk;
  const uint64_t hi = lo + A.chunk < A.n ? lo + A.chunk : A.n;
  const bool has_ts = TS || (GENERIC && A.t_start != nullptr && A.f.start_time != 0);
  const uint32_t KP = A.kp_mask + 1u;
  const uint32_t cell_none = (1u << A.cell_bits) - 1u;
  const unsigned long long value_limit = A.value_limit;
  uint32_t err = 0, used = 0;
  const uint64_t nfull = hi > lo ? (hi - lo) / TILE : 0;
  const uint64_t ntiles = hi > lo ? (hi - lo + TILE - 1) / TILE : 0;

  struct Rows { uint64_t k[RPT], k2[HAS2 ? RPT : 1], v[RPT]; int64_t t[RPT], ts[TS ? RPT : 1]; };
  auto row_index = [&](uint64_t base, int j) -> uint64_t { return base + (uint64_t)(j >> 1) * (2 * kPartThreads) + 2 * threadIdx.x + (j & 1); };
  auto load_tile = [&](Rows &r, uint64_t tile) {  // workgroup-uniform branches only
    const uint64_t base = lo + tile * TILE;
    if (tile < nfull) {
#pragma unroll
      for (int j = 0; j < RPT; j += 2) {
        const uint64_t i = row_index(base, j);
        const ulonglong2 k = *reinterpret_cast<const ulonglong2 *>(A.key + i);
        const longlong2 t = *reinterpret_cast<const longlong2 *>(A.t_end + i);
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(A.value + i);
        r.k[j] = k.x; r.k[j + 1] = k.y; r.t[j] = t.x; r.t[j + 1] = t.y; r.v[j] = v.x; r.v[j + 1] = v.y;
        if (HAS2) { const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(A.key2 + i); r.k2[j] = k2.x; r.k2[j + 1] = k2.y; }
        if (TS) { const longlong2 ts = *reinterpret_cast<const longlong2 *>(A.t_start + i); r.ts[TS ? j : 0] = ts.x; r.ts[TS ? j + 1 : 0] = ts.y; }
      }
    } else if (tile < ntiles) {
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const uint64_t i = row_index(base, j);
        const bool in = i < hi;
        r.k[j] = in ? A.key[i] : TAD_KEY_SKIP;
        r.t[j] = in ? A.t_end[i] : 0;
        r.v[j] = in ? A.value[i] : 0;
        if (HAS2) r.k2[j] = in ? A.key2[i] : TAD_KEY_SKIP;
        if (TS) r.ts[TS ? j : 0] = in ? A.t_start[i] : 0;
      }
    }
  };

  auto process = [&](Rows &r, uint64_t tile) {
    const uint64_t base = lo + tile * TILE;
    uint32_t &s_njobs = s_njobs2[tile & 1u], &s_nspill = s_nspill2[tile & 1u];
    // ---- append: every record joins its partition's queue (LDS), or spills to the end of the region ----
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int64_t te = r.t[j];
      bool kept = true;
      if (GENERIC && (A.f.end_time != 0 || has_ts)) {
        const uint64_t i = row_index(base, j);
        const int64_t ts = TS ? r.ts[TS ? j : 0] : ((has_ts && i < hi) ? A.t_start[i] : 0);
        kept = p_time_kept(te, ts, has_ts, A.f);
      }
      uint32_t bucket = 0;
      bool on_lattice;
      if (GENERIC) {
        on_lattice = p_bucket(A.L, te, bucket);
      } else {
        const uint64_t d = (uint64_t)te - (uint64_t)A.L.t0;
        const uint64_t bq = A.L.mode == 0 ? d : __umul64hi(d, A.L.magic);
        on_lattice = (d >> 32) == 0 && bq < A.L.nb && bq * (uint64_t)A.L.step == d;
        bucket = (uint32_t)bq;
      }
#pragma unroll
      for (int h = 0; h < (HAS2 ? 2 : 1); ++h) {
        const uint64_t k = h == 0 ? r.k[j] : r.k2[HAS2 ? j : 0];
        if (kept && k != TAD_KEY_SKIP && k < A.K) {  // same predicate as pass A: the slot is reserved
          uint32_t cell = cell_none;
          unsigned long long vrec = r.v[j];
          if (on_lattice) {
            used++;
            const bool big = r.v[j] >= value_limit || (A.narrow_limit != 0 && r.v[j] >= A.narrow_limit);
            if (!big || A.narrow_limit != 0) cell = bucket * KP + ((uint32_t)k & A.kp_mask);
            if (big) {
              const unsigned long long o = atomicAdd(A.ovf_count, 1ull);
              if (o < A.ovf_cap) { A.ovf[o].val = r.v[j]; A.ovf[o].gcell = (unsigned long long)bucket * A.K + k; }
              else err |= DEV_ERR_OVERFLOW_LIST;
              if (A.ovf_keys != nullptr) atomicOr(A.ovf_keys + (k >> 5), 1u << (k & 31u));
              if (A.narrow_limit != 0) vrec = A.narrow_limit;   // the record stays, with the sentinel value: the tile cell becomes all ones
            }
          } else {
            err |= DEV_ERR_OFF_LATTICE;
          }
          const uint32_t p = (uint32_t)(k >> A.shift_part);
          const unsigned long long rec = (vrec << A.cell_bits) | cell;
#if TAD_WAVE_AGG
          // Rows that arrive sorted (a GROUP BY result, a view ordered by its key; the first rows of any table whose ids were handed out in
          // order of first appearance) send most lanes of a wavefront to ONE partition: 64 atomics on one LDS word, a queue of ~20 slots
          // for hundreds of records.  The lanes that share the first active lane's partition take, when they are many, one slice of the
          // region's top together — one atomic, consecutive slots, one coalesced store; hashed keys never meet the condition.
          {
            uint32_t p0 = __builtin_amdgcn_readfirstlane(p);
            unsigned long long grp = __ballot(p == p0);
            if (__popcll(grp) < kWaveAggMin) {   // (the first lane may hold the odd row out — an old key among new ones: ask the last lane too)
              p0 = __builtin_amdgcn_readlane(p, 63 - __clzll(__ballot(true)));
              grp = __ballot(p == p0);
            }
            if (__popcll(grp) >= kWaveAggMin && p == p0) {
              const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(grp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)grp, 0u));
              uint32_t k0 = 0;
              if (rank == 0) k0 = atomicAdd(&nsp[p0], (uint32_t)__popcll(grp));
              k0 = (uint32_t)__shfl((int)k0, (int)__builtin_ctzll(grp)) + rank;   // (from the group's OWN first lane: right whichever lanes execute this together)
              if (k0 < gend[p0] - gcur[p0]) A.recs[gend[p0] - 1u - k0] = rec;
              else err |= DEV_ERR_REGION_FULL;
              continue;
            }
          }
#endif
          const uint32_t word = atomicAdd(&cnt[p], 1u);
          const uint32_t pos = word & 0xFFFFu, qo = word >> 16, qc = (cnt[p + 1] >> 16) - qo;
          // the append that makes the queue hold a whole piece registers it for this tile's emit phase (round 5: there used to be a scan over all
          // queues behind a barrier of its own).  A queue leaves the emit phase with fewer than SEC records, so its count passes SEC - 1 exactly
          // once in a tile in which it ends up with a whole piece or more (the emit phase writes ALL its whole pieces), and never otherwise.
          if (pos == (uint32_t)SEC - 1u) jobs[atomicAdd(&s_njobs, 1u)] = (uint16_t)p;
          if (pos < qc) q[qo + pos] = rec;
          else {  // queue full (a burst, or a hot key): top of the region
            const uint32_t k = atomicAdd(&nsp[p], 1u);
            if (k < gend[p] - gcur[p]) {                                  // (gcur only moves in the emit phase)
              const uint32_t at = gend[p] - 1u - k;
#if TAD_WAVE_AGG
              const unsigned long long spl = __ballot(true);      // one update of the parking counter per wavefront, not per record
              const uint32_t srank = __builtin_amdgcn_mbcnt_hi((uint32_t)(spl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)spl, 0u));
              uint32_t sl = 0;
              if (srank == 0) sl = atomicAdd(&s_nspill, (uint32_t)__popcll(spl));
              sl = (uint32_t)__shfl((int)sl, (int)__builtin_ctzll(spl)) + srank;
#else
              const uint32_t sl = atomicAdd(&s_nspill, 1u);
#endif
              if (sl < kSpillSlots) { s_spill_rec[sl] = rec; s_spill_at[sl] = at; }   // stored in the emit phase
              else A.recs[at] = rec;                                       // a hot key: more spills in one tile than the buffer holds
            } else err |= DEV_ERR_REGION_FULL;                             // sampled regions only: the region is full
          }
        } else if (kept && k != TAD_KEY_SKIP) {
          err |= DEV_ERR_KEY_RANGE;   // pass A reports it too, but only for the rows it reads: with a sampled histogram that is one row in sixteen
        }
      }
    }
    load_tile(r, tile + 2);  // this register set is free again: lands during the rest of this tile and the whole next one
    lds_barrier();
    // ---- emit: SEC consecutive lanes write the queue's whole aligned pieces (one store instruction each); the remainder
    // slides down to the front of the queue ----
    const uint32_t nj = s_njobs;
    for (uint32_t j = threadIdx.x / SEC; j < nj; j += kPartThreads / SEC) {
      const uint32_t p = jobs[j], sl = threadIdx.x & (SEC - 1u);
      const uint32_t word = cnt[p], qo = word >> 16, qcap = (cnt[p + 1] >> 16) - qo;
      const uint32_t c0 = word & 0xFFFFu, c = c0 < qcap ? c0 : qcap, g = gcur[p];      // (the count runs past the queue's depth for the records that spilled)
      unsigned long long *qp = q + qo;
      const uint32_t whole = c & ~(uint32_t)(SEC - 1);
      const uint32_t room = gend[p] - g, sp = nsp[p];
      const bool fits = sp <= room && whole <= room - sp;   // always with exact regions; a region sized from a sampled histogram may be full
      if (fits) { for (uint32_t o = 0; o < whole; o += SEC) A.recs[g + o + sl] = qp[o + sl]; }
      else err |= DEV_ERR_REGION_FULL;
      const bool mv = whole + sl < c;
      const unsigned long long tail = mv ? qp[whole + sl] : 0ull;
      if (mv) qp[sl] = tail;  // (one wavefront, LDS in order: every lane has read before any lane writes)
      if (sl == 0) { cnt[p] = (qo << 16) | (c - whole); gcur[p] = g + whole; }
    }
    {   // the parked spills of this tile
      const uint32_t ns = s_nspill < kSpillSlots ? s_nspill : kSpillSlots;
      for (uint32_t i = threadIdx.x; i < ns; i += kPartThreads) A.recs[s_spill_at[i]] = s_spill_rec[i];
    }
    lds_barrier();
    if (threadIdx.x == 0) { s_njobs = 0; s_nspill = 0; }  // this parity's pair: next written in the append phase of tile + 2, two barriers from here
  };

  Rows R0, R1;
  load_tile(R0, 0);
  load_tile(R1, 1);
  lds_barrier();
  for (uint64_t tile = 0; tile < ntiles; tile += 2) {
    process(R0, tile);
    if (tile + 1 < ntiles) process(R1, tile + 1);
  }
  // ---- close the regions: leftovers, then (exact regions) `no cell` fillers up to the spilled records; sampled regions keep
  // their slack untouched and record where the valid records end / the spilled ones start ----
  for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) {
    const uint32_t word = cnt[p], qo = word >> 16, qcap = (cnt[p + 1] >> 16) - qo;
    const uint32_t c = (word & 0xFFFFu) < qcap ? (word & 0xFFFFu) : qcap;
    uint32_t g = gcur[p];
    const uint32_t room = gend[p] - g;
    const uint32_t e = gend[p] - (nsp[p] < room ? nsp[p] : room);   // first spilled record
    if (A.fin != nullptr) {
      if (c <= e - g && g <= e) { for (uint32_t i = 0; i < c; ++i) A.recs[g + i] = q[qo + i]; g += c; }
      else err |= DEV_ERR_REGION_FULL;
      uint32_t *fo = A.fin + ((size_t)blockIdx.x * F + p) * 2;
      fo[0] = g; fo[1] = e;
    } else if (c <= e - g && g <= e) {
      for (uint32_t i = 0; i < c; ++i) A.recs[g + i] = q[qo + i];
      for (g += c; g < e; ++g) A.recs[g] = ~0ull;
    } else {
      err |= DEV_ERR_REGION_FULL;   // exact counts that are not this batch's (a caller's tad_key_hist): nothing is written past the region
    }
  }
  block_count_rows(used, err, A.ctr);
}

// ------------------------------------------------------------------------------------------------
// pass C — aggregate the records of a partition in an LDS tile, write the tile out.
// LDS carve: vals[KP*T] u64 | flags[KP*T] u8
//
// Work unit = a SLICE of at most kSliceRecords records of one partition, so that a partition swollen by a hot key
// (real flow tables have heavy hitters: one key with half the rows made this pass 30x slower when it was one
// workgroup per partition) is spread over many CUs.  A partition with one slice — every partition of a uniform
// table — stores its tile directly; the slices of a split partition merge into the (pre-zeroed) grid tile with
// agent-scope integer atomics, which commute, so the aggregates are bit-exact either way.
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t kSliceRecords = 1u << 17;

// Geometry of pass C, decided by the plan: a partition is KP = 2^shift_part keys x T buckets; its cells are processed
// in n_chunks rounds of TB buckets (TB * KP <= kTileCells cells per LDS tile).  n_chunks == 1 whenever the whole
// KP x T tile fits (C2); wide grids (many keys and/or many buckets) take larger key blocks — so that pass B keeps a
// few thousand partitions at most — and several rounds over the partition's records, which are L2 / MALL resident
// after the first round (a partition holds a few hundred KB of records).
struct TileGeom {
  int shift_part;
  int cell_bits;
  uint32_t nparts;
  uint32_t tb;        // buckets per round
  uint32_t n_chunks;  // ceil(T / tb)
  uint32_t par_rounds;  // 1: every round of a slice is its own workgroup (grid = slices x rounds), see k_tile_aggregate
  uint32_t slice_len;   // record slots per slice (kSliceRecords; twice that when the regions carry the slack of a sampled histogram)
  uint32_t kt;          // settle mode: keys per tile (rounds split the partition by key sub-range); 0 = bucket rounds
};

// NARROW (settle mode with `max`, round 4): a tile cell is ONE 32-bit word, value + 1 (0 = absent) instead of 8 bytes + a flag byte: 2.25x the
// keys per tile, three key rounds instead of six at C4 (1024 keys x 100 buckets) — every round streams the partition's records again, so the
// pass costs what its rounds cost.  A value >= 2^32 - 2 does not fit a cell.  Pass B sends it to the overflow list (as it does values that do
// not fit a packed record), marks its key in a bitmap and leaves a record with the SENTINEL value 2^32 - 2 in the stream, so that the cell
// becomes the all-ones word without a single extra instruction in this kernel's hot loop (a compare + branch there was measured: +0.09 ms).
// The tile leaves exactly the marked keys to k_dbscan_scan_redo — with their series contiguous behind the redo list, the all-ones cells flagged:
// the redo pass reads those few cells from the grid once the fold has put the real values there, instead of gathering the whole column.
// `max` only: a sum can outgrow 32 bits without any operand doing so.
template <bool OPMAX, bool SETTLE, bool NARROW>
__device__ __forceinline__ void tile_aggregate_body(const unsigned long long *__restrict__ recs,
                                                    const unsigned long long *__restrict__ part_start,
                                                    SliceTable st, TileGeom tg, Grid g,
                                                    const uint32_t *__restrict__ offs32, const uint32_t *__restrict__ fin, int G,
                                                    SettleArgs sa, const unsigned long long *__restrict__ ovf_count_in) {
  // Rounds as workgroups: a partition whose KP x T block needs R > 1 LDS tiles is read by R workgroups, one per bucket
  // round, instead of R times by one.  The R workgroups of a slice get block ids x + 8 * (R * j + r): the same XCD
  // (blocks are dealt round-robin over the 8 XCDs) and adjacent in dispatch order, so they stream the same records at
  // the same time and all but the first reader hit that XCD's L2.
  uint32_t s_idx = blockIdx.x, r_lo = 0, r_hi = tg.n_chunks;
  if (tg.par_rounds && tg.n_chunks > 1) {
    const uint32_t x = blockIdx.x & 7u, y = blockIdx.x >> 3;
    r_lo = y % tg.n_chunks;
    r_hi = r_lo + 1;
    s_idx = (y / tg.n_chunks) * 8u + x;
  }
  if (s_idx >= *st.n_slices) return;
  const uint32_t p = st.slice_part[s_idx];
  const uint32_t first = st.slice_first[p];
  const uint32_t nsl = (p + 1 < tg.nparts ? st.slice_first[p + 1] : *st.n_slices) - first;
  const bool split = nsl > 1;
  const int shift_part = tg.shift_part;
  const uint32_t KP = 1u << shift_part;
  const uint32_t T = (uint32_t)g.T;
  const uint64_t k0 = (uint64_t)p << shift_part;
  // (the grid tile of a split partition was zeroed by k_part_offsets: its slices merge with atomics)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_nlist, s_lbase, s_nredo, s_rbase;
  __shared__ uint32_t s_ovfw[40];   // settle mode: the words of the overflow-key bitmap that cover this round's keys (<= 1024 keys + alignment)
  const uint32_t cell_none = (1u << tg.cell_bits) - 1u;
  const unsigned long long plo = part_start[p], phi = part_start[p + 1];
  const unsigned long long lo = plo + (unsigned long long)(s_idx - first) * tg.slice_len;
  const unsigned long long hi = lo + tg.slice_len < phi ? lo + tg.slice_len : phi;
  for (uint32_t chunk = r_lo; chunk < r_hi; ++chunk) {
    // SETTLE: round `chunk` holds the keys [chunk * kt, (chunk + 1) * kt) of the partition with ALL their buckets — tile cell =
    // bucket * kt + key within the sub-range — instead of all keys with the buckets [b_lo, b_lo + nb)
    const uint32_t KT = SETTLE ? tg.kt : KP;                       // keys per tile row
    const uint32_t kt0 = SETTLE ? chunk * KT : 0u;                  // first key of the tile within the partition
    const uint32_t b_lo = SETTLE ? 0u : chunk * tg.tb;
    const uint32_t nb = SETTLE ? T : (b_lo + tg.tb <= T ? tg.tb : T - b_lo);
    const uint32_t cells = SETTLE ? nb * KT : nb << shift_part;
    const uint32_t c_lo = b_lo << shift_part;  // first partition-local cell of this round
    unsigned long long *vals = reinterpret_cast<unsigned long long *>(smem);
    uint32_t *vals32 = reinterpret_cast<uint32_t *>(smem);          // NARROW: value + 1, 0 = absent
    uint8_t *flags = smem + (size_t)(SETTLE ? tg.tb * KT : tg.tb << shift_part) * (NARROW ? 4 : 8);   // (NARROW: no flag bytes; the settle bookkeeping starts here)
    auto cell_present = [&](uint32_t c) -> bool { return NARROW ? vals32[c] != 0u : (flags[c] & FLAG_PRESENT) != 0; };
    auto cell_value = [&](uint32_t c) -> unsigned long long { return NARROW ? (unsigned long long)(vals32[c] - 1u) : vals[c]; };   // (of a present cell)
    if (chunk != r_lo) __syncthreads();  // the previous round's tile has been written out
    if (NARROW) {
      for (uint32_t c = threadIdx.x; c < cells; c += kPartThreads) vals32[c] = 0u;
    } else {
      for (uint32_t c = threadIdx.x; c < cells; c += kPartThreads) vals[c] = 0ull;
      for (uint32_t c = threadIdx.x; c < (cells + 3) / 4; c += kPartThreads) reinterpret_cast<uint32_t *>(flags)[c] = 0u;
    }
    __syncthreads();
    auto apply = [&](unsigned long long r) {
      const uint32_t cg = (uint32_t)r & cell_none;
      uint32_t c = cg - c_lo;  // wraps for cells before this round: the unsigned compare rejects them
      if (SETTLE) {
        const uint32_t kk = (cg & (KP - 1u)) - kt0;               // wraps for keys before this round
        if (cg == cell_none || kk >= KT) return;
        c = __umul24(cg >> shift_part, KT) + kk;   // (bucket < 2^16, KT <= 256: the 24-bit multiply issues at full rate, v_mul_lo_u32 at a quarter)
      } else if (c >= cells) return;   // (also rejects `no cell`: the all-ones cell is >= KP * T, plan_tiles reserves it)
      const unsigned long long v = r >> tg.cell_bits;
      if (NARROW) { atomicMax(&vals32[c], (uint32_t)v + 1u); return; }     // (pass B kept values >= 2^32 - 1 out of the records)
      if (OPMAX) atomicMax(&vals[c], v);
      else atomicAdd(&vals[c], v);
      flags[c] = FLAG_PRESENT;
    };
    constexpr int U = 8;
    // Settle mode reads one bit per key of the overflow-key bitmap in its per-key pass: as a global load inside that pass it was a memory round trip on
    // the critical path of every tile (profiling build, round 5: 10 us of a C4 tile's 40 in that pass).  The words are requested HERE, stay in a
    // register while the records stream, and go to LDS behind the walk's barrier.
    uint32_t ovf_w = 0;
    const uint64_t ovf_k0 = (k0 + (SETTLE ? chunk * tg.kt : 0u)) & ~(uint64_t)31;
    if (SETTLE && sa.ovf_keys != nullptr && threadIdx.x < 40u) {
      const uint64_t wi = (ovf_k0 >> 5) + threadIdx.x;
      if (wi < (g.K + 31) / 32) ovf_w = sa.ovf_keys[wi];
    }
    if (fin != nullptr) {
      // Regions sized from a sampled histogram: the partition's record space [plo, phi) is tiled by one region per pass-B
      // workgroup; a region holds valid records in [start, fin_lo) and [fin_hi, end) (spilled ones), untouched slack
      // between.  Wavefront w takes the regions w, w + 16, ...; the bounds of its regions are fetched up front (one lane
      // each) and broadcast, so that the record loads of consecutive regions are not serialised behind a bounds load.
      const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
      const uint32_t F = tg.nparts;
      for (uint32_t w0 = wave; w0 < (uint32_t)G; w0 += 64u * (kPartThreads / 64)) {
        const uint32_t wm = w0 + lane * (kPartThreads / 64);       // the region lane `lane` fetches the bounds of
        unsigned long long b_start = 0, b_lo = 0, b_hi = 0, b_end = 0;
        if (wm < (uint32_t)G) {
          b_start = plo + offs32[(size_t)wm * F + p];
          b_end = wm + 1 < (uint32_t)G ? plo + offs32[(size_t)(wm + 1) * F + p] : phi;
          b_lo = fin[((size_t)wm * F + p) * 2];
          b_hi = fin[((size_t)wm * F + p) * 2 + 1];
        }
        for (uint32_t j = 0; j < 64u && w0 + j * (kPartThreads / 64) < (uint32_t)G; ++j) {
          const unsigned long long r_start = __shfl(b_start, (int)j), r_lo = __shfl(b_lo, (int)j), r_hi = __shfl(b_hi, (int)j),
                                   r_end = __shfl(b_end, (int)j);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            unsigned long long a = half == 0 ? r_start : r_hi, b = half == 0 ? r_lo : r_end;
            a = a < r_start ? r_start : a;   // (a job whose region overflowed is redone, but must not read out of bounds first)
            b = b > r_end ? r_end : b;
            a = a < lo ? lo : a;             // clip to this slice
            b = b > hi ? hi : b;
            constexpr int UR = TAD_REGION_UR;   // 12: a region of the uniform C2 table holds ~500 +- 22 records: one batch of 768 covers it
            for (unsigned long long i = a + lane; i < b; i += UR * 64) {
              unsigned long long r[UR];
#pragma unroll
              for (int u = 0; u < UR; ++u) r[u] = i + u * 64 < b ? recs[i + u * 64] : ~0ull;
#pragma unroll
              for (int u = 0; u < UR; ++u) apply(r[u]);
            }
          }
        }
      }
    }
    // exact regions: the slice is one contiguous run of records.  A wavefront takes chunks of 512 consecutive records (its eight
    // loads cover one contiguous 4 KB, immediate offsets off one address) instead of eight rows 8 KB apart of a workgroup-wide
    // stride: C4 pass C 0.75 -> 0.70 ms, same box alternating (profiles/r3_v6_passC_wave_chunks_ab.log).  Fewer instructions per
    // record are NOT what that bought: scalar region bounds (-40 % instructions in the walk above) and dropping the flag write for
    // `max` changed nothing (profiles/r3_v6_passC_fewer_instructions_ab.log) — the pass follows the bytes it pulls through L2.
    if (fin == nullptr) {
      const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
      constexpr unsigned long long kChunk = 64ull * U;
      unsigned long long i = lo + (unsigned long long)wave * kChunk + lane;
      for (; i + (U - 1) * 64 < hi; i += kChunk * (kPartThreads / 64)) {
        const unsigned long long *q = recs + i;
        unsigned long long r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = q[u * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) apply(r[u]);
      }
      if (i < hi) {   // this wavefront's last, ragged chunk (`no cell` for the slots past the end)
        unsigned long long r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = i + u * 64 < hi ? recs[i + u * 64] : ~0ull;
#pragma unroll
        for (int u = 0; u < U; ++u) apply(r[u]);
      }
    }
    __syncthreads();
    // SETTLE: k_dbscan_scan's per-key pass on the LDS tile (one thread per key, buckets in order: the same operations in the
    // same order, so n / mean / M2 are that kernel's bits), then only the columns of unsettled keys are written
    uint8_t *settled = flags + (((size_t)((SETTLE && !NARROW) ? tg.tb * KT : 0u)) + 3 & ~(size_t)3);   // [KT] u8, then the tile's listed keys [KT] u32
    uint32_t *tile_list = reinterpret_cast<uint32_t *>(settled + ((KT + 3u) & ~3u));
    uint32_t *tile_redo = tile_list + KT;                            // [KT] the tile's keys for the redo list
    bool skip_cols = false;
    if (SETTLE) {
      // (without the per-key bitmap any record on the overflow list sends the whole job to the redo path)
      skip_cols = !split && (sa.ovf_keys != nullptr || *ovf_count_in == 0ull);
      if (threadIdx.x == 0) { s_nlist = 0; s_nredo = 0; }
      if (threadIdx.x < 40u) s_ovfw[threadIdx.x] = ovf_w;
      __syncthreads();
      // Four threads per key (adjacent lanes), each over every fourth bucket with its own first value as the shift of its sums;
      // the partials meet in two xor-shuffles (sums re-based onto lane 0's shift: exact algebra, no division).  One thread per key
      // was a 100-step FP64 dependency chain on 3 of the workgroup's 16 wavefronts: +0.26 ms on pass C at C4.
      for (uint32_t w = threadIdx.x; w < ((KT + 15u) & ~15u) * 4u; w += kPartThreads) {
        const uint32_t kk = w >> 2, part = w & 3u;
        const uint64_t k = k0 + kt0 + kk;
        const bool live = kk < KT && kt0 + kk < KP && k < g.K;
        uint32_t n = 0;
        // (min / max start at +-infinity instead of at the first value: the same result for n > 0 — every value is finite — and unused for n == 0)
        double mn = __builtin_huge_val(), mx = -__builtin_huge_val(), x0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (live && skip_cols) {
          for (uint32_t b = part; b < nb; b += 4u) {
            const uint32_t c = __umul24(b, KT) + kk;
            if (cell_present(c)) {
              const double x = (double)cell_value(c);
              if (n == 0) x0 = x;
              mn = fmin(mn, x);
              mx = fmax(mx, x);
              const double d = x - x0;
              s1 += d;
              s2 += d * d;
              n++;
            }
          }
        }
#pragma unroll
        for (int d = 1; d <= 2; d <<= 1) {       // merge with the partner's partial; the lower lane's shift is kept
          const uint32_t on = __shfl_xor(n, d);
          const double omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d), ox0 = __shfl_xor(x0, d), os1 = __shfl_xor(s1, d), os2 = __shfl_xor(s2, d);
          const bool low = (part & (uint32_t)d) == 0;
          const uint32_t na = low ? n : on, nb2 = low ? on : n;                   // a = the lower lane's partial, b = the higher one's
          const double amn = low ? mn : omn, amx = low ? mx : omx, ax0 = low ? x0 : ox0, as1 = low ? s1 : os1, as2 = low ? s2 : os2;
          const double bmn = low ? omn : mn, bmx = low ? omx : mx, bx0 = low ? ox0 : x0, bs1 = low ? os1 : s1, bs2 = low ? os2 : s2;
          if (na == 0) { n = nb2; mn = bmn; mx = bmx; x0 = bx0; s1 = bs1; s2 = bs2; }
          else if (nb2 == 0) { n = na; mn = amn; mx = amx; x0 = ax0; s1 = as1; s2 = as2; }
          else {
            const double sh = bx0 - ax0, dnb = (double)nb2;                        // b's values are (its d) + sh relative to a's shift
            n = na + nb2; mn = fmin(amn, bmn); mx = fmax(amx, bmx); x0 = ax0;
            s1 = as1 + (bs1 + dnb * sh);
            s2 = as2 + (bs2 + 2.0 * sh * bs1 + dnb * (sh * sh));
          }
        }
        if (part != 0 || kk >= KT) continue;
        if (!live) { settled[kk] = 1; continue; }
        // a key with a value on the overflow list is incomplete in the tile: its column is written, the fold completes it, the scan redoes it
        const bool key_ovf = sa.ovf_keys != nullptr && ((s_ovfw[(uint32_t)((k - ovf_k0) >> 5)] >> (k & 31u)) & 1u) != 0;
        if (!skip_cols || key_ovf) { sa.st.n_pts[k] = kSettleRedo; settled[kk] = 0; tile_redo[atomicAdd(&s_nredo, 1u)] = (uint32_t)k; continue; }
        const bool slow = n > 0 && (!(mx - mn <= sa.eps) || n < (uint32_t)sa.min_samples);
        sa.st.n_pts[k] = n;
        sa.st.n_anom[k] = 0;
        const double dn = (double)(n ? n : 1);
        sa.st.key_mean[k] = n ? x0 + s1 / dn : 0.0;
        sa.st.key_m2[k] = n ? fmax(s2 - s1 * (s1 / dn), 0.0) : 0.0;
        settled[kk] = slow ? 0 : 1;
        if (slow) tile_list[atomicAdd(&s_nlist, 1u)] = (uint32_t)k;   // ~2 % of the keys: a few per tile (LDS)
      }
      __syncthreads();
      if (threadIdx.x == 0 && s_nlist) s_lbase = atomicAdd(sa.count, s_nlist);   // one global reservation per tile
      if (threadIdx.x == 64 && s_nredo) s_rbase = atomicAdd(sa.redo_count, s_nredo);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < s_nlist; i += kPartThreads) sa.list[s_lbase + i] = tile_list[i];
      for (uint32_t i = threadIdx.x; i < s_nredo; i += kPartThreads) sa.redo_list[s_rbase + i] = tile_redo[i];
      if (sa.rs_val != nullptr && s_nredo) {   // the redo keys' series, contiguous per redo entry (whole series only: not for a split partition)
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        for (uint32_t i = wave; i < s_nredo; i += kPartThreads / 64) {
          const uint32_t e = s_rbase + i;
          if (e >= sa.rs_cap) continue;
          if (skip_cols) {
            const uint32_t kk = tile_redo[i] - (uint32_t)(k0 + kt0);
            for (uint32_t b = lane; b < nb; b += 64) {
              const uint32_t c = __umul24(b, KT) + kk;
              const bool pr = cell_present(c);
              sa.rs_val[(size_t)e * nb + b] = pr ? cell_value(c) : 0ull;
              // bit 1: the cell's aggregate is on the overflow list (NARROW: the all-ones word) -> read it from the grid after the fold
              sa.rs_flag[(size_t)e * nb + b] = pr ? (uint8_t)(FLAG_PRESENT | ((NARROW && vals32[c] == 0xFFFFFFFFu) ? 2 : 0)) : 0;
            }
          }
          if (lane == 0) sa.rs_has[e] = skip_cols ? (NARROW ? 1 : 2) : 0;   // 2: values beyond the record range are absent from the tile altogether
        }
      }
      if (sa.cs_val != nullptr) {   // the listed keys' series, contiguous per list entry: a wavefront per key, lanes over the buckets
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        for (uint32_t i = wave; i < s_nlist; i += kPartThreads / 64) {
          const uint32_t e = s_lbase + i;
          if (e >= sa.cs_cap) continue;
          const uint32_t kk = tile_list[i] - (uint32_t)(k0 + kt0);
          for (uint32_t b = lane; b < nb; b += 64) {
            const uint32_t c = __umul24(b, KT) + kk;
            const bool pr = cell_present(c);
            sa.cs_val[(size_t)e * nb + b] = pr ? cell_value(c) : 0ull;
            sa.cs_flag[(size_t)e * nb + b] = pr ? FLAG_PRESENT : 0;
          }
          if (lane == 0) sa.cs_has[e] = 1;
        }
      }
    }
    if (SETTLE) {   // 256 lanes per bucket row (every 256th key of the tile per lane), four bucket rows per trip: no division by KT
      for (uint32_t kk = threadIdx.x & 255u; kk < KT; kk += 256u) {
        const uint64_t k = k0 + kt0 + kk;
        if (kt0 + kk < KP && k < g.K && !(skip_cols && settled[kk])) {
          for (uint32_t b = threadIdx.x >> 8; b < nb; b += kPartThreads / 256) {
            const uint32_t c = __umul24(b, KT) + kk;
            const uint64_t gc = (uint64_t)b * g.K + k;
            const bool pr = cell_present(c);
            const unsigned long long cv = pr ? cell_value(c) : 0ull;
            if (!split) {
              g.val[gc] = cv;
              g.flag[gc] = pr ? FLAG_PRESENT : 0;
            } else if (pr) {
              if (OPMAX) __hip_atomic_fetch_max(g.val + gc, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else __hip_atomic_fetch_add(g.val + gc, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              g.flag[gc] = FLAG_PRESENT;
            }
          }
        }
      }
    } else {
      for (uint32_t c = threadIdx.x; c < cells; c += kPartThreads) {  // consecutive lanes -> consecutive keys of one bucket
        const uint32_t b = b_lo + (c >> shift_part), kk = c & (KP - 1);
        const uint64_t k = k0 + kk;
        if (k >= g.K) continue;
        const uint64_t gc = (uint64_t)b * g.K + k;
        if (!split) {
          g.val[gc] = vals[c];
          g.flag[gc] = flags[c];
        } else if (flags[c]) {
          if (OPMAX) __hip_atomic_fetch_max(g.val + gc, vals[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_fetch_add(g.val + gc, vals[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g.flag[gc] = FLAG_PRESENT;
        }
      }
    }
  }
}

template <bool OPMAX, bool SETTLE = false, bool NARROW = false>
__global__ __launch_bounds__(kPartThreads) void k_tile_aggregate(const unsigned long long *__restrict__ recs,
                                                                 const unsigned long long *__restrict__ part_start,
                                                                 SliceTable st, TileGeom tg, Grid g,
                                                                 const uint32_t *__restrict__ offs32, const uint32_t *__restrict__ fin, int G,
                                                                 SettleArgs sa, const unsigned long long *__restrict__ ovf_count) {
  static_assert(!NARROW || (OPMAX && SETTLE), "32-bit tile cells: settle mode with max only");
  tile_aggregate_body<OPMAX, SETTLE, NARROW>(recs, part_start, st, tg, g, offs32, fin, G, sa, ovf_count);
}

// records whose value did not fit the packed form: fold them into the finished grid (agent-scope integer atomics).
// (Round 4 tried to let the LAST workgroup of the tile pass do this — a ticket behind a device-scope fence in every workgroup — to
// save the launch: on this multi-XCD chip an agent-scope release writes the XCD's dirty L2 lines back, and ~1600 workgroups doing
// that with the grid tiles in flight took the C2 job from 1.28 to 3.0 ms (profiles/r4_v5_ab_c2.log).  A launch of its own it stays.)
template <bool OPMAX>
__global__ __launch_bounds__(256) void k_apply_overflow(const OverflowRec *__restrict__ ovf,
                                                        const unsigned long long *__restrict__ ovf_count, uint32_t cap, Grid g) {
  unsigned long long n = *ovf_count;
  if (n > cap) n = cap;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
    const OverflowRec r = ovf[i];
    if (OPMAX) __hip_atomic_fetch_max(g.val + r.gcell, r.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(g.val + r.gcell, r.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    g.flag[r.gcell] = FLAG_PRESENT;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------


bool part_plan_bins(uint64_t n, uint64_t K, bool has2, PartPlan *pl) {
  if (K == 0 || n == 0 || n * (has2 ? 2 : 1) >= (1ull << 32)) return false;
  int s = 0;
  while (((K + (1ull << s) - 1) >> s) > kMaxBins) ++s;
  pl->shift_bin = s;
  pl->nbins = (uint32_t)((K + (1ull << s) - 1) >> s);
  pl->G = 256;  // one workgroup per CU for passes A and B
  uint64_t chunk = (n + pl->G - 1) / pl->G;
  chunk = (chunk + 1) & ~1ull;  // even: 16-byte loads stay aligned in every workgroup
  pl->chunk = chunk;
  return true;
}

bool part_plan_tiles(uint64_t K, uint64_t T, bool has2, PartPlan *pl) {
  if (T == 0 || T >= (1ull << 16)) return false;
  // Key block: as small as keeps the partition count <= kMaxParts (few partitions = long runs in pass B), but not
  // smaller than the largest block whose whole KP x T tile fits one LDS tile (then pass C needs a single round).
  int sp_fit = -1;
  for (int c = 13; c >= 0; --c)
    if (((uint64_t)T << c) <= kTileCells) { sp_fit = c; break; }
  int sp = sp_fit >= pl->shift_bin ? sp_fit : pl->shift_bin;
  while (((K + (1ull << sp) - 1) >> sp) > kMaxParts && sp < 13) ++sp;
  if (((K + (1ull << sp) - 1) >> sp) > kMaxParts) return false;
  while (sp > pl->shift_bin && (1ull << (sp - 1)) >= K) --sp;  // no wider than the key space
  // Whole 128-byte lines from the write-combining pass B need >= 18 queue slots per partition in LDS (part_plan_wc):
  // widen the key block until the partitions are few enough, if pass C then needs at most twice the bucket rounds
  // (they run as parallel workgroups sharing an XCD's L2, k_tile_aggregate) and no more than 4.  C2: 1563 partitions of
  // 64 keys -> 782 of 128 keys, 2 rounds: partition pass 0.715 -> 0.59 ms, pass C 0.26 -> 0.29 ms.
  {
    auto parts_of = [&](int c) { return (K + (1ull << c) - 1) >> c; };
    auto rounds_of = [&](int c) { const uint64_t tb = kTileCells >> c; return tb ? (T + tb - 1) / tb : (uint64_t)1 << 30; };
    const uint64_t line_parts = kLdsBudget / (8 * 18 + 18);   // (18 slots: part_plan_wc)
    if (parts_of(sp) > line_parts) {
      int c = sp;
      while (c < 13 && parts_of(c) > line_parts) ++c;
      if (parts_of(c) <= line_parts && (1ull << c) <= kTileCells && rounds_of(c) <= 2 * rounds_of(sp) && rounds_of(c) <= 4) sp = c;
    }
  }
  // Too many partitions even for 64-byte sectors (the queues need >= 9 slots each: ~1770 partitions): one more doubling of the
  // key block puts pass B on the write-combining kernel if pass C then needs at most 8 bucket rounds.  C4 (1e6 keys x 100
  // buckets): 1954 partitions of 512 keys, sort-by-tile pass B 1.00-1.09 ms + pass C (3 rounds) 0.50 ms -> 977 of 1024 keys,
  // write-combining pass B 0.78-0.81 ms + pass C (6 rounds) 0.69 ms: 2.14-2.29 -> 2.09-2.18 ms per job, same boxes
  // (profiles/r3_v2_c4_keyblock_ab.log; 2048-key blocks: pass B 0.65 ms but 13 rounds, pass C 1.8 ms).
  {
    auto parts_of = [&](int c) { return (K + (1ull << c) - 1) >> c; };
    auto rounds_of = [&](int c) { const uint64_t tb = kTileCells >> c; return tb ? (T + tb - 1) / tb : (uint64_t)1 << 30; };
    const uint64_t sector_parts = kLdsBudget / (8 * 9 + 18);
    if (parts_of(sp) > sector_parts && sp < 13 && parts_of(sp + 1) <= sector_parts && (1ull << (sp + 1)) <= kTileCells && rounds_of(sp + 1) <= 8) ++sp;
  }
  if ((1ull << sp) > kTileCells) return false;                 // one bucket of the block must fit a tile
  pl->shift_part = sp;
  pl->KP = 1u << sp;
  pl->nparts = (uint32_t)((K + pl->KP - 1) >> sp);
  pl->bins_per_part = 1u << (sp - pl->shift_bin);
  const uint64_t cells_all = (uint64_t)pl->KP * T;
  int cb = kMinCellBits;
  while (cb < kMaxCellBits && (1ull << cb) - 1 <= cells_all) ++cb;   // all-ones is reserved
  if ((1ull << cb) - 1 <= cells_all) return false;
  pl->cell_bits = cb;
  pl->tb = (uint32_t)(kTileCells >> sp);
  if (pl->tb > T) pl->tb = (uint32_t)T;
  pl->n_chunks = (uint32_t)((T + pl->tb - 1) / pl->tb);
  pl->tb = (uint32_t)((T + pl->n_chunks - 1) / pl->n_chunks);   // same number of rounds, balanced (100 buckets: 34 + 33 + 33, not 34 + 34 + 32 ... + 1)
  pl->agg_lds = ((size_t)pl->tb * pl->KP * 9 + 15) & ~(size_t)15;
  // pass B: records per tile limited by LDS: 10 B per slot + 16 B per partition
  const size_t fixed = ((size_t)pl->nparts + 4) * 16 + 64;
  pl->rpt = 0;
  const int mult = has2 ? 2 : 1;
  for (int r : {10, 8, 4, 2}) {   // (12 rows per thread spill registers at 1024 threads: measured slower)
    const size_t slots = (size_t)r * kPartThreads * mult;
    if (slots * 10 + fixed <= kLdsBudget) { pl->rpt = r; pl->part_lds = (slots * 10 + fixed + 15) & ~(size_t)15; break; }
  }
  return pl->rpt != 0;
}

// Sparse tables (second-resolution lattices, per-connection keys: tad_sparse.hip) use passes A and B only: the records of a key block are
// sorted by (key, bucket) in LDS afterwards, no tile has to hold the block's buckets.  The key block is the smallest that leaves the
// write-combining pass its queues (64-byte sectors: <= ~1770 partitions) — the sort streams a partition once per LDS round, so small
// partitions are cheap ones.  Record = value << cell_bits | (bucket * KP + key-in-block) with cell_bits = log2(KP) + bit_width(T): the sort's key
// (key-in-block << bit_width(T) | bucket) fits the same bits.  At most 28 cell bits (>= 36 value bits: wider values raise
// DEV_ERR_OVERFLOW_LIST and the job is redone by the LSD sort, like every shape this plan refuses).
bool part_plan_sparse(uint64_t K, uint64_t T, bool has2, PartPlan *pl) {
  (void)has2;
  if (T == 0 || T >= (1ull << 32) || K == 0) return false;
  int tbits = 0;
  while ((T >> tbits) != 0) ++tbits;
  auto parts_of = [&](int c) { return (K + (1ull << c) - 1) >> c; };
  const uint64_t sector_parts = kLdsBudget / (8 * 9 + 18);
  int sp = pl->shift_bin;
  while (sp < 13 && parts_of(sp) > sector_parts) ++sp;
  if (parts_of(sp) > sector_parts) return false;
  int cb = sp + tbits;
  if (cb > 28) return false;
  if (cb < kMinCellBits) cb = kMinCellBits;
  pl->shift_part = sp;
  pl->KP = 1u << sp;
  pl->nparts = (uint32_t)parts_of(sp);
  pl->bins_per_part = 1u << (sp - pl->shift_bin);
  pl->cell_bits = cb;
  pl->sp_tbits = tbits;
  pl->tb = 0; pl->n_chunks = 0; pl->settle_kt = 0; pl->narrow = false; pl->agg_lds = 0;
  pl->rpt = 0; pl->part_lds = 0;     // (the sort-by-tile pass B is not used: without the write-combining queues the LSD sort runs)
  return true;
}

// Settle mode of pass C (DBSCAN jobs): kt = the most keys whose WHOLE series fit one LDS tile (9 B per cell + 5 B per key of
// settle bookkeeping); the partition's rounds then split it by key sub-range.  Adopted when that does not take more rounds than the
// bucket rounds it replaces (every round streams the partition's records again).
bool part_plan_settle(uint64_t T, PartPlan *pl, bool narrow) {
  pl->settle_kt = 0;
  pl->narrow = false;
  if (T == 0) return false;
  const size_t cell = narrow ? 4 : 9;     // narrow: one 32-bit word per cell (value + 1), `max` only
  uint32_t kt = (uint32_t)((kLdsBudget - 64) / (T * cell + 9));   // + settled u8, work-list u32, redo-list u32 per key
  if (kt > pl->KP) kt = pl->KP;
  if (kt > 1024) kt = 1024;
  if (kt < 8) return false;
  const uint32_t rounds = (pl->KP + kt - 1) / kt;
  if (rounds > pl->n_chunks + 1 && rounds > 1) return false;
  kt = (pl->KP + rounds - 1) / rounds;             // balanced rounds
  pl->settle_kt = kt;
  pl->n_chunks = rounds;
  pl->tb = (uint32_t)T;
  pl->agg_lds = (((size_t)T * kt * cell + 3) & ~(size_t)3) + (((size_t)kt + 3) & ~(size_t)3) + (size_t)kt * 8 + 16;
  pl->agg_lds = (pl->agg_lds + 15) & ~(size_t)15;
  pl->narrow = narrow;
  return true;
}

static bool aligned16(const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool columns_aligned16(const void *key, const void *key2, const void *t_end, const void *value) {
  return aligned16(key) && aligned16(key2) && aligned16(t_end) && aligned16(value);
}

// write-combining pass B: F * (8 * cap + 18) bytes of LDS.  With >= 18 slots per partition the queues emit whole
// 128-byte lines (16 records), with 9..17 slots 64-byte sectors; below 9 slots (more than ~1800 partitions) and for
// tables with so few partitions that the sort-by-tile pass already writes long runs, the old pass runs.
// partition_pass (tad_plan): 1 = always the sort-by-tile pass, 2 = the write-combining pass whenever its queues fit LDS.
void part_plan_wc(uint64_t slots, bool aligned, bool has2, int partition_pass, PartPlan *pl) {
  pl->wc_cap = 0;
  pl->wc_sec = 0;
  pl->wc_rpt = 0;
  pl->pad_slots = 0;
  if (partition_pass == 1) return;
  if (!aligned || pl->nparts == 0) return;
  const size_t per = (kLdsBudget - 16) / pl->nparts;
  if (per < kWcFixedBytes + 8 * 9) return;
  uint32_t cap = (uint32_t)((per - kWcFixedBytes) / 8);
  if (cap > 64) cap = 64;
  // Whole 128-byte lines: 15 leftovers + room for a tile's arrivals.  Round 3 asked for 22 slots (at C4's 977 partitions = 18 slots lines measured
  // 0.99 against 0.80 ms for 64-byte sectors); that loss was the global store of the ~1 % spilled records between the LDS appends draining the
  // prefetch, not the spills themselves: with the spills parked in LDS (round 4) lines win from 18 slots on — C4 pass B 0.797 -> 0.707 ms, job
  // 1.867 -> 1.772 ms, same box alternating (profiles/r4_v10_ab_c4_lines18.log).  partition_pass 3 keeps the 64-byte sectors for the A/B.
  uint32_t sec = (cap >= 18 && partition_pass != 3) ? 16 : 8;
  if (sec == 8 && cap > 16) cap = 16;
  // the sort-by-tile pass writes runs of (tile slots / partitions) records: long runs beat 64-byte sectors
  const bool forced = partition_pass >= 2;
  if (!forced && sec == 8 && (uint64_t)pl->rpt * kPartThreads * (has2 ? 2 : 1) / pl->nparts >= 24) return;
  // few, large partitions: a tile brings more records per partition than a queue can take (and the sort pass writes long runs)
  // (the mean arrivals of a tile must fit the room behind SEC - 1 leftovers; what exceeds it now and then is parked in LDS and stored in
  // the emit phase — until round 4 the rule asked for twice the room, because a spill was a global store between the LDS appends)
  if (!forced && 2.0 * kPartThreads * (has2 ? 2 : 1) / (double)pl->nparts > (double)(cap - (sec - 1))) return;
  // rows per thread: 4 when a queue holding SEC - 1 leftovers still has room for twice the expected arrivals of a tile
  const double lam4 = 4.0 * kPartThreads * (has2 ? 2 : 1) / (double)pl->nparts;
  const uint32_t rpt = !has2 && (double)(cap - (sec - 1)) >= 2.0 * lam4 + 4.0 ? 4 : 2;
  const uint64_t pad = (uint64_t)(sec - 1) * pl->G * pl->nparts;
  if (slots + pad >= (1ull << 32)) return;
  pl->wc_cap = cap;
  pl->wc_sec = sec;
  pl->wc_rpt = rpt;
  pl->pad_slots = pad;
}

bool launch_meta_hist(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                      const int64_t *t_start, uint64_t n, uint64_t K, RowFilter f, const PartPlan &pl,
                      MetaPartial *partials, uint32_t *binhist, DevCounters *ctr, bool sample_hist) {
  const bool vec = aligned16(key) && aligned16(key2) && aligned16(t_end);
  const bool has2 = key2 != nullptr;
  // the histogram is sampled with 16-byte loads only; a time-window filter (anomaly_detection.py:581-586) is applied to the sampled rows
  const bool sh = sample_hist && vec && (t_start == nullptr || f.start_time == 0 || aligned16(t_start));
#define TAD_MH(V, H2, SH)                                                                                              \
  do {                                                                                                                 \
    allow_big_lds(reinterpret_cast<const void *>(k_meta_hist<V, H2, SH>), kLdsBudget);                                 \
    hipLaunchKernelGGL((k_meta_hist<V, H2, SH>), dim3(pl.G), dim3(kPartThreads), (size_t)pl.nbins * 4, s, key, key2, t_end, t_start, n, \
                       pl.chunk, K, f, pl.shift_bin, pl.nbins, partials, binhist, ctr);                                \
  } while (0)
  if (vec) {
    if (sh) { if (has2) TAD_MH(true, true, true); else TAD_MH(true, false, true); }
    else { if (has2) TAD_MH(true, true, false); else TAD_MH(true, false, false); }
  } else { if (has2) TAD_MH(false, true, false); else TAD_MH(false, false, false); }
#undef TAD_MH
  return sh;
}

// 6 sigma of every region summed by Cauchy-Schwarz over R = G * nparts regions whose estimates total <= slots
uint64_t sampled_slots_bound(uint64_t slots, const PartPlan &pl) {
  const double R = (double)pl.G * (double)pl.nparts, scale = (double)(kSampleMask + 2);   // (the sampling interval, and one to spare)
  static_assert(kCapSigmas <= 6.0, "the bound below sums 6 sigma per region");
  return slots + (uint64_t)(6.0 * sqrt(scale * R * (double)slots) + R * (kCapIntervals * scale + 16.0)) + 1024;   // sampled_capacity's constant + the rounding to 16, per region
}

// slice table inside slice_mem: slice_part[max_slices] | slice_first[nparts] | n_slices
static SliceTable slice_table(void *slice_mem, uint64_t slots, const PartPlan &pl) {
  const uint32_t max_slices = (uint32_t)((size_t)pl.nparts + (size_t)(slots / kSliceRecords) + 1);
  SliceTable st;
  st.slice_part = static_cast<uint32_t *>(slice_mem);
  st.slice_first = st.slice_part + max_slices;
  st.n_slices = st.slice_first + pl.nparts;
  return st;
}
// regions sized from a sampled histogram hold ~2.4x the slots of their records: slices three times as long keep one slice per
// partition for uniform tables (a split partition merges into the grid with atomics instead of plain stores)
// Round 5: ... and never shorter than 1.5x the MEAN partition.  Slices exist to spread a partition swollen by a hot key over many CUs; when every
// partition is long (C5 on one GPU: 1e9 rows / 1954 partitions = 512k records each) cutting all of them made every tile a pre-zeroed one that its four
// slices merged into with global atomics (k_part_offsets wrote the 2.25 GB grid once to zero it, pass C 5.6 GB for 2.25 GB of tiles, `profiles/r5_m1_pmc_c5.json`)
// although the partitions x rounds alone are 15 632 workgroups.  A partition well above the mean is still split.
static uint32_t slice_len_of(bool sampled, uint64_t slots, uint32_t nparts) {
  const uint64_t base = sampled ? 3ull * kSliceRecords : kSliceRecords;
  const uint64_t mean = nparts ? slots / nparts : 0;
  const uint64_t want = mean + mean / 2;
  const uint64_t len = want > base ? want : base;
  return (uint32_t)(len < 0xFFFFFFF0ull ? len : 0xFFFFFFF0ull);
}

void launch_part_offsets(hipStream_t s, const uint32_t *binhist, const PartPlan &pl, uint32_t *offs32, uint32_t *total,
                         unsigned long long *part_start, bool sampled, const MetaPartial *partials, uint64_t n, uint64_t slots, void *slice_mem,
                         Grid g, DevCounters *ctr) {
  static_assert(kMaxParts <= 8 * 256, "k_part_tail holds 8 partitions per thread");
  OffsetsArgs A;
  A.binhist = binhist; A.nbins = pl.nbins; A.bins_per_part = pl.bins_per_part; A.nparts = pl.nparts;
  // sampled regions are rounded to 16 records (128-byte lines) whatever pass B runs, so that they start line-aligned
  A.round_mask = sampled ? 15u : (pl.wc_cap ? pl.wc_sec - 1u : 0u);
  A.G = pl.G; A.partials = sampled ? partials : nullptr; A.n = n; A.chunk = pl.chunk;
  A.offs32 = offs32; A.total = total; A.part_start = part_start;
  A.st = slice_table(slice_mem, slots, pl);
  A.slice_len = g.val == nullptr ? 0xFFFFFFFFu : slice_len_of(sampled, slots, pl.nparts);   // (no grid: the sparse sort reads whole partitions, nothing is split or pre-zeroed)
  A.g = g; A.shift_part = pl.shift_part; A.ctr = sampled ? ctr : nullptr;
  hipLaunchKernelGGL(k_part_offsets, dim3((pl.nparts + 3) / 4), dim3(256), 0, s, A);
  hipLaunchKernelGGL(k_part_tail, dim3(1), dim3(256), 0, s, A);
}

void launch_partition(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                      const int64_t *t_start, const uint64_t *value, uint64_t n, uint64_t K, RowFilter f, Lattice L,
                      const PartPlan &pl, const uint32_t *offs32, const unsigned long long *part_start, void *recs,
                      OverflowRec *ovf, unsigned long long *ovf_count, uint32_t ovf_cap, DevCounters *ctr, uint32_t *fin, uint32_t *ovf_keys) {
  PartArgs A;
  A.fin = fin; A.G = pl.G;
  A.value_limit = 1ull << (64 - pl.cell_bits);
  A.narrow_limit = pl.narrow ? 0xFFFFFFFEull : 0ull;     // (value + 1 of every in-range value stays below the all-ones word)
  A.ovf_keys = ovf_keys;
  A.key = key; A.key2 = key2; A.t_end = t_end; A.t_start = t_start; A.value = value;
  A.n = n; A.chunk = pl.chunk; A.K = K; A.f = f; A.L = L;
  A.shift_part = pl.shift_part; A.kp_mask = pl.KP - 1; A.nparts = pl.nparts; A.cell_bits = pl.cell_bits;
  A.offs32 = offs32; A.part_start = part_start;
  A.recs = static_cast<unsigned long long *>(recs); A.ctr = ctr;
  A.ovf = ovf; A.ovf_count = ovf_count; A.ovf_cap = ovf_cap;
  const bool has2 = key2 != nullptr;
  const bool vec = aligned16(key) && aligned16(key2) && aligned16(t_end) && aligned16(value);
  // fast path: 16-byte loads, no time-window filter, bucket by one multiply-high
  const bool generic = !vec || L.mode == 2 || f.end_time != 0 || (f.start_time != 0 && t_start != nullptr);
  if (pl.wc_cap) {  // write-combining variant (the plan checked the alignment)
    const size_t wlds = ((size_t)pl.nparts * (8 * (size_t)pl.wc_cap + kWcFixedBytes) + 4 + 15) & ~(size_t)15;
    const bool ts16 = generic && t_start != nullptr && f.start_time != 0 && aligned16(t_start);   // start times prefetched with the tile
#define TAD_WC1(RPT, SEC, H2, GEN, TS)                                                                                  \
  do {                                                                                                                \
    allow_big_lds(reinterpret_cast<const void *>(k_partition_wc<RPT, SEC, H2, GEN, TS>), kLdsBudget);                  \
    hipLaunchKernelGGL((k_partition_wc<RPT, SEC, H2, GEN, TS>), dim3(pl.G), dim3(kPartThreads), wlds, s, A, pl.wc_cap, pl.G); \
  } while (0)
#define TAD_WC(RPT, SEC, H2, GEN) do { if (GEN && ts16) TAD_WC1(RPT, SEC, H2, GEN, GEN); else TAD_WC1(RPT, SEC, H2, GEN, false); } while (0)
#define TAD_WC_SEC(RPT, H2, GEN) do { if (pl.wc_sec == 16) TAD_WC(RPT, 16, H2, GEN); else TAD_WC(RPT, 8, H2, GEN); } while (0)
    if (pl.wc_rpt == 4 && !has2) { if (generic) TAD_WC_SEC(4, false, true); else TAD_WC_SEC(4, false, false); }
    else if (has2) { if (generic) TAD_WC_SEC(2, true, true); else TAD_WC_SEC(2, true, false); }
    else { if (generic) TAD_WC_SEC(2, false, true); else TAD_WC_SEC(2, false, false); }
#undef TAD_WC_SEC
#undef TAD_WC
#undef TAD_WC1
    return;
  }
  int rpt = pl.rpt;
  if (generic && rpt > 4) rpt = 4;
  const size_t fixed = ((size_t)pl.nparts + 4) * 16 + 64;
  const size_t lds = ((size_t)rpt * kPartThreads * (has2 ? 2 : 1) * 10 + fixed + 15) & ~(size_t)15;
#define TAD_PART(RPT, H2, V, GEN)                                                                                       \
  do {                                                                                                                \
    allow_big_lds(reinterpret_cast<const void *>(k_partition<RPT, H2, V, GEN>), kLdsBudget);                           \
    hipLaunchKernelGGL((k_partition<RPT, H2, V, GEN>), dim3(pl.G), dim3(kPartThreads), lds, s, A);                       \
  } while (0)
  if (!generic) {
    switch (rpt) {
      case 10: if (has2) TAD_PART(8, true, true, false); else TAD_PART(10, false, true, false); break;
      case 8: if (has2) TAD_PART(8, true, true, false); else TAD_PART(8, false, true, false); break;
      case 6: case 4: if (has2) TAD_PART(4, true, true, false); else TAD_PART(4, false, true, false); break;
      default: if (has2) TAD_PART(2, true, true, false); else TAD_PART(2, false, true, false); break;
    }
  } else if (vec) {
    if (rpt >= 4) { if (has2) TAD_PART(4, true, true, true); else TAD_PART(4, false, true, true); }
    else { if (has2) TAD_PART(2, true, true, true); else TAD_PART(2, false, true, true); }
  } else {
    if (rpt >= 4) { if (has2) TAD_PART(4, true, false, true); else TAD_PART(4, false, false, true); }
    else { if (has2) TAD_PART(2, true, false, true); else TAD_PART(2, false, false, true); }
  }
#undef TAD_PART
}

size_t slice_table_bytes(uint64_t slots, const PartPlan &pl) {
  const size_t max_slices = (size_t)pl.nparts + (size_t)(slots / kSliceRecords) + 1;
  return (max_slices + pl.nparts + 4) * sizeof(uint32_t);
}

void launch_tile_aggregate(hipStream_t s, const void *recs, const unsigned long long *part_start, const PartPlan &pl,
                           uint64_t slots, void *slice_mem, Grid g, bool op_max, const OverflowRec *ovf,
                           const unsigned long long *ovf_count, uint32_t ovf_cap, const uint32_t *offs32, const uint32_t *fin, SettleArgs settle) {
  const unsigned long long *rr = static_cast<const unsigned long long *>(recs);
  const uint32_t max_slices = (uint32_t)((size_t)pl.nparts + (size_t)(slots / kSliceRecords) + 1);
  const SliceTable st = slice_table(slice_mem, slots, pl);   // built by k_part_offsets (with the pre-zeroed tiles of split partitions)
  const uint32_t slice_len = slice_len_of(fin != nullptr, slots, pl.nparts);
  // the rounds of a partition run as parallel workgroups on one XCD (measured: C2 pass C 0.39 -> 0.29 ms against sequential rounds)
  const uint32_t par = pl.n_chunks > 1 ? 1u : 0u;
  const bool settle_on = settle.on != 0 && pl.settle_kt != 0;
  TileGeom tg{pl.shift_part, pl.cell_bits, pl.nparts, pl.tb, pl.n_chunks, par, slice_len, settle_on ? pl.settle_kt : 0u};
  const uint32_t blocks1 = par ? ((max_slices + 7u) / 8u) * 8u * pl.n_chunks : max_slices;
  const SettleArgs none{};
#define TAD_TA(OPMAX, SET)                                                                                                                            \
  do {                                                                                                                                               \
    allow_big_lds(reinterpret_cast<const void *>(k_tile_aggregate<OPMAX, SET>), kLdsBudget);                                                         \
    hipLaunchKernelGGL((k_tile_aggregate<OPMAX, SET>), dim3(blocks1), dim3(kPartThreads), pl.agg_lds, s, rr, part_start, st, tg, g, offs32, fin, pl.G, \
                       SET ? settle : none, ovf_count);                                                                                               \
    hipLaunchKernelGGL((k_apply_overflow<OPMAX>), dim3(64), dim3(256), 0, s, ovf, ovf_count, ovf_cap, g);                                             \
  } while (0)
  if (settle_on && pl.narrow && op_max) {
    allow_big_lds(reinterpret_cast<const void *>(k_tile_aggregate<true, true, true>), kLdsBudget);
    hipLaunchKernelGGL((k_tile_aggregate<true, true, true>), dim3(blocks1), dim3(kPartThreads), pl.agg_lds, s, rr, part_start, st, tg, g, offs32, fin, pl.G, settle, ovf_count);
    hipLaunchKernelGGL((k_apply_overflow<true>), dim3(64), dim3(256), 0, s, ovf, ovf_count, ovf_cap, g);
  } else if (settle_on) { if (op_max) TAD_TA(true, true); else TAD_TA(false, true); }
  else { if (op_max) TAD_TA(true, false); else TAD_TA(false, false); }
#undef TAD_TA
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_stage0_part() { return reinterpret_cast<const void *>(&k_part_tail); }

}  // namespace tad
