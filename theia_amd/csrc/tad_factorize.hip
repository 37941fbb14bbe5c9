// tad_factorize.hip — the GROUP BY keys of the job, factorised on the GPU (ingest).
//
// The reference leaves the grouping to ClickHouse (anomaly_detection.py:507-614: GROUP BY over string / integer columns); the engine
// wants dense key ids.  theia_amd/anomaly_detection.py:prepare_columns evaluates the SQL's string predicates on the DISTINCT values
// of each string column and then has to turn the rows' key TUPLES — dictionary codes of the string columns, ports, protocol,
// flowStartSeconds: up to eight 8-byte integers per row — into ids.  pandas does that at 3e6-1.5e7 rows/s on one core in front of an
// engine that aggregates 7e10 rows/s.  Here: an open-addressing hash table in HBM, one 8-byte word per slot
//     word = fingerprint (32 bits of the tuple's hash) << 32 | virtual row of a row that holds the tuple     (all ones = empty)
// A row claims an empty slot with ONE compare-and-swap (fingerprint and representative row appear together: no reader ever sees a
// half-written slot, nobody spins); a row that finds its fingerprint compares its tuple with the representative row's tuple in the
// input columns (exact: a fingerprint match alone is not equality) and, if it comes EARLIER in the table, lowers the slot's row
// with an atomic min — the fingerprint sits in the high half, so the minimum is taken among rows of this very tuple.  After the
// pass every slot names the first row of its tuple.  Ids in order of first appearance (what pandas.factorize gives, so that the
// GPU path and the pandas path produce identical key ids and key tables): a bitmap of the first rows, a scan of its popcounts,
// id(first row r) = first rows before r.  The insert pass leaves every row's slot behind (4 bytes); after the scan a pass over the SLOTS turns every
// claimed word into its id, and the pass over the rows is slot -> id.
// Pod mode (the UNION ALL of the inbound and the outbound view, :556-565) passes two tuples per row: the table runs over the
// virtual rows [side a: 0 .. n) ++ [side b: n .. 2n), the side is part of the tuple.
#include "tad_internal.h"

namespace tad {

static constexpr int kFzBlock = 256;
static constexpr unsigned long long kFzEmpty = ~0ull;

__device__ __forceinline__ uint64_t fz_mix(uint64_t x) {   // splitmix64 finaliser
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// A probe LOOKS at a slot with a plain cached load.  On this multi-XCD chip an agent-scope load is a transaction with the memory side for every
// row (the L2s of the eight XCDs are not coherent with each other, so device-scope accesses bypass them): 1e8 of them were the insert pass
// (3 of its 4 ms for one key column, profiles/r4_v29_*).  A stale view is harmless here: a slot's fingerprint never changes once claimed, so a
// non-empty word is trusted for WHICH slot it is (its row half may be stale-high: the atomic min below is then merely redundant), and a word
// that looks empty is only ever claimed with a compare-and-swap, which is performed at the memory side and returns the truth.
__device__ __forceinline__ unsigned long long fz_peek(const unsigned long long *slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct FzArgs {
  const long long *a[kFzMaxCols];
  const long long *b[kFzMaxCols];   // all NULL: one tuple per row
  const uint8_t *keep_a, *keep_b;   // NULL = every row
  uint64_t n;
  int n_cols;
  uint32_t sides;                   // 1 or 2
};

__device__ __forceinline__ bool fz_kept(const FzArgs &A, uint64_t v) {
  const bool sb = v >= A.n;
  const uint8_t *keep = sb ? A.keep_b : A.keep_a;
  return keep == nullptr || keep[sb ? v - A.n : v] != 0;
}
__device__ __forceinline__ long long fz_value(const FzArgs &A, uint64_t v, int c) {
  const bool sb = v >= A.n;
  return (sb ? A.b[c] : A.a[c])[sb ? v - A.n : v];
}
__device__ __forceinline__ uint64_t fz_hash(const FzArgs &A, uint64_t v, long long (&t)[kFzMaxCols]) {
  uint64_t h = v >= A.n ? 0x9E3779B97F4A7C15ull : 0ull;   // the side is part of the tuple
  for (int c = 0; c < A.n_cols; ++c) {
    t[c] = fz_value(A, v, c);
    h = fz_mix(h ^ (uint64_t)t[c]) + 0x632BE59BD9B4E019ull * (uint64_t)(c + 1);
  }
  return fz_mix(h);
}
__device__ __forceinline__ bool fz_same(const FzArgs &A, uint64_t v, const long long (&t)[kFzMaxCols], uint64_t rep) {
  if ((v >= A.n) != (rep >= A.n)) return false;
  // every column of the representative row is loaded before any is compared: with an early exit per column the compare was one memory round
  // trip per key column (rows that match — all but the first of a key — never take the exit anyway)
  bool same = true;
#pragma unroll
  for (int c = 0; c < kFzMaxCols; ++c)
    if (c < A.n_cols) same = same & (fz_value(A, rep, c) == t[c]);
  return same;
}

static constexpr uint32_t kFzMaxProbe = 32;
enum : uint32_t { FZ_FLAG_GROW = 1u, FZ_FLAG_BAD_INPUT = 2u };

// The table starts SMALL (round 4, factorize_first_slots: 2^20 slots = 8 MB, then 2^24, then 2 n): most jobs have far fewer keys than rows, a
// table that fits the L2 / MALL turns every probe from an HBM transaction into a cache hit, and clearing + scanning 2 n slots (2 GB each at 1e8
// rows) was a third of the old pass.  A pass that finds its table filling up (a probe sequence > 32 slots, or more than half the slots claimed —
// counted with one atomic per wavefront) raises a flag; every thread leaves at its next row, the later kernels return at once, and the host —
// which reads the key count at the end anyway — repeats the call with the next size.
// every kept virtual row into the table; on return a slot's low half = the smallest virtual row holding its tuple, slot_of[v] = v's slot
// max_probe: kFzMaxProbe on the small tables; unlimited on the full-size one (2 n slots: it cannot fill up, and with tens of millions of distinct
// keys a linear-probing cluster longer than 32 DOES occur — 5e7 keys at load 0.37: ~400 expected — which round 6's 5e7-connection ingest met)
__global__ __launch_bounds__(kFzBlock) void k_fz_insert(FzArgs A, unsigned long long *__restrict__ table, uint64_t mask, uint32_t *__restrict__ slot_of,
                                                        uint32_t *__restrict__ flags, unsigned long long *__restrict__ claims, uint32_t max_probe) {
  const uint64_t V = A.n * A.sides;
  uint32_t claimed = 0;
  uint32_t round = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v < V; v += (uint64_t)gridDim.x * kFzBlock, ++round) {
    // the table is too small: the pass is being abandoned (asked at the memory side, so only every eighth row of a thread)
    if ((round & 7u) == 0u && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (!fz_kept(A, v)) continue;
    long long t[kFzMaxCols];
    const uint64_t h = fz_hash(A, v, t);
    const unsigned long long mine = ((h >> 32) << 32) | v;     // (v < 2^32 - 1: never the empty word)
    uint32_t probes = 0;
    for (uint64_t s = h & mask;; s = (s + 1) & mask) {
      unsigned long long w = fz_peek(table + s);
      if (w == kFzEmpty) {
        w = atomicCAS(table + s, kFzEmpty, mine);
        if (w == kFzEmpty) { ++claimed; slot_of[v] = (uint32_t)s; break; }      // claimed
      }
      if ((w >> 32) == (mine >> 32) && fz_same(A, v, t, w & 0xffffffffull)) {
        if (mine < w) atomicMin(table + s, mine);                // an earlier row of the same tuple (same high half: the min stays in the class)
        slot_of[v] = (uint32_t)s;
        break;
      }
      if (++probes > max_probe) { atomicOr(flags, FZ_FLAG_GROW); break; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) claimed += __shfl_down(claimed, o);     // one atomic per wavefront for the claim count
  if ((threadIdx.x & 63) == 0 && claimed) {
    const unsigned long long before = atomicAdd(claims, (unsigned long long)claimed);
    if (max_probe != 0xFFFFFFFFu && 2 * (before + claimed) > mask + 1) atomicOr(flags, FZ_FLAG_GROW);
  }
}

// bitmap of the first rows (bits zeroed by the caller)
__global__ __launch_bounds__(kFzBlock) void k_fz_mark(const unsigned long long *__restrict__ table, uint64_t slots, uint32_t *__restrict__ bits) {
  for (uint64_t s = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * kFzBlock) {
    const unsigned long long w = table[s];
    if (w != kFzEmpty) atomicOr(bits + ((w & 0xffffffffull) >> 5), 1u << (w & 31ull));
  }
}

__global__ __launch_bounds__(kFzBlock) void k_fz_popc(const uint32_t *__restrict__ bits, uint64_t words, uint32_t *__restrict__ cnt) {
  for (uint64_t i = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; i < words; i += (uint64_t)gridDim.x * kFzBlock) cnt[i] = (uint32_t)__popc(bits[i]);
}

// After the scan every claimed slot learns its id: table[s] = id (the fingerprint has done its work), first_row[id] = the slot's first row.
// A pass over the SLOTS — 2^20 of them in the common case — so that the pass over the rows below is slot -> id, one random read instead of
// three (table word, bitmap word, offset).
__global__ __launch_bounds__(kFzBlock) void k_fz_ids(unsigned long long *__restrict__ table, uint64_t slots, const uint32_t *__restrict__ bits,
                                                      const unsigned long long *__restrict__ off, uint64_t *__restrict__ first_row, uint64_t first_row_cap,
                                                      const uint32_t *__restrict__ flags) {
  if (*flags != 0u) return;     // the insert pass gave up: the host repeats with a larger table
  for (uint64_t s = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * kFzBlock) {
    const unsigned long long w = table[s];
    if (w == kFzEmpty) continue;
    const uint64_t rep = w & 0xffffffffull;
    const uint64_t id = off[rep >> 5] + (uint64_t)__popc(bits[rep >> 5] & ((1u << (rep & 31ull)) - 1u));
    table[s] = id;
    if (id < first_row_cap) first_row[id] = rep;
  }
}

// id of every row (TAD_KEY_SKIP for rows that are not kept): slot -> id
__global__ __launch_bounds__(kFzBlock) void k_fz_lookup(FzArgs A, const unsigned long long *__restrict__ table, const uint32_t *__restrict__ slot_of,
                                                         uint64_t *__restrict__ key_a, uint64_t *__restrict__ key_b, const uint32_t *__restrict__ flags) {
  if (*flags != 0u) return;     // the insert pass gave up: slot_of is incomplete, the host repeats with a larger table
  constexpr int U = 4;          // four rows per thread in flight (slot, then the table word)
  const uint64_t V = A.n * A.sides, stride = (uint64_t)gridDim.x * kFzBlock;
  for (uint64_t v0 = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v0 < V; v0 += U * stride) {
    bool kept[U];
    uint32_t sl[U];
    unsigned long long id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t v = v0 + u * stride;
      kept[u] = v < V && fz_kept(A, v);
      sl[u] = kept[u] ? slot_of[v] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = kept[u] ? table[sl[u]] : (unsigned long long)TAD_KEY_SKIP;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t v = v0 + u * stride;
      if (v >= V) break;
      *(v >= A.n ? key_b + (v - A.n) : key_a + v) = id[u];
    }
  }
}

// k_fz_lookup that also leaves pass A's key-bin histogram behind (tad_factorize_hist, include/tad.h): the ids are in this kernel's registers
// anyway, so Stage 0 of the job that follows need not read the key column a second time to size pass B's regions exactly (C4: 0.19 ms and
// 0.89 GB of its 1.55 ms).  Workgroup g takes the rows [g * chunk, (g + 1) * chunk) of BOTH sides — pass B's own row chunking
// (part_plan_bins) — and counts bin = id >> shift in LDS; shift and nbins follow from the key count, which only exists on the device at this
// point (the same rule as part_plan_bins: the smallest shift with at most kMaxBins bins).
static constexpr int kFzHistThreads = 1024;
__global__ __launch_bounds__(kFzHistThreads) void k_fz_lookup_hist(FzArgs A, const unsigned long long *__restrict__ table, const uint32_t *__restrict__ slot_of,
                                                                  uint64_t *__restrict__ key_a, uint64_t *__restrict__ key_b, const uint32_t *__restrict__ flags,
                                                                  const unsigned long long *__restrict__ num_keys, uint64_t chunk, uint32_t *__restrict__ bins) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fz_hist[];
  uint32_t *fz_hist = reinterpret_cast<uint32_t *>(smem_fz_hist);
  if (*flags != 0u) return;
  const uint64_t K = *num_keys;
  int shift = 0;
  while (((K + (1ull << shift) - 1) >> shift) > kMaxBins) ++shift;
  const uint32_t nbins = (uint32_t)((K + (1ull << shift) - 1) >> shift);
  for (uint32_t i = threadIdx.x; i < nbins; i += kFzHistThreads) fz_hist[i] = 0u;
  __syncthreads();
  constexpr int U = 4;          // four rows per thread in flight (slot, then the table word)
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < A.n ? lo + chunk : A.n;
  for (uint32_t side = 0; side < A.sides; ++side) {
    uint64_t *__restrict__ out = side == 0 ? key_a : key_b;
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (uint64_t)U * kFzHistThreads) {
      bool kept[U];
      uint32_t sl[U];
      unsigned long long id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t i = i0 + (uint64_t)u * kFzHistThreads;
        kept[u] = i < hi && fz_kept(A, i + side * A.n);
        sl[u] = kept[u] ? slot_of[i + side * A.n] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) id[u] = kept[u] ? table[sl[u]] : (unsigned long long)TAD_KEY_SKIP;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t i = i0 + (uint64_t)u * kFzHistThreads;
        if (i >= hi) break;
        out[i] = id[u];
        if (kept[u]) atomicAdd(&fz_hist[(uint32_t)(id[u] >> shift)], 1u);
      }
    }
  }
  __syncthreads();
  uint32_t *dst = bins + (size_t)blockIdx.x * nbins;
  for (uint32_t i = threadIdx.x; i < nbins; i += kFzHistThreads) dst[i] = fz_hist[i];
}

uint64_t factorize_table_slots(uint64_t virtual_rows) {   // the full size: load factor <= 1/2 whatever the input
  uint64_t s = 1024;
  while (s < 2 * virtual_rows) s <<= 1;
  return s;
}
uint64_t factorize_first_slots(uint64_t virtual_rows) {
  const uint64_t full = factorize_table_slots(virtual_rows);
  return full < (1ull << 20) ? full : (1ull << 20);
}
uint64_t factorize_next_slots(uint64_t virtual_rows, uint64_t slots) {   // after a pass that asked for more room; == slots: nothing larger exists
  const uint64_t full = factorize_table_slots(virtual_rows);
  if (slots >= full) return full;
  const uint64_t next = slots < (1ull << 24) ? (1ull << 24) : full;
  return next < full ? next : full;
}

// temp layout: table[slots] u64 | slot_of[V] u32 | bits[words] u32 | cnt[words] u32 | off[words + 1] u64 | scan scratch | claims u64 + flags u32
size_t factorize_temp_bytes(uint64_t virtual_rows, uint64_t slots) {
  const uint64_t words = (virtual_rows + 31) / 32;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up(slots * 8) + up(virtual_rows * 4) + up(words * 4) + up(words * 4) + up((words + 1) * 8) + up(scan_scratch_elems(words ? words : 1) * 8) + 256;
}

struct FzTemp {
  unsigned long long *table; uint32_t *slot_of, *bits, *cnt; unsigned long long *off, *scratch, *claims; uint32_t *flags;
};
static FzTemp fz_temp(void *temp, uint64_t V, uint64_t slots) {
  const uint64_t words = (V + 31) / 32;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  unsigned char *p = static_cast<unsigned char *>(temp);
  FzTemp t;
  t.table = reinterpret_cast<unsigned long long *>(p); p += up(slots * 8);
  t.slot_of = reinterpret_cast<uint32_t *>(p); p += up(V * 4);
  t.bits = reinterpret_cast<uint32_t *>(p); p += up(words * 4);
  t.cnt = reinterpret_cast<uint32_t *>(p); p += up(words * 4);
  t.off = reinterpret_cast<unsigned long long *>(p); p += up((words + 1) * 8);
  t.scratch = reinterpret_cast<unsigned long long *>(p); p += up(scan_scratch_elems(words ? words : 1) * 8);
  t.claims = reinterpret_cast<unsigned long long *>(p);
  t.flags = reinterpret_cast<uint32_t *>(p + 8);
  return t;
}
static dim3 fz_grid(uint64_t items) { const uint64_t b = (items + kFzBlock - 1) / kFzBlock; return dim3((unsigned)(b < 16384 ? (b ? b : 1) : 16384)); }

// One attempt with a table of `slots` slots: all launches, no synchronisation.  (*flags_dev_out)[0] != 0 afterwards: the table was too small —
// key ids, first rows and the count were not written; repeat with factorize_next_slots.  num_keys_dev: one u64 on the device.
void launch_factorize(hipStream_t s, const long long *const *cols_a, const uint8_t *keep_a, const long long *const *cols_b, const uint8_t *keep_b, uint64_t n,
                      int n_cols, uint64_t slots, void *temp, uint64_t *key_a, uint64_t *key_b, uint64_t *first_row, uint64_t first_row_cap,
                      unsigned long long *num_keys_dev, uint32_t **flags_dev_out, uint32_t *hist_bins, int hist_workgroups, uint64_t hist_chunk) {
  FzArgs A{};
  for (int c = 0; c < n_cols; ++c) { A.a[c] = cols_a[c]; A.b[c] = cols_b != nullptr ? cols_b[c] : nullptr; }
  A.keep_a = keep_a; A.keep_b = keep_b; A.n = n; A.n_cols = n_cols; A.sides = cols_b != nullptr ? 2u : 1u;
  const uint64_t V = n * A.sides, words = (V + 31) / 32;
  const FzTemp t = fz_temp(temp, V, slots);
  *flags_dev_out = t.flags;
  hipMemsetAsync(t.table, 0xFF, slots * 8, s);
  hipMemsetAsync(t.bits, 0, words * 4, s);
  hipMemsetAsync(t.claims, 0, 16, s);
  hipLaunchKernelGGL(k_fz_insert, fz_grid(V), dim3(kFzBlock), 0, s, A, t.table, slots - 1, t.slot_of, t.flags, t.claims,
                     slots >= factorize_table_slots(V) ? 0xFFFFFFFFu : kFzMaxProbe);
  hipLaunchKernelGGL(k_fz_mark, fz_grid(slots), dim3(kFzBlock), 0, s, t.table, slots, t.bits);
  hipLaunchKernelGGL(k_fz_popc, fz_grid(words), dim3(kFzBlock), 0, s, t.bits, words, t.cnt);
  launch_scan(s, t.cnt, t.off, words, t.scratch, num_keys_dev);
  hipLaunchKernelGGL(k_fz_ids, fz_grid(slots), dim3(kFzBlock), 0, s, t.table, slots, t.bits, t.off, first_row, first_row_cap, t.flags);
  if (hist_bins != nullptr) {     // the lookup pass leaves pass A's key-bin histogram behind (one workgroup per pass-B workgroup)
    static bool once = (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fz_lookup_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxBins * 4)), true);
    (void)once;
    hipLaunchKernelGGL(k_fz_lookup_hist, dim3((unsigned)hist_workgroups), dim3(kFzHistThreads), (size_t)kMaxBins * 4, s, A, t.table, t.slot_of, key_a, key_b, t.flags,
                       num_keys_dev, hist_chunk, hist_bins);
  } else {
    hipLaunchKernelGGL(k_fz_lookup, fz_grid(V), dim3(kFzBlock), 0, s, A, t.table, t.slot_of, key_a, key_b, t.flags);
  }
}

// ------------------------------------------------------------------------------------------------
// Arrow string column -> dictionary codes (ABI 10): the step in FRONT of the tuples above.  ClickHouse delivers the GROUP BY columns of
// the job (pod names, namespaces, labels, IPs, service port names: anomaly_detection.py:52-137) as Arrow `string` columns — n + 1 offsets
// and the bytes.  The host used to dictionary-encode them batch by batch (Arrow's C++ kernel, ~1e7 rows/s per column, the slowest stage
// of the whole job at 1e8 rows); here the column is encoded where the tuples are factorised anyway.  Same table as above — one 8-byte
// word per slot, fingerprint << 32 | first row, ONE compare-and-swap per claim, atomic min for the first row — over variable-length keys:
//   * a row's bytes are read as 8-byte words with two ALIGNED loads and a funnel shift (a string starts at any byte; an aligned word that
//     holds at least one byte of the string lies inside the buffer's pages, so nothing outside the allocation's pages is touched and a
//     second word is only loaded when the chunk crosses into it);
//   * equal fingerprints are confirmed by comparing the bytes with the representative row's (lengths first);
//   * the table starts SMALL (2^20 slots = 8 MB: resident in L2 / MALL for the common case of <= ~2e5 distinct values) instead of 2 n
//     slots; a pass that finds it filling up (a probe sequence > 32 slots, or more than half the slots claimed) raises a flag, every
//     thread leaves at its next row, and the host repeats the pass with the full-size table — one extra synchronisation in the rare
//     high-cardinality case, decided on the device;
//   * the insert pass leaves every row's SLOT (4 bytes) behind, so the second pass reads 4 bytes + a cached table word per row instead
//     of hashing and comparing the strings again.
// Codes are ids in order of first appearance (what pyarrow's dictionary_encode and pandas.factorize give): bitmap of the first rows,
// scan of its popcounts — the kernels above.  A null row (Arrow validity bitmap) encodes like the empty string, which is what the
// host path did (`fill_null("")`, theia_amd/clickhouse.py).
// ------------------------------------------------------------------------------------------------
struct StrArgs {
  const void *off;        // n + 1 offsets into data
  const uint8_t *data;
  const uint8_t *valid;   // Arrow validity bitmap (bit valid_off + i), NULL = no nulls
  uint64_t valid_off;
  uint64_t n;
  uint64_t data_bytes;
  int off64;              // offsets are int64 (large_string) instead of int32
};

static constexpr uint32_t kSeMaxProbe = 32;
enum : uint32_t { SE_FLAG_GROW = FZ_FLAG_GROW, SE_FLAG_BAD_OFFSETS = FZ_FLAG_BAD_INPUT };

// [b, b + len) of row v; false if the offsets are not usable
__device__ __forceinline__ bool se_span(const StrArgs &A, uint64_t v, uint64_t &b, uint32_t &len) {
  uint64_t e;
  if (A.off64) {
    const long long *o = static_cast<const long long *>(A.off);
    b = (uint64_t)o[v]; e = (uint64_t)o[v + 1];
  } else {
    const int *o = static_cast<const int *>(A.off);
    b = (uint64_t)(uint32_t)o[v]; e = (uint64_t)(uint32_t)o[v + 1];
  }
  if (e < b || e > A.data_bytes || e - b > 0xFFFFFFFFull) return false;
  len = (uint32_t)(e - b);
  if (A.valid != nullptr) {
    const uint64_t bit = A.valid_off + v;
    if (((A.valid[bit >> 3] >> (bit & 7)) & 1u) == 0) len = 0;   // null = ""
  }
  return true;
}

// m (1..8) bytes at p as a little-endian word, bytes beyond m zero.  Two aligned loads without a branch between them (a conditional second load
// made every chunk of a compare its own memory round trip): when the chunk does not reach into the next word, the first word is loaded twice.
__device__ __forceinline__ uint64_t se_load(const uint8_t *p, uint32_t m) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint64_t *w = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
  const uint32_t skip = (uint32_t)(a & 7);           // bytes of w[0] in front of p
  const bool two = skip + m > 8;                      // (skip >= 1 then: the left shift below is < 64)
  const uint64_t lo = w[0], hi = w[two ? 1 : 0];
  uint64_t x = lo >> (skip * 8);
  if (two) x |= hi << ((8 - skip) * 8);
  if (m < 8) x &= (1ull << (m * 8)) - 1ull;
  return x;
}

__device__ __forceinline__ uint64_t se_hash(const uint8_t *p, uint32_t len) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
  for (uint32_t i = 0; i < len; i += 8) h = fz_mix(h ^ se_load(p + i, len - i < 8 ? len - i : 8)) + 0x632BE59BD9B4E019ull;
  return fz_mix(h);
}

// own(at, m): m bytes of the lane's own string at offset `at`; q: the representative row's bytes in global memory.  What the compare costs is
// the number of load instructions — 64 lanes, 64 different representatives, 64 different cache lines per instruction, all from L2 — not their
// latency (3.4 of the insert pass's 5.1 ms, profiles/r4_v33_*; loading the chunks four at a time changed nothing).  So the representative's bytes
// are fetched as ALIGNED 16-byte words, each exactly once (a 29-byte name is 2-3 loads; chunk by chunk through se_load it was 8: every aligned word
// twice), and each 8-byte half is compared with the bytes of the own string it covers.
template <class Own>
__device__ __forceinline__ bool se_same_as(Own own, const uint8_t *q, uint32_t len) {
  if (len == 0) return true;                               // (nothing to read: the lengths are equal)
  const uintptr_t a = reinterpret_cast<uintptr_t>(q);
  const ulonglong2 *w = reinterpret_cast<const ulonglong2 *>(a & ~(uintptr_t)15);
  const int skip = (int)(a & 15);                          // bytes of w[0] in front of the string
  const uint32_t nw = ((uint32_t)skip + len + 15u) >> 4;   // aligned 16-byte words that hold a byte of the string (each inside the buffer's pages)
  bool same = true;
  for (uint32_t k0 = 0; k0 < nw && same; k0 += 2) {
    ulonglong2 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = w[k0 + u < nw ? k0 + u : k0];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (k0 + u >= nw) break;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int start = (int)(16u * (k0 + u)) + 8 * h - skip;        // offset in the string of this 8-byte half's first byte (may be < 0)
        const int s0 = start < 0 ? 0 : start;
        const int e0 = start + 8 < (int)len ? start + 8 : (int)len;
        if (e0 <= s0) continue;                                         // the half lies before or behind the string
        const uint32_t m = (uint32_t)(e0 - s0);
        uint64_t x = (h == 0 ? v[u].x : v[u].y) >> (8 * (s0 - start));
        if (m < 8) x &= (1ull << (m * 8)) - 1ull;
        same = same && x == own((uint32_t)s0, m);
      }
    }
  }
  return same;
}
__device__ __forceinline__ bool se_same(const uint8_t *p, const uint8_t *q, uint32_t len) {
  return se_same_as([&](uint32_t at, uint32_t m) { return se_load(p + at, m); }, q, len);
}

// A block's rows are CONSECUTIVE rows of the column, so their bytes are one contiguous range [off[r0], off[r0 + 256)) of `data`: it is copied
// into LDS with 16-byte loads (consecutive lanes on consecutive 16 bytes: the column's bytes cross the memory system once, in whole lines),
// and every lane then hashes and compares its own string from LDS.  Per-lane 8-byte global loads at a ~29-byte stride — the first version —
// made every wave-level load touch ~15 cache lines, five times over per row, and ran at 0.07 of the HBM peak (profiles/r4_v28_*).
// A block whose 256 rows hold more than kSeStage bytes (long labels) reads its strings from global memory lane by lane instead.
static constexpr uint32_t kSeStage = 24 * 1024;

// m (1..8) bytes at byte offset `at` of an 8-byte aligned LDS buffer, bytes beyond m zero
__device__ __forceinline__ uint64_t se_load_lds(const uint64_t *buf, uint32_t at, uint32_t m) {
  const uint32_t skip = at & 7u;
  uint64_t x = buf[at >> 3] >> (skip * 8);
  if (skip + m > 8) x |= buf[(at >> 3) + 1] << ((8 - skip) * 8);
  if (m < 8) x &= (1ull << (m * 8)) - 1ull;
  return x;
}

// every row into the table; slot_of[v] = the slot of v's string.  flags: SE_FLAG_GROW / SE_FLAG_BAD_OFFSETS; claims: slots claimed
__global__ __launch_bounds__(kFzBlock) void k_se_insert(StrArgs A, unsigned long long *__restrict__ table, uint64_t mask, uint32_t *__restrict__ slot_of,
                                                        uint32_t *__restrict__ flags, unsigned long long *__restrict__ claims, uint32_t max_probe) {
  __shared__ __attribute__((aligned(16))) uint64_t s_bytes[kSeStage / 8 + 4];
  __shared__ uint64_t s_lo, s_hi;
  __shared__ uint32_t s_stop;
  uint32_t claimed = 0;
  const uint32_t tid = threadIdx.x;
  for (uint64_t base = (uint64_t)blockIdx.x * kFzBlock; base < A.n; base += (uint64_t)gridDim.x * kFzBlock) {
    const uint64_t v = base + tid;
    const uint32_t rows = A.n - base < kFzBlock ? (uint32_t)(A.n - base) : kFzBlock;
    uint64_t b = 0; uint32_t len = 0;
    bool ok = true;
    if (tid < rows) ok = se_span(A, v, b, len);
    if (tid == 0) {
      // the block's byte range from its first and last row's RAW offsets (a null row reads as "" but its bytes may still be there)
      uint64_t lo, hi;
      if (A.off64) { const long long *o = static_cast<const long long *>(A.off); lo = (uint64_t)o[base]; hi = (uint64_t)o[base + rows]; }
      else { const int *o = static_cast<const int *>(A.off); lo = (uint64_t)(uint32_t)o[base]; hi = (uint64_t)(uint32_t)o[base + rows]; }
      s_lo = lo; s_hi = hi;
      s_stop = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // someone found the table too small (or the offsets bad)
    }
    if (!ok) atomicOr(flags, SE_FLAG_BAD_OFFSETS);      // (the host fails the call; this lane skips its row)
    __syncthreads();                                    // (also: the previous round's strings are no longer read)
    if (s_stop != 0u) break;                            // block-uniform
    const uint64_t lo = s_lo, hi = s_hi;
    const bool bad_range = hi < lo || hi > A.data_bytes;
    if (bad_range && tid == 0) atomicOr(flags, SE_FLAG_BAD_OFFSETS);
    const uintptr_t abs_lo = reinterpret_cast<uintptr_t>(A.data) + lo;
    const uintptr_t abs_a = abs_lo & ~(uintptr_t)15;    // the aligned 16-byte word that holds the range's first byte
    const uint64_t span = !bad_range && hi > lo ? (reinterpret_cast<uintptr_t>(A.data) + hi) - abs_a : 0;
    const bool block_staged = !bad_range && span <= kSeStage;     // block-uniform (from the shared bounds)
    if (block_staged && span) {
      const uint4 *src = reinterpret_cast<const uint4 *>(abs_a);
      uint4 *dst = reinterpret_cast<uint4 *>(s_bytes);
      for (uint32_t i = tid; (uint64_t)i * 16 < span; i += kFzBlock) dst[i] = src[i];   // (the last word may reach past `hi`: same aligned 16 bytes, same page)
    }
    __syncthreads();
    // a lane reads its string from the stage when it lies inside the staged range (always, for monotone offsets), else from global memory
    const bool staged = block_staged && b >= lo && b + len <= hi;
    if (ok && tid < rows) {
      const uint8_t *p = A.data + b;
      const uint32_t at = staged ? (uint32_t)((reinterpret_cast<uintptr_t>(A.data) + b) - abs_a) : 0u;      // own string's offset in the stage
      uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
      if (staged) {
        for (uint32_t i = 0; i < len; i += 8) h = fz_mix(h ^ se_load_lds(s_bytes, at + i, len - i < 8 ? len - i : 8)) + 0x632BE59BD9B4E019ull;
        h = fz_mix(h);
      } else {
        h = se_hash(p, len);
      }
      const unsigned long long mine = ((h >> 32) << 32) | v;
      uint32_t probes = 0;
#if defined(TAD_SE_PROF_NOTABLE)      // measurement builds (tools/build_variants.py): stage + hash only
      slot_of[v] = (uint32_t)(h & mask);
      continue;
#endif
      for (uint64_t s = h & mask;; s = (s + 1) & mask) {
        unsigned long long w = fz_peek(table + s);
        if (w == kFzEmpty) {
          w = atomicCAS(table + s, kFzEmpty, mine);
          if (w == kFzEmpty) { ++claimed; slot_of[v] = (uint32_t)s; break; }
        }
        if ((w >> 32) == (mine >> 32)) {
          uint64_t rb; uint32_t rlen;
          const uint64_t rep = w & 0xffffffffull;
#if defined(TAD_SE_PROF_NOCOMPARE)    // measurement builds: a fingerprint match is taken for equality
          bool same = true; rb = 0; rlen = len;
          if (false) {
#else
          bool same = se_span(A, rep, rb, rlen) && rlen == len;
          if (same) {
#endif
            const uint8_t *q = A.data + rb;
            if (staged) same = se_same_as([&](uint32_t o, uint32_t m) { return se_load_lds(s_bytes, at + o, m); }, q, len);
            else same = se_same(p, q, len);
          }
          if (same) {
            if (mine < w) atomicMin(table + s, mine);
            slot_of[v] = (uint32_t)s;
            break;
          }
        }
        if (++probes > max_probe) { atomicOr(flags, SE_FLAG_GROW); break; }
      }
    }
  }
  // one atomic per wavefront for the claim count (never one per claim)
  for (int o = 32; o > 0; o >>= 1) claimed += __shfl_down(claimed, o);
  if ((threadIdx.x & 63) == 0 && claimed) {
    const unsigned long long before = atomicAdd(claims, (unsigned long long)claimed);
    if (max_probe != 0xFFFFFFFFu && 2 * (before + claimed) > mask + 1) atomicOr(flags, SE_FLAG_GROW);
  }
}

// code of every row: slot -> id (k_fz_ids has turned the table's words into ids and written the first rows)
__global__ __launch_bounds__(kFzBlock) void k_se_codes(uint64_t n, const unsigned long long *__restrict__ table, const uint32_t *__restrict__ slot_of,
                                                       long long *__restrict__ codes, const uint32_t *__restrict__ flags) {
  if (*flags != 0u) return;     // the insert pass gave up (table too small / bad offsets): slot_of is not complete, the host repeats or fails
  constexpr int U = 4;
  const uint64_t stride = (uint64_t)gridDim.x * kFzBlock;
  for (uint64_t v0 = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v0 < n; v0 += U * stride) {
    uint32_t sl[U];
    unsigned long long id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sl[u] = v0 + u * stride < n ? slot_of[v0 + u * stride] : 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = v0 + u * stride < n ? table[sl[u]] : 0ull;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (v0 + u * stride < n) codes[v0 + u * stride] = (long long)id[u];
  }
}

// One attempt with a table of `slots` slots (factorize_first_slots / factorize_next_slots; temp: factorize_temp_bytes(n, slots)), all launches, no
// synchronisation.  (*flags_dev_out)[0] != 0 afterwards: bit 0 (grow) -> repeat with the next size (codes / first_row / num_values_dev are not
// written then), bit 1 -> the offsets are malformed.
void launch_encode_strings(hipStream_t s, const void *offsets, int off64, const uint8_t *data, uint64_t data_bytes, const uint8_t *valid, uint64_t valid_off,
                           uint64_t n, uint64_t slots, void *temp, long long *codes, uint64_t *first_row, uint64_t first_row_cap,
                           unsigned long long *num_values_dev, uint32_t **flags_dev_out) {
  StrArgs A{};
  A.off = offsets; A.data = data; A.valid = valid; A.valid_off = valid_off; A.n = n; A.data_bytes = data_bytes; A.off64 = off64;
  const uint64_t words = (n + 31) / 32;
  const FzTemp t = fz_temp(temp, n, slots);
  *flags_dev_out = t.flags;
  hipMemsetAsync(t.table, 0xFF, slots * 8, s);
  hipMemsetAsync(t.bits, 0, words * 4, s);
  hipMemsetAsync(t.claims, 0, 16, s);
  hipLaunchKernelGGL(k_se_insert, fz_grid(n), dim3(kFzBlock), 0, s, A, t.table, slots - 1, t.slot_of, t.flags, t.claims,
                     slots >= factorize_table_slots(n) ? 0xFFFFFFFFu : kSeMaxProbe);
  // (after a raised flag the table is incomplete: the passes below still run — over a bitmap of n bits, harmless — and k_se_codes returns
  // at once; the host reads flags and the count in ONE synchronisation and repeats the attempt with the next table size if asked to)
  hipLaunchKernelGGL(k_fz_mark, fz_grid(slots), dim3(kFzBlock), 0, s, t.table, slots, t.bits);
  hipLaunchKernelGGL(k_fz_popc, fz_grid(words), dim3(kFzBlock), 0, s, t.bits, words, t.cnt);
  launch_scan(s, t.cnt, t.off, words, t.scratch, num_values_dev);
  hipLaunchKernelGGL(k_fz_ids, fz_grid(slots), dim3(kFzBlock), 0, s, t.table, slots, t.bits, t.off, first_row, first_row_cap, t.flags);
  hipLaunchKernelGGL(k_se_codes, fz_grid(n), dim3(kFzBlock), 0, s, n, t.table, t.slot_of, codes, t.flags);
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_factorize() { return reinterpret_cast<const void *>(&k_fz_mark); }

}  // namespace tad
