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
// id(first row r) = first rows before r.  A second pass over the rows looks every tuple up again and writes its id.
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
  for (int c = 0; c < A.n_cols; ++c)
    if (fz_value(A, rep, c) != t[c]) return false;
  return true;
}

// every kept virtual row into the table; on return a slot's low half = the smallest virtual row holding its tuple
__global__ __launch_bounds__(kFzBlock) void k_fz_insert(FzArgs A, unsigned long long *__restrict__ table, uint64_t mask) {
  const uint64_t V = A.n * A.sides;
  for (uint64_t v = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v < V; v += (uint64_t)gridDim.x * kFzBlock) {
    if (!fz_kept(A, v)) continue;
    long long t[kFzMaxCols];
    const uint64_t h = fz_hash(A, v, t);
    const unsigned long long mine = ((h >> 32) << 32) | v;     // (v < 2^32 - 1: never the empty word)
    for (uint64_t s = h & mask;; s = (s + 1) & mask) {
      unsigned long long w = __hip_atomic_load(table + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w == kFzEmpty) {
        w = atomicCAS(table + s, kFzEmpty, mine);
        if (w == kFzEmpty) break;                               // claimed
      }
      if ((w >> 32) == (mine >> 32) && fz_same(A, v, t, w & 0xffffffffull)) {
        if (mine < w) atomicMin(table + s, mine);                // an earlier row of the same tuple (same high half: the min stays in the class)
        break;
      }
    }
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

// id of every row (TAD_KEY_SKIP for rows that are not kept); first_row[id] by the row that is its tuple's first
__global__ __launch_bounds__(kFzBlock) void k_fz_lookup(FzArgs A, const unsigned long long *__restrict__ table, uint64_t mask, const uint32_t *__restrict__ bits,
                                                         const unsigned long long *__restrict__ off, uint64_t *__restrict__ key_a, uint64_t *__restrict__ key_b,
                                                         uint64_t *__restrict__ first_row, uint64_t first_row_cap) {
  const uint64_t V = A.n * A.sides;
  for (uint64_t v = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v < V; v += (uint64_t)gridDim.x * kFzBlock) {
    uint64_t *out = v >= A.n ? key_b + (v - A.n) : key_a + v;
    if (!fz_kept(A, v)) { *out = TAD_KEY_SKIP; continue; }
    long long t[kFzMaxCols];
    const uint64_t h = fz_hash(A, v, t);
    uint64_t rep = 0;
    for (uint64_t s = h & mask;; s = (s + 1) & mask) {
      const unsigned long long w = table[s];
      if (w == kFzEmpty) { rep = v; break; }                    // (cannot happen after k_fz_insert; never loop forever)
      if ((w >> 32) == (h >> 32) && fz_same(A, v, t, w & 0xffffffffull)) { rep = w & 0xffffffffull; break; }
    }
    const uint64_t id = off[rep >> 5] + (uint64_t)__popc(bits[rep >> 5] & ((1u << (rep & 31ull)) - 1u));
    *out = id;
    if (rep == v && id < first_row_cap) first_row[id] = v;
  }
}

uint64_t factorize_table_slots(uint64_t virtual_rows) {
  uint64_t s = 1024;
  while (s < 2 * virtual_rows) s <<= 1;
  return s;
}

// temp layout: table[slots] u64 | bits[words] u32 | cnt[words] u32 | off[words + 1] u64 | scan scratch
size_t factorize_temp_bytes(uint64_t virtual_rows) {
  const uint64_t slots = factorize_table_slots(virtual_rows), words = (virtual_rows + 31) / 32;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up(slots * 8) + up(words * 4) + up(words * 4) + up((words + 1) * 8) + up(scan_scratch_elems(words ? words : 1) * 8) + 256;
}

// num_keys_dev: one u64 on the device
void launch_factorize(hipStream_t s, const long long *const *cols_a, const uint8_t *keep_a, const long long *const *cols_b, const uint8_t *keep_b, uint64_t n,
                      int n_cols, void *temp, uint64_t *key_a, uint64_t *key_b, uint64_t *first_row, uint64_t first_row_cap, unsigned long long *num_keys_dev) {
  FzArgs A{};
  for (int c = 0; c < n_cols; ++c) { A.a[c] = cols_a[c]; A.b[c] = cols_b != nullptr ? cols_b[c] : nullptr; }
  A.keep_a = keep_a; A.keep_b = keep_b; A.n = n; A.n_cols = n_cols; A.sides = cols_b != nullptr ? 2u : 1u;
  const uint64_t V = n * A.sides;
  const uint64_t slots = factorize_table_slots(V), words = (V + 31) / 32;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  unsigned char *p = static_cast<unsigned char *>(temp);
  unsigned long long *table = reinterpret_cast<unsigned long long *>(p); p += up(slots * 8);
  uint32_t *bits = reinterpret_cast<uint32_t *>(p); p += up(words * 4);
  uint32_t *cnt = reinterpret_cast<uint32_t *>(p); p += up(words * 4);
  unsigned long long *off = reinterpret_cast<unsigned long long *>(p); p += up((words + 1) * 8);
  unsigned long long *scratch = reinterpret_cast<unsigned long long *>(p);
  hipMemsetAsync(table, 0xFF, slots * 8, s);
  hipMemsetAsync(bits, 0, words * 4, s);
  auto grid = [](uint64_t items) { const uint64_t b = (items + kFzBlock - 1) / kFzBlock; return dim3((unsigned)(b < 16384 ? (b ? b : 1) : 16384)); };
  hipLaunchKernelGGL(k_fz_insert, grid(V), dim3(kFzBlock), 0, s, A, table, slots - 1);
  hipLaunchKernelGGL(k_fz_mark, grid(slots), dim3(kFzBlock), 0, s, table, slots, bits);
  hipLaunchKernelGGL(k_fz_popc, grid(words), dim3(kFzBlock), 0, s, bits, words, cnt);
  launch_scan(s, cnt, off, words, scratch, num_keys_dev);
  hipLaunchKernelGGL(k_fz_lookup, grid(V), dim3(kFzBlock), 0, s, A, table, slots - 1, bits, off, key_a, key_b, first_row, first_row_cap);
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
enum : uint32_t { SE_FLAG_GROW = 1u, SE_FLAG_BAD_OFFSETS = 2u };

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

// m (1..8) bytes at p as a little-endian word, bytes beyond m zero
__device__ __forceinline__ uint64_t se_load(const uint8_t *p, uint32_t m) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint64_t *w = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
  const uint32_t skip = (uint32_t)(a & 7);           // bytes of w[0] in front of p
  uint64_t x = w[0] >> (skip * 8);
  if (skip + m > 8) x |= w[1] << ((8 - skip) * 8);   // (skip >= 1 here: the shift is < 64)
  if (m < 8) x &= (1ull << (m * 8)) - 1ull;
  return x;
}

__device__ __forceinline__ uint64_t se_hash(const uint8_t *p, uint32_t len) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
  for (uint32_t i = 0; i < len; i += 8) h = fz_mix(h ^ se_load(p + i, len - i < 8 ? len - i : 8)) + 0x632BE59BD9B4E019ull;
  return fz_mix(h);
}

__device__ __forceinline__ bool se_same(const uint8_t *p, const uint8_t *q, uint32_t len) {
  for (uint32_t i = 0; i < len; i += 8) {
    const uint32_t m = len - i < 8 ? len - i : 8;
    if (se_load(p + i, m) != se_load(q + i, m)) return false;
  }
  return true;
}

// every row into the table; slot_of[v] = the slot of v's string.  flags: SE_FLAG_GROW / SE_FLAG_BAD_OFFSETS; claims: slots claimed
__global__ __launch_bounds__(kFzBlock) void k_se_insert(StrArgs A, unsigned long long *__restrict__ table, uint64_t mask, uint32_t *__restrict__ slot_of,
                                                        uint32_t *__restrict__ flags, unsigned long long *__restrict__ claims) {
  uint32_t claimed = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v < A.n; v += (uint64_t)gridDim.x * kFzBlock) {
    if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;      // someone found the table too small (or the offsets bad)
    uint64_t b; uint32_t len;
    if (!se_span(A, v, b, len)) { atomicOr(flags, SE_FLAG_BAD_OFFSETS); break; }
    const uint8_t *p = A.data + b;
    const uint64_t h = se_hash(p, len);
    const unsigned long long mine = ((h >> 32) << 32) | v;
    uint32_t probes = 0;
    for (uint64_t s = h & mask;; s = (s + 1) & mask) {
      unsigned long long w = __hip_atomic_load(table + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w == kFzEmpty) {
        w = atomicCAS(table + s, kFzEmpty, mine);
        if (w == kFzEmpty) { ++claimed; slot_of[v] = (uint32_t)s; break; }
      }
      if ((w >> 32) == (mine >> 32)) {
        uint64_t rb; uint32_t rlen;
        const uint64_t rep = w & 0xffffffffull;
        if (se_span(A, rep, rb, rlen) && rlen == len && se_same(p, A.data + rb, len)) {
          if (mine < w) atomicMin(table + s, mine);
          slot_of[v] = (uint32_t)s;
          break;
        }
      }
      if (++probes > kSeMaxProbe) { atomicOr(flags, SE_FLAG_GROW); break; }
    }
  }
  // one atomic per wavefront for the claim count (never one per claim)
  for (int o = 32; o > 0; o >>= 1) claimed += __shfl_down(claimed, o);
  if ((threadIdx.x & 63) == 0 && claimed) {
    const unsigned long long before = atomicAdd(claims, (unsigned long long)claimed);
    if (2 * (before + claimed) > mask + 1) atomicOr(flags, SE_FLAG_GROW);
  }
}

// code of every row from its slot; first_row[id] by the row that is its string's first
__global__ __launch_bounds__(kFzBlock) void k_se_codes(uint64_t n, const unsigned long long *__restrict__ table, const uint32_t *__restrict__ slot_of,
                                                       const uint32_t *__restrict__ bits, const unsigned long long *__restrict__ off,
                                                       long long *__restrict__ codes, uint64_t *__restrict__ first_row, uint64_t first_row_cap,
                                                       const uint32_t *__restrict__ flags) {
  if (*flags != 0u) return;     // the insert pass gave up (table too small / bad offsets): slot_of is not complete, the host repeats or fails
  for (uint64_t v = (uint64_t)blockIdx.x * kFzBlock + threadIdx.x; v < n; v += (uint64_t)gridDim.x * kFzBlock) {
    const uint64_t rep = table[slot_of[v]] & 0xffffffffull;
    const uint64_t id = off[rep >> 5] + (uint64_t)__popc(bits[rep >> 5] & ((1u << (rep & 31ull)) - 1u));
    codes[v] = (long long)id;
    if (rep == v && id < first_row_cap) first_row[id] = v;
  }
}

uint64_t encode_strings_small_slots(uint64_t n) {
  const uint64_t full = factorize_table_slots(n);
  return full < (1ull << 20) ? full : (1ull << 20);
}

// temp layout: table[slots] u64 | slot_of[n] u32 | bits[words] u32 | cnt[words] u32 | off[words + 1] u64 | scan scratch | flags u32 + claims u64
size_t encode_strings_temp_bytes(uint64_t n, uint64_t slots) {
  const uint64_t words = (n + 31) / 32;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up(slots * 8) + up(n * 4) + up(words * 4) + up(words * 4) + up((words + 1) * 8) + up(scan_scratch_elems(words ? words : 1) * 8) + 256;
}

// One attempt with a table of `slots` slots, all launches, no synchronisation.  (*flags_dev_out)[0] != 0 afterwards: bit 0 (grow) -> repeat with
// factorize_table_slots(n) slots (codes / first_row / num_values_dev are not written then), bit 1 -> the offsets are malformed.
void launch_encode_strings(hipStream_t s, const void *offsets, int off64, const uint8_t *data, uint64_t data_bytes, const uint8_t *valid, uint64_t valid_off,
                           uint64_t n, uint64_t slots, void *temp, long long *codes, uint64_t *first_row, uint64_t first_row_cap,
                           unsigned long long *num_values_dev, uint32_t **flags_dev_out) {
  StrArgs A{};
  A.off = offsets; A.data = data; A.valid = valid; A.valid_off = valid_off; A.n = n; A.data_bytes = data_bytes; A.off64 = off64;
  const uint64_t words = (n + 31) / 32;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  unsigned char *p = static_cast<unsigned char *>(temp);
  unsigned long long *table = reinterpret_cast<unsigned long long *>(p); p += up(slots * 8);
  uint32_t *slot_of = reinterpret_cast<uint32_t *>(p); p += up(n * 4);
  uint32_t *bits = reinterpret_cast<uint32_t *>(p); p += up(words * 4);
  uint32_t *cnt = reinterpret_cast<uint32_t *>(p); p += up(words * 4);
  unsigned long long *off = reinterpret_cast<unsigned long long *>(p); p += up((words + 1) * 8);
  unsigned long long *scratch = reinterpret_cast<unsigned long long *>(p); p += up(scan_scratch_elems(words ? words : 1) * 8);
  unsigned long long *claims = reinterpret_cast<unsigned long long *>(p);
  uint32_t *flags = reinterpret_cast<uint32_t *>(p + 8);
  *flags_dev_out = flags;
  hipMemsetAsync(table, 0xFF, slots * 8, s);
  hipMemsetAsync(bits, 0, words * 4, s);
  hipMemsetAsync(claims, 0, 16, s);
  auto grid = [](uint64_t items) { const uint64_t b = (items + kFzBlock - 1) / kFzBlock; return dim3((unsigned)(b < 16384 ? (b ? b : 1) : 16384)); };
  hipLaunchKernelGGL(k_se_insert, grid(n), dim3(kFzBlock), 0, s, A, table, slots - 1, slot_of, flags, claims);
  // (after a raised flag the table is incomplete: the passes below still run — over a bitmap of n bits, harmless — and k_se_codes returns
  // at once; the host reads flags and the count in ONE synchronisation and repeats the attempt with the full-size table if asked to)
  hipLaunchKernelGGL(k_fz_mark, grid(slots), dim3(kFzBlock), 0, s, table, slots, bits);
  hipLaunchKernelGGL(k_fz_popc, grid(words), dim3(kFzBlock), 0, s, bits, words, cnt);
  launch_scan(s, cnt, off, words, scratch, num_values_dev);
  hipLaunchKernelGGL(k_se_codes, grid(n), dim3(kFzBlock), 0, s, n, table, slot_of, bits, off, codes, first_row, first_row_cap, flags);
}

}  // namespace tad
