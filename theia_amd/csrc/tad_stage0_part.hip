// tad_stage0_part.hip — Stage 0 v2: GROUP BY (key, flowEndSeconds) without random HBM atomics.
//
// Why: on MI355X the direct scatter (tad_kernels.hip:k_scatter) is bound by L2-miss read-modify-write
// transactions (~23.5e9/s measured, tools/ubench_scatter.hip) — 5.5 ms for 1e8 rows while the 24 B/row
// column stream alone takes 0.43 ms.  Here every random access lands in LDS instead:
//
//   pass A  k_meta_hist     one streaming read of the key/time columns (16 B/row): derives the
//                           flowEndSeconds lattice (min, max, gcd) AND a per-workgroup histogram of rows
//                           per key bin (bin = key >> shift_bin, <= 16384 bins, LDS counters).
//   (tiny)  k_part_*        bins -> partitions of KP = 2^shift_part consecutive keys whose KP x T tile of
//                           the point grid fits in LDS; exclusive offsets per (workgroup, partition):
//                           the partition pass needs no global atomics and its output order is fixed.
//   pass B  k_partition     streams the rows once more (24 B/row), turns each into a 16-byte record
//                           {value, tile-local cell}, groups a tile of S records by partition in LDS
//                           (LDS histogram + scan) and copies the runs out (16 B/row written).
//   pass C  k_tile_aggregate  one workgroup per partition: LDS u64 atomics (add wraps mod 2^64 / unsigned
//                           max) over its records, then writes its KP x T tile of the time-major grid
//                           (values + presence flags) with coalesced stores.  No grid memset needed.
//
// Integer add/max are associative and commutative, so the aggregates are bit-identical to v1 and to
// ClickHouse's sum()/max() over UInt64 whatever the record order.
#include "tad_internal.h"

namespace tad {

static constexpr int kPartThreads = 1024;
static constexpr uint32_t kCellPoison = 0xFFFFFFFFu;
static constexpr size_t kLdsBudget = 156 * 1024;  // dynamic LDS per workgroup; the rest of the CU's 160 KiB is for static __shared__

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

__device__ __forceinline__ bool p_row_kept(int64_t te, const int64_t *t_start, uint64_t i, RowFilter f) {
  if (f.end_time != 0 && !(te < f.end_time)) return false;
  if (f.start_time != 0 && t_start != nullptr && !(t_start[i] >= f.start_time)) return false;
  return true;
}

__device__ __forceinline__ bool p_bucket(const Lattice &L, int64_t te, uint64_t &bucket) {
  const uint64_t d = (uint64_t)te - (uint64_t)L.t0;
  uint64_t b;
  if (L.mode == 0) b = d;
  else if (L.mode == 1) { if (d >> 32) return false; b = __umul64hi(d, L.magic); }
  else b = d / (uint64_t)L.step;
  if (b >= L.nb || b * (uint64_t)L.step != d) return false;
  bucket = b;
  return true;
}

// ------------------------------------------------------------------------------------------------
// pass A — lattice partial + per-workgroup key-bin histogram.  Workgroup b owns rows
// [b * chunk, (b+1) * chunk): the SAME chunking as pass B, so the histogram row b is exactly what
// workgroup b of pass B will emit.
// ------------------------------------------------------------------------------------------------
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
  PMeta acc{0, 0, 0, 0, 0};
  uint32_t err = 0;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kPartThreads) {
    const int64_t te = t_end[i];
    const uint64_t k1 = key[i];
    const uint64_t k2 = key2 != nullptr ? key2[i] : TAD_KEY_SKIP;
    if ((k1 == TAD_KEY_SKIP && k2 == TAD_KEY_SKIP) || !p_row_kept(te, t_start, i, f)) continue;
    bool counted = false;
    if (k1 != TAD_KEY_SKIP) {
      if (k1 < K) { atomicAdd(&hist[(uint32_t)(k1 >> shift_bin)], 1u); counted = true; }
      else err |= DEV_ERR_KEY_RANGE;
    }
    if (k2 != TAD_KEY_SKIP) {
      if (k2 < K) { atomicAdd(&hist[(uint32_t)(k2 >> shift_bin)], 1u); counted = true; }
      else err |= DEV_ERR_KEY_RANGE;
    }
    if (!counted) continue;
    if (acc.used == 0) {
      acc.tmin = acc.tmax = acc.tref = te;
      acc.g = 0;
    } else {
      acc.tmin = te < acc.tmin ? te : acc.tmin;
      acc.tmax = te > acc.tmax ? te : acc.tmax;
      const uint64_t d = p_absdiff(te, acc.tref);
      if (d != 0 && (acc.g == 0 || d % acc.g != 0)) acc.g = p_gcd_u64(acc.g, d);
    }
    acc.used++;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    PMeta o;
    o.tmin = __shfl_down((long long)acc.tmin, d);
    o.tmax = __shfl_down((long long)acc.tmax, d);
    o.tref = __shfl_down((long long)acc.tref, d);
    o.g = __shfl_down((unsigned long long)acc.g, d);
    o.used = __shfl_down((unsigned long long)acc.used, d);
    acc = pmeta_merge(acc, o);
    err |= __shfl_down(err, d);
  }
  __shared__ PMeta s_acc[kPartThreads / 64];
  __shared__ uint32_t s_err[kPartThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_acc[wave] = acc; s_err[wave] = err; }
  __syncthreads();  // also: every histogram update of this workgroup is done
  if (threadIdx.x == 0) {
    PMeta a = s_acc[0];
    uint32_t e = s_err[0];
    for (int w = 1; w < kPartThreads / 64; ++w) { a = pmeta_merge(a, s_acc[w]); e |= s_err[w]; }
    MetaPartial p;
    p.tmin = a.tmin; p.tmax = a.tmax; p.tref = a.tref; p.g = a.g; p.used = a.used;
    partials[blockIdx.x] = p;
    if (e) atomicOr(&ctr->err, e);
  }
  uint32_t *out = binhist + (size_t)blockIdx.x * nbins;
  for (uint32_t i = threadIdx.x; i < nbins; i += kPartThreads) out[i] = hist[i];
}

// bins -> partition totals (one thread per partition)
__global__ void k_part_counts(const uint32_t *__restrict__ binhist, uint32_t nbins, int G, uint32_t bins_per_part,
                              uint32_t nparts, uint32_t *__restrict__ part_cnt) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nparts) return;
  uint64_t s = 0;
  const uint32_t b0 = p * bins_per_part;
  const uint32_t b1 = b0 + bins_per_part < nbins ? b0 + bins_per_part : nbins;
  for (int g = 0; g < G; ++g)
    for (uint32_t b = b0; b < b1; ++b) s += binhist[(size_t)g * nbins + b];
  part_cnt[p] = (uint32_t)s;  // v2 requires n_rows < 2^32 (checked on the host)
}

// exclusive offsets per (workgroup, partition): offs[g * nparts + p] = start[p] + sum_{g' < g} cnt[g'][p]
__global__ void k_part_offsets(const uint32_t *__restrict__ binhist, uint32_t nbins, int G, uint32_t bins_per_part,
                               uint32_t nparts, const unsigned long long *__restrict__ part_start,
                               unsigned long long *__restrict__ offs) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nparts) return;
  unsigned long long run = part_start[p];
  const uint32_t b0 = p * bins_per_part;
  const uint32_t b1 = b0 + bins_per_part < nbins ? b0 + bins_per_part : nbins;
  for (int g = 0; g < G; ++g) {
    offs[(size_t)g * nparts + p] = run;
    uint32_t s = 0;
    for (uint32_t b = b0; b < b1; ++b) s += binhist[(size_t)g * nbins + b];
    run += s;
  }
}

// ------------------------------------------------------------------------------------------------
// pass B — group records by partition through LDS.
// LDS carve (dynamic): rec[S] (16 B) | hist[F] u32 | off[F+1] u32 | cur[F] u64 | part[S] u16
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
  uint32_t nparts;
  const unsigned long long *offs;  // [G][nparts]
  ulonglong2 *recs;   // out: {value, cell}
  DevCounters *ctr;
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
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < kPartThreads / 64; ++w) {
    if (w < wave) base += s_wave[w];
    tot += s_wave[w];
  }
  uint32_t run = base + incl - sum;
  for (uint32_t j = 0; j < per; ++j)
    if (b0 + j < n) { const uint32_t c = a[b0 + j]; a[b0 + j] = run; run += c; }
  if (threadIdx.x == 0) a[n] = tot;
  __syncthreads();
}

template <int RPT, bool HAS2>
__global__ __launch_bounds__(kPartThreads) void k_partition(PartArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t S = (uint32_t)RPT * kPartThreads * (HAS2 ? 2 : 1);  // record slots per tile
  const uint32_t F = A.nparts;
  ulonglong2 *rec = reinterpret_cast<ulonglong2 *>(smem);
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem + (size_t)S * 16);
  uint32_t *off = hist + F;
  unsigned long long *cur = reinterpret_cast<unsigned long long *>(off + (F + 1) + 1);  // hist F + off F+1 + 1 pad = even
  uint16_t *part = reinterpret_cast<uint16_t *>(cur + F);
  __shared__ uint32_t s_wave[kPartThreads / 64];

  const unsigned long long *my_offs = A.offs + (size_t)blockIdx.x * F;
  for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) { cur[p] = my_offs[p]; hist[p] = 0; }
  __syncthreads();

  const uint64_t lo = (uint64_t)blockIdx.x * A.chunk;
  const uint64_t hi = lo + A.chunk < A.n ? lo + A.chunk : A.n;
  uint32_t err = 0, used = 0;
  constexpr int NSLOT = RPT * (HAS2 ? 2 : 1);

  for (uint64_t base = lo; base < hi; base += (uint64_t)RPT * kPartThreads) {
    // ---- phase 1: load rows, make records, rank them inside their partition (LDS atomic) ----
    uint64_t r_val[NSLOT];
    uint32_t r_cell[NSLOT], r_part[NSLOT], r_rank[NSLOT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const uint64_t i = base + (uint64_t)j * kPartThreads + threadIdx.x;
      uint64_t k1 = TAD_KEY_SKIP, k2 = TAD_KEY_SKIP, v = 0;
      int64_t te = 0;
      bool kept = false;
      if (i < hi) {
        k1 = A.key[i];
        te = A.t_end[i];
        v = A.value[i];
        if (HAS2) k2 = A.key2[i];
        kept = p_row_kept(te, A.t_start, i, A.f);
      }
#pragma unroll
      for (int h = 0; h < (HAS2 ? 2 : 1); ++h) {
        const int slot = j * (HAS2 ? 2 : 1) + h;
        const uint64_t k = h == 0 ? k1 : k2;
        r_part[slot] = 0xFFFFFFFFu;
        r_val[slot] = v;
        r_cell[slot] = kCellPoison;
        r_rank[slot] = 0;
        if (kept && k != TAD_KEY_SKIP && k < A.K) {  // same predicate as pass A: the slot is reserved
          uint64_t bucket;
          if (p_bucket(A.L, te, bucket)) {
            r_cell[slot] = (uint32_t)bucket * (A.kp_mask + 1u) + ((uint32_t)k & A.kp_mask);
            used++;
          } else {
            err |= DEV_ERR_OFF_LATTICE;  // only possible with a wrong caller-supplied lattice hint
          }
          const uint32_t p = (uint32_t)(k >> A.shift_part);
          r_part[slot] = p;
          r_rank[slot] = atomicAdd(&hist[p], 1u);
        }
      }
    }
    __syncthreads();
    // ---- phase 2: exclusive scan of the tile's partition histogram ----
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) off[p] = hist[p];
    __syncthreads();
    lds_exclusive_scan(off, F, s_wave);
    // ---- phase 3: place the records in partition order ----
#pragma unroll
    for (int slot = 0; slot < NSLOT; ++slot) {
      if (r_part[slot] != 0xFFFFFFFFu) {
        const uint32_t pos = off[r_part[slot]] + r_rank[slot];
        rec[pos] = make_ulonglong2(r_val[slot], (unsigned long long)r_cell[slot]);
        part[pos] = (uint16_t)r_part[slot];
      }
    }
    __syncthreads();
    // ---- phase 4: copy the runs out (consecutive lanes -> consecutive records of one partition) ----
    const uint32_t total = off[F];
    for (uint32_t idx = threadIdx.x; idx < total; idx += kPartThreads) {
      const uint32_t p = part[idx];
      A.recs[cur[p] + (idx - off[p])] = rec[idx];
    }
    __syncthreads();
    // ---- phase 5: advance the cursors, clear the histogram ----
    for (uint32_t p = threadIdx.x; p < F; p += kPartThreads) { cur[p] += hist[p]; hist[p] = 0; }
    __syncthreads();
  }
  unsigned long long u = used;
  for (int d = 32; d >= 1; d >>= 1) { u += __shfl_down(u, d); err |= __shfl_down(err, d); }
  if ((threadIdx.x & 63) == 0) {
    if (u) atomicAdd(&A.ctr->rows_used, u);
    if (err) atomicOr(&A.ctr->err, err);
  }
}

// ------------------------------------------------------------------------------------------------
// pass C — one workgroup per partition: aggregate its records in an LDS tile, write the tile out.
// LDS carve: vals[KP*T] u64 | flags[KP*T] u8
// ------------------------------------------------------------------------------------------------
template <bool OPMAX>
__global__ __launch_bounds__(kPartThreads) void k_tile_aggregate(const ulonglong2 *__restrict__ recs,
                                                                 const unsigned long long *__restrict__ part_start,
                                                                 int shift_part, Grid g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t KP = 1u << shift_part;
  const uint32_t cells = KP * (uint32_t)g.T;
  unsigned long long *vals = reinterpret_cast<unsigned long long *>(smem);
  uint8_t *flags = smem + (size_t)cells * 8;
  for (uint32_t c = threadIdx.x; c < cells; c += kPartThreads) vals[c] = 0ull;
  for (uint32_t c = threadIdx.x; c < (cells + 3) / 4; c += kPartThreads) reinterpret_cast<uint32_t *>(flags)[c] = 0u;
  __syncthreads();
  const uint32_t p = blockIdx.x;
  const unsigned long long lo = part_start[p], hi = part_start[p + 1];
  for (unsigned long long i = lo + threadIdx.x; i < hi; i += kPartThreads) {
    const ulonglong2 r = recs[i];
    const uint32_t c = (uint32_t)r.y;
    if (c == kCellPoison) continue;
    if (OPMAX) atomicMax(&vals[c], r.x);
    else atomicAdd(&vals[c], r.x);
    flags[c] = FLAG_PRESENT;
  }
  __syncthreads();
  const uint64_t k0 = (uint64_t)p << shift_part;
  for (uint32_t c = threadIdx.x; c < cells; c += kPartThreads) {
    const uint32_t b = c >> shift_part, kk = c & (KP - 1);
    const uint64_t k = k0 + kk;
    if (k < g.K) {
      g.val[(uint64_t)b * g.K + k] = vals[c];
      g.flag[(uint64_t)b * g.K + k] = flags[c];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static constexpr uint32_t kMaxBins = 16384;

bool part_plan_bins(uint64_t n, uint64_t K, PartPlan *pl) {
  if (K == 0 || n == 0 || n >= (1ull << 32)) return false;
  int s = 0;
  while (((K + (1ull << s) - 1) >> s) > kMaxBins) ++s;
  pl->shift_bin = s;
  pl->nbins = (uint32_t)((K + (1ull << s) - 1) >> s);
  pl->G = 256;  // one workgroup per CU for passes A and B
  uint64_t chunk = (n + pl->G - 1) / pl->G;
  pl->chunk = chunk;
  return true;
}

bool part_plan_tiles(uint64_t K, uint64_t T, bool has2, PartPlan *pl) {
  if (T == 0 || T >= (1ull << 31)) return false;
  // largest power-of-two key tile whose KP x T (u64 + flag byte) fits in LDS
  int sp = -1;
  for (int c = 16; c >= pl->shift_bin; --c)
    if (((uint64_t)T << c) * 9 + 16 <= kLdsBudget && ((uint64_t)T << c) < (1ull << 31)) { sp = c; break; }
  if (sp < 0) return false;
  while (sp > pl->shift_bin && (1ull << (sp - 1)) >= K) --sp;  // no wider than the key space
  pl->shift_part = sp;
  pl->KP = 1u << sp;
  pl->nparts = (uint32_t)((K + pl->KP - 1) >> sp);
  if (pl->nparts > 65535) return false;  // part[] is u16
  pl->bins_per_part = 1u << (sp - pl->shift_bin);
  pl->agg_lds = ((size_t)pl->KP * T * 9 + 15) & ~(size_t)15;
  // pass B: records per tile limited by LDS
  const size_t fixed = (size_t)pl->nparts * 4 + ((size_t)pl->nparts + 2) * 4 + (size_t)pl->nparts * 8 + 64;
  pl->rpt = 0;
  const int mult = has2 ? 2 : 1;
  for (int r : {6, 4, 2, 1}) {
    const size_t slots = (size_t)r * kPartThreads * mult;
    if (slots * 18 + fixed <= kLdsBudget) { pl->rpt = r; pl->part_lds = (slots * 18 + fixed + 15) & ~(size_t)15; break; }
  }
  return pl->rpt != 0;
}

void launch_meta_hist(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                      const int64_t *t_start, uint64_t n, uint64_t K, RowFilter f, const PartPlan &pl,
                      MetaPartial *partials, uint32_t *binhist, DevCounters *ctr) {
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void *>(k_meta_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget); attr = true; }
  hipLaunchKernelGGL(k_meta_hist, dim3(pl.G), dim3(kPartThreads), (size_t)pl.nbins * 4, s, key, key2, t_end, t_start, n,
                     pl.chunk, K, f, pl.shift_bin, pl.nbins, partials, binhist, ctr);
}

void launch_part_counts(hipStream_t s, const uint32_t *binhist, const PartPlan &pl, uint32_t *part_cnt) {
  hipLaunchKernelGGL(k_part_counts, dim3((pl.nparts + 255) / 256), dim3(256), 0, s, binhist, pl.nbins, pl.G, pl.bins_per_part,
                     pl.nparts, part_cnt);
}

void launch_part_offsets(hipStream_t s, const uint32_t *binhist, const PartPlan &pl, const unsigned long long *part_start,
                         unsigned long long *offs) {
  hipLaunchKernelGGL(k_part_offsets, dim3((pl.nparts + 255) / 256), dim3(256), 0, s, binhist, pl.nbins, pl.G, pl.bins_per_part,
                     pl.nparts, part_start, offs);
}

void launch_partition(hipStream_t s, const uint64_t *key, const uint64_t *key2, const int64_t *t_end,
                      const int64_t *t_start, const uint64_t *value, uint64_t n, uint64_t K, RowFilter f, Lattice L,
                      const PartPlan &pl, const unsigned long long *offs, void *recs, DevCounters *ctr) {
  PartArgs A;
  A.key = key; A.key2 = key2; A.t_end = t_end; A.t_start = t_start; A.value = value;
  A.n = n; A.chunk = pl.chunk; A.K = K; A.f = f; A.L = L;
  A.shift_part = pl.shift_part; A.kp_mask = pl.KP - 1; A.nparts = pl.nparts;
  A.offs = offs; A.recs = static_cast<ulonglong2 *>(recs); A.ctr = ctr;
  const bool has2 = key2 != nullptr;
#define TAD_PART(RPT, H2)                                                                                             \
  do {                                                                                                                \
    static bool attr = false;                                                                                         \
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void *>(k_partition<RPT, H2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget); attr = true; } \
    hipLaunchKernelGGL((k_partition<RPT, H2>), dim3(pl.G), dim3(kPartThreads), pl.part_lds, s, A);                     \
  } while (0)
  switch (pl.rpt) {
    case 6: if (has2) TAD_PART(6, true); else TAD_PART(6, false); break;
    case 4: if (has2) TAD_PART(4, true); else TAD_PART(4, false); break;
    case 2: if (has2) TAD_PART(2, true); else TAD_PART(2, false); break;
    default: if (has2) TAD_PART(1, true); else TAD_PART(1, false); break;
  }
#undef TAD_PART
}

void launch_tile_aggregate(hipStream_t s, const void *recs, const unsigned long long *part_start, const PartPlan &pl,
                           Grid g, bool op_max) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_aggregate<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_aggregate<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    attr = true;
  }
  if (op_max)
    hipLaunchKernelGGL((k_tile_aggregate<true>), dim3(pl.nparts), dim3(kPartThreads), pl.agg_lds, s,
                       static_cast<const ulonglong2 *>(recs), part_start, pl.shift_part, g);
  else
    hipLaunchKernelGGL((k_tile_aggregate<false>), dim3(pl.nparts), dim3(kPartThreads), pl.agg_lds, s,
                       static_cast<const ulonglong2 *>(recs), part_start, pl.shift_part, g);
}

}  // namespace tad
