// tad_ingest.hip — the last step of columnar ingest: what ClickHouse's ArrowStream delivers -> the engine's 8-byte device columns.
//
// The reference reads the GROUP BY result through one JDBC connection into Spark rows (anomaly_detection.py:651-662).  Here the raw rows arrive as
// Arrow record batches over several connections (theia_amd/clickhouse.py:fetch_flows_device); string columns come as Arrow DICTIONARY arrays — what
// ClickHouse sends for LowCardinality columns under output_format_arrow_low_cardinality_as_dictionary — with a dictionary per record batch.  Per batch
// and column the host only looks at the DISTINCT values (it maps the batch's dictionary into the column's job-wide one); the rows are touched here:
//   k_widen      dst[i] = table ? table[src[i]] : src[i], src of 8 / 16 / 32 / 64 bits (dictionary indices, UInt32 DateTime, UInt16 ports, first-row gathers)
//   k_mask_rows  keep[i] = AND_t masks[t][codes[t][i]] — the SQL's string predicates (anomaly_detection.py:507-614), evaluated on the distinct values by the
//                host, applied to the rows as a gather
// Both are plain streaming kernels: 1-8 B in, 8 (1) B out per row, bound by HBM; a grid-stride loop over 2048 workgroups fills the 256 CUs.
#include <hip/hip_runtime.h>

#include "tad_internal.h"

namespace tad {

namespace {
constexpr int kIngestBlock = 256;
constexpr int kIngestGrid = 2048;

template <typename T>
__global__ __launch_bounds__(kIngestBlock) void k_widen(const T *__restrict__ src, uint64_t n, const long long *__restrict__ table, uint64_t table_len,
                                                       long long *__restrict__ dst, unsigned int *__restrict__ err) {
  const uint64_t stride = (uint64_t)gridDim.x * kIngestBlock;
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * kIngestBlock + threadIdx.x; i < n; i += stride) {
    const T v = src[i];
    if (table != nullptr) {
      const uint64_t ix = (uint64_t)(long long)v;        // a negative index becomes huge: out of range
      if (ix < table_len) dst[i] = table[ix];
      else bad = true;
    } else {
      dst[i] = (long long)v;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(err, 1u);
}

struct MaskArgs {
  const long long *codes[kMaskMaxTerms];
  const uint8_t *masks[kMaskMaxTerms];
  uint64_t len[kMaskMaxTerms];
  int n_terms;
};

__global__ __launch_bounds__(kIngestBlock) void k_mask_rows(MaskArgs A, uint64_t n, int combine, uint8_t *__restrict__ keep, unsigned int *__restrict__ err) {
  const uint64_t stride = (uint64_t)gridDim.x * kIngestBlock;
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * kIngestBlock + threadIdx.x; i < n; i += stride) {
    uint8_t k = combine ? keep[i] : (uint8_t)1;
    for (int t = 0; t < A.n_terms; ++t) {
      const uint64_t c = (uint64_t)A.codes[t][i];
      if (c < A.len[t]) k &= A.masks[t][c] != 0 ? (uint8_t)1 : (uint8_t)0;
      else { bad = true; k = 0; }
    }
    keep[i] = k;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(err, 1u);
}

unsigned grid_for(uint64_t n) {
  const uint64_t b = (n + kIngestBlock - 1) / kIngestBlock;
  return (unsigned)(b < (uint64_t)kIngestGrid ? (b ? b : 1) : kIngestGrid);
}
}  // namespace

// src: DEVICE, n elements of `bits` bits (is_signed: sign-extend); table NULL = widen only.  *err |= 1 on an index outside the table.
void launch_widen(hipStream_t s, const void *src, int bits, bool is_signed, uint64_t n, const long long *table, uint64_t table_len, long long *dst, unsigned int *err) {
  if (n == 0) return;
  const unsigned g = grid_for(n);
#define TAD_WIDEN(T) hipLaunchKernelGGL((k_widen<T>), dim3(g), dim3(kIngestBlock), 0, s, static_cast<const T *>(src), n, table, table_len, dst, err)
  if (bits == 8) { if (is_signed) TAD_WIDEN(int8_t); else TAD_WIDEN(uint8_t); }
  else if (bits == 16) { if (is_signed) TAD_WIDEN(int16_t); else TAD_WIDEN(uint16_t); }
  else if (bits == 32) { if (is_signed) TAD_WIDEN(int32_t); else TAD_WIDEN(uint32_t); }
  else { if (is_signed) TAD_WIDEN(long long); else TAD_WIDEN(unsigned long long); }
#undef TAD_WIDEN
}

void launch_mask_rows(hipStream_t s, uint64_t n, int n_terms, const long long *const *codes, const uint8_t *const *masks, const uint64_t *mask_len, bool combine,
                      uint8_t *keep, unsigned int *err) {
  if (n == 0) return;
  MaskArgs A{};
  A.n_terms = n_terms;
  for (int t = 0; t < n_terms; ++t) { A.codes[t] = codes[t]; A.masks[t] = masks[t]; A.len[t] = mask_len[t]; }
  hipLaunchKernelGGL(k_mask_rows, dim3(grid_for(n)), dim3(kIngestBlock), 0, s, A, n, combine ? 1 : 0, keep, err);
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_ingest() { return reinterpret_cast<const void *>(&k_mask_rows); }

}  // namespace tad
