// tad_capi_ingest.cpp — the ingest entry points of include/tad.h: rows bucketed by owner for the all-to-all(v),
// key tuples -> dense ids, Arrow string columns -> dictionary codes, Arrow buffers -> 8-byte device columns, row masks, the synthetic table.
#include "tad_engine.h"

using namespace tad;
using namespace tadh;

extern "C" {

int tad_shard_rows(tad_engine *eng, const tad_columns *cols, uint32_t world, uint64_t *out_key_id, int64_t *out_flow_end_s,
                   uint64_t *out_value, uint64_t *counts) {
  if (!eng) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_shard_rows: engine is NULL");
  if (!cols || !counts || !shard_world_ok(world)) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_shard_rows: bad arguments (1 <= world <= 1024)");
  if (cols->memory != TAD_MEM_DEVICE || cols->key_id2 || cols->flow_start_s)
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_shard_rows: device columns with one key per row only");
  const uint64_t n = cols->n_rows;
  if (n && (!cols->key_id || !cols->flow_end_s || !cols->value || !out_key_id || !out_flow_end_s || !out_value))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_shard_rows: key_id, flow_end_s, value and the three outputs are required");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_shard_rows: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t s = e->stream;
  int rc;
  if ((rc = ensure(e, e->scan_scratch, (size_t)world * 16)) != TAD_OK) return rc;
  unsigned long long *d_counts = static_cast<unsigned long long *>(e->scan_scratch.p);
  unsigned long long *d_cursor = d_counts + world;
  HIP_TRY(e, hipMemsetAsync(d_counts, 0, (size_t)world * 8, s));
  launch_shard_count(s, cols->key_id, n, world, d_counts);
  std::vector<unsigned long long> h(world), off(world);
  HIP_TRY(e, hipMemcpyAsync(h.data(), d_counts, (size_t)world * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  unsigned long long run = 0;
  for (uint32_t d = 0; d < world; ++d) { off[d] = run; run += h[d]; counts[d] = h[d]; }
  HIP_TRY(e, hipMemcpyAsync(d_cursor, off.data(), (size_t)world * 8, hipMemcpyHostToDevice, s));
  launch_shard_scatter(s, cols->key_id, cols->flow_end_s, cols->value, n, world, d_cursor, out_key_id, out_flow_end_s, out_value);
  HIP_TRY(e, hipStreamSynchronize(s));   // `off` goes out of scope; the caller may hand the buffers to a collective on another stream
  HIP_TRY(e, hipGetLastError());
  return TAD_OK;
}

}  // extern "C"

// tad_factorize / tad_factorize_hist
static int factorize_impl(tad_engine *eng, const tad_key_columns *kc, uint64_t *key_id, uint64_t *key_id2, uint64_t *first_row, uint64_t first_row_cap,
                          uint64_t *num_keys, tad_key_hist *hist) {
  if (!eng) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_factorize: engine is NULL");
  if (!kc || !num_keys || kc->n_cols < 1 || kc->n_cols > kFzMaxCols || !kc->cols_a || (kc->n_rows && !key_id) || (kc->cols_b && kc->n_rows && !key_id2) ||
      (first_row_cap && !first_row))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_factorize: bad arguments (1..%d key columns, key_id / key_id2 / first_row buffers)", kFzMaxCols);
  const uint64_t n = kc->n_rows;
  const uint32_t sides = kc->cols_b ? 2 : 1;
  *num_keys = 0;
  uint32_t *hist_bins = nullptr;
  PartPlan hpl{};
  if (hist != nullptr) {
    hist_bins = hist->bins;
    hist->n_rows = hist->num_keys = hist->chunk_rows = 0;
    hist->workgroups = hist->nbins = hist->shift = hist->sides = 0;
    // pass B's row chunking does not depend on the key count (part_plan_bins): the kernel needs it before the count exists on the host
    if (hist_bins == nullptr || !part_plan_bins(n, 1, sides == 2, &hpl)) hist_bins = nullptr;
  }
  if (n == 0) return TAD_OK;
  if (n * sides >= 0xFFFFFFFFull) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_factorize: %llu virtual rows do not fit 32-bit row indices", (unsigned long long)(n * sides));
  for (int c = 0; c < kc->n_cols; ++c)
    if (!kc->cols_a[c] || (kc->cols_b && !kc->cols_b[c])) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_factorize: key column %d is NULL", c);
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_factorize: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t s = e->stream;
  const bool host = kc->memory == TAD_MEM_HOST;
  // host inputs are staged behind the table block in the same scratch: columns, masks, outputs
  const size_t col_bytes = (n * 8 + 255) & ~(size_t)255, mask_bytes = (n + 255) & ~(size_t)255, fr_bytes = (first_row_cap * 8 + 255) & ~(size_t)255;
  const size_t stage = host ? (size_t)kc->n_cols * sides * col_bytes + 2 * mask_bytes + sides * col_bytes + fr_bytes : 0;
  // the table starts small and grows when the device says so (tad_factorize.hip): 2^20 -> 2^24 -> 2 n slots
  for (uint64_t slots = factorize_first_slots(n * sides);;) {
    const size_t tb = factorize_temp_bytes(n * sides, slots);
    if (tb + stage + 64 > e->ws_limit)
      return fail(e, TAD_ERR_GRID_TOO_LARGE, "tad_factorize needs %llu bytes of scratch > workspace limit %llu", (unsigned long long)(tb + stage), (unsigned long long)e->ws_limit);
    int rc;
    if ((rc = ensure(e, e->sp_temp, tb + stage + 64)) != TAD_OK) return rc;
    unsigned char *base = static_cast<unsigned char *>(e->sp_temp.p);
    unsigned long long *nk_dev = reinterpret_cast<unsigned long long *>(base + tb);
    const long long *ca[kFzMaxCols] = {}, *cb[kFzMaxCols] = {};
    const uint8_t *ka = kc->keep_a, *kb = kc->keep_b;
    uint64_t *d_key = key_id, *d_key2 = key_id2, *d_fr = first_row;
    if (host) {
      unsigned char *p = base + tb + 64;
      for (int c = 0; c < kc->n_cols; ++c) {
        HIP_TRY(e, hipMemcpyAsync(p, kc->cols_a[c], n * 8, hipMemcpyHostToDevice, s)); ca[c] = reinterpret_cast<const long long *>(p); p += col_bytes;
        if (sides == 2) { HIP_TRY(e, hipMemcpyAsync(p, kc->cols_b[c], n * 8, hipMemcpyHostToDevice, s)); cb[c] = reinterpret_cast<const long long *>(p); p += col_bytes; }
      }
      if (ka) { HIP_TRY(e, hipMemcpyAsync(p, ka, n, hipMemcpyHostToDevice, s)); ka = p; }
      p += mask_bytes;
      if (kb) { HIP_TRY(e, hipMemcpyAsync(p, kb, n, hipMemcpyHostToDevice, s)); kb = p; }
      p += mask_bytes;
      d_key = reinterpret_cast<uint64_t *>(p); p += col_bytes;
      if (sides == 2) { d_key2 = reinterpret_cast<uint64_t *>(p); p += col_bytes; }
      d_fr = reinterpret_cast<uint64_t *>(p);
    } else {
      for (int c = 0; c < kc->n_cols; ++c) { ca[c] = reinterpret_cast<const long long *>(kc->cols_a[c]); if (sides == 2) cb[c] = reinterpret_cast<const long long *>(kc->cols_b[c]); }
    }
    uint32_t *flags_dev = nullptr;
    launch_factorize(s, ca, ka, sides == 2 ? cb : nullptr, kb, n, kc->n_cols, slots, base, d_key, d_key2, d_fr, first_row_cap, nk_dev, &flags_dev,
                     hist_bins, hpl.G, hpl.chunk);
    unsigned long long nk = 0;
    uint32_t flags = 0;
    HIP_TRY(e, hipMemcpyAsync(&nk, nk_dev, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipMemcpyAsync(&flags, flags_dev, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    HIP_TRY(e, hipGetLastError());
    if (flags != 0) {      // more keys than this table takes: once more with the next size (a host batch is staged again: the block may have moved)
      const uint64_t next = factorize_next_slots(n * sides, slots);
      if (next == slots) return fail(e, TAD_ERR_HIP, "tad_factorize: the full-size table filled up");
      slots = next;
      continue;
    }
    if (host) {
      HIP_TRY(e, hipMemcpyAsync(key_id, d_key, n * 8, hipMemcpyDeviceToHost, s));
      if (sides == 2) HIP_TRY(e, hipMemcpyAsync(key_id2, d_key2, n * 8, hipMemcpyDeviceToHost, s));
      const uint64_t m = nk < first_row_cap ? nk : first_row_cap;
      if (m) HIP_TRY(e, hipMemcpyAsync(first_row, d_fr, m * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
    }
    *num_keys = nk;
    if (hist_bins != nullptr && nk != 0 && part_plan_bins(n, nk, sides == 2, &hpl)) {   // what the kernel derived from the device-side count
      hist->n_rows = n; hist->num_keys = nk; hist->chunk_rows = hpl.chunk;
      hist->workgroups = (uint32_t)hpl.G; hist->nbins = hpl.nbins; hist->shift = (uint32_t)hpl.shift_bin; hist->sides = sides;
    }
    return TAD_OK;
  }
}

extern "C" {

int tad_factorize(tad_engine *eng, const tad_key_columns *kc, uint64_t *key_id, uint64_t *key_id2, uint64_t *first_row, uint64_t first_row_cap,
                  uint64_t *num_keys) {
  return factorize_impl(eng, kc, key_id, key_id2, first_row, first_row_cap, num_keys, nullptr);
}

int tad_factorize_hist(tad_engine *eng, const tad_key_columns *kc, uint64_t *key_id, uint64_t *key_id2, uint64_t *first_row, uint64_t first_row_cap,
                       uint64_t *num_keys, tad_key_hist *hist) {
  if (eng && !hist) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_factorize_hist: hist is NULL");
  return factorize_impl(eng, kc, key_id, key_id2, first_row, first_row_cap, num_keys, hist);
}

int tad_encode_strings(tad_engine *eng, const tad_string_column *col, int64_t *codes, uint64_t *first_row, uint64_t first_row_cap, uint64_t *num_values) {
  if (!eng) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_encode_strings: engine is NULL");
  if (!col || !num_values || (col->offset_bits != 32 && col->offset_bits != 64) || (col->n_rows && (!col->offsets || !codes)) || (first_row_cap && !first_row) ||
      (col->data_bytes && !col->data))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_encode_strings: bad arguments (offsets of 32 or 64 bits, data, codes / first_row buffers)");
  const uint64_t n = col->n_rows;
  *num_values = 0;
  if (n == 0) return TAD_OK;
  if (n >= 0xFFFFFFFFull) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_encode_strings: %llu rows do not fit 32-bit row indices", (unsigned long long)n);
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_encode_strings: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t s = e->stream;
  const bool host = col->memory == TAD_MEM_HOST;
  const int off64 = col->offset_bits == 64;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  // host inputs are staged behind the table block: offsets, bytes (+ 8 of slack: the last aligned word), validity, codes, first rows
  const size_t off_bytes = up((n + 1) * (off64 ? 8 : 4)), data_bytes = up(col->data_bytes + 8);
  const size_t val_bytes = col->validity ? up((col->validity_offset + n + 7) / 8) : 0, code_bytes = up(n * 8), fr_bytes = up(first_row_cap * 8);
  const size_t stage = host ? off_bytes + data_bytes + val_bytes + code_bytes + fr_bytes : 0;
  for (uint64_t slots = factorize_first_slots(n);;) {
    const size_t tb = factorize_temp_bytes(n, slots);
    if (tb + stage + 64 > e->ws_limit)
      return fail(e, TAD_ERR_GRID_TOO_LARGE, "tad_encode_strings needs %llu bytes of scratch > workspace limit %llu", (unsigned long long)(tb + stage), (unsigned long long)e->ws_limit);
    int rc;
    if ((rc = ensure(e, e->sp_temp, tb + stage + 64)) != TAD_OK) return rc;
    unsigned char *base = static_cast<unsigned char *>(e->sp_temp.p);
    unsigned long long *nv_dev = reinterpret_cast<unsigned long long *>(base + tb);
    const void *d_off = col->offsets;
    const uint8_t *d_data = col->data, *d_valid = col->validity;
    long long *d_codes = reinterpret_cast<long long *>(codes);
    uint64_t *d_fr = first_row;
    if (host) {
      unsigned char *p = base + tb + 64;
      HIP_TRY(e, hipMemcpyAsync(p, col->offsets, (n + 1) * (off64 ? 8 : 4), hipMemcpyHostToDevice, s)); d_off = p; p += off_bytes;
      if (col->data_bytes) HIP_TRY(e, hipMemcpyAsync(p, col->data, col->data_bytes, hipMemcpyHostToDevice, s));
      d_data = p; p += data_bytes;
      if (col->validity) { HIP_TRY(e, hipMemcpyAsync(p, col->validity, (col->validity_offset + n + 7) / 8, hipMemcpyHostToDevice, s)); d_valid = p; p += val_bytes; }
      d_codes = reinterpret_cast<long long *>(p); p += code_bytes;
      d_fr = reinterpret_cast<uint64_t *>(p);
    }
    uint32_t *flags_dev = nullptr;
    launch_encode_strings(s, d_off, off64, d_data, col->data_bytes, d_valid, col->validity_offset, n, slots, base, d_codes, d_fr, first_row_cap, nv_dev, &flags_dev);
    unsigned long long nv = 0;
    uint32_t flags = 0;
    HIP_TRY(e, hipMemcpyAsync(&nv, nv_dev, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipMemcpyAsync(&flags, flags_dev, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    HIP_TRY(e, hipGetLastError());
    if (flags & 2u) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_encode_strings: offsets decrease or point beyond data_bytes");
    if (flags & 1u) {      // more distinct values than this table takes: once more with the next size (2^20 -> 2^24 -> 2 n slots)
      const uint64_t next = factorize_next_slots(n, slots);
      if (next == slots) return fail(e, TAD_ERR_HIP, "tad_encode_strings: the full-size table filled up");
      slots = next;
      continue;
    }
    if (host) {
      HIP_TRY(e, hipMemcpyAsync(codes, d_codes, n * 8, hipMemcpyDeviceToHost, s));
      const uint64_t m = nv < first_row_cap ? nv : first_row_cap;
      if (m) HIP_TRY(e, hipMemcpyAsync(first_row, d_fr, m * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
    }
    *num_values = nv;
    return TAD_OK;
  }
}

int tad_synth_generate(tad_engine *eng, uint64_t seed, uint64_t first_row, uint64_t n_rows, uint64_t num_keys,
                       uint64_t n_buckets, uint64_t *key_id, int64_t *flow_end_s, uint64_t *value) {
  if (!eng || num_keys == 0 || n_buckets == 0 || (n_rows && (!key_id || !flow_end_s || !value)))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_synth_generate: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_synth_generate: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  launch_synth(e->stream, seed, first_row, n_rows, num_keys, n_buckets, key_id, flow_end_s, value);
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  HIP_TRY(e, hipGetLastError());
  return TAD_OK;
}

// ---- columnar ingest: Arrow buffers in host memory -> 8-byte device columns ----
int tad_widen_column(tad_engine *eng, const void *src, int32_t src_bits, int32_t src_signed, tad_mem src_memory, uint64_t n, const int64_t *table,
                     uint64_t table_len, int64_t *dst) {
  if (!eng) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_widen_column: engine is NULL");
  if ((src_bits != 8 && src_bits != 16 && src_bits != 32 && src_bits != 64) || (n && (!src || !dst)) || (table_len && !table))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_widen_column: bad arguments (src of 8 / 16 / 32 / 64 bits, src / dst buffers, table)");
  if (n == 0) return TAD_OK;
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_widen_column: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t s = e->stream;
  const size_t bytes = (size_t)n * (size_t)(src_bits / 8);
  if (src_memory == TAD_MEM_HOST && src_bits == 64 && table == nullptr) {      // nothing to convert: the copy is the column
    HIP_TRY(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    return TAD_OK;
  }
  int rc;
  if ((rc = ensure(e, e->counters, kTailBytes)) != TAD_OK) return rc;
  unsigned int *err = reinterpret_cast<unsigned int *>(e->counters.p);
  HIP_TRY(e, hipMemsetAsync(err, 0, 4, s));
  const void *d_src = src;
  if (src_memory == TAD_MEM_HOST) {
    if ((rc = ensure(e, e->in_key, bytes)) != TAD_OK) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->in_key.p, src, bytes, hipMemcpyHostToDevice, s));
    d_src = e->in_key.p;
  }
  launch_widen(s, d_src, src_bits, src_signed != 0, n, reinterpret_cast<const long long *>(table), table ? table_len : 0, reinterpret_cast<long long *>(dst), err);
  unsigned int herr = 0;
  HIP_TRY(e, hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  HIP_TRY(e, hipGetLastError());
  if (herr) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_widen_column: an index lies outside the table of %llu entries", (unsigned long long)table_len);
  return TAD_OK;
}

int tad_mask_rows(tad_engine *eng, uint64_t n, int32_t n_terms, const int64_t *const *codes, const uint8_t *const *masks, const uint64_t *mask_len,
                  int32_t combine, uint8_t *keep) {
  if (!eng) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_mask_rows: engine is NULL");
  if (n_terms < 0 || n_terms > kMaskMaxTerms || (n_terms && (!codes || !masks || !mask_len)) || (n && !keep))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_mask_rows: bad arguments (0..%d terms, keep buffer)", kMaskMaxTerms);
  for (int t = 0; t < n_terms; ++t)
    if (n && (!codes[t] || !masks[t])) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_mask_rows: term %d is NULL", t);
  if (n == 0) return TAD_OK;
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_mask_rows: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t s = e->stream;
  int rc;
  if ((rc = ensure(e, e->counters, kTailBytes)) != TAD_OK) return rc;
  unsigned int *err = reinterpret_cast<unsigned int *>(e->counters.p);
  HIP_TRY(e, hipMemsetAsync(err, 0, 4, s));
  launch_mask_rows(s, n, n_terms, reinterpret_cast<const long long *const *>(codes), masks, mask_len, combine != 0, keep, err);
  unsigned int herr = 0;
  HIP_TRY(e, hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  HIP_TRY(e, hipGetLastError());
  if (herr) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_mask_rows: a code lies outside its mask");
  return TAD_OK;
}

}  // extern "C"
