// tad_capi.cpp — the job of include/tad.h: tad_run / tad_aggregate / tad_run_stream on a job context (tad_engine.h).  Replaces one run of
// anomaly_detection() (plugins/anomaly-detection/anomaly_detection.py:647-710): Stage 0 GROUP BY -> per-key sigma -> detector -> compaction.
#include "tad_engine.h"

using namespace tad;
using namespace tadh;

namespace tadh {

// reciprocals of the point counts 1..T for the exact-division FMA sequence (tad_internal.h:div_by_count);
// 1.0 / n on the host is IEEE division = the correctly rounded reciprocal the sequence needs.
int ensure_rcp_table(JobCtx *e, uint64_t T) {
  const uint64_t want = T + 2;
  if (want <= e->rcp_n) return TAD_OK;
  uint64_t cap = want < 1024 ? 1024 : want + want / 4;
  int rc = ensure(e, e->rcp_table, cap * sizeof(double));
  if (rc != TAD_OK) return rc;
  std::vector<double> h(cap);
  h[0] = 0.0;
  for (uint64_t i = 1; i < cap; ++i) h[i] = 1.0 / (double)i;
  HIP_TRY(e, hipMemcpyAsync(e->rcp_table.p, h.data(), cap * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));  // h goes out of scope
  e->rcp_n = cap;
  return TAD_OK;
}

int ensure_key_buffers(JobCtx *e, uint64_t K) {
  int rc;
  const uint64_t k = K ? K : 1;
  if ((rc = ensure(e, e->sigma, k * sizeof(double))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->n_pts, k * sizeof(uint32_t))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->n_anom, k * sizeof(uint32_t))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->off, (k + 1) * sizeof(unsigned long long))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->scan_scratch, scan_scratch_elems(k) * sizeof(unsigned long long))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->key_mean, k * sizeof(double))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->key_m2, k * sizeof(double))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->counters, kTailBytes)) != TAD_OK) return rc;
  return TAD_OK;
}

// Runs sigma + detector + scan on grid g.  On return *rows = number of rows emit will write.
// stats_done: Stage 0 v2's tile pass already produced sigma / n_pts / (EWMA) n_anom / moments inputs / counters.
}  // namespace tadh

namespace {

int detect_and_count(JobCtx *e, Grid g, JobParams &jp, DevCounters *ctr, uint64_t *rows, bool stats_done = false) {
  hipStream_t s = e->stream;
  int rc;
  if ((rc = ensure_key_buffers(e, g.K)) != TAD_OK) return rc;
  if ((rc = ensure_rcp_table(e, g.T)) != TAD_OK) return rc;
  double *sigma = static_cast<double *>(e->sigma.p);
  uint32_t *n_pts = static_cast<uint32_t *>(e->n_pts.p);
  uint32_t *n_anom = static_cast<uint32_t *>(e->n_anom.p);
  unsigned long long *off = static_cast<unsigned long long *>(e->off.p);

  const bool ewma = jp.algo == TAD_ALGO_EWMA;
  const bool drop = jp.algo == TAD_ALGO_DROP;
  // DBSCAN ignores sigma for its verdicts (anomaly_detection.py:325-349) — it is only an output column of the anomalous
  // rows.  The tile kernel then delivers the per-key counts / moments itself and k_emit streams stddev_samp for the keys
  // that have rows: no separate per-key walk over the whole grid (C4: -0.44 ms).  emit-all jobs keep the general path.
  const bool db_fused = jp.algo == TAD_ALGO_DBSCAN && !jp.all_points && !stats_done && dbscan_uses_list(g);
  jp.lazy_sigma = db_fused;
  if (drop) {   // mean / std / verdicts / counters in one kernel (pandas' pairwise arithmetic, not Spark's streaming update)
    if ((rc = ensure(e, e->calc, (g.K * g.T ? g.K * g.T : 1) * sizeof(double))) != TAD_OK) return rc;
    launch_drop(s, g, jp.drop_nsigma, jp.drop_min_samples, static_cast<double *>(e->calc.p), sigma, n_pts,
                static_cast<double *>(e->key_mean.p), static_cast<double *>(e->key_m2.p), ctr);
  } else if (!stats_done && !db_fused)
    launch_key_sigma(s, g, jp.alpha, ewma && !jp.all_points, static_cast<const double *>(e->rcp_table.p), sigma, n_pts, n_anom, ctr, static_cast<double *>(e->key_mean.p),
                     static_cast<double *>(e->key_m2.p));
  if (jp.algo == TAD_ALGO_DBSCAN) {
    if ((rc = ensure(e, e->aux, dbscan_scratch_bytes(g))) != TAD_OK) return rc;
    if (dbscan_uses_list(g)) {
      DbscanStats dst{nullptr, nullptr, nullptr, nullptr};
      if (db_fused) dst = DbscanStats{n_pts, n_anom, static_cast<double *>(e->key_mean.p), static_cast<double *>(e->key_m2.p)};
      if (launch_dbscan(s, g, jp.eps, jp.min_samples, e->aux.p, dst, jp.settled && db_fused) != 0)
        return fail(e, TAD_ERR_HIP, "DBSCAN launch failed");
    } else {
      return fail(e, TAD_ERR_GRID_TOO_LARGE, "DBSCAN: series of %llu buckets are not supported", (unsigned long long)g.T);
    }
  } else if (jp.algo == TAD_ALGO_ARIMA) {
    if ((rc = ensure(e, e->calc, g.K * g.T * sizeof(double))) != TAD_OK) return rc;
    const size_t wsb = arima_workspace_bytes(g);
    if ((rc = ensure(e, e->aux, wsb)) != TAD_OK) return rc;
    // The fit yields to whole-CU jobs of other contexts (PauseHold): its wavefronts suspend their fits while the engine's pause word is raised
    // and the kernel is relaunched here — after the word has cleared, or after 2 ms at the latest, so that a steady stream of short jobs
    // time-slices with the fit instead of starving it.  This job's own claim is dropped for the duration (it would pause itself) and taken back for the emit.
    const bool held = e->hold && e->hold->held;
    if (held) e->hold->release();
    const unsigned int *yielded_dev = nullptr;
    if (launch_arima(s, g, sigma, n_pts, jp.maxiter, static_cast<double *>(e->calc.p), ctr, e->aux.p, wsb, e->eng->pause_dev, &yielded_dev) != 0)
      return fail(e, TAD_ERR_HIP, "ARIMA launch failed");
    while (yielded_dev != nullptr && e->eng->pause_dev != nullptr) {
      unsigned int y = 0;
      HIP_TRY(e, hipMemcpyAsync(&y, yielded_dev, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
      if (y == 0) break;
      const auto t0 = std::chrono::steady_clock::now();
      while (__atomic_load_n(&e->eng->pause_count, __ATOMIC_ACQUIRE) != 0 && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2))
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      // still raised after 2 ms (short jobs arrive back to back): this launch runs 24 optimiser cycles (~1 ms) before it looks at the word
      const uint32_t grace = __atomic_load_n(&e->eng->pause_count, __ATOMIC_ACQUIRE) != 0 ? 24u : 0u;
      if (launch_arima_fit(s, g, sigma, n_pts, jp.maxiter, static_cast<double *>(e->calc.p), ctr, e->aux.p, e->eng->pause_dev, &yielded_dev, grace) != 0)
        return fail(e, TAD_ERR_HIP, "ARIMA launch failed");
      e->arima_relaunches++;
    }
    if (held) e->hold->acquire();
  }
  const uint32_t *cnt = n_anom;
  if (jp.all_points && jp.algo != TAD_ALGO_ARIMA && !drop) cnt = n_pts;
  else if (db_fused) {}                                                             // the tile kernel counted the noise points
  else if (!ewma || jp.all_points) launch_count_flags(s, g, jp.all_points, n_anom);  // ARIMA / DROP all_points: skips no-result keys
  launch_scan_moments(s, cnt, off, g.K, static_cast<unsigned long long *>(e->scan_scratch.p), dev_total(e), n_pts,
                      static_cast<const double *>(e->key_mean.p), static_cast<const double *>(e->key_m2.p), dev_moments(e), db_fused ? ctr : nullptr);
  HIP_TRY(e, hipMemcpyAsync(e->tail_host, e->counters.p, kTailBytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  HIP_TRY(e, hipGetLastError());
  *rows = *e->total_host;
  return TAD_OK;
}

}  // namespace

namespace tadh {

void emit_rows(JobCtx *e, Grid g, Lattice L, const JobParams &jp, OutRows out, uint64_t rows) {
  const int kind = jp.algo == TAD_ALGO_EWMA ? 0 : (jp.algo == TAD_ALGO_ARIMA ? 1 : (jp.algo == TAD_ALGO_DROP ? 3 : (jp.lazy_sigma ? 4 : 2)));
  // DBSCAN job: only keys of the detector's work list (still in e->aux) can have rows
  if (kind == 4 && !jp.all_points &&
      launch_emit_dbscan_list(e->stream, g, L, e->aux.p, static_cast<const uint32_t *>(e->n_anom.p), static_cast<const unsigned long long *>(e->off.p), out))
    return;
  launch_emit(e->stream, g, L, kind, jp.all_points, jp.alpha, static_cast<const double *>(e->sigma.p),
              static_cast<const uint32_t *>(e->n_pts.p), static_cast<const double *>(kind == 3 ? e->key_mean.p : e->calc.p),
              static_cast<const unsigned long long *>(e->off.p), out, rows, e->plan.ewma_emit, e->plan.ewma_emit_rows);
}

}  // namespace tadh

namespace {

int make_result(JobCtx *e, uint64_t rows, bool with_anomaly, tad_mem out_memory, ResultPriv **out, OutRows *dev_rows,
                ResultBlock *dev_block) {
  ResultPriv *rp = new (std::nothrow) ResultPriv();
  if (!rp) return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory");
  memset(rp, 0, sizeof *rp);
  const size_t bytes = result_bytes(rows, with_anomaly);
  int rc = alloc_device_block(e, bytes, dev_block);
  if (rc != TAD_OK) { delete rp; return rc; }
  carve(dev_block->base, rows, with_anomaly, dev_rows);
  rp->pub.n_rows = rows;
  rp->pub.memory = out_memory;
  *out = rp;
  return TAD_OK;
}

// after emit: hand the device block to the caller, or copy it to a host block
int finish_result(JobCtx *e, ResultPriv *rp, uint64_t rows, bool with_anomaly, ResultBlock dev_block, OutRows dev_rows) {
  if (rp->pub.memory == TAD_MEM_DEVICE) {
    rp->block = dev_block.base;
    rp->block_cap = dev_block.cap;
    rp->pub.key_id = reinterpret_cast<uint64_t *>(dev_rows.key_id);
    rp->pub.flow_end_s = reinterpret_cast<int64_t *>(dev_rows.flow_end_s);
    rp->pub.throughput = dev_rows.throughput;
    rp->pub.algo_calc = dev_rows.algo_calc;
    rp->pub.stddev = dev_rows.stddev;
    rp->pub.anomaly = dev_rows.anomaly;
    return TAD_OK;
  }
  const size_t bytes = result_bytes(rows, with_anomaly);
  void *h = malloc(bytes);
  if (!h) { release_block(e, dev_block.base, dev_block.cap); return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory for %zu result bytes", bytes); }
  hipError_t r = hipMemcpyAsync(h, dev_block.base, bytes, hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
  release_block(e, dev_block.base, dev_block.cap);
  if (r != hipSuccess) { free(h); return fail(e, TAD_ERR_HIP, "result copy failed: %s", hipGetErrorString(r)); }
  OutRows ho;
  carve(h, rows, with_anomaly, &ho);
  rp->block = h;
  rp->block_cap = bytes;
  rp->pub.key_id = reinterpret_cast<uint64_t *>(ho.key_id);
  rp->pub.flow_end_s = reinterpret_cast<int64_t *>(ho.flow_end_s);
  rp->pub.throughput = ho.throughput;
  rp->pub.algo_calc = ho.algo_calc;
  rp->pub.stddev = ho.stddev;
  rp->pub.anomaly = ho.anomaly;
  return TAD_OK;
}

int stage_column(JobCtx *e, DevBuf &buf, const void *src, uint64_t n, tad_mem mem, const void **dev) {
  if (!src) { *dev = nullptr; return TAD_OK; }
  if (mem == TAD_MEM_DEVICE) { *dev = src; return TAD_OK; }
  int rc = ensure(e, buf, n * 8);
  if (rc != TAD_OK) return rc;
  HIP_TRY(e, hipMemcpyAsync(buf.p, src, n * 8, hipMemcpyHostToDevice, e->stream));
  *dev = buf.p;
  return TAD_OK;
}

size_t state_bytes(uint64_t K) { return (size_t)K * (4 + 8 * 4 + 1) + 64; }

StreamState state_view(const tad_state *st, int which) {
  unsigned char *b = static_cast<unsigned char *>(st->block[which]);
  StreamState v;
  v.avg = reinterpret_cast<double *>(b);
  v.m2 = v.avg + st->K;
  v.ewma = v.m2 + st->K;
  v.last_t = reinterpret_cast<long long *>(v.ewma + st->K);
  v.n = reinterpret_cast<uint32_t *>(v.last_t + st->K);
  v.seen = reinterpret_cast<unsigned char *>(v.n + st->K);
  return v;
}

int run_job_locked(JobCtx *e, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_result **out, tad_points **points_out,
                   tad_state *stream, int depth);
int run_sparse_classes(JobCtx *e, const tad_job *job, const JobParams &jp, bool op_max, uint64_t n_rows_in, uint64_t rows_used, uint64_t K, Lattice L,
                       uint64_t P, uint32_t tmax, tad_mem out_memory, tad_result **out);
int sparse_points_direct(JobCtx *e, uint64_t n_rows_in, uint64_t rows_used, Lattice L, uint64_t P, DevCounters *ctr, tad_mem out_memory,
                         tad_points **points_out);

// The job (points_out == nullptr), Stage 0 alone (points_out != nullptr), or one streaming batch (stream != nullptr).
int run_job(tad_engine *eng, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_result **out, tad_points **points_out,
            tad_state *stream = nullptr) {
  tad_engine *e = eng;
  const bool points_mode = points_out != nullptr;
  if (stream && e && job && cols) {
    if (job->algo != TAD_ALGO_EWMA) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run_stream: only the EWMA detector has a streaming form");
    // k_stream writes the candidate state for keys < cols->num_keys and the double buffer flips as a whole: a batch
    // that declares fewer keys than the state holds would drop the others' state
    if (cols->num_keys != stream->K) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run_stream: batch declares %llu keys, the state holds %llu (they must be equal)",
                                                 (unsigned long long)cols->num_keys, (unsigned long long)stream->K);
  }
  if (!e) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_run: engine is NULL");
  if (!job || !cols || (!out && !points_out)) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run: job, cols and out must not be NULL");
  if (out) *out = nullptr;
  if (points_out) *points_out = nullptr;
  if (!points_mode && job->algo != TAD_ALGO_EWMA && job->algo != TAD_ALGO_ARIMA && job->algo != TAD_ALGO_DBSCAN && job->algo != TAD_ALGO_DROP)
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "invalid request: Throughput Anomaly Detector algorithm type should be 'EWMA' or 'ARIMA' or 'DBSCAN'");
  if (job->agg_flow < TAD_AGG_NONE || job->agg_flow > TAD_AGG_EXTERNAL)
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "invalid request: Throughput Anomaly Detector aggregated flow type should be 'pod' or 'external' or 'svc'");
  if (job->start_time != 0 && job->end_time != 0 && job->end_time <= job->start_time)
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "invalid request: EndInterval should be after StartInterval");
  if (cols->n_rows > 0 && (!cols->key_id || !cols->flow_end_s || !cols->value))
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run: key_id, flow_end_s and value columns are required");
  if (cols->n_rows > 0 && cols->num_keys == 0)
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run: num_keys is 0 but there are rows");
  if (cols->n_buckets > 0 && cols->step < 1)
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run: lattice hint needs step >= 1");
  if (job->ewma_alpha < 0.0 || job->ewma_alpha > 1.0 || job->dbscan_eps < 0.0 || job->dbscan_min_samples < 0 || job->arima_maxiter < 0 ||
      job->drop_nsigma < 0.0 || job->drop_min_samples < 0)
    return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run: detector parameter out of range");

  // one job context = one job in flight; a streaming state is advanced by one batch at a time
  std::unique_lock<std::mutex> state_lk;
  if (stream) state_lk = std::unique_lock<std::mutex>(stream->mu);
  Lease lease(eng, job->id, !points_mode && job->algo == TAD_ALGO_ARIMA);
  if (!lease.c) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_run: no job context available");
  PauseHold hold(eng);     // (declared after the lease: dropped before the context goes back to the pool)
  lease.c->hold = &hold;
  return run_job_locked(lease.c, job, cols, out_memory, out, points_out, stream, 0);
}

// the validated job on the context the caller holds; depth > 0: a length class of a skewed sparse table run as a job of its own
int run_job_locked(JobCtx *e, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_result **out, tad_points **points_out,
                   tad_state *stream, int depth) {
  const bool points_mode = points_out != nullptr;
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t s = e->stream;
  if (depth == 0) {
    e->done.store(0);
    e->total.store(4);
    e->arima_relaunches = 0;
  }

  JobParams jp;
  jp.algo = job->algo;
  jp.alpha = job->ewma_alpha == 0.0 ? 0.5 : job->ewma_alpha;
  jp.eps = job->dbscan_eps == 0.0 ? 250000000.0 : job->dbscan_eps;
  jp.min_samples = job->dbscan_min_samples == 0 ? 4 : job->dbscan_min_samples;
  jp.maxiter = job->arima_maxiter == 0 ? 50 : job->arima_maxiter;
  jp.drop_nsigma = job->drop_nsigma == 0.0 ? 3.0 : job->drop_nsigma;
  jp.drop_min_samples = job->drop_min_samples == 0 ? 3 : job->drop_min_samples;
  jp.all_points = (job->flags & TAD_FLAG_EMIT_ALL_POINTS) != 0;
  const bool op_max = job->value_op == TAD_OP_MAX || (job->value_op == TAD_OP_AUTO && job->agg_flow == TAD_AGG_NONE);
  const uint64_t n = cols->n_rows;
  const uint64_t K = cols->num_keys;
  RowFilter rf{job->start_time, job->end_time};

  int rc;
  const void *d_key, *d_key2, *d_te, *d_ts, *d_val;
  if ((rc = stage_column(e, e->in_key, cols->key_id, n, cols->memory, &d_key)) != TAD_OK) return rc;
  if ((rc = stage_column(e, e->in_key2, cols->key_id2, n, cols->memory, &d_key2)) != TAD_OK) return rc;
  if ((rc = stage_column(e, e->in_te, cols->flow_end_s, n, cols->memory, &d_te)) != TAD_OK) return rc;
  if ((rc = stage_column(e, e->in_ts, cols->flow_start_s, n, cols->memory, &d_ts)) != TAD_OK) return rc;
  if ((rc = stage_column(e, e->in_val, cols->value, n, cols->memory, &d_val)) != TAD_OK) return rc;

  if ((rc = ensure(e, e->counters, kTailBytes)) != TAD_OK) return rc;
  DevCounters *ctr = static_cast<DevCounters *>(e->counters.p);

  HIP_TRY(e, hipEventRecord(e->ev[0], s));
  if (depth == 0) HIP_TRY(e, hipEventRecord(e->ev[6], s));   // (class jobs of a skewed sparse table re-record ev[0..5])
  // ---- time lattice ----
  // lat_mode 0: the caller's hint; 1: derived — v2 samples the gcd (pass A) and pass B verifies every row, v1 derives it
  // exactly; 2: exact derivation (k_meta).  A row off the lattice (wrong hint / sample missed a residue) moves to the next mode.
  int lat_mode = cols->n_buckets > 0 ? 0 : 1;
  Lattice L = make_lattice(cols->t0, lat_mode == 0 ? cols->step : 1, cols->n_buckets);
  bool empty = (n == 0 || K == 0);
  // Stage 0 strategy: v2 (partition + LDS tiles) for big batches, v1 (direct atomics) otherwise / as fallback.
  const tad_plan plan = e->plan;   // (the engine's plan when the job was admitted)
  const bool force_v1 = plan.stage0 == 1;
  const bool force_v2 = plan.stage0 == 2;
  const bool has2 = cols->key_id2 != nullptr;
  bool force_v1_retry = false;
  bool force_wide_tiles = plan.tile_cells == 1;   // set when the overflow list filled up under 32-bit tile cells (many values >= 2^32 - 1): 8-byte cells next
  // pass A may histogram a SAMPLE of the rows (1/16 of the key column, plus the chunk ends, instead of all of it): pass B's regions are then sized from
  // the estimate with 6 sigma of slack; a region that still turns out too small (keys arriving in bursts the sample missed)
  // raises DEV_ERR_REGION_FULL and the job is redone with the exact histogram.  tad_plan.histogram = 1 disables it.
  bool force_exact_hist = plan.histogram == 1;
  bool learnt_exact_hist = false, probing_sampled_hist = false;   // (JobCtx::Learnt: the exact histogram on the last job's word / the sample on probation)
  bool kh_rejected = false;   // the caller's key-bin histogram did not describe the batch (a region overflowed): the job counts for itself
  bool sparse_lsd = plan.sparse_sort == 1;   // set when the partition + LDS-sort form of the sparse Stage 0 met a heavy key bin or a value too wide for its records
  // what the context's last job learnt about a table of this shape: skip the attempt that is known to fail
  {
    const JobCtx::Learnt &lt = e->learnt;
    if (depth == 0 && lt.valid && lt.n == n && lt.K == K && lt.has2 == has2 && lt.algo == (int)job->algo && lt.op == (int)op_max) {
      if (lt.exact_hist) {
        if (lt.exact_uses >= lt.exact_backoff) probing_sampled_hist = true;   // time to try the sample again
        else { force_exact_hist = true; learnt_exact_hist = true; }
      }
      if (lt.wide_tiles) force_wide_tiles = true;
    }
  }
  // retries: wrong hint -> derive (0 -> 1); sampled lattice too coarse / saw no live row -> exact (1 -> 2); overflow list
  // full -> Stage 0 v1.  Each transition happens at most once, so 8 attempts cover every path.
  for (int attempt = 0; attempt < 11; ++attempt) {
    const bool hinted = lat_mode == 0;
    HIP_TRY(e, hipMemsetAsync(ctr, 0, kTailMoments, s));    // counters, row total, overflow-list count
    jp.settled = false;
    bool narrow_tiles = false;
    PartPlan pl{};
    bool v2 = !empty && !force_v1 && !force_v1_retry && (force_v2 || n >= (1ull << 22)) && part_plan_bins(n, K, has2, &pl);
    if (v2 && e->hold) e->hold->acquire();   // pass B / pass C workgroups need whole CUs: ARIMA fits of other jobs in flight make room (PauseHold)
    if ((rc = ensure(e, e->meta, sizeof(MetaPartial) * kMetaBlocks)) != TAD_OK) return rc;
    int meta_blocks = 0;
    bool hist_sampled = false;
    // The caller's key-bin histogram (tad_factorize_hist's by-product): pass A then only samples the time lattice and pass B's regions are
    // sized EXACTLY from the caller's counts.  Taken when it provably describes this batch and this job: same rows, keys, sides and row
    // chunking, no time-window filter (the histogram counted every kept row), the lattice still derived from a sample (lat_mode 1 or a hint).
    const tad_key_hist *kh = cols->key_hist;
    bool use_kh = v2 && depth == 0 && !kh_rejected && kh != nullptr && kh->bins != nullptr && lat_mode != 2 && kh->n_rows == n && kh->num_keys == K &&
                  kh->sides == (has2 ? 2u : 1u) && kh->workgroups == (uint32_t)pl.G && kh->nbins == pl.nbins && kh->shift == (uint32_t)pl.shift_bin &&
                  kh->chunk_rows == pl.chunk && rf.end_time == 0 && !(d_ts != nullptr && rf.start_time != 0);
    const uint32_t *binhist = nullptr;
    if (v2) {
      // pass A: lattice partials + per-workgroup key-bin histogram in one read of the key/time columns
      if ((rc = ensure(e, e->binhist, (size_t)pl.G * pl.nbins * 4)) != TAD_OK) return rc;
      hist_sampled = launch_meta_hist(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts, n, K, rf,
                                      pl, static_cast<MetaPartial *>(e->meta.p), static_cast<uint32_t *>(e->binhist.p), ctr,
                                      // small regions (many keys: C4 has ~50 records per workgroup and 128-key block) make pass C's walk
                                      // over the regions cost more than the sampled pass A saves: sample only when a region of a
                                      // 128-key block is expected to hold a few hundred records
                                      // (lat_mode 2 re-derives the lattice with k_meta, which reuses the partials buffer the sampling ratios live in)
                                      use_kh ||       // (the sampled pass: its histogram lands in e->binhist and is not used)
                                      (!force_exact_hist && lat_mode != 2 && sampled_slots_bound(n * (has2 ? 2 : 1), pl) < (1ull << 32) &&
                                          (plan.histogram == 2 ||      // (A/B: sampled wherever it is possible at all)
                                           n * (has2 ? 2 : 1) / ((uint64_t)pl.G * ((K >> kSampleBlockShift) ? (K >> kSampleBlockShift) : 1)) >= 384)));
      meta_blocks = pl.G;
      binhist = static_cast<const uint32_t *>(e->binhist.p);
      if (use_kh && hist_sampled) { binhist = kh->bins; hist_sampled = false; }   // exact counts, from the caller
      else use_kh = false;          // (pass A could not sample — unaligned columns — and counted every row itself)
    }
    if (!hinted && !empty && (!v2 || lat_mode == 2)) {
      meta_blocks = (int)((n + 255) / 256);
      if (meta_blocks > kMetaBlocks) meta_blocks = kMetaBlocks;
      launch_meta(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts, n, rf,
                  static_cast<MetaPartial *>(e->meta.p), meta_blocks);
    }
    if (!hinted && !empty) {
      HIP_TRY(e, hipMemcpyAsync(e->meta_host, e->meta.p, sizeof(MetaPartial) * meta_blocks, hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
      int64_t tmin = 0, tmax = 0, tref = 0;
      uint64_t g = 0, used = 0;
      for (int b = 0; b < meta_blocks; ++b) {
        const MetaPartial &p = e->meta_host[b];
        if (p.used == 0) continue;
        if (used == 0) { tmin = p.tmin; tmax = p.tmax; tref = p.tref; g = p.g; }
        else {
          if (p.tmin < tmin) tmin = p.tmin;
          if (p.tmax > tmax) tmax = p.tmax;
          const uint64_t d = p.tref >= tref ? (uint64_t)p.tref - (uint64_t)tref : (uint64_t)tref - (uint64_t)p.tref;
          g = host_gcd(host_gcd(g, p.g), d);
        }
        used += p.used;
      }
      if (used == 0 && v2 && lat_mode == 1) {
        // pass A only SAMPLES the time column: every live row (not TAD_KEY_SKIP, inside the time window) may sit in an
        // unsampled stretch of a big, mostly filtered table.  "No live row" is only believed from the exact pass.
        lat_mode = 2;
        continue;
      }
      if (used == 0) { empty = true; }
      else {
        const uint64_t span = (uint64_t)tmax - (uint64_t)tmin;
        // the lattice must contain tmin and tmax whatever the sample saw
        const uint64_t step = host_gcd(host_gcd(g, span), (uint64_t)tref - (uint64_t)tmin);
        L = make_lattice(tmin, (int64_t)(step == 0 ? 1 : step), span / (step == 0 ? 1 : step) + 1);
      }
    }
    HIP_TRY(e, hipEventRecord(e->ev[1], s));
    if (depth == 0) e->done.store(1);
    if (empty) { L = make_lattice(0, 1, 0); v2 = false; }

    // ---- Stage 0: GROUP BY (key, flowEndSeconds) into the time-major point grid ----
    uint64_t cells = empty ? 0 : K * L.nb;
    const bool cells_overflow = !empty && L.nb != 0 && cells / L.nb != K;
    // (ARIMA: predictions + 60 B per cell of workspace, arima_workspace_bytes; DROP: one double per cell)
    uint64_t need = cells * 9 + (jp.algo == TAD_ALGO_ARIMA ? cells * 80 + (1ull << 22) : (jp.algo == TAD_ALGO_DROP ? cells * 8 : 0));
    // Sparse tables (few points per key on a fine lattice: second-resolution timestamps, per-connection keys): the dense
    // K x T grid would be mostly empty or not fit at all — sort the rows by (key, time) instead and lay each key's points
    // out by rank (tad_sparse.hip).  Chosen when the rows could fill at most 1/8 of a large grid, or the grid does not fit.
    const uint64_t slots_all = n * (has2 ? 2 : 1);
    // (first[], len[] and the class offsets are 32-bit indices into the sorted point list: 2^32 slots and beyond stay dense or fail cleanly)
    bool sparse = !empty && !stream && K <= 0xFFFFFFFFull && slots_all < (1ull << 32) &&
                  (plan.sparse == 2 ||
                   (plan.sparse != 1 && (cells_overflow || need > e->ws_limit || (cells >= (1ull << 24) && slots_all < cells / 8))));
    Grid sparse_grid{};
    bool sp_part = false;
    if (sparse && use_kh) {
      // the sparse sort plans LDS rounds of exactly known sizes from the histogram (k_ss_plan): only pass A's own count is trusted with that
      kh_rejected = true;
      continue;
    }
    if (sparse) {
      // Big sparse tables (pass A ran with its key-bin histogram): the dense path's partition pass brings every key block's rows together as
      // 8-byte records, a workgroup per key sub-range sorts them in LDS (tad_sparse.hip: launch_sparse_sort) — the columns are read once and
      // the records move through HBM once, where the LSD sort moves 16-byte pairs once per digit.  Needs the exact histogram.
      PartPlan spl = pl;
      sp_part = v2 && !sparse_lsd && part_plan_sparse(K, L.nb, has2, &spl);
      if (sp_part) {
        part_plan_wc(slots_all, columns_aligned16(d_key, d_key2, d_te, d_val), has2, 2, &spl);
        if (spl.wc_cap == 0 || slots_all + spl.pad_slots >= (1ull << 32)) sp_part = false;
      }
      if (sp_part && hist_sampled) { force_exact_hist = true; continue; }
      v2 = false;
      // (the partition sort: comp_a = the records by round, val_a = the staged ranks until the sorted list — if anyone needs it — takes their place;
      //  the b buffers = the staged points; a round's place is its block's record offset, fillers of pass B included)
      const uint64_t stage_slots = slots_all + (sp_part ? spl.pad_slots : 0);
      if ((rc = ensure(e, e->sp_comp_a, stage_slots * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->sp_comp_b, stage_slots * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->sp_val_a, stage_slots * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->sp_val_b, stage_slots * 8)) != TAD_OK) return rc;
      size_t tb = sparse_sort_temp_bytes(slots_all);
      if (sp_part && sparse_part_temp_bytes(spl) > tb) tb = sparse_part_temp_bytes(spl);
      if ((uint64_t)slots_all * 32 + tb > e->ws_limit)   // the four sort buffers count against the workspace too: fail cleanly, not in hipMalloc
        return fail(e, TAD_ERR_GRID_TOO_LARGE, "sparse Stage 0 needs %llu bytes of sort buffers for %llu row slots > workspace limit %llu",
                    (unsigned long long)(slots_all * 32 + tb), (unsigned long long)slots_all, (unsigned long long)e->ws_limit);
      if ((rc = ensure(e, e->sp_temp, tb + 64)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->sp_first, K * 4 + 64)) != TAD_OK) return rc;
      unsigned long long *d_runs = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(e->sp_temp.p) + tb);   // [0] runs, [1] tmax
      HIP_TRY(e, hipMemsetAsync(d_runs, 0, 16, s));
      HIP_TRY(e, hipEventRecord(e->ev[2], s));
      unsigned long long *ucomp = static_cast<unsigned long long *>(e->sp_comp_a.p), *uval = static_cast<unsigned long long *>(e->sp_val_a.p);
      // (the sort covers bit_width(span) time bits: a row beyond the lattice's last bucket raises DEV_ERR_OFF_LATTICE like a row before t0)
      const uint64_t span = L.nb ? (L.nb - 1) * (uint64_t)L.step : 0;
      if (sp_part) {
        const uint64_t slots = slots_all + spl.pad_slots;
        if ((rc = ensure(e, e->part_total, (size_t)spl.nparts * 4)) != TAD_OK) return rc;
        if ((rc = ensure(e, e->part_start, ((size_t)spl.nparts + 1) * 8)) != TAD_OK) return rc;
        if ((rc = ensure(e, e->part_offs32, (size_t)spl.G * spl.nparts * 4)) != TAD_OK) return rc;
        if ((rc = ensure(e, e->recs, (size_t)slots * 8)) != TAD_OK) return rc;
        if ((rc = ensure(e, e->slices, slice_table_bytes(slots, spl))) != TAD_OK) return rc;
        uint32_t *offs32 = static_cast<uint32_t *>(e->part_offs32.p);
        unsigned long long *part_start = static_cast<unsigned long long *>(e->part_start.p);
        launch_part_offsets(s, binhist, spl, offs32, static_cast<uint32_t *>(e->part_total.p), part_start, false,
                            static_cast<const MetaPartial *>(e->meta.p), n, slots, e->slices.p, Grid{});
        // (no overflow list: a value that does not fit the record raises DEV_ERR_OVERFLOW_LIST and the LSD sort redoes the job)
        launch_partition(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts, (const uint64_t *)d_val, n, K,
                         rf, L, spl, offs32, part_start, e->recs.p, nullptr, dev_ovf_count(e), 0, ctr, nullptr, nullptr);
        launch_sparse_sort(s, e->recs.p, part_start, binhist, spl, K, L.step, op_max,
                           ucomp, static_cast<unsigned long long *>(e->sp_comp_b.p), static_cast<unsigned long long *>(e->sp_val_b.p),
                           reinterpret_cast<uint32_t *>(uval), e->sp_temp.p, d_runs, ctr);
      } else if (launch_sparse_group(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts, (const uint64_t *)d_val, n, K,
                                     rf, L.t0, span, op_max, ucomp, uval, static_cast<unsigned long long *>(e->sp_comp_b.p),
                                     static_cast<unsigned long long *>(e->sp_val_b.p), e->sp_temp.p, tb, d_runs, ctr) != 0)
        return fail(e, TAD_ERR_HIP, "sparse Stage 0: sort / reduce failed");
      if (depth == 0) e->sp_by_partition = sp_part;
      // first[] / the longest series from the device-resident point count; then ONE round trip for both numbers
      // (the partition sort counted both itself and leaves its points in the stages: the sorted list is only built for those who read it)
      if (!sp_part) launch_sparse_tmax(s, ucomp, slots_all, d_runs, static_cast<uint32_t *>(e->sp_first.p), reinterpret_cast<unsigned int *>(d_runs + 1));
      unsigned long long runs_tmax[2] = {0, 0};
      HIP_TRY(e, hipMemcpyAsync(runs_tmax, d_runs, 16, hipMemcpyDeviceToHost, s));
      if (sp_part) HIP_TRY(e, hipMemcpyAsync(e->ctr_host, ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
      if (sp_part) {
        const uint32_t er = e->ctr_host->err;
        if (er & DEV_ERR_KEY_RANGE)
          return fail(e, TAD_ERR_KEY_RANGE, "a key id is >= num_keys (%llu) and is not TAD_KEY_SKIP", (unsigned long long)K);
        if (er & DEV_ERR_OFF_LATTICE) {
          if (lat_mode < 2) { lat_mode = (lat_mode == 0) ? 1 : 2; continue; }
          return fail(e, TAD_ERR_HIP, "internal error: a row fell off the derived time lattice");
        }
        if (er & (DEV_ERR_OVERFLOW_LIST | DEV_ERR_SPARSE_ROUND)) { sparse_lsd = true; continue; }   // a heavy key bin / a value wider than the record
      }
      const uint64_t P = runs_tmax[0];    // the filtered-out slots sort last and the reduction drops them
      const unsigned int tmax = (unsigned int)runs_tmax[1];
      cells = K * (uint64_t)tmax;
      need = cells * 17 + (jp.algo == TAD_ALGO_ARIMA ? cells * 80 + (1ull << 22) : (jp.algo == TAD_ALGO_DROP ? cells * 8 : 0));
      // Skewed series lengths (one key with a day of seconds next to many short-lived ones): K x Tmax does not fit although the
      // points do.  The keys are split into length classes that run as jobs of their own (run_sparse_classes).
      if (P && depth == 0 && (need > e->ws_limit || plan.sparse_classes == 1)) {
        if (sp_part) {
          launch_sparse_compact(s, spl, e->sp_temp.p, static_cast<const unsigned long long *>(e->sp_comp_b.p), static_cast<const unsigned long long *>(e->sp_val_b.p), ucomp, uval);
          launch_sparse_tmax(s, ucomp, slots_all, d_runs, static_cast<uint32_t *>(e->sp_first.p), reinterpret_cast<unsigned int *>(d_runs + 1));
        }
        HIP_TRY(e, hipMemcpyAsync(e->ctr_host, ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        const DevCounters c0 = *e->ctr_host;
        if (c0.err & DEV_ERR_KEY_RANGE)
          return fail(e, TAD_ERR_KEY_RANGE, "a key id is >= num_keys (%llu) and is not TAD_KEY_SKIP", (unsigned long long)K);
        if (c0.err & DEV_ERR_OFF_LATTICE) {
          if (lat_mode < 2) { lat_mode = (lat_mode == 0) ? 1 : 2; continue; }
          return fail(e, TAD_ERR_HIP, "internal error: a row fell off the derived time lattice");
        }
        if (points_mode) return sparse_points_direct(e, n, c0.rows_used, L, P, ctr, out_memory, points_out);   // Stage 0 alone needs no grid
        return run_sparse_classes(e, job, jp, op_max, n, c0.rows_used, K, L, P, tmax, out_memory, out);
      }
      if (need > e->ws_limit)
        return fail(e, TAD_ERR_GRID_TOO_LARGE, "sparse point grid needs %llu bytes (%llu keys x longest series %u points) > workspace limit %llu",
                    (unsigned long long)need, (unsigned long long)K, tmax, (unsigned long long)e->ws_limit);
      if ((rc = ensure(e, e->grid_val, (cells ? cells : 1) * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->grid_flag, cells ? cells : 1)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->sp_times, (cells ? cells : 1) * 8)) != TAD_OK) return rc;
      sparse_grid = Grid{static_cast<unsigned long long *>(e->grid_val.p), static_cast<uint8_t *>(e->grid_flag.p), tmax ? K : 0, tmax,
                         static_cast<const long long *>(e->sp_times.p)};
      if (cells) {
        HIP_TRY(e, hipMemsetAsync(sparse_grid.flag, 0, cells, s));
        if (sp_part)
          launch_sparse_place_staged(s, spl, e->sp_temp.p, static_cast<const unsigned long long *>(e->sp_comp_b.p), static_cast<const unsigned long long *>(e->sp_val_b.p),
                                     reinterpret_cast<const uint32_t *>(uval), L.t0, sparse_grid, static_cast<long long *>(e->sp_times.p));
        else
          launch_sparse_place(s, ucomp, uval, P, static_cast<const uint32_t *>(e->sp_first.p), L.t0, sparse_grid, static_cast<long long *>(e->sp_times.p));
      }
      HIP_TRY(e, hipEventRecord(e->ev[3], s));
    }
    if (!sparse && cells_overflow) return fail(e, TAD_ERR_GRID_TOO_LARGE, "grid of %llu keys x %llu buckets overflows", (unsigned long long)K, (unsigned long long)L.nb);
    if (!sparse && need > e->ws_limit) {
      if (lat_mode == 1 && v2) { lat_mode = 2; continue; }  // a too-fine sampled step cannot happen (it is a multiple of the true one); be safe
      return fail(e, TAD_ERR_GRID_TOO_LARGE,
                  "dense point grid needs %llu bytes (%llu keys x %llu time buckets, step %lld s) > workspace limit %llu",
                  (unsigned long long)need, (unsigned long long)K, (unsigned long long)L.nb, (long long)L.step, (unsigned long long)e->ws_limit);
    }
    if (!sparse) {
      if ((rc = ensure(e, e->grid_val, cells * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->grid_flag, cells)) != TAD_OK) return rc;
    }
    Grid g{static_cast<unsigned long long *>(e->grid_val.p), static_cast<uint8_t *>(e->grid_flag.p), empty ? 0 : K, L.nb, nullptr};
    if (sparse) g = sparse_grid;
    if (v2 && !part_plan_tiles(K, L.nb, has2, &pl)) v2 = false;  // tile does not fit LDS: direct scatter
    const bool stats_done = false;
    if (sparse) {
      // the rank grid is already filled
    } else if (v2) {
      part_plan_wc(hist_sampled ? sampled_slots_bound(n * (has2 ? 2 : 1), pl) : n * (has2 ? 2 : 1), columns_aligned16(d_key, d_key2, d_te, d_val), has2, plan.partition_pass, &pl);
      // nparts is only known now: the bound is recomputed with the final plan (part_plan_bins' G, part_plan_tiles' nparts)
      const uint64_t slots = hist_sampled ? sampled_slots_bound(n * (has2 ? 2 : 1), pl) : n * (has2 ? 2 : 1) + pl.pad_slots;
      if (hist_sampled && slots >= (1ull << 32)) { force_exact_hist = true; continue; }
      uint32_t *fin = nullptr;
      if (hist_sampled) {
        if ((rc = ensure(e, e->part_fin, (size_t)pl.G * pl.nparts * 8)) != TAD_OK) return rc;
        fin = static_cast<uint32_t *>(e->part_fin.p);
      }
      if ((rc = ensure(e, e->part_total, (size_t)pl.nparts * 4)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->part_start, ((size_t)pl.nparts + 1) * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->part_offs32, (size_t)pl.G * pl.nparts * 4)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->recs, (size_t)slots * 8)) != TAD_OK) return rc;
      if ((rc = ensure(e, e->ovf, 16 + (size_t)kOverflowCap * sizeof(OverflowRec))) != TAD_OK) return rc;
      unsigned long long *ovf_count = dev_ovf_count(e);     // in the job tail: zeroed with the counters, one fill per attempt
      OverflowRec *ovf = reinterpret_cast<OverflowRec *>(static_cast<unsigned char *>(e->ovf.p) + 16);
      if ((rc = ensure_key_buffers(e, K)) != TAD_OK) return rc;
      if ((rc = ensure_rcp_table(e, L.nb)) != TAD_OK) return rc;
      uint32_t *offs32 = static_cast<uint32_t *>(e->part_offs32.p);
      unsigned long long *part_start = static_cast<unsigned long long *>(e->part_start.p);
      if ((rc = ensure(e, e->slices, slice_table_bytes(slots, pl))) != TAD_OK) return rc;
      launch_part_offsets(s, binhist, pl, offs32, static_cast<uint32_t *>(e->part_total.p), part_start,
                          hist_sampled, static_cast<const MetaPartial *>(e->meta.p), n, slots, e->slices.p, g, ctr);
      // (per-key statistics run as their own kernel: fusing them into the tile pass measured slower on MI355X — one
      // wavefront per tile walks a 250-step FP64 dependency chain while the CU's other wavefronts have nothing left to do)
      // DBSCAN job: pass C in settle mode — key rounds, the detector's per-key pass on the LDS tile, grid columns of unsettled keys only.
      // Decided BEFORE pass B: with `max` the tile cells are 32-bit words (value + 1; three key rounds instead of six at C4) and pass B keeps
      // values >= 2^32 - 1 out of the records (overflow list + a bitmap of their keys, which alone are left to k_dbscan_scan).
      SettleArgs settle{};
      jp.settled = false;
      uint32_t *ovf_keys = nullptr;
      if (jp.algo == TAD_ALGO_DBSCAN && !jp.all_points && !points_mode && !stream && dbscan_uses_list(g) && part_plan_settle(L.nb, &pl, op_max && !force_wide_tiles)) {
        if ((rc = ensure(e, e->aux, dbscan_scratch_bytes(g))) != TAD_OK) return rc;
        if ((rc = ensure(e, e->ovf_keys, ((size_t)(K + 31) / 32) * 4 + 64)) != TAD_OK) return rc;
        ovf_keys = static_cast<uint32_t *>(e->ovf_keys.p);
        HIP_TRY(e, hipMemsetAsync(ovf_keys, 0, ((size_t)(K + 31) / 32) * 4, s));
        unsigned int *cnt = static_cast<unsigned int *>(e->aux.p);
        HIP_TRY(e, hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned int), s));    // work-list and redo-list lengths
        settle.redo_list = dbscan_redo_list(g, e->aux.p);
        settle.redo_count = cnt + 1;
        settle.st = DbscanStats{static_cast<uint32_t *>(e->n_pts.p), static_cast<uint32_t *>(e->n_anom.p), static_cast<double *>(e->key_mean.p),
                                static_cast<double *>(e->key_m2.p)};
        settle.list = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(e->aux.p) + 64);
        settle.count = cnt;
        settle.eps = jp.eps;
        settle.min_samples = jp.min_samples;
        settle.on = 1;
        settle.ovf_keys = ovf_keys;
        dbscan_compact_series(g, e->aux.p, &settle.cs_val, &settle.cs_flag, &settle.cs_has, &settle.cs_cap);
        dbscan_redo_series(g, e->aux.p, &settle.rs_val, &settle.rs_flag, &settle.rs_has, &settle.rs_cap);
        jp.settled = true;
        narrow_tiles = pl.narrow;
      }
      HIP_TRY(e, hipEventRecord(e->ev[2], s));
      launch_partition(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts,
                       (const uint64_t *)d_val, n, K, rf, L, pl, offs32, part_start, e->recs.p, ovf, ovf_count, kOverflowCap, ctr, fin, ovf_keys);
      HIP_TRY(e, hipEventRecord(e->ev[3], s));
      launch_tile_aggregate(s, e->recs.p, part_start, pl, slots, e->slices.p, g, op_max, ovf, ovf_count, kOverflowCap,
                            hist_sampled ? offs32 : nullptr, fin, settle);
    } else {
      if (cells) {
        HIP_TRY(e, hipMemsetAsync(g.val, 0, cells * 8, s));
        HIP_TRY(e, hipMemsetAsync(g.flag, 0, cells, s));
      }
      HIP_TRY(e, hipEventRecord(e->ev[2], s));
      if (!empty)
        launch_scatter(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts,
                       (const uint64_t *)d_val, n, rf, L, g, op_max, ctr);
      HIP_TRY(e, hipEventRecord(e->ev[3], s));
    }
    HIP_TRY(e, hipEventRecord(e->ev[5], s));
    if (depth == 0) e->done.store(2);

    // ---- Stage 1+2: sigma, detector, count, scan ----
    uint64_t rows = 0;
    ResultPriv *rp = nullptr;
    OutRows dev_rows{};
    ResultBlock dev_block;
    if (points_mode) {   // every present point: counts = n_pts
      if ((rc = ensure_key_buffers(e, g.K)) != TAD_OK) return rc;
      if ((rc = ensure_rcp_table(e, g.T)) != TAD_OK) return rc;
      launch_key_sigma(s, g, 0.5, false, static_cast<const double *>(e->rcp_table.p), static_cast<double *>(e->sigma.p),
                       static_cast<uint32_t *>(e->n_pts.p), static_cast<uint32_t *>(e->n_anom.p), ctr, static_cast<double *>(e->key_mean.p),
                       static_cast<double *>(e->key_m2.p));
      launch_moments(s, g.K, static_cast<const uint32_t *>(e->n_pts.p), static_cast<const double *>(e->key_mean.p),
                     static_cast<const double *>(e->key_m2.p), dev_moments(e));
      unsigned long long *off = static_cast<unsigned long long *>(e->off.p);
      launch_scan(s, static_cast<const uint32_t *>(e->n_pts.p), off, g.K, static_cast<unsigned long long *>(e->scan_scratch.p), dev_total(e));
      HIP_TRY(e, hipMemcpyAsync(e->tail_host, e->counters.p, kTailBytes, hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
      HIP_TRY(e, hipGetLastError());
      rows = *e->total_host;
    } else if (stream) {   // continue the per-key recurrences from the stored state; the next state stays a candidate
      if ((rc = ensure_key_buffers(e, g.K)) != TAD_OK) return rc;
      launch_stream(s, g, L, jp.alpha, jp.all_points, false, state_view(stream, stream->cur), state_view(stream, stream->cur ^ 1),
                    static_cast<uint32_t *>(e->n_anom.p), nullptr, OutRows{}, ctr);
      unsigned long long *off = static_cast<unsigned long long *>(e->off.p);
      launch_scan(s, static_cast<const uint32_t *>(e->n_anom.p), off, g.K, static_cast<unsigned long long *>(e->scan_scratch.p), dev_total(e));
      HIP_TRY(e, hipMemcpyAsync(e->tail_host, e->counters.p, kTailBytes, hipMemcpyDeviceToHost, s));
      HIP_TRY(e, hipStreamSynchronize(s));
      HIP_TRY(e, hipGetLastError());
      rows = *e->total_host;
      for (int b = 0; b < kMomentBlocks; ++b) e->moments_host[b] = Moments{0.0, 0.0, 0.0};
    } else {
      if ((rc = detect_and_count(e, g, jp, ctr, &rows, stats_done)) != TAD_OK) return rc;
    }
    const DevCounters c = *e->ctr_host;
    if (c.err & DEV_ERR_KEY_RANGE)
      return fail(e, TAD_ERR_KEY_RANGE, "a key id is >= num_keys (%llu) and is not TAD_KEY_SKIP", (unsigned long long)K);
    if (c.err & DEV_ERR_LATE_ROW)
      return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run_stream: a row is not newer than the last flowEndSeconds of its key's state; state unchanged");
    if (c.err & DEV_ERR_REGION_FULL) {   // a region sized from the sampled histogram was too small: exact histogram
      if (use_kh) { kh_rejected = true; continue; }     // ... or the caller's histogram is not this batch's: pass A counts
      if (!force_exact_hist) { force_exact_hist = true; continue; }
      return fail(e, TAD_ERR_HIP, "internal error: a partition region overflowed with an exact histogram");
    }
    if (c.err & DEV_ERR_OVERFLOW_LIST) {  // more than kOverflowCap values >= 2^49: the packed records do not pay off, use v1
      if (narrow_tiles && !force_wide_tiles) { force_wide_tiles = true; continue; }   // (... or >= 2^32 - 1 under 32-bit tile cells: 8-byte cells first)
      if (!force_v1_retry) { force_v1_retry = true; continue; }
      return fail(e, TAD_ERR_HIP, "internal error: overflow list full on the v1 path");
    }
    if (c.err & DEV_ERR_OFF_LATTICE) {
      if (lat_mode < 2) { lat_mode = (lat_mode == 0) ? 1 : 2; continue; }  // wrong hint -> derive; sampled gcd too coarse -> exact
      return fail(e, TAD_ERR_HIP, "internal error: a row fell off the derived time lattice");
    }
    if (depth == 0) e->done.store(3);

    if (points_mode) {
      PointsPriv *pp = new (std::nothrow) PointsPriv();
      if (!pp) return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory");
      memset(pp, 0, sizeof *pp);
      const uint64_t r = rows ? rows : 1;
      const size_t bytes = (size_t)r * 24;
      ResultBlock blk;
      if ((rc = alloc_device_block(e, bytes, &blk)) != TAD_OK) { delete pp; return rc; }
      unsigned char *d = static_cast<unsigned char *>(blk.base);
      if (rows)
        launch_emit_points(s, g, L, static_cast<const unsigned long long *>(e->off.p), reinterpret_cast<unsigned long long *>(d),
                           reinterpret_cast<long long *>(d + r * 8), reinterpret_cast<unsigned long long *>(d + r * 16));
      {
        const hipError_t er = hipEventRecord(e->ev[4], s);
        if (er != hipSuccess) { release_block(e, blk.base, blk.cap); delete pp; return fail(e, TAD_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(er)); }
      }
      unsigned char *base = d;
      if (out_memory == TAD_MEM_HOST) {
        void *h = malloc(bytes);
        if (!h) { release_block(e, blk.base, blk.cap); delete pp; return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory for %zu bytes of points", bytes); }
        hipError_t hr = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
        if (hr == hipSuccess) hr = hipStreamSynchronize(s);
        release_block(e, blk.base, blk.cap);
        if (hr != hipSuccess) { free(h); delete pp; return fail(e, TAD_ERR_HIP, "points copy failed: %s", hipGetErrorString(hr)); }
        base = static_cast<unsigned char *>(h);
        pp->block = h; pp->block_cap = bytes;
      } else {
        const hipError_t hr = hipStreamSynchronize(s);
        if (hr != hipSuccess) { release_block(e, blk.base, blk.cap); delete pp; return fail(e, TAD_ERR_HIP, "kernel failure: %s", hipGetErrorString(hr)); }
        pp->block = blk.base; pp->block_cap = blk.cap;
      }
      {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
          if (out_memory == TAD_MEM_HOST) free(pp->block); else release_block(e, pp->block, pp->block_cap);
          delete pp;
          return fail(e, TAD_ERR_HIP, "kernel failure: %s", hipGetErrorString(le));
        }
      }
      pp->pub.n_points = rows;
      pp->pub.key_id = reinterpret_cast<uint64_t *>(base);
      pp->pub.flow_end_s = reinterpret_cast<int64_t *>(base + r * 8);
      pp->pub.value = reinterpret_cast<uint64_t *>(base + r * 16);
      pp->pub.memory = out_memory;
      tad_stats &st = pp->pub.stats;
      st.rows_in = n; st.rows_used = c.rows_used; st.n_keys = c.n_keys; st.n_points = c.n_points;
      st.t0 = L.t0; st.step = L.step; st.n_buckets = L.nb;
      {
        double mn = 0.0, mean = 0.0, m2 = 0.0;
        if (g.K)
          for (int b = 0; b < kMomentBlocks; ++b) {
            const Moments &p = e->moments_host[b];
            if (p.n == 0.0) continue;
            if (mn == 0.0) { mn = p.n; mean = p.mean; m2 = p.m2; continue; }
            const double nn = mn + p.n, dd = p.mean - mean;
            mean = mean + dd * (p.n / nn);
            m2 = m2 + p.m2 + dd * dd * (mn * p.n / nn);
            mn = nn;
          }
        st.pts_mean = mean; st.pts_m2 = m2;
      }
      hipEventElapsedTime(&st.ms_meta, e->ev[0], e->ev[1]);
      hipEventElapsedTime(&st.ms_stage0, e->ev[1], e->ev[5]);
      hipEventElapsedTime(&st.ms_scatter, e->ev[2], e->ev[3]);
      hipEventElapsedTime(&st.ms_detect, e->ev[5], e->ev[4]);
      hipEventElapsedTime(&st.ms_total, e->ev[0], e->ev[4]);
      st.stage0_path = sparse ? (sp_part ? 8 : 4) : (v2 ? (pl.wc_cap ? 3 : 2) : 1);
      st.stage0_attempts = attempt + 1;
      st.hist_sampled = (v2 && hist_sampled) ? 1 : (use_kh ? 2 : 0);
      e->done.store(4);
      *points_out = &pp->pub;
      return TAD_OK;
    }

    // ---- Stage 3: emit ----
    if ((rc = make_result(e, rows, jp.all_points, out_memory, &rp, &dev_rows, &dev_block)) != TAD_OK) return rc;
    if (rows && stream)
      launch_stream(s, g, L, jp.alpha, jp.all_points, true, state_view(stream, stream->cur), state_view(stream, stream->cur ^ 1),
                    nullptr, static_cast<const unsigned long long *>(e->off.p), dev_rows, ctr);
    else if (rows)
      emit_rows(e, g, L, jp, dev_rows, rows);
    {
      const hipError_t er = hipEventRecord(e->ev[4], s);
      if (er != hipSuccess) {
        release_block(e, dev_block.base, dev_block.cap);
        delete rp;
        return fail(e, TAD_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(er));
      }
    }
    if ((rc = finish_result(e, rp, rows, jp.all_points, dev_block, dev_rows)) != TAD_OK) { delete rp; return rc; }
    hipError_t le = hipStreamSynchronize(s);
    if (le == hipSuccess) le = hipGetLastError();
    if (le != hipSuccess) { tad_result_free(e->eng, &rp->pub); return fail(e, TAD_ERR_HIP, "kernel failure: %s", hipGetErrorString(le)); }

    tad_stats &st = rp->pub.stats;
    st.rows_in = n;
    st.rows_used = c.rows_used;
    st.n_keys = c.n_keys;
    st.n_points = c.n_points;
    st.keys_no_result = c.keys_no_result;
    st.kalman_steps = c.kalman_steps;
    st.arima_fits = c.arima_fits;
    st.arima_nan_fits = c.arima_nan_fits;
    st.t0 = L.t0; st.step = L.step; st.n_buckets = L.nb;
    {
      double mn = 0.0, mean = 0.0, m2 = 0.0;  // Chan merge of the block partials, fixed order
      if (g.K)
        for (int b = 0; b < kMomentBlocks; ++b) {
          const Moments &p = e->moments_host[b];
          if (p.n == 0.0) continue;
          if (mn == 0.0) { mn = p.n; mean = p.mean; m2 = p.m2; continue; }
          const double nn = mn + p.n, d = p.mean - mean;
          mean = mean + d * (p.n / nn);
          m2 = m2 + p.m2 + d * d * (mn * p.n / nn);
          mn = nn;
        }
      st.pts_mean = mean;
      st.pts_m2 = m2;
    }
    st.n_anomalies = rows;
    if (jp.all_points) {
      // count verdicts host- or device-side? cheap: the emit kernel wrote them; count on the host copy if there is one
      st.n_anomalies = 0;
      if (rows) {
        std::vector<uint8_t> tmp;
        const uint8_t *a = rp->pub.anomaly;
        if (out_memory == TAD_MEM_DEVICE) {
          tmp.resize(rows);
          const hipError_t cr = hipMemcpy(tmp.data(), rp->pub.anomaly, rows, hipMemcpyDeviceToHost);
          if (cr != hipSuccess) { tad_result_free(e->eng, &rp->pub); return fail(e, TAD_ERR_HIP, "verdict copy failed: %s", hipGetErrorString(cr)); }
          a = tmp.data();
        }
        for (uint64_t i = 0; i < rows; ++i) st.n_anomalies += a[i];
      }
    }
    hipEventElapsedTime(&st.ms_meta, e->ev[0], e->ev[1]);
    hipEventElapsedTime(&st.ms_stage0, e->ev[1], e->ev[5]);
    hipEventElapsedTime(&st.ms_scatter, e->ev[2], e->ev[3]);
    hipEventElapsedTime(&st.ms_detect, e->ev[5], e->ev[4]);
    st.stage0_path = sparse ? (sp_part ? 8 : 4) : (v2 ? (pl.wc_cap ? 3 : 2) : 1);
    st.stage0_attempts = attempt + 1;
    st.hist_sampled = (v2 && hist_sampled) ? 1 : (use_kh ? 2 : 0);
    st.host_syncs = (hinted || empty) ? 2 : 3;
    st.job_context = e->index;
    st.arima_relaunches = e->arima_relaunches;
    hipEventElapsedTime(&st.ms_total, e->ev[0], e->ev[4]);
    if (depth == 0 && !points_mode && !stream) {
      JobCtx::Learnt &w = e->learnt;
      const bool exact_now = force_exact_hist && plan.histogram != 1;
      if (learnt_exact_hist) w.exact_uses++;                                           // same table shape, the exact histogram once more
      else if (probing_sampled_hist) { w.exact_uses = 0; w.exact_backoff = exact_now ? (w.exact_backoff < 64 ? w.exact_backoff * 2 : 64) : 8; }
      else { w.exact_uses = 0; w.exact_backoff = 8; }
      w.valid = true; w.n = n; w.K = K; w.has2 = has2; w.algo = (int)job->algo; w.op = (int)op_max;
      w.exact_hist = exact_now;
      w.wide_tiles = force_wide_tiles && plan.tile_cells != 1;
    }
    strncpy(rp->pub.id, job->id, sizeof rp->pub.id - 1);
    if (stream && g.K) stream->cur ^= 1;   // the batch succeeded: the candidate state becomes current (an empty batch wrote none)
    if (depth == 0) e->done.store(4);
    *out = &rp->pub;
    return TAD_OK;
  }
  return fail(e, TAD_ERR_HIP, "internal error: Stage 0 did not settle on a lattice / strategy after 6 attempts");
}

// Stage 0 alone on a sparse table whose rank grid does not fit: the sorted unique points (e->sp_comp_a / e->sp_val_a) are
// the answer — three columns out, counters and moments from the same pass (tad_sparse.hip:k_sparse_points_out).
int sparse_points_direct(JobCtx *e, uint64_t n_rows_in, uint64_t rows_used, Lattice L, uint64_t P, DevCounters *ctr, tad_mem out_memory,
                         tad_points **points_out) {
  hipStream_t s = e->stream;
  int rc;
  if ((rc = ensure(e, e->counters, kTailBytes)) != TAD_OK) return rc;
  PointsPriv *pp = new (std::nothrow) PointsPriv();
  if (!pp) return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory");
  memset(pp, 0, sizeof *pp);
  const size_t bytes = (size_t)P * 24;
  ResultBlock blk;
  if ((rc = alloc_device_block(e, bytes, &blk)) != TAD_OK) { delete pp; return rc; }
  unsigned char *d = static_cast<unsigned char *>(blk.base);
  launch_sparse_points_out(s, static_cast<const unsigned long long *>(e->sp_comp_a.p), static_cast<const unsigned long long *>(e->sp_val_a.p), P, L.t0,
                           reinterpret_cast<unsigned long long *>(d), reinterpret_cast<long long *>(d + P * 8),
                           reinterpret_cast<unsigned long long *>(d + P * 16), dev_moments(e), ctr);
  hipError_t hr = hipMemcpyAsync(e->tail_host, e->counters.p, kTailBytes, hipMemcpyDeviceToHost, s);
  if (hr == hipSuccess) hr = hipEventRecord(e->ev[7], s);
  void *h = nullptr;
  if (hr == hipSuccess && out_memory == TAD_MEM_HOST) {
    h = malloc(bytes);
    if (!h) { release_block(e, blk.base, blk.cap); delete pp; return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory for %zu bytes of points", bytes); }
    hr = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
  }
  if (hr == hipSuccess) hr = hipStreamSynchronize(s);
  if (hr == hipSuccess) hr = hipGetLastError();
  if (hr != hipSuccess) {
    release_block(e, blk.base, blk.cap);
    free(h);
    delete pp;
    return fail(e, TAD_ERR_HIP, "sparse Stage 0, points: %s", hipGetErrorString(hr));
  }
  unsigned char *base = d;
  if (out_memory == TAD_MEM_HOST) {
    release_block(e, blk.base, blk.cap);
    base = static_cast<unsigned char *>(h);
    pp->block = h; pp->block_cap = bytes;
  } else {
    pp->block = blk.base; pp->block_cap = blk.cap;
  }
  pp->pub.n_points = P;
  pp->pub.key_id = reinterpret_cast<uint64_t *>(base);
  pp->pub.flow_end_s = reinterpret_cast<int64_t *>(base + P * 8);
  pp->pub.value = reinterpret_cast<uint64_t *>(base + P * 16);
  pp->pub.memory = out_memory;
  tad_stats &st = pp->pub.stats;
  const DevCounters c = *e->ctr_host;
  st.rows_in = n_rows_in; st.rows_used = rows_used; st.n_keys = c.n_keys; st.n_points = c.n_points;
  st.t0 = L.t0; st.step = L.step; st.n_buckets = L.nb;
  double mn = 0.0, mean = 0.0, m2 = 0.0;
  for (int b = 0; b < kMomentBlocks; ++b) {
    const Moments &p = e->moments_host[b];
    if (p.n == 0.0) continue;
    if (mn == 0.0) { mn = p.n; mean = p.mean; m2 = p.m2; continue; }
    const double nn = mn + p.n, dd = p.mean - mean;
    mean = mean + dd * (p.n / nn);
    m2 = m2 + p.m2 + dd * dd * (mn * p.n / nn);
    mn = nn;
  }
  st.pts_mean = mean; st.pts_m2 = m2;
  hipEventElapsedTime(&st.ms_total, e->ev[6], e->ev[7]);
  st.ms_stage0 = st.ms_total;
  st.stage0_path = e->sp_by_partition ? 10 : 7;
  st.stage0_attempts = 1;
  e->done.store(4);
  *points_out = &pp->pub;
  return TAD_OK;
}

// A sparse table whose K x Tmax rank grid does not fit (skewed series lengths): the keys are split into classes by series
// length (tad_sparse.hip), every class is handed to run_job_locked as a points table of its own — renumbered dense key ids,
// (key, time) order kept, one row per point, so its Stage 0 only re-sorts what is sorted — and the row sets are merged back in
// ORIGINAL key order.  Detectors are per key, so the rows are the rows of the single-grid run, bit for bit; the job-wide
// moments are Chan-merged in class order (telemetry).  On entry the sorted unique points are in e->sp_comp_a / e->sp_val_a
// (P of them), e->sp_first[k] = first point of key k; the class jobs reuse every engine buffer, so the parent's state moves
// to a block of its own first.
int run_sparse_classes(JobCtx *e, const tad_job *job, const JobParams &jp, bool op_max, uint64_t n_rows_in, uint64_t rows_used, uint64_t K, Lattice L,
                       uint64_t P, uint32_t tmax, tad_mem out_memory, tad_result **out) {
  hipStream_t s = e->stream;
  int rc;
  const uint32_t nclass = sparse_class_count(tmax);
  // per-key arrays: len u32 | member u32 | pts u32 | key_off u64[K + 1] | pt_off u64[K + 1]
  const size_t kpad = (size_t)((K + 3) & ~3ull);
  if ((rc = ensure(e, e->sp_cls, kpad * 12 + (kpad + 4) * 16 + 64)) != TAD_OK) return rc;
  if ((rc = ensure(e, e->scan_scratch, scan_scratch_elems(K ? K : 1) * sizeof(unsigned long long))) != TAD_OK) return rc;
  uint32_t *len = static_cast<uint32_t *>(e->sp_cls.p), *member = len + kpad, *pts = member + kpad;
  unsigned long long *key_off = reinterpret_cast<unsigned long long *>(pts + kpad), *pt_off = key_off + kpad + 4;
  unsigned long long *scratch = static_cast<unsigned long long *>(e->scan_scratch.p);
  const unsigned long long *ucomp = static_cast<const unsigned long long *>(e->sp_comp_a.p), *uval = static_cast<const unsigned long long *>(e->sp_val_a.p);
  const uint32_t *first = static_cast<const uint32_t *>(e->sp_first.p);
  HIP_TRY(e, hipMemsetAsync(len, 0, (size_t)K * 4, s));
  launch_sparse_len(s, ucomp, P, first, len);

  // the class tables: three 8-byte columns per point, class after class, then the key maps (class key -> original key)
  const uint64_t kmax = K < P ? K : P;   // keys with points
  ResultBlock blk;
  if ((rc = alloc_device_block(e, (size_t)P * 24 + (size_t)kmax * 4 + 256, &blk)) != TAD_OK) return rc;
  unsigned long long *c_key = static_cast<unsigned long long *>(blk.base);
  long long *c_t = reinterpret_cast<long long *>(c_key + P);
  unsigned long long *c_val = reinterpret_cast<unsigned long long *>(c_t + P);
  uint32_t *c_map = reinterpret_cast<uint32_t *>(c_val + P);
  struct Cls { uint64_t keys, points, key0, pt0; tad_result *res; };
  std::vector<Cls> cls;
  auto release = [&]() {
    for (Cls &c : cls) if (c.res) { tad_result_free(e->eng, c.res); c.res = nullptr; }
    release_block(e, blk.base, blk.cap);
  };
  uint64_t key0 = 0, pt0 = 0;
  for (uint32_t c = 0; c < nclass; ++c) {
    launch_sparse_class_counts(s, len, K, c, member, pts);
    launch_scan(s, member, key_off, K, scratch);
    launch_scan(s, pts, pt_off, K, scratch);
    unsigned long long kc = 0, pc = 0;
    hipError_t hr = hipMemcpyAsync(&kc, key_off + K, 8, hipMemcpyDeviceToHost, s);
    if (hr == hipSuccess) hr = hipMemcpyAsync(&pc, pt_off + K, 8, hipMemcpyDeviceToHost, s);
    if (hr == hipSuccess) hr = hipStreamSynchronize(s);
    if (hr != hipSuccess) { release(); return fail(e, TAD_ERR_HIP, "length classes: %s", hipGetErrorString(hr)); }
    if (kc == 0) continue;
    launch_sparse_class_columns(s, ucomp, uval, P, first, len, c, key_off, pt_off, L.t0, c_key + pt0, c_t + pt0, c_val + pt0, c_map + key0);
    cls.push_back(Cls{kc, pc, key0, pt0, nullptr});
    key0 += kc;
    pt0 += pc;
  }
  if (pt0 != P || key0 > kmax) { release(); return fail(e, TAD_ERR_HIP, "internal error: length classes cover %llu of %llu points", (unsigned long long)pt0, (unsigned long long)P); }
  {
    const hipError_t hr = hipStreamSynchronize(s);   // the class jobs below overwrite the sort buffers the kernels above read
    if (hr != hipSuccess) { release(); return fail(e, TAD_ERR_HIP, "length classes: %s", hipGetErrorString(hr)); }
  }

  // one job per class (filters are applied, every (key, time) is unique: the operator no longer matters)
  tad_job sub = *job;
  sub.start_time = 0;
  sub.end_time = 0;
  sub.value_op = op_max ? TAD_OP_MAX : TAD_OP_SUM;
  uint64_t rows = 0;
  for (Cls &c : cls) {
    tad_columns cc;
    memset(&cc, 0, sizeof cc);
    cc.n_rows = c.points;
    cc.num_keys = c.keys;
    cc.key_id = reinterpret_cast<const uint64_t *>(c_key + c.pt0);
    cc.flow_end_s = reinterpret_cast<const int64_t *>(c_t + c.pt0);
    cc.value = reinterpret_cast<const uint64_t *>(c_val + c.pt0);
    cc.memory = TAD_MEM_DEVICE;
    if ((rc = run_job_locked(e, &sub, &cc, TAD_MEM_DEVICE, &c.res, nullptr, nullptr, 1)) != TAD_OK) { release(); return rc; }
    rows += c.res->n_rows;
  }
  e->done.store(3);

  // merge: rows of original key k start at off[k] = rows of all smaller original keys (whatever their class)
  if ((rc = ensure_key_buffers(e, K)) != TAD_OK) { release(); return rc; }
  if ((rc = ensure(e, e->aux, (size_t)(K ? K : 1) * 8)) != TAD_OK) { release(); return rc; }
  uint32_t *cnt = static_cast<uint32_t *>(e->n_anom.p);
  unsigned long long *off = static_cast<unsigned long long *>(e->off.p), *first_row = static_cast<unsigned long long *>(e->aux.p);
  ResultPriv *rp = nullptr;
  OutRows dev_rows;
  ResultBlock dev_block;
  if ((rc = make_result(e, rows, jp.all_points, out_memory, &rp, &dev_rows, &dev_block)) != TAD_OK) { release(); return rc; }
  hipError_t hr = hipMemsetAsync(cnt, 0, (size_t)K * 4, s);
  for (Cls &c : cls)
    launch_class_count_rows(s, reinterpret_cast<const unsigned long long *>(c.res->key_id), c.res->n_rows, c_map + c.key0, cnt, first_row);
  launch_scan(s, cnt, off, K, static_cast<unsigned long long *>(e->scan_scratch.p));
  for (Cls &c : cls) {
    OutRows src{reinterpret_cast<unsigned long long *>(c.res->key_id), reinterpret_cast<long long *>(c.res->flow_end_s), c.res->throughput,
                c.res->algo_calc, c.res->stddev, c.res->anomaly};
    launch_class_gather(s, src, c.res->n_rows, c_map + c.key0, off, first_row, dev_rows);
  }
  if (hr == hipSuccess) hr = hipEventRecord(e->ev[7], s);
  if (hr == hipSuccess) hr = hipStreamSynchronize(s);
  if (hr == hipSuccess) hr = hipGetLastError();
  if (hr != hipSuccess) {
    release_block(e, dev_block.base, dev_block.cap);
    delete rp;
    release();
    return fail(e, TAD_ERR_HIP, "length classes, merge: %s", hipGetErrorString(hr));
  }
  if ((rc = finish_result(e, rp, rows, jp.all_points, dev_block, dev_rows)) != TAD_OK) { delete rp; release(); return rc; }

  tad_stats &st = rp->pub.stats;
  st.rows_in = n_rows_in;
  st.rows_used = rows_used;
  st.t0 = L.t0; st.step = L.step; st.n_buckets = L.nb;
  double mn = 0.0, mean = 0.0, m2 = 0.0;   // Chan merge of the classes' (n_points, mean, M2), class order
  float ms_classes = 0.0f;
  for (const Cls &c : cls) {
    const tad_stats &cs = c.res->stats;
    st.n_keys += cs.n_keys;
    st.n_points += cs.n_points;
    st.n_anomalies += cs.n_anomalies;
    st.keys_no_result += cs.keys_no_result;
    st.kalman_steps += cs.kalman_steps;
    st.arima_fits += cs.arima_fits;
    ms_classes += cs.ms_total;
    const double pn = (double)cs.n_points;
    if (pn == 0.0) continue;
    if (mn == 0.0) { mn = pn; mean = cs.pts_mean; m2 = cs.pts_m2; continue; }
    const double nn = mn + pn, d = cs.pts_mean - mean;
    mean = mean + d * (pn / nn);
    m2 = m2 + cs.pts_m2 + d * d * (mn * pn / nn);
    mn = nn;
  }
  st.pts_mean = mean;
  st.pts_m2 = m2;
  hipEventElapsedTime(&st.ms_total, e->ev[6], e->ev[7]);
  st.ms_detect = ms_classes;                       // the class jobs, each with its own (small) Stage 0
  st.ms_stage0 = st.ms_total - ms_classes;         // sort + reduce + class tables + merge
  st.stage0_path = e->sp_by_partition ? 9 : 6;
  st.stage0_attempts = 1;
  strncpy(rp->pub.id, job->id, sizeof rp->pub.id - 1);
  release();
  e->done.store(4);
  *out = &rp->pub;
  return TAD_OK;
}

}  // namespace

extern "C" {

int tad_run(tad_engine *e, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_result **out) {
  if (e && !out) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run: job, cols and out must not be NULL");
  return run_job(e, job, cols, out_memory, out, nullptr);
}

int tad_run_stream(tad_engine *e, tad_state *st, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_result **out) {
  if (!e) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_run_stream: engine is NULL");
  if (!st || !out) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_run_stream: state and out must not be NULL");
  return run_job(e, job, cols, out_memory, out, nullptr, st);
}

int tad_state_create(tad_engine *eng, uint64_t num_keys, tad_state **out) {
  if (!eng || !out || num_keys == 0) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_state_create: bad arguments");
  *out = nullptr;
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_state_create: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  tad_state *st = new (std::nothrow) tad_state();
  if (!st) return fail(e, TAD_ERR_OUT_OF_MEMORY, "out of host memory");
  st->K = num_keys;
  for (int i = 0; i < 2; ++i) {
    hipError_t r = hipMalloc(&st->block[i], state_bytes(num_keys));
    if (r == hipSuccess) r = hipMemsetAsync(st->block[i], 0, state_bytes(num_keys), e->stream);   // n = 0, avg = m2 = ewma = 0, unseen
    if (r != hipSuccess) {
      for (int j = 0; j <= i; ++j) if (st->block[j]) hipFree(st->block[j]);
      delete st;
      return fail(e, TAD_ERR_OUT_OF_MEMORY, "tad_state_create: %s", hipGetErrorString(r));
    }
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  *out = st;
  return TAD_OK;
}

void tad_state_destroy(tad_engine *e, tad_state *st) {
  if (!st) return;
  { std::lock_guard<std::mutex> lk(st->mu); }   // a batch on this state has returned (it synchronises its stream before it does)
  if (e) hipSetDevice(e->device);
  for (int i = 0; i < 2; ++i) if (st->block[i]) hipFree(st->block[i]);
  delete st;
}

int tad_state_export(tad_engine *eng, const tad_state *st, uint32_t *n, double *avg, double *m2, double *ewma, int64_t *last_t) {
  if (!eng || !st) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_state_export: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_state_export: no job context available");
  std::lock_guard<std::mutex> state_lk(st->mu);
  HIP_TRY(e, hipSetDevice(e->device));
  const StreamState v = state_view(st, st->cur);
  if (n) HIP_TRY(e, hipMemcpy(n, v.n, st->K * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (avg) HIP_TRY(e, hipMemcpy(avg, v.avg, st->K * sizeof(double), hipMemcpyDeviceToHost));
  if (m2) HIP_TRY(e, hipMemcpy(m2, v.m2, st->K * sizeof(double), hipMemcpyDeviceToHost));
  if (ewma) HIP_TRY(e, hipMemcpy(ewma, v.ewma, st->K * sizeof(double), hipMemcpyDeviceToHost));
  if (last_t) HIP_TRY(e, hipMemcpy(last_t, v.last_t, st->K * sizeof(long long), hipMemcpyDeviceToHost));
  return TAD_OK;
}

int tad_aggregate(tad_engine *e, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_points **out) {
  if (e && !out) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_aggregate: job, cols and out must not be NULL");
  return run_job(e, job, cols, out_memory, nullptr, out);
}


}  // extern "C"
