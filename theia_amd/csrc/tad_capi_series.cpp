// tad_capi_series.cpp — the per-series entry points of include/tad.h: the reference's pure functions (anomaly_detection.py:146-349) on one key,
// values in time order, through the same device kernels as tad_run (a 1-key series table).
#include "tad_engine.h"

using namespace tad;
using namespace tadh;

namespace {

// Fill the engine's grid with one series: K = 1, T = n, every point present.
int series_grid(JobCtx *e, const uint64_t *x, uint64_t n, Grid *g) {
  int rc;
  if ((rc = ensure(e, e->grid_val, (n ? n : 1) * 8)) != TAD_OK) return rc;
  if ((rc = ensure(e, e->grid_flag, n ? n : 1)) != TAD_OK) return rc;
  if ((rc = ensure(e, e->counters, kTailBytes)) != TAD_OK) return rc;
  if (n) {
    HIP_TRY(e, hipMemcpyAsync(e->grid_val.p, x, n * 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->grid_flag.p, FLAG_PRESENT, n, e->stream));
  }
  HIP_TRY(e, hipMemsetAsync(e->counters.p, 0, sizeof(DevCounters), e->stream));
  g->val = static_cast<unsigned long long *>(e->grid_val.p);
  g->flag = static_cast<uint8_t *>(e->grid_flag.p);
  g->K = 1;
  g->T = n;
  g->times = nullptr;
  return TAD_OK;
}

// Emit every point of a one-key grid with given sigma; copies verdicts / calc to the host.
int series_emit_all(JobCtx *e, Grid g, const JobParams &jp, bool has_sigma, double sigma, double *calc_out, uint8_t *verdict_out) {
  const uint64_t n = g.T;
  int rc;
  if ((rc = ensure(e, e->sigma, sizeof(double))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->n_pts, sizeof(uint32_t))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->off, 2 * sizeof(unsigned long long))) != TAD_OK) return rc;
  const uint32_t npts = has_sigma ? (uint32_t)(n < 2 ? 2 : n) : (uint32_t)(n < 1 ? 0 : 1);  // n_pts >= 2 <=> sigma is defined
  const unsigned long long off[2] = {0ull, (unsigned long long)n};
  HIP_TRY(e, hipMemcpyAsync(e->sigma.p, &sigma, sizeof sigma, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->n_pts.p, &npts, sizeof npts, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->off.p, off, sizeof off, hipMemcpyHostToDevice, e->stream));
  ResultBlock blk;
  if ((rc = alloc_device_block(e, result_bytes(n, true), &blk)) != TAD_OK) return rc;
  OutRows o;
  carve(blk.base, n, true, &o);
  JobParams all = jp;
  all.all_points = true;
  emit_rows(e, g, make_lattice(0, 1, n), all, o);
  hipError_t r = hipSuccess;
  if (calc_out && n) r = hipMemcpyAsync(calc_out, o.algo_calc, n * 8, hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess && verdict_out && n) r = hipMemcpyAsync(verdict_out, o.anomaly, n, hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
  release_block(e, blk.base, blk.cap);
  if (r != hipSuccess) return fail(e, TAD_ERR_HIP, "series copy failed: %s", hipGetErrorString(r));
  return TAD_OK;
}

JobParams series_params(tad_algo algo, double alpha, double eps, int min_samples, int maxiter) {
  JobParams jp;
  jp.algo = algo;
  jp.alpha = alpha == 0.0 ? 0.5 : alpha;
  jp.eps = eps == 0.0 ? 250000000.0 : eps;
  jp.min_samples = min_samples == 0 ? 4 : min_samples;
  jp.maxiter = maxiter == 0 ? 50 : maxiter;
  jp.drop_nsigma = 3.0;
  jp.drop_min_samples = 3;
  jp.all_points = true;
  return jp;
}

}  // namespace

extern "C" {

int tad_series_ewma(tad_engine *eng, const uint64_t *x, uint64_t n, double alpha, double *out) {
  if (!eng || (n && (!x || !out))) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_ewma: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_ewma: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  Grid g;
  int rc = series_grid(e, x, n, &g);
  if (rc != TAD_OK || n == 0) return rc;
  return series_emit_all(e, g, series_params(TAD_ALGO_EWMA, alpha, 0, 0, 0), false, 0.0, out, nullptr);
}

int tad_series_ewma_anomaly(tad_engine *eng, const uint64_t *x, uint64_t n, double alpha, int has_stddev, double stddev,
                            uint8_t *verdict) {
  if (!eng || (n && (!x || !verdict))) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_ewma_anomaly: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_ewma_anomaly: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  Grid g;
  int rc = series_grid(e, x, n, &g);
  if (rc != TAD_OK || n == 0) return rc;
  return series_emit_all(e, g, series_params(TAD_ALGO_EWMA, alpha, 0, 0, 0), has_stddev != 0, stddev, nullptr, verdict);
}

int tad_series_stddev(tad_engine *eng, const uint64_t *x, uint64_t n, int *has_stddev, double *stddev) {
  if (!eng || !has_stddev || !stddev || (n && !x)) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_stddev: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_stddev: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  *has_stddev = 0;
  *stddev = 0.0;
  Grid g;
  int rc = series_grid(e, x, n, &g);
  if (rc != TAD_OK || n == 0) return rc;
  if ((rc = ensure(e, e->sigma, sizeof(double))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->n_pts, sizeof(uint32_t))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->n_anom, sizeof(uint32_t))) != TAD_OK) return rc;
  if ((rc = ensure_rcp_table(e, g.T)) != TAD_OK) return rc;
  launch_key_sigma(e->stream, g, 0.5, false, static_cast<const double *>(e->rcp_table.p), static_cast<double *>(e->sigma.p), static_cast<uint32_t *>(e->n_pts.p),
                   static_cast<uint32_t *>(e->n_anom.p), static_cast<DevCounters *>(e->counters.p), nullptr, nullptr);
  double sg = 0.0;
  HIP_TRY(e, hipMemcpyAsync(&sg, e->sigma.p, sizeof sg, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  *has_stddev = n >= 2;
  *stddev = sg;
  return TAD_OK;
}

int tad_series_dbscan_anomaly(tad_engine *eng, const uint64_t *x, uint64_t n, double eps, int min_samples, uint8_t *verdict) {
  if (!eng || (n && (!x || !verdict)) || eps < 0.0 || min_samples < 0)
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_dbscan_anomaly: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_dbscan_anomaly: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  Grid g;
  int rc = series_grid(e, x, n, &g);
  if (rc != TAD_OK || n == 0) return rc;
  JobParams jp = series_params(TAD_ALGO_DBSCAN, 0, eps, min_samples, 0);
  if ((rc = ensure(e, e->aux, dbscan_scratch_bytes(g))) != TAD_OK) return rc;
  if (launch_dbscan(e->stream, g, jp.eps, jp.min_samples, e->aux.p) != 0) return fail(e, TAD_ERR_HIP, "DBSCAN launch failed");
  return series_emit_all(e, g, jp, false, 0.0, nullptr, verdict);
}

int tad_series_drop(tad_engine *eng, const uint64_t *x, uint64_t n, double nsigma, int min_samples, int *has_result,
                    double *mean, double *stddev, uint8_t *verdict) {
  if (!eng || !has_result || !mean || !stddev || (n && (!x || !verdict)) || nsigma < 0.0 || min_samples < 0)
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_drop: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_drop: no job context available");
  HIP_TRY(e, hipSetDevice(e->device));
  *has_result = 0;
  *mean = 0.0;
  *stddev = 0.0;
  const int ms = min_samples == 0 ? 3 : min_samples;
  if (n == 0 || n < (uint64_t)ms || n < 2) return TAD_OK;   // drop_detection_udf.py:44-45
  Grid g;
  int rc = series_grid(e, x, n, &g);
  if (rc != TAD_OK) return rc;
  if ((rc = ensure_key_buffers(e, 1)) != TAD_OK) return rc;
  if ((rc = ensure(e, e->calc, n * sizeof(double))) != TAD_OK) return rc;
  launch_drop(e->stream, g, nsigma == 0.0 ? 3.0 : nsigma, ms, static_cast<double *>(e->calc.p), static_cast<double *>(e->sigma.p),
              static_cast<uint32_t *>(e->n_pts.p), static_cast<double *>(e->key_mean.p), static_cast<double *>(e->key_m2.p),
              static_cast<DevCounters *>(e->counters.p));
  std::vector<uint8_t> flags(n);
  HIP_TRY(e, hipMemcpyAsync(mean, e->key_mean.p, sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipMemcpyAsync(stddev, e->sigma.p, sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipMemcpyAsync(flags.data(), g.flag, n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  HIP_TRY(e, hipGetLastError());
  *has_result = 1;
  for (uint64_t i = 0; i < n; ++i) verdict[i] = (flags[i] & FLAG_ANOMALY) ? 1 : 0;
  return TAD_OK;
}

}  // extern "C"

namespace {

// calculate_arima / calculate_arima_anomaly on one series, on the context the caller holds.  pred_out (n doubles) may be NULL.
int series_arima_locked(JobCtx *e, const uint64_t *x, uint64_t n, int maxiter, int has_stddev, double stddev,
                        uint8_t *verdict, uint64_t *n_verdict, double *pred_out) {
  HIP_TRY(e, hipSetDevice(e->device));
  *n_verdict = 1;
  verdict[0] = 0;
  if (n <= 3) return TAD_OK;  // anomaly_detection.py:232-234 -> None -> [False] (:284-287)
  Grid g;
  int rc = series_grid(e, x, n, &g);
  if (rc != TAD_OK) return rc;
  JobParams jp = series_params(TAD_ALGO_ARIMA, 0, 0, 0, maxiter);
  if ((rc = ensure(e, e->sigma, sizeof(double))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->n_pts, sizeof(uint32_t))) != TAD_OK) return rc;
  if ((rc = ensure(e, e->calc, n * sizeof(double))) != TAD_OK) return rc;
  // sigma as given by the caller; n_pts carries the real length for ARIMA
  const double sg = has_stddev ? stddev : __builtin_inf();  // no sigma -> no point can exceed it
  const uint32_t npts = (uint32_t)n;
  HIP_TRY(e, hipMemcpyAsync(e->sigma.p, &sg, sizeof sg, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->n_pts.p, &npts, sizeof npts, hipMemcpyHostToDevice, e->stream));
  const size_t wsb = arima_workspace_bytes(g);
  if ((rc = ensure(e, e->aux, wsb)) != TAD_OK) return rc;
  DevCounters *ctr = static_cast<DevCounters *>(e->counters.p);
  if (launch_arima(e->stream, g, static_cast<const double *>(e->sigma.p), static_cast<const uint32_t *>(e->n_pts.p), jp.maxiter,
                   static_cast<double *>(e->calc.p), ctr, e->aux.p, wsb) != 0)
    return fail(e, TAD_ERR_HIP, "ARIMA launch failed");
  HIP_TRY(e, hipMemcpyAsync(e->ctr_host, ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, e->stream));
  std::vector<uint8_t> flags(n);
  HIP_TRY(e, hipMemcpyAsync(flags.data(), g.flag, n, hipMemcpyDeviceToHost, e->stream));
  if (pred_out) HIP_TRY(e, hipMemcpyAsync(pred_out, e->calc.p, n * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  HIP_TRY(e, hipGetLastError());
  if (e->ctr_host->keys_no_result) return TAD_OK;  // calculate_arima returned None
  *n_verdict = n;
  for (uint64_t i = 0; i < n; ++i) verdict[i] = (flags[i] & FLAG_ANOMALY) ? 1 : 0;
  return TAD_OK;
}

}  // namespace

extern "C" {

int tad_series_arima(tad_engine *eng, const uint64_t *x, uint64_t n, int maxiter, int *has_result, double *out) {
  if (!eng || !has_result || (n && (!x || !out)) || maxiter < 0) return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_arima: bad arguments");
  uint64_t nv = 0;
  std::vector<uint8_t> verdict(n ? n : 1);
  std::vector<double> pred(n ? n : 1);
  // ONE critical section: the predictions are read from the engine's calc buffer before any other thread can run
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_arima: no job context available");
  const int rc = series_arima_locked(e, x, n, maxiter, 0, 0.0, verdict.data(), &nv, pred.data());
  if (rc != TAD_OK) return rc;
  *has_result = (nv == n && n > 3) ? 1 : 0;
  if (*has_result) memcpy(out, pred.data(), n * 8);
  return TAD_OK;
}

int tad_series_arima_anomaly(tad_engine *eng, const uint64_t *x, uint64_t n, int maxiter, int has_stddev, double stddev,
                             uint8_t *verdict, uint64_t *n_verdict) {
  if (!eng || !n_verdict || !verdict || (n && !x) || maxiter < 0)
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_series_arima_anomaly: bad arguments");
  Lease lease(eng);
  JobCtx *e = lease.c;
  if (!e) return fail(eng, TAD_ERR_OUT_OF_MEMORY, "tad_series_arima_anomaly: no job context available");
  return series_arima_locked(e, x, n, maxiter, has_stddev, stddev, verdict, n_verdict, nullptr);
}

}  // extern "C"
