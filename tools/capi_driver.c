/* capi_driver.c — plain-C caller of libtad_mi355x.so, the way a cgo shim sees it (include/tad.h is C).
 * Runs the reference's golden series (anomaly_detection_test.py:199-217, 3 keys x 30 points of it) through
 * tad_run with HOST columns and prints the anomalous rows.  Exit code 0 = ok, 3 = no GPU (expected on a
 * CPU-only box: the library has no CPU fallback), anything else = failure.
 * build: gcc -std=c11 -Iinclude tools/capi_driver.c -Ltheia_amd/lib -ltad_mi355x -Wl,-rpath,$PWD/theia_amd/lib -o /tmp/capi_driver */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tad.h"

int main(int argc, char **argv) {
  const char *algo_name = argc > 1 ? argv[1] : "EWMA";
  tad_engine *e = NULL;
  tad_engine_opts opts;
  memset(&opts, 0, sizeof opts);
  int rc = tad_engine_create(&opts, &e);
  if (rc != TAD_OK) {
    fprintf(stderr, "tad_engine_create failed: %d: %s\n", rc, tad_last_error(NULL));
    return rc == TAD_ERR_NO_DEVICE ? 3 : 1;
  }
  enum { K = 3, T = 30, N = K * T * 2 };
  static uint64_t key[N], val[N];
  static int64_t tend[N];
  uint64_t n = 0;
  for (int k = 0; k < K; ++k)
    for (int t = 0; t < T; ++t)
      for (int dup = 0; dup < 2; ++dup) { /* two rows per point: sum() must add them */
        key[n] = (uint64_t)k;
        tend[n] = 1660202814 + 60 * (int64_t)t;
        val[n] = 2000000000ull + 1000ull * (uint64_t)t + (uint64_t)dup + ((t == 20 + k) ? 30000000000ull : 0ull);
        ++n;
      }
  tad_job job;
  memset(&job, 0, sizeof job);
  job.algo = !strcmp(algo_name, "ARIMA") ? TAD_ALGO_ARIMA : !strcmp(algo_name, "DBSCAN") ? TAD_ALGO_DBSCAN : TAD_ALGO_EWMA;
  job.agg_flow = TAD_AGG_SVC;
  strncpy(job.id, "capi-driver", sizeof job.id - 1);
  tad_columns cols;
  memset(&cols, 0, sizeof cols);
  cols.n_rows = n;
  cols.key_id = key;
  cols.flow_end_s = tend;
  cols.value = val;
  cols.num_keys = K;
  cols.memory = TAD_MEM_HOST;
  tad_result *res = NULL;
  rc = tad_run(e, &job, &cols, TAD_MEM_HOST, &res);
  if (rc != TAD_OK) {
    fprintf(stderr, "tad_run failed: %d: %s\n", rc, tad_last_error(e));
    tad_engine_destroy(e);
    return 1;
  }
  printf("id=%s algo=%s keys=%llu points=%llu anomalies=%llu lattice=(%lld,%lld,%llu)\n", res->id, algo_name,
         (unsigned long long)res->stats.n_keys, (unsigned long long)res->stats.n_points,
         (unsigned long long)res->stats.n_anomalies, (long long)res->stats.t0, (long long)res->stats.step,
         (unsigned long long)res->stats.n_buckets);
  for (uint64_t i = 0; i < res->n_rows; ++i)
    printf("row key=%llu t=%lld x=%.17g calc=%.17g sd=%.17g\n", (unsigned long long)res->key_id[i],
           (long long)res->flow_end_s[i], res->throughput[i], res->algo_calc[i], res->stddev[i]);
  /* an illegal job must come back as TAD_ERR_INVALID_ARGUMENT with the controller's message */
  job.algo = (tad_algo)7;
  tad_result *bad = NULL;
  rc = tad_run(e, &job, &cols, TAD_MEM_HOST, &bad);
  printf("illegal algo -> %d: %s\n", rc, tad_last_error(e));
  tad_result_free(e, res);
  tad_engine_destroy(e);
  return rc == TAD_ERR_INVALID_ARGUMENT ? 0 : 1;
}
