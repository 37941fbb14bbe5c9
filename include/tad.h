/*
 * tad.h — C ABI of libtad_mi355x.so, the MI355X (gfx950) throughput-anomaly-detection engine.
 *
 * This is the drop-in boundary for ONE path of antrea-io/theia: the Throughput Anomaly
 * Detection job.  The reference has no FFI for it — the boundary there is a process boundary:
 *   - pkg/controller/anomalydetector/controller.go:525-698 builds the job's argument vector
 *     (--algo, --start_time, --end_time, --agg-flow, --pod-label, ... --id) and launches a
 *     SparkApplication;
 *   - plugins/anomaly-detection/anomaly_detection.py:647-710 (anomaly_detection) runs it:
 *     ClickHouse GROUP BY (:507-614) -> per-key series + stddev_samp (:664-684) ->
 *     EWMA / ARIMA / DBSCAN per key (:146-349) -> explode + keep anomalies (:352-421) ->
 *     append to default.tadetector (:713-726).
 * A cgo host (theia-manager) binds exactly the entry points below instead of launching Spark;
 * INTEGRATION.md shows that binding.  Every struct is plain C: pointers, sizes, enums.  No
 * callbacks, no retained caller pointers after return, no exceptions across the boundary.
 *
 * Strings never cross the boundary: the host dictionary-encodes the mode's key columns
 * (anomaly_detection.py:109-137 DF_GROUP_COLUMNS / DF_AGG_GRP_COLUMNS_*) into dense uint64
 * key ids and evaluates the string predicates of the SQL (:507-614); a row the predicates reject
 * carries TAD_KEY_SKIP.
 */
#ifndef THEIA_TAD_H
#define THEIA_TAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAD_ABI_VERSION 12
#define TAD_KEY_SKIP UINT64_MAX /* row (or its second key) does not take part */

/* ---- error codes (0 = ok, negative = failure; text via tad_last_error) ---- */
enum {
  TAD_OK = 0,
  TAD_ERR_INVALID_ARGUMENT = -1, /* maps to the controller's illegal-argument FAILED state,
                                    controller.go:505-514 */
  TAD_ERR_NO_DEVICE = -2,        /* no gfx950 device / HIP runtime failure at create */
  TAD_ERR_OUT_OF_MEMORY = -3,
  TAD_ERR_HIP = -4,              /* a HIP call or kernel failed */
  TAD_ERR_KEY_RANGE = -5,        /* a key id >= num_keys (and != TAD_KEY_SKIP) */
  TAD_ERR_GRID_TOO_LARGE = -6,   /* num_keys x time-lattice does not fit the workspace limit */
  TAD_ERR_BUSY = -7
};

/* --algo, controller.go:527-533 ("EWMA" | "ARIMA" | "DBSCAN"); anomaly_detection.py:697-709 */
/* TAD_ALGO_DROP: the abnormal-traffic-drop detector of the reference's Snowflake backend
 * (snowflake/udfs/udfs/drop_detection/drop_detection_udf.py:42-56): per key mean / sample std of the aggregated counts,
 * anomaly outside mean +- drop_nsigma * std, keys with fewer than drop_min_samples points yield nothing.
 * Result rows: throughput = the count, algo_calc = the key's mean, stddev = its std. */
typedef enum { TAD_ALGO_EWMA = 0, TAD_ALGO_ARIMA = 1, TAD_ALGO_DBSCAN = 2, TAD_ALGO_DROP = 3 } tad_algo;

/* --agg-flow, controller.go:560-620; anomaly_detection.py:617-628 (aggType literal).
 * NONE  : per-connection keys, max(throughput)   (anomaly_detection.py:52-61)
 * POD / SVC / EXTERNAL : sum(throughput)         (:63-106)                              */
typedef enum { TAD_AGG_NONE = 0, TAD_AGG_POD = 1, TAD_AGG_SVC = 2, TAD_AGG_EXTERNAL = 3 } tad_agg_flow;

/* Stage-0 aggregate over rows that share (key, flowEndSeconds). UInt64 semantics of ClickHouse:
 * SUM wraps mod 2^64, MAX is unsigned (create_table.sh:74 `throughput UInt64`). */
typedef enum { TAD_OP_AUTO = 0, TAD_OP_MAX = 1, TAD_OP_SUM = 2 } tad_value_op;

typedef enum { TAD_MEM_HOST = 0, TAD_MEM_DEVICE = 1 } tad_mem;

/* tad_job.flags */
#define TAD_FLAG_EMIT_ALL_POINTS 1u /* result = every point (plotDF before the filter of :394),
                                       with its verdict in tad_result.anomaly; for inspection/tests */

typedef struct tad_engine tad_engine; /* opaque; one per GPU; runs up to max_jobs_in_flight jobs concurrently (ABI 12) */

/* Plan overrides (ABI 7).  Every field 0 = the engine decides from the shape of the batch, which is what a production host
 * passes.  A non-zero field forces one of the strategies the engine would otherwise choose between: the parity tests run
 * every strategy on the same table, A/B measurements time them on the same box.  ABI <= 6 read TAD_* environment variables
 * per job for this — process-global state that a host with several workers (controller.go:199-201) cannot scope to a job
 * or an engine; since ABI 7 the library reads no environment variable at all. */
typedef struct {
  int32_t stage0;            /* 1 = direct atomic scatter into the grid, 2 = partition + LDS tiles whatever the batch size */
  int32_t partition_pass;    /* 1 = sort-by-tile pass B, 2 = write-combining pass B whenever its queues fit LDS, 3 = as 2 but 64-byte sectors even where whole 128-byte lines fit (A/B) */
  int32_t histogram;         /* 1 = exact per-workgroup histogram in pass A (regions of pass B never sized from a sample), 2 = sampled wherever possible (A/B) */
  int32_t sparse;            /* 1 = never, 2 = always the sort-based Stage 0 for sparse tables */
  int32_t sparse_classes;    /* 1 = always run a sparse table as length classes of keys */
  int32_t ewma_emit;         /* 1 = lane-per-key emit for the EWMA job instead of the LDS-staged one */
  uint32_t ewma_emit_rows;   /* LDS rows per wavefront of the staged EWMA emit (<= 4096); 0 = sized from the row count */
  int32_t reserved0;         /* must be 0 (ABI 8-11: one_sync — the one-synchronisation form of a job was removed in ABI 12: a placement-neutral
                                A/B put it at 1.0 % of a C2 / C4 job, profiles/r6_a1_ab1_*.log) */
  int32_t tile_cells;        /* 1 = 8-byte tile cells in the settle mode of DBSCAN jobs with `max` (ABI 8; default: 32-bit cells, value + 1) */
  int32_t sparse_sort;       /* sparse tables (ABI 9; was `reserved`): 1 = always the LSD radix sort, 2 = the partition pass + LDS sort wherever its plan fits
                                (0: when pass A ran with its key-bin histogram, i.e. >= 2^22 rows) */
  int32_t reserved1;         /* must be 0 (ABI 11: placement — the placement search of pass B's record buffer was removed in ABI 12: 7-21 ms per
                                search to win <= 0.04 ms per job on the SAME column buffers; a controller brings new columns with every job,
                                profiles/r6_a1_cold_*.log) */
} tad_plan;

typedef struct {
  int32_t device;            /* HIP device ordinal */
  void *stream;              /* hipStream_t to run on, or NULL: the engine creates its own */
  uint64_t workspace_limit;  /* bytes of HBM ONE job may use for its grid and Stage-0 buffers; 0 = 3/4 of what is free at create.  Per job in
                                flight: contexts keep their (grow-only) buffers between jobs, and when an allocation fails the idle contexts'
                                buffers are given back to the device before the job fails */
  tad_plan plan;             /* all zero in production */
  int32_t max_jobs_in_flight;/* ABI 12: job contexts (own HIP stream, events, workspace) the engine may create — tad_run calls from that many
                                threads run concurrently on the GPU, further callers wait for a context.  0 = 4 (controller.go:199-201 runs four
                                workers; Spark ran one pod per job), 1 = jobs serialise (ABI <= 11), max 16.  Forced to 1 when `stream` is
                                given.  Contexts are created on demand: a serial caller only ever uses the first */
  int32_t reserved;          /* must be 0 */
} tad_engine_opts;

/* Mirrors the job's argument vector (anomaly_detection.py:781-870). */
typedef struct {
  tad_algo algo;
  tad_agg_flow agg_flow;
  tad_value_op value_op;     /* AUTO: MAX for TAD_AGG_NONE, SUM otherwise */
  int64_t start_time;        /* epoch seconds; 0 = unset. Row kept iff flow_start_s >= start_time
                                (:581-583). Ignored when columns.flow_start_s is NULL. */
  int64_t end_time;          /* epoch seconds; 0 = unset. Row kept iff flow_end_s < end_time (:584-586) */
  double ewma_alpha;         /* 0 -> 0.5 (:157) */
  double dbscan_eps;         /* 0 -> 250000000 (:342) */
  int32_t dbscan_min_samples;/* 0 -> 4 (:342) */
  int32_t arima_maxiter;     /* 0 -> 50 (statsmodels fit() default) */
  double drop_nsigma;        /* 0 -> 3 (drop_detection_udf.py:49-50) */
  int32_t drop_min_samples;  /* 0 -> 3 (:44) */
  uint32_t flags;            /* TAD_FLAG_* */
  char id[64];               /* --id, echoed into the result (tadetector.id, :503) */
} tad_job;

/* By-product of tad_factorize_hist (ABI 12): how many of each Stage-0 workgroup's rows fall into each key bin — what pass A of tad_run would
 * otherwise count with a second read of the key column when it sizes pass B's regions exactly (many-key tables: DBSCAN on 1e6 keys spends
 * 0.19 ms and 0.89 GB of its 1.55 ms there).  The factorisation has every row's id in registers when it writes key_id; counting there is
 * free.  `bins` is DEVICE memory of TAD_KEY_HIST_BYTES bytes supplied by the caller; the other fields are filled in by tad_factorize_hist
 * and checked by tad_run against its own plan — a histogram that does not belong to the batch (other row count, key count, sides) or that
 * the job cannot use (a time-window filter drops rows the histogram counted; a small batch has no pass A) is ignored.  A histogram of the right
 * shape whose COUNTS are another batch's (a stale one) costs an attempt, never memory or rows: pass B writes nothing past a region, reports the
 * region it found full, and the job is redone with pass A's own count (tad_stats.stage0_attempts).  The counts must be those tad_factorize_hist
 * wrote: their sum is not checked against n_rows. */
#define TAD_KEY_HIST_BYTES ((uint64_t)256 * 16384 * 4)
typedef struct {
  uint64_t n_rows, num_keys;   /* the batch it was taken from */
  uint64_t chunk_rows;         /* rows per workgroup */
  uint32_t workgroups, nbins, shift, sides;
  uint32_t *bins;              /* DEVICE [workgroups][nbins]: rows of workgroup g whose key id >> shift == b */
} tad_key_hist;

/* One columnar batch of flow rows (the columns the SQL of :507-614 touches, after the host's
 * dictionary encoding).  All arrays have n_rows entries; memory says where they live. */
typedef struct {
  uint64_t n_rows;
  const uint64_t *key_id;       /* dense id of the row's key, < num_keys, or TAD_KEY_SKIP */
  const uint64_t *key_id2;      /* optional (NULL): second key of the same row — pod mode's
                                   UNION ALL of inbound + outbound (:556-565) */
  const int64_t *flow_end_s;    /* flowEndSeconds, epoch seconds (DateTime) */
  const int64_t *flow_start_s;  /* optional (NULL): flowStartSeconds for the start_time filter */
  const uint64_t *value;        /* throughput (UInt64) — or any UInt64 column, e.g. octetDeltaCount */
  uint64_t num_keys;            /* size of the key dictionary (ids are 0..num_keys-1) */
  tad_mem memory;
  /* Optional time lattice: flow_end_s = t0 + step*bucket, bucket < n_buckets.  n_buckets == 0:
   * the engine derives (min, gcd of differences, max) itself with one extra pass over flow_end_s. */
  int64_t t0;
  int64_t step;
  uint64_t n_buckets;
  const tad_key_hist *key_hist; /* optional (NULL): tad_factorize_hist's by-product for THIS batch (ABI 12) */
} tad_columns;

/* Per-run counters and stage timings (for CompletedStages/TotalStages-style progress and bench). */
typedef struct {
  uint64_t rows_in;        /* n_rows */
  uint64_t rows_used;      /* rows that passed the filters (each key of a 2-key row counts) */
  uint64_t n_keys;         /* keys with >= 1 point */
  uint64_t n_points;       /* distinct (key, flowEndSeconds) points = P */
  uint64_t n_anomalies;    /* A (0 => caller writes the sentinel row, :395-420) */
  uint64_t keys_no_result; /* ARIMA keys that yield no rows (n<=3, x<=0, constant; :232-234,260-264) */
  uint64_t kalman_steps;   /* ARIMA: filter time-steps over all likelihood evaluations */
  uint64_t arima_fits;     /* ARIMA: number of (key,t) fits */
  uint64_t arima_nan_fits; /* ARIMA: fits whose prediction is not finite (the optimiser walked into a non-finite likelihood: Box-Cox
                              with a strongly negative lambda next to the 1e6 diffuse prior).  Python's abs(x - nan) > sigma is False,
                              so such a point is never an anomaly (anomaly_detection.py:306-307) — counted so that an operator can
                              tell voided fits from clean ones */
  double pts_mean;         /* mean and sum of squared deviations (M2) of the aggregated point values of */
  double pts_m2;           /* this shard: (n_points, mean, M2) triples Chan-merge across GPUs into the global
                              mean / sigma the multi-GPU host reports (telemetry; the reference has none) */
  int64_t t0, step;        /* the time lattice used.  stage0_path >= 4 (sparse tables) does not place rows on a lattice: there t0 is the
                              smallest flowEndSeconds, and step / n_buckets are the caller's hint or pass A's (min, max, gcd) estimate,
                              reported for information — rows are never rejected for being off it (the dense path does reject) */
  uint64_t n_buckets;
  float ms_meta;           /* lattice derivation pass */
  float ms_stage0;         /* Stage 0 after the lattice pass: v1 grid clear + k_scatter; v2 offsets +
                              k_partition + k_tile_aggregate */
  float ms_scatter;        /* the dominant Stage-0 kernel alone: k_scatter (v1) or k_partition (v2) */
  float ms_detect;         /* per-key sigma + detector + compaction */
  float ms_total;          /* device time of the whole run, HIP events on the engine stream */
  int32_t stage0_path;     /* 1 = direct atomic scatter, 2 = partition (sort-by-tile pass B) + LDS tiles, 3 = partition (write-combining pass B) + LDS tiles,
                              4 = sparse table: sort by (key, time) + rank grid (time proportional to the rows, not to keys x lattice),
                              (5 was the two-level partition of ABI 6: measured no faster than the single-level plan, removed)
                              6 = sparse table with skewed series lengths: as 4, then one job per length class of keys (<= 16, <= 64, ... points),
                                  rows merged back in key order (the K x longest-series rank grid would not fit the workspace),
                              7 = tad_aggregate on such a table: the sorted unique points are the result, no grid at all,
                              8 / 9 / 10 (ABI 9) = as 4 / 6 / 7 with the rows sorted through the key-block partition pass + one LDS sort per key sub-range
                                  instead of the LSD radix sort (big sparse tables: the columns are read once, 8-byte records move through HBM once) */
  int32_t stage0_attempts; /* times Stage 0 ran before it settled: 1 normally; more after a wrong lattice hint, a sampled lattice or
                              a sampled histogram that proved too optimistic (every fallback is exact), an overflow-list fallback */
  int32_t hist_sampled;    /* 1: pass B's regions were sized from a SAMPLE of the key column (1/16 of pass A's reads); 2 (ABI 12): from the
                              caller's tad_key_hist (exact; pass A only sampled the time lattice); 0: from pass A's own exact histogram */
  int32_t host_syncs;      /* host synchronisations of the attempt that produced the result: 3 = lattice derivation, row count, result;
                              2 with a lattice hint (or an empty batch) */
  int32_t job_context;     /* ABI 12: index of the job context (stream + workspace) that ran the job; 0 for a serial caller */
  int32_t arima_relaunches;/* ABI 12: times this job's ARIMA fit kernel was relaunched after its wavefronts had retired early to make room for
                              another job's whole-CU workgroups (pass B / pass C need 1024 threads and up to 156 KB of LDS per workgroup and
                              cannot be placed beside the fit's long-lived wavefronts); 0 when the job ran alone.  Results do not depend on it */
} tad_stats;

/* Anomalous points only (anomaly_detection.py:394), ordered by (key_id, flow_end_s).
 * Columns = what the mode-independent part of a tadetector row needs (create_table.sh:363-384):
 * flowEndSeconds, throughputStandardDeviation, algoCalc, throughput; the host expands key_id
 * back into the mode's string columns and adds aggType / algoType / id / anomaly="true". */
typedef struct {
  uint64_t n_rows;         /* rows in the arrays below: stats.n_anomalies, or stats.n_points with
                              TAD_FLAG_EMIT_ALL_POINTS */
  uint64_t *key_id;
  int64_t *flow_end_s;
  double *throughput;      /* float(x): correctly rounded uint64 -> double (:161) */
  double *algo_calc;       /* EWMA value / ARIMA prediction / 0.0 for DBSCAN (:312-322) */
  double *stddev;          /* the key's stddev_samp (:674-684) */
  uint8_t *anomaly;        /* NULL unless TAD_FLAG_EMIT_ALL_POINTS: verdict per emitted point */
  tad_mem memory;          /* where the arrays live (same as the request's out_memory) */
  tad_stats stats;
  char id[64];
} tad_result;

/* ---- engine life cycle ---- */
int tad_abi_version(void);
int tad_engine_create(const tad_engine_opts *opts, tad_engine **out);
void tad_engine_destroy(tad_engine *e);
/* Replace the engine's plan overrides (NULL = all zero); serialised with the jobs, takes effect with the next one. */
int tad_engine_set_plan(tad_engine *e, const tad_plan *plan);
/* Thread-safe; the returned string is owned by the engine (or static when e == NULL). */
const char *tad_last_error(tad_engine *e);

/* ---- the job: replaces the SparkApplication run (anomaly_detection.py:647-710) ----
 * Callable from several OS threads (controller.go:199-201 runs 4 workers): each call takes one of the engine's job contexts and runs
 * on that context's stream, so up to max_jobs_in_flight jobs overlap on the GPU (a short EWMA job does not queue behind a long ARIMA
 * job: ARIMA jobs run on a low-priority stream); callers beyond that wait for a context.  Results are bit-identical to a serial run
 * (contexts share no buffers).  out_memory selects host or device result arrays. */
int tad_run(tad_engine *e, const tad_job *job, const tad_columns *cols, tad_mem out_memory,
            tad_result **out);
void tad_result_free(tad_engine *e, tad_result *r);
/* ---- Stage 0 alone: the GROUP BY the reference pushes into ClickHouse (anomaly_detection.py:507-614) ----
 * Aggregated points ordered by (key_id, flow_end_s); value keeps the full UInt64 (sum wraps, max unsigned).
 * Used by row-sharded multi-GPU ingest: every GPU pre-aggregates its slice of the rows, the partial points travel to
 * the key owners (one all-to-all), and tad_run over the partials with the same value_op gives bit-identical
 * aggregates (integer add / max are associative).  job->algo and the detector parameters are ignored. */
typedef struct {
  uint64_t n_points;
  uint64_t *key_id;
  int64_t *flow_end_s;
  uint64_t *value;
  tad_mem memory;
  tad_stats stats;       /* rows_in, rows_used, n_keys, n_points, lattice, stage timings */
} tad_points;
int tad_aggregate(tad_engine *e, const tad_job *job, const tad_columns *cols, tad_mem out_memory, tad_points **out);
void tad_points_free(tad_engine *e, tad_points *p);

/* ---- row-sharded ingest: bucket rows by the owner of their key for ONE all-to-all(v) ----
 * The reference has no such call: Spark's shuffle (anomaly_detection.py:664-684, groupby(key)) plays this role.  With G
 * GPUs, key k is owned by rank k mod G under the local id k / G.  tad_shard_rows writes the rows of `cols` grouped by
 * destination rank — destination 0's rows first — with LOCAL key ids, into three DEVICE arrays of cols->n_rows elements
 * each, and the rows per destination into counts[world] (HOST): the send splits of the all-to-all(v).  Rows whose key is
 * TAD_KEY_SKIP are dropped.  cols->memory must be TAD_MEM_DEVICE; key_id2 / flow_start_s must be NULL (pre-aggregate pod-mode
 * tables with tad_aggregate first: its points have one key).  The order of rows inside a destination is unspecified
 * (Stage 0 aggregates with commutative operators).  1 <= world <= 1024. */
int tad_shard_rows(tad_engine *e, const tad_columns *cols, uint32_t world, uint64_t *out_key_id, int64_t *out_flow_end_s,
                   uint64_t *out_value, uint64_t *counts);

/* ---- ingest: the GROUP BY key tuples of the job factorised on the GPU (ABI 8) ----
 * The reference groups in ClickHouse over string / integer columns (anomaly_detection.py:52-137, 507-614); the engine wants dense
 * key ids.  The host evaluates the SQL's string predicates on the distinct values of each string column (keep masks) and hands the
 * rows' key TUPLES over as n_cols (<= 8) columns of 8-byte integers: dictionary codes of the string columns, ports, protocol,
 * flowStartSeconds.  Out: key_id[i] = dense id of row i's tuple, ids in order of FIRST APPEARANCE (what pandas.factorize gives:
 * the GPU path and the host path of theia_amd/anomaly_detection.py:prepare_columns produce identical ids and key tables),
 * TAD_KEY_SKIP where keep[i] == 0; first_row[k] (k < first_row_cap) = the virtual row where key k first appears — the host reads
 * the key's column values there; *num_keys.
 * Pod mode (the UNION ALL of the inbound and the outbound view, :556-565) passes a second tuple per row (cols_b / keep_b, ids into
 * key_id2): ids are assigned over the virtual rows [side a: 0 .. n) ++ [side b: n .. 2n) and the side is part of the tuple.
 * n_rows * sides must be < 2^32 - 1.  All arrays (inputs and outputs) live in kc->memory. */
typedef struct {
  uint64_t n_rows;
  int32_t n_cols;                /* 1..8 */
  const int64_t *const *cols_a;  /* n_cols pointers to n_rows values each */
  const uint8_t *keep_a;         /* NULL = every row */
  const int64_t *const *cols_b;  /* NULL = one tuple per row */
  const uint8_t *keep_b;
  tad_mem memory;
} tad_key_columns;
int tad_factorize(tad_engine *e, const tad_key_columns *kc, uint64_t *key_id, uint64_t *key_id2, uint64_t *first_row,
                  uint64_t first_row_cap, uint64_t *num_keys);
/* ... and the same with the key-bin histogram of the ids as a by-product (tad_key_hist above): hist->bins must point to TAD_KEY_HIST_BYTES
 * bytes of DEVICE memory (whatever kc->memory is); the other fields are outputs.  hist->n_rows == 0 afterwards: no histogram (empty batch,
 * or more than 2^32 - 1 row slots). */
int tad_factorize_hist(tad_engine *e, const tad_key_columns *kc, uint64_t *key_id, uint64_t *key_id2, uint64_t *first_row,
                       uint64_t first_row_cap, uint64_t *num_keys, tad_key_hist *hist);

/* ---- ingest, one step earlier (ABI 10): an Arrow string column -> dictionary codes on the GPU ----
 * ClickHouse delivers the job's GROUP BY columns (sourcePodName, destinationPodName, pod labels, namespaces, destinationIP,
 * destinationServicePortName: anomaly_detection.py:52-137, 507-614) as strings; tad_factorize above wants integer columns.  The host's
 * dictionary encode (theia_amd/clickhouse.py:query_columns) was the slowest stage of a 1e8-row job.  In: one column in Arrow's layout —
 * n_rows + 1 offsets (int32 for `string`, int64 for `large_string`; pass the pointer already advanced by a sliced array's offset) into
 * `data`, an optional validity bitmap (a null row encodes like the empty string, which is what the host path did).  Out: codes[i] = id of
 * row i's string, ids in order of FIRST APPEARANCE (what Arrow's dictionary_encode and pandas.factorize give: same codes and
 * dictionaries as the host path); first_row[k] (k < first_row_cap) = the row where value k first appears — the host reads the
 * dictionary's strings there; *num_values.  n_rows < 2^32 - 1.  All arrays live in col->memory.  Offsets that decrease or point beyond
 * data_bytes are TAD_ERR_INVALID_ARGUMENT (checked on the device while the rows are read). */
typedef struct {
  uint64_t n_rows;
  const void *offsets;        /* n_rows + 1 */
  int32_t offset_bits;        /* 32 or 64 */
  const uint8_t *data;
  uint64_t data_bytes;        /* bytes behind `data` that offsets may address */
  const uint8_t *validity;    /* NULL = no nulls; else bit (validity_offset + i) of the bitmap is row i */
  uint64_t validity_offset;
  tad_mem memory;
} tad_string_column;
int tad_encode_strings(tad_engine *e, const tad_string_column *col, int64_t *codes, uint64_t *first_row, uint64_t first_row_cap,
                       uint64_t *num_values);

/* ---- ingest, columnar (ABI 12): Arrow record batches in host memory -> the 8-byte device columns of tad_factorize / tad_run ----
 * ClickHouse's HTTP interface (the reference's JDBC URL points at it, anomaly_detection.py:730-731) streams the raw rows as Arrow record
 * batches; with `toLowCardinality(col)` and output_format_arrow_low_cardinality_as_dictionary = 1 the string columns arrive as Arrow
 * DICTIONARY arrays, a dictionary per batch.  The host maps each batch's dictionary (its distinct values only) into the column's job-wide
 * one and evaluates the SQL's string predicates (anomaly_detection.py:507-614) on the distinct values; the rows are touched on the GPU:
 * tad_widen_column: dst[i] = table ? table[src[i]] : src[i], for i < n.  src holds n integers of src_bits (8 / 16 / 32 / 64) bits, sign-extended
 *   when src_signed, in HOST (staged by the call) or DEVICE memory; table (int64[table_len], DEVICE) and dst (int64[n], DEVICE).  Covers the
 *   dictionary indices of a batch (table = the batch's remap), UInt32 DateTime and UInt16 port columns (table NULL) and gathers of a device
 *   column at the rows tad_factorize reports in first_row (src = first_row, table = the column).  An index outside the table is
 *   TAD_ERR_INVALID_ARGUMENT.
 * tad_mask_rows: keep[i] = AND over t < n_terms of (masks[t][codes[t][i]] != 0), ANDed into the previous keep[i] when combine != 0.  codes[t]
 *   (int64[n]), masks[t] (uint8[mask_len[t]]) and keep (uint8[n]) are DEVICE memory; the pointer arrays themselves are host memory.  n_terms <= 8.
 * tad_host_alloc / tad_host_free: page-locked host memory (a reader receives the HTTP body straight into it; copies from it run at PCIe rate). */
int tad_widen_column(tad_engine *e, const void *src, int32_t src_bits, int32_t src_signed, tad_mem src_memory, uint64_t n, const int64_t *table,
                     uint64_t table_len, int64_t *dst);
int tad_mask_rows(tad_engine *e, uint64_t n, int32_t n_terms, const int64_t *const *codes, const uint8_t *const *masks, const uint64_t *mask_len,
                  int32_t combine, uint8_t *keep);
int tad_host_alloc(tad_engine *e, uint64_t bytes, void **ptr);
int tad_host_free(tad_engine *e, void *ptr);

/* ---- streaming EWMA: per-key running state kept in HBM between batches ----
 * The batch job re-reads the whole window and judges every point against the stddev_samp of the WHOLE series
 * (anomaly_detection.py:664-684, 168-212).  A long-running detector appends: tad_state holds, per key, Spark's streaming
 * moments (n, avg, m2 — the same update as the batch job), the last EWMA value and the last
 * flowEndSeconds seen.  tad_run_stream aggregates ONE new batch (Stage 0 as in tad_run), continues both recurrences over
 * the key's new points in time order and emits the points with |x - ewma| > stddev_samp(points seen so far, this one
 * included) — the running sigma, the only one an append-only detector can know.  After the last batch the state equals
 * what the batch job computes over the concatenated table bit for bit (same operations in the same order): n, avg, m2
 * give its stddev_samp, ewma its last EWMA value.  A row not newer than its key's last_t is rejected
 * (TAD_ERR_INVALID_ARGUMENT) and the state is left untouched.  job->algo must be TAD_ALGO_EWMA, and cols->num_keys must
 * EQUAL the num_keys the state was created with (a batch addresses the state's whole key space; keys without rows in
 * the batch keep their state) — anything else is TAD_ERR_INVALID_ARGUMENT. */
typedef struct tad_state tad_state;
int tad_state_create(tad_engine *e, uint64_t num_keys, tad_state **out);
void tad_state_destroy(tad_engine *e, tad_state *s);
/* copies the state to HOST arrays of num_keys entries each (any may be NULL) */
int tad_state_export(tad_engine *e, const tad_state *s, uint32_t *n, double *avg, double *m2, double *ewma, int64_t *last_t);
int tad_run_stream(tad_engine *e, tad_state *s, const tad_job *job, const tad_columns *cols, tad_mem out_memory,
                   tad_result **out);

/* Stage counter for Status.CompletedStages / TotalStages (controller.go:426-453); callable while
 * tad_run executes on another thread.  tad_progress: the sum over the jobs in flight (with none: the job that finished last).
 * tad_job_progress (ABI 12): the job whose tad_job.id equals `id`; *total = 0 when no such job is in flight (finished or not yet
 * admitted).  tad_jobs_in_flight: contexts busy right now. */
int tad_progress(tad_engine *e, int32_t *done, int32_t *total);
int tad_job_progress(tad_engine *e, const char *id, int32_t *done, int32_t *total);
int tad_jobs_in_flight(tad_engine *e);

/* ---- per-series entry points: the reference's pure functions, one key, values in time order.
 * Same device kernels as tad_run (a 1-key series table).  x, out arrays are HOST memory. ---- */
/* calculate_ewma (:146-165): out[n]. */
int tad_series_ewma(tad_engine *e, const uint64_t *x, uint64_t n, double alpha, double *out);
/* calculate_ewma_anomaly (:168-212): verdict[n] in {0,1}. has_stddev == 0 models stddev None. */
int tad_series_ewma_anomaly(tad_engine *e, const uint64_t *x, uint64_t n, double alpha,
                            int has_stddev, double stddev, uint8_t *verdict);
/* stddev_samp over the series (:674-684). *has_stddev = 0 when n < 2 (Spark returns null). */
int tad_series_stddev(tad_engine *e, const uint64_t *x, uint64_t n, int *has_stddev, double *stddev);
/* calculate_dbscan_anomaly (:325-349): verdict[n] = (label == -1). */
int tad_series_dbscan_anomaly(tad_engine *e, const uint64_t *x, uint64_t n, double eps,
                              int min_samples, uint8_t *verdict);
/* DropDetection.end_partition (drop_detection_udf.py:42-56) on one partition: *has_result = 0 when n < min_samples. */
int tad_series_drop(tad_engine *e, const uint64_t *x, uint64_t n, double nsigma, int min_samples, int *has_result,
                    double *mean, double *stddev, uint8_t *verdict);
/* calculate_arima (:215-264): out[n]; *has_result = 0 reproduces the `return None` cases. */
int tad_series_arima(tad_engine *e, const uint64_t *x, uint64_t n, int maxiter, int *has_result,
                     double *out);
/* calculate_arima_anomaly (:267-309): verdict[n]; *n_verdict = 1 and verdict[0] = 0 when ARIMA
 * returned None (:284-287). */
int tad_series_arima_anomaly(tad_engine *e, const uint64_t *x, uint64_t n, int maxiter,
                             int has_stddev, double stddev, uint8_t *verdict, uint64_t *n_verdict);

/* ---- deterministic synthetic flow table, generated straight into HBM ----
 * Rows [first_row, first_row + n_rows) of the table (seed, num_keys, n_buckets); the three output
 * arrays are DEVICE memory with n_rows entries.  Definition: theia_amd/csrc/tad_synth.hip. */
int tad_synth_generate(tad_engine *e, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                       uint64_t num_keys, uint64_t n_buckets, uint64_t *key_id,
                       int64_t *flow_end_s, uint64_t *value);

/* ---- device memory helpers for hosts without a HIP binding (cgo) ---- */
int tad_device_alloc(tad_engine *e, uint64_t bytes, void **ptr);
int tad_device_free(tad_engine *e, void *ptr);
int tad_copy_to_device(tad_engine *e, void *dst, const void *src, uint64_t bytes);
int tad_copy_to_host(tad_engine *e, void *dst, const void *src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* THEIA_TAD_H */
