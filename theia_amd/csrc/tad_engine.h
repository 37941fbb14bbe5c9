// tad_engine.h — PRIVATE header of libtad_mi355x.so's host side: the engine, its pool of job contexts, and what the four translation
// units of the C ABI share (tad_engine.cpp: life cycle, pool, memory; tad_capi.cpp: the job; tad_capi_ingest.cpp: the ingest entry
// points; tad_capi_series.cpp: the per-series entry points).  HIP only: there is no CPU fallback in this library (the CPU oracle under
// oracle/ is test infrastructure and is never linked or called from here).
//
// Threading (controller.go:199-201 runs four workers, Spark ran one pod per job): an engine owns a small POOL of job
// contexts.  A context is everything one job in flight needs — a HIP stream (two: normal and low priority), its events, its pinned
// read-back blocks and its grow-only workspace buffers — so jobs submitted from different threads run concurrently on the GPU, each
// on its own stream, and never touch each other's memory.  tad_run takes an idle context (creating one up to
// tad_engine_opts.max_jobs_in_flight, else waiting), runs, and gives it back.  Serial callers always get context 0 and see the
// behaviour of the single-mutex engine of ABI <= 11.
#ifndef THEIA_TAD_ENGINE_H
#define THEIA_TAD_ENGINE_H

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "tad_internal.h"

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct FreeBlock {
  void *p;
  size_t cap;
};

struct JobCtx;

struct tad_engine {
  int device = 0;
  uint64_t ws_limit = 0;       // per job in flight
  int max_ctx = 1;
  hipStream_t user_stream = nullptr;   // tad_engine_opts.stream: context 0 runs on it (and the pool has that one context)
  int prio_normal = 0, prio_low = 0, prio_high = 0;   // hipDeviceGetStreamPriorityRange: ARIMA jobs (seconds of FP64) run on the low-priority stream of their
                                       // context so that the short HBM-bound jobs of other contexts are dispatched ahead of their workgroups
  std::mutex mu;               // protects plan, ctxs, the busy flags and last_done / last_total
  std::condition_variable cv;  // a context became idle
  tad_plan plan{};             // plan overrides (tests / A-B measurements); all zero = the engine decides
  std::vector<JobCtx *> ctxs;
  int creating = 0;            // contexts being created outside the lock (counted against max_ctx)
  int32_t last_done = 0, last_total = 0;   // progress of the job that finished last (tad_progress with nothing in flight)
  std::mutex err_mu;           // protects err
  std::string err;
  // Whole-CU jobs vs. the ARIMA fit.  A workgroup of pass B / pass C needs a whole CU; the fit kernel of an ARIMA job in flight on another
  // context keeps every CU populated with long-lived wavefronts, so such a workgroup would wait for the fit's whole grid (212 ms measured)
  // whatever the stream priorities.  pause_count = jobs in flight that are in a whole-CU phase; *pause_dev (DEVICE memory) is 0 / non-zero
  // accordingly, written on the 0 <-> 1 transitions by a 4-byte fill on signal_stream (one stream, under pause_mu: the writes cannot
  // pass each other).  The fit polls it every optimiser cycle and suspends while it is raised (tad_arima.hip); its host loop relaunches it
  // (detect_and_count).  The word lives in device memory because 2048 wavefronts polling a page-locked HOST word once per cycle (1.6e7
  // reads/s over the host link) doubled the fit's time (C3 266 -> 492 ms, profiles/r6_a4_*); an agent-scope load from HBM costs nothing
  // measurable.
  std::mutex pause_mu;
  int pause_count = 0;         // under pause_mu (read without it by the fit's host loop: a hint, re-checked by the kernel)
  int *pause_dev = nullptr;
  hipStream_t signal_stream = nullptr;
  std::mutex pool_mu;          // protects free_blocks
  std::vector<FreeBlock> free_blocks;  // recycled device result blocks (a result may be freed from any thread)
};

// One job in flight.  Everything below is touched by the thread that holds the context only (busy == true), except done / total / id.
struct PauseHold;
struct JobCtx {
  tad_engine *eng = nullptr;
  PauseHold *hold = nullptr;   // the running job's claim on whole CUs (run_job); NULL for the small entry points
  int index = 0;               // position in eng->ctxs (tad_stats.job_context)
  bool busy = false;           // under eng->mu
  int device = 0;
  hipStream_t stream = nullptr;        // the stream of the running job: stream_normal or stream_low
  hipStream_t stream_normal = nullptr, stream_low = nullptr;
  bool own_streams = false;
  uint64_t ws_limit = 0;
  tad_plan plan{};             // the engine's plan when the job was admitted
  std::atomic<int32_t> done{0}, total{0};
  char id[64] = {};            // tad_job.id of the running job (tad_job_progress); under eng->mu
  // grow-only device scratch
  DevBuf grid_val, grid_flag, sigma, n_pts, n_anom, off, scan_scratch, calc, counters, meta, aux, key_mean, key_m2;   // counters: the job tail (kTailBytes)
  DevBuf in_key, in_key2, in_te, in_ts, in_val;
  DevBuf rcp_table;           // rcp_table[n] = RN(1/n), n = 0..rcp_n-1
  uint64_t rcp_n = 0;
  DevBuf binhist, part_total, part_start, part_offs32, recs, ovf, slices;  // Stage 0 v2 (ovf: 8-byte count + overflow records)
  DevBuf sp_comp_a, sp_comp_b, sp_val_a, sp_val_b, sp_temp, sp_first, sp_times;  // Stage 0 sparse (sort + rank grid)
  DevBuf sp_cls;                                                                  // Stage 0 sparse, length classes: per-key class arrays
  int arima_relaunches = 0;       // times the running job's ARIMA fit was relaunched after it had yielded to whole-CU jobs (tad_stats.arima_relaunches)
  bool sp_by_partition = false;   // the running job's sparse Stage 0 went through the partition pass + LDS sort (stage0_path 8 / 9 / 10 instead of 4 / 6 / 7)
  DevBuf part_fin;                                                                // Stage 0 v2, sampled histogram: final cursors of the (workgroup, partition) regions
  DevBuf ovf_keys;                                                                // Stage 0 v2, settle mode: bitmap of the keys with a value on the overflow list
  hipEvent_t ev[8] = {};
  tad::MetaPartial *meta_host = nullptr;    // pinned
  // The job's tail — what the host reads when a job's kernels are done — is ONE block on the device (e->counters: DevCounters |
  // row total | overflow-list count | pad to 128 B | kMomentBlocks moment partials) and ONE pinned block here: one copy per job.
  unsigned char *tail_host = nullptr;        // pinned, kTailBytes
  tad::DevCounters *ctr_host = nullptr;           // = tail_host
  unsigned long long *total_host = nullptr;  // = tail_host + 64
  tad::Moments *moments_host = nullptr;           // = tail_host + 128
  // what the last job of this context learnt about its table, reused when the next job has the same shape (nothing speculative: both only
  // skip an attempt that is known to fail)
  struct Learnt {
    bool valid = false;
    uint64_t n = 0, K = 0;
    bool has2 = false;
    int algo = 0, op = 0;
    bool exact_hist = false;   // the sampled histogram proved too optimistic for this table: go straight to the exact one ...
    uint32_t exact_uses = 0, exact_backoff = 8;   // ... for `exact_backoff` jobs, then the sample is tried again (a sorted table may be followed by
                                                  // hashed ones of the same shape); a probe that fails doubles the interval, up to 64
    bool wide_tiles = false;   // 32-bit tile cells overflowed the list for this table: go straight to 8-byte cells
  } learnt;
};

// per-key running state of the streaming EWMA detector: two copies (the count pass writes the candidate next state,

// per-key running state of the streaming EWMA detector: two copies (the count pass writes the candidate next state,
// it becomes current only when the batch succeeds)
struct tad_state {
  uint64_t K = 0;
  void *block[2] = {nullptr, nullptr};
  int cur = 0;
  mutable std::mutex mu;     // batches of one state are serial (tad_run_stream from two threads on one state)
};

// A job's claim on whole CUs: raised when its Stage 0 takes the partition path, dropped while its own ARIMA fit runs, dropped for good when
// the job returns.
struct PauseHold {
  tad_engine *eng;
  bool held = false;
  explicit PauseHold(tad_engine *e) : eng(e) {}
  void acquire() {
    if (held || !eng->pause_dev) return;
    std::lock_guard<std::mutex> lk(eng->pause_mu);
    if (eng->pause_count++ == 0) (void)hipMemsetAsync(eng->pause_dev, 1, 4, eng->signal_stream);
    held = true;
  }
  void release() {
    if (!held) return;
    std::lock_guard<std::mutex> lk(eng->pause_mu);
    if (--eng->pause_count == 0) (void)hipMemsetAsync(eng->pause_dev, 0, 4, eng->signal_stream);
    held = false;
  }
  ~PauseHold() { release(); }
  PauseHold(const PauseHold &) = delete;
  PauseHold &operator=(const PauseHold &) = delete;
};

namespace tadh {
using namespace tad;

extern thread_local std::string g_static_err;   // tad_last_error(NULL): the message of a failed tad_engine_create

// the job tail: what the host reads when a job's kernels are done, ONE block on the device and one pinned block on the host
constexpr size_t kTailCtr = 0, kTailTotal = 64, kTailOvfCount = 72, kTailMoments = 128;
constexpr size_t kTailBytes = kTailMoments + sizeof(Moments) * kMomentBlocks;
inline unsigned long long *dev_total(JobCtx *e) { return reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(e->counters.p) + kTailTotal); }
inline unsigned long long *dev_ovf_count(JobCtx *e) { return reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(e->counters.p) + kTailOvfCount); }
inline Moments *dev_moments(JobCtx *e) { return reinterpret_cast<Moments *>(static_cast<unsigned char *>(e->counters.p) + kTailMoments); }

constexpr int kMetaBlocks = 2048;
constexpr int kSampleBlockShift = 7;   // (C4 with 1024-key blocks and a sampled histogram: pass A 0.21 -> 0.10 ms but pass C 0.68 -> 0.86 ms, profiles/r3_v2_c4_keyblock_ab.log)
constexpr int kDefaultJobsInFlight = 4;   // controller.go:199-201 / pkg/controller/util.go:43: four workers
constexpr int kMaxJobsInFlight = 16;
constexpr uint32_t kOverflowCap = 1u << 20;  // Stage 0 v2: rows with a value >= 2^49 per run before falling back to v1

bool plan_ok(const tad_plan &p);
int fail(tad_engine *e, int code, const char *fmt, ...);
int fail(JobCtx *c, int code, const char *fmt, ...);
int fail(std::nullptr_t, int code, const char *fmt, ...);

#define HIP_TRY(e, call)                                                                         \
  do {                                                                                           \
    hipError_t err__ = (call);                                                                   \
    if (err__ != hipSuccess)                                                                     \
      return fail((e), err__ == hipErrorOutOfMemory ? TAD_ERR_OUT_OF_MEMORY : TAD_ERR_HIP,       \
                  "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
  } while (0)


// every grow-only buffer of a context, for trimming and teardown
template <typename F> void for_each_buf(JobCtx *c, F f) {
  DevBuf *bufs[] = {&c->grid_val, &c->grid_flag, &c->sigma, &c->n_pts, &c->n_anom, &c->off, &c->scan_scratch, &c->calc, &c->counters, &c->meta, &c->aux,
                    &c->key_mean, &c->key_m2, &c->rcp_table, &c->binhist, &c->part_total, &c->part_start, &c->part_offs32, &c->recs, &c->ovf, &c->slices,
                    &c->sp_comp_a, &c->sp_comp_b, &c->sp_val_a, &c->sp_val_b, &c->sp_temp, &c->sp_first, &c->sp_times, &c->sp_cls, &c->part_fin, &c->ovf_keys,
                    &c->in_key, &c->in_key2, &c->in_te, &c->in_ts, &c->in_val};
  for (DevBuf *b : bufs) f(*b);
}

void drop_buffers(JobCtx *c);
void trim_idle(tad_engine *eng, JobCtx *self);
int ensure(JobCtx *e, DevBuf &b, size_t bytes);          // grow-only device scratch
void preload_code_objects();
JobCtx *ctx_create(tad_engine *eng, bool first);
void ctx_destroy(JobCtx *c);

// An idle context (the lowest-numbered one: a serial caller always gets context 0 and its warm buffers), a new one while the pool may
// grow, else wait.  low_priority: the job's kernels go to the context's low-priority stream (ARIMA).
struct Lease {
  tad_engine *eng;
  JobCtx *c = nullptr;
  Lease(tad_engine *eng_, const char *id = nullptr, bool low_priority = false) : eng(eng_) {
    std::unique_lock<std::mutex> lk(eng->mu);
    for (;;) {
      for (JobCtx *x : eng->ctxs)
        if (!x->busy) { c = x; break; }
      if (c) break;
      if ((int)eng->ctxs.size() + eng->creating < eng->max_ctx) {
        ++eng->creating;
        lk.unlock();      // (stream / pinned-memory creation outside the lock)
        hipSetDevice(eng->device);
        JobCtx *n = ctx_create(eng, false);
        lk.lock();
        --eng->creating;
        if (n) { n->index = (int)eng->ctxs.size(); eng->ctxs.push_back(n); c = n; break; }
        if (eng->ctxs.empty()) return;   // cannot happen (context 0 is made by tad_engine_create); c stays NULL
      }
      eng->cv.wait(lk);
    }
    c->busy = true;
    c->plan = eng->plan;
    c->done.store(0);
    c->total.store(0);
    memset(c->id, 0, sizeof c->id);
    if (id) strncpy(c->id, id, sizeof c->id - 1);
    c->stream = low_priority ? c->stream_low : c->stream_normal;
    c->hold = nullptr;
  }
  ~Lease() {
    if (!c) return;
    {
      std::lock_guard<std::mutex> lk(eng->mu);
      if (c->total.load() != 0) { eng->last_done = c->done.load(); eng->last_total = c->total.load(); }
      c->busy = false;
      c->id[0] = 0;
      c->hold = nullptr;
    }
    eng->cv.notify_one();
  }
  Lease(const Lease &) = delete;
  Lease &operator=(const Lease &) = delete;
};


// recycled device result blocks (engine-wide: a result is freed by whoever holds it)
void release_block(tad_engine *eng, void *p, size_t cap);
inline void release_block(JobCtx *e, void *p, size_t cap) { release_block(e->eng, p, cap); }

Lattice make_lattice(int64_t t0, int64_t step, uint64_t nb);
uint64_t host_gcd(uint64_t a, uint64_t b);

struct ResultBlock {
  void *base = nullptr;
  size_t cap = 0;
};

// every column of a result block starts on a 32-byte boundary (stride = rows rounded up to 4): k_emit stores four rows at a time
size_t result_bytes(uint64_t rows, bool with_anomaly);
void carve(void *base, uint64_t rows, bool with_anomaly, OutRows *o);
int alloc_device_block(JobCtx *e, size_t bytes, ResultBlock *rb);

struct ResultPriv {  // lives right behind the public struct
  tad_result pub;
  void *block;
  size_t block_cap;
};

struct PointsPriv {  // tad_points + its storage
  tad_points pub;
  void *block;
  size_t block_cap;
};

struct JobParams {
  tad_algo algo;
  double alpha, eps;
  int min_samples, maxiter;
  double drop_nsigma;
  int drop_min_samples;
  bool all_points;
  bool lazy_sigma = false;   // set by detect_and_count: the stddev column is computed by the emit kernel (DBSCAN jobs)
  bool settled = false;      // set by Stage 0: pass C ran in settle mode (SettleArgs) — the DBSCAN scan only walks the keys it marked
};

// (tad_capi.cpp) shared with the per-series entry points
int ensure_rcp_table(JobCtx *e, uint64_t T);
int ensure_key_buffers(JobCtx *e, uint64_t K);
void emit_rows(JobCtx *e, Grid g, Lattice L, const JobParams &jp, OutRows out, uint64_t rows = 0);

}  // namespace tadh

#endif  // THEIA_TAD_ENGINE_H
