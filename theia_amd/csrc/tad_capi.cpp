// tad_capi.cpp — the C ABI of include/tad.h on top of the gfx950 kernels.  HIP only: there is no
// CPU fallback in this library (the CPU oracle under oracle/ is test infrastructure and is never
// linked or called from here).
//
// Threading (controller.go:199-201 runs four workers, Spark ran one pod per job): an engine owns a small POOL of job
// contexts.  A context is everything one job in flight needs — a HIP stream (two: normal and low priority), its events, its pinned
// read-back blocks and its grow-only workspace buffers — so jobs submitted from different threads run concurrently on the GPU, each
// on its own stream, and never touch each other's memory.  tad_run takes an idle context (creating one up to
// tad_engine_opts.max_jobs_in_flight, else waiting), runs, and gives it back.  Serial callers always get context 0 and see the
// behaviour of the single-mutex engine of ABI <= 11.
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

using namespace tad;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct FreeBlock {
  void *p;
  size_t cap;
};

thread_local std::string g_static_err;

}  // namespace

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
  MetaPartial *meta_host = nullptr;    // pinned
  // The job's tail — what the host reads when a job's kernels are done — is ONE block on the device (e->counters: DevCounters |
  // row total | overflow-list count | pad to 128 B | kMomentBlocks moment partials) and ONE pinned block here: one copy per job.
  unsigned char *tail_host = nullptr;        // pinned, kTailBytes
  DevCounters *ctr_host = nullptr;           // = tail_host
  unsigned long long *total_host = nullptr;  // = tail_host + 64
  Moments *moments_host = nullptr;           // = tail_host + 128
  // what the last job of this context learnt about its table, reused when the next job has the same shape (nothing speculative: both only
  // skip an attempt that is known to fail)
  struct Learnt {
    bool valid = false;
    uint64_t n = 0, K = 0;
    bool has2 = false;
    int algo = 0, op = 0;
    bool exact_hist = false;   // the sampled histogram proved too optimistic for this table: go straight to the exact one
    bool wide_tiles = false;   // 32-bit tile cells overflowed the list for this table: go straight to 8-byte cells
  } learnt;
};

// per-key running state of the streaming EWMA detector: two copies (the count pass writes the candidate next state,
// it becomes current only when the batch succeeds)
namespace {
constexpr size_t kTailCtr = 0, kTailTotal = 64, kTailOvfCount = 72, kTailMoments = 128;
constexpr size_t kTailBytes = kTailMoments + sizeof(Moments) * kMomentBlocks;
inline unsigned long long *dev_total(JobCtx *e) { return reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(e->counters.p) + kTailTotal); }
inline unsigned long long *dev_ovf_count(JobCtx *e) { return reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(e->counters.p) + kTailOvfCount); }
inline Moments *dev_moments(JobCtx *e) { return reinterpret_cast<Moments *>(static_cast<unsigned char *>(e->counters.p) + kTailMoments); }
}  // namespace

struct tad_state {
  uint64_t K = 0;
  void *block[2] = {nullptr, nullptr};
  int cur = 0;
  mutable std::mutex mu;     // batches of one state are serial (tad_run_stream from two threads on one state)
};

namespace {

constexpr int kMetaBlocks = 2048;
constexpr int kSampleBlockShift = 7;   // (C4 with 1024-key blocks and a sampled histogram: pass A 0.21 -> 0.10 ms but pass C 0.68 -> 0.86 ms, profiles/r3_v2_c4_keyblock_ab.log)
constexpr int kDefaultJobsInFlight = 4;   // controller.go:199-201 / pkg/controller/util.go:43: four workers
constexpr int kMaxJobsInFlight = 16;

bool plan_ok(const tad_plan &p) {
  return p.stage0 >= 0 && p.stage0 <= 2 && p.partition_pass >= 0 && p.partition_pass <= 3 && p.histogram >= 0 && p.histogram <= 2 && p.sparse >= 0 &&
         p.sparse <= 2 && p.sparse_classes >= 0 && p.sparse_classes <= 1 && p.ewma_emit >= 0 && p.ewma_emit <= 1 && p.ewma_emit_rows <= 4096 && p.tile_cells >= 0 && p.tile_cells <= 1 && p.sparse_sort >= 0 && p.sparse_sort <= 2 && p.reserved0 == 0 && p.reserved1 == 0;
}
constexpr uint32_t kOverflowCap = 1u << 20;  // Stage 0 v2: rows with a value >= 2^49 per run before falling back to v1

int vfail(tad_engine *e, int code, const char *fmt, va_list ap) {
  char buf[512];
  vsnprintf(buf, sizeof buf, fmt, ap);
  if (e) {
    std::lock_guard<std::mutex> lk(e->err_mu);
    e->err = buf;
  } else {
    g_static_err = buf;
  }
  return code;
}
int fail(tad_engine *e, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  const int rc = vfail(e, code, fmt, ap);
  va_end(ap);
  return rc;
}
int fail(JobCtx *c, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  const int rc = vfail(c ? c->eng : nullptr, code, fmt, ap);
  va_end(ap);
  return rc;
}
int fail(std::nullptr_t, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  const int rc = vfail(nullptr, code, fmt, ap);
  va_end(ap);
  return rc;
}

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

// give the workspace of a context back to the device (the context is idle and held by the caller, or is being destroyed)
void drop_buffers(JobCtx *c) {
  for_each_buf(c, [](DevBuf &b) {
    if (b.p) hipFree(b.p);
    b = DevBuf{};
  });
  c->rcp_n = 0;
}

// An allocation failed: the idle contexts of the engine and the recycled result blocks give their memory back, then the caller retries once.
// (Contexts are grow-only for speed; the sum over a pool may exceed what a single big job plus the others' leftovers can share.)
void trim_idle(tad_engine *eng, JobCtx *self) {
  std::vector<JobCtx *> held;
  {
    std::lock_guard<std::mutex> lk(eng->mu);
    for (JobCtx *c : eng->ctxs)
      if (c != self && !c->busy) { c->busy = true; held.push_back(c); }
  }
  for (JobCtx *c : held) drop_buffers(c);
  {
    std::lock_guard<std::mutex> lk(eng->pool_mu);
    for (auto &fb : eng->free_blocks) hipFree(fb.p);
    eng->free_blocks.clear();
  }
  (void)hipGetLastError();
  {
    std::lock_guard<std::mutex> lk(eng->mu);
    for (JobCtx *c : held) c->busy = false;
  }
  eng->cv.notify_all();
}

int ensure(JobCtx *e, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return TAD_OK;
  if (b.p) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipFree(b.p));
    b = DevBuf{};
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t r = hipMalloc(&b.p, want);
  if (r != hipSuccess) {
    want = bytes;
    r = hipMalloc(&b.p, want);
  }
  if (r != hipSuccess) {
    (void)hipGetLastError();
    trim_idle(e->eng, e);
    r = hipMalloc(&b.p, want);
  }
  if (r != hipSuccess) {
    b.p = nullptr;
    (void)hipGetLastError();
    return fail(e, TAD_ERR_OUT_OF_MEMORY, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(r));
  }
  b.cap = want;
  return TAD_OK;
}

// Resolve one kernel of every translation unit: the lazy loader brings the unit's code object onto the device.
void preload_code_objects() {
  const void *anchors[] = {code_anchor_arima(), code_anchor_dbscan(), code_anchor_drop(), code_anchor_factorize(), code_anchor_ingest(), code_anchor_kernels(), code_anchor_shard(), code_anchor_sparse(), code_anchor_stage0_part(), code_anchor_synth()};
  for (const void *k : anchors) {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, k);
  }
  (void)hipGetLastError();
}

// ---- the pool ----
JobCtx *ctx_create(tad_engine *eng, bool first) {
  JobCtx *c = new (std::nothrow) JobCtx();
  if (!c) return nullptr;
  c->eng = eng;
  c->device = eng->device;
  c->ws_limit = eng->ws_limit;
  bool ok = true;
  if (first && eng->user_stream) {
    c->stream_normal = c->stream_low = eng->user_stream;
  } else {
    c->own_streams = true;
    ok = hipStreamCreateWithPriority(&c->stream_normal, hipStreamNonBlocking, eng->prio_normal) == hipSuccess;
    if (ok && eng->prio_low != eng->prio_normal) ok = hipStreamCreateWithPriority(&c->stream_low, hipStreamNonBlocking, eng->prio_low) == hipSuccess;
    else c->stream_low = c->stream_normal;
  }
  c->stream = c->stream_normal;
  for (auto &ev : c->ev) ok = ok && hipEventCreate(&ev) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&c->meta_host), sizeof(MetaPartial) * kMetaBlocks, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&c->tail_host), kTailBytes, hipHostMallocDefault) == hipSuccess;
  if (ok) {
    memset(c->tail_host, 0, kTailBytes);
    c->ctr_host = reinterpret_cast<DevCounters *>(c->tail_host + kTailCtr);
    c->total_host = reinterpret_cast<unsigned long long *>(c->tail_host + kTailTotal);
    c->moments_host = reinterpret_cast<Moments *>(c->tail_host + kTailMoments);
  }
  if (!ok) {
    (void)hipGetLastError();
    for (auto &ev : c->ev) if (ev) hipEventDestroy(ev);
    if (c->meta_host) hipHostFree(c->meta_host);
    if (c->tail_host) hipHostFree(c->tail_host);
    if (c->own_streams) {
      if (c->stream_low && c->stream_low != c->stream_normal) hipStreamDestroy(c->stream_low);
      if (c->stream_normal) hipStreamDestroy(c->stream_normal);
    }
    delete c;
    return nullptr;
  }
  return c;
}

void ctx_destroy(JobCtx *c) {
  if (c->stream_normal) hipStreamSynchronize(c->stream_normal);
  if (c->stream_low && c->stream_low != c->stream_normal) hipStreamSynchronize(c->stream_low);
  drop_buffers(c);
  for (auto &ev : c->ev) if (ev) hipEventDestroy(ev);
  if (c->meta_host) hipHostFree(c->meta_host);
  if (c->tail_host) hipHostFree(c->tail_host);
  if (c->own_streams) {
    if (c->stream_low && c->stream_low != c->stream_normal) hipStreamDestroy(c->stream_low);
    if (c->stream_normal) hipStreamDestroy(c->stream_normal);
  }
  delete c;
}

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
      if ((int)eng->ctxs.size() < eng->max_ctx) {
        lk.unlock();      // (stream / pinned-memory creation outside the lock)
        hipSetDevice(eng->device);
        JobCtx *n = ctx_create(eng, false);
        lk.lock();
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

}  // namespace

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

namespace {

// recycled device result blocks (engine-wide: a result is freed by whoever holds it)
void release_block(tad_engine *eng, void *p, size_t cap) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(eng->pool_mu);
    if (eng->free_blocks.size() < 16) { eng->free_blocks.push_back({p, cap}); return; }
  }
  hipSetDevice(eng->device);
  hipFree(p);
}
inline void release_block(JobCtx *e, void *p, size_t cap) { release_block(e->eng, p, cap); }

Lattice make_lattice(int64_t t0, int64_t step, uint64_t nb) {
  Lattice L;
  L.t0 = t0;
  L.step = step < 1 ? 1 : step;
  L.nb = nb;
  L.magic = 0;
  if (L.step == 1) {
    L.mode = 0;
  } else {
    // ceil(2^64 / step) = floor((2^64 - 1) / step) + 1 (step >= 2 never divides 2^64 - 1 + 1 exactly
    // unless it is a power of two, for which floor((2^64-1)/step) + 1 = 2^64/step as well)
    L.magic = UINT64_MAX / (uint64_t)L.step + 1;
    // the multiply-high quotient is exact for dividends < 2^32 and divisors < 2^32
    const bool small = (uint64_t)L.step < (1ull << 32) &&
                       (nb == 0 || (nb - 1) <= (UINT32_MAX / (uint64_t)L.step));
    L.mode = small ? 1 : 2;
  }
  return L;
}

uint64_t host_gcd(uint64_t a, uint64_t b) {
  while (b) { uint64_t r = a % b; a = b; b = r; }
  return a;
}

struct ResultBlock {
  void *base = nullptr;
  size_t cap = 0;
};

// every column starts on a 32-byte boundary (stride = rows rounded up to 4): k_emit stores four rows at a time
size_t result_bytes(uint64_t rows, bool with_anomaly) {
  const uint64_t r = ((rows ? rows : 1) + 3) & ~3ull;
  return (size_t)r * 8 * 5 + (with_anomaly ? (size_t)((r + 15) & ~15ull) : 0);
}

void carve(void *base, uint64_t rows, bool with_anomaly, OutRows *o) {
  const uint64_t r = ((rows ? rows : 1) + 3) & ~3ull;
  unsigned char *p = static_cast<unsigned char *>(base);
  o->key_id = reinterpret_cast<unsigned long long *>(p); p += r * 8;
  o->flow_end_s = reinterpret_cast<long long *>(p); p += r * 8;
  o->throughput = reinterpret_cast<double *>(p); p += r * 8;
  o->algo_calc = reinterpret_cast<double *>(p); p += r * 8;
  o->stddev = reinterpret_cast<double *>(p); p += r * 8;
  o->anomaly = with_anomaly ? p : nullptr;
}

int alloc_device_block(JobCtx *e, size_t bytes, ResultBlock *rb) {
  {
    std::lock_guard<std::mutex> lk(e->eng->pool_mu);
    std::vector<FreeBlock> &fb = e->eng->free_blocks;
    for (size_t i = 0; i < fb.size(); ++i) {
      if (fb[i].cap >= bytes && fb[i].cap <= 2 * bytes + (1 << 20)) {
        rb->base = fb[i].p;
        rb->cap = fb[i].cap;
        fb.erase(fb.begin() + i);
        return TAD_OK;
      }
    }
  }
  void *p = nullptr;
  hipError_t r = hipMalloc(&p, bytes);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    trim_idle(e->eng, e);
    r = hipMalloc(&p, bytes);
  }
  if (r != hipSuccess) { (void)hipGetLastError(); return fail(e, TAD_ERR_OUT_OF_MEMORY, "hipMalloc(result, %zu) failed: %s", bytes, hipGetErrorString(r)); }
  rb->base = p;
  rb->cap = bytes;
  return TAD_OK;
}

struct ResultPriv {  // lives right behind the public struct
  tad_result pub;
  void *block;
  size_t block_cap;
};

}  // namespace

extern "C" {

int tad_abi_version(void) { return TAD_ABI_VERSION; }

const char *tad_last_error(tad_engine *e) {
  if (!e) return g_static_err.c_str();
  std::lock_guard<std::mutex> lk(e->err_mu);
  static thread_local std::string copy;
  copy = e->err;
  return copy.c_str();
}

int tad_engine_create(const tad_engine_opts *opts, tad_engine **out) {
  if (!out) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_engine_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t r = hipGetDeviceCount(&ndev);
  if (r != hipSuccess || ndev == 0)
    return fail(nullptr, TAD_ERR_NO_DEVICE, "no HIP device available (%s)", r != hipSuccess ? hipGetErrorString(r) : "count = 0");
  const int dev = opts ? opts->device : 0;
  if (dev < 0 || dev >= ndev) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "device %d out of range (have %d)", dev, ndev);
  if (opts && !plan_ok(opts->plan)) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_engine_create: a tad_plan field is out of range");
  if (opts && (opts->max_jobs_in_flight < 0 || opts->max_jobs_in_flight > kMaxJobsInFlight || opts->reserved != 0))
    return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_engine_create: max_jobs_in_flight must be 0 (default %d) .. %d", kDefaultJobsInFlight, kMaxJobsInFlight);
  tad_engine *e = new (std::nothrow) tad_engine();
  if (!e) return fail(nullptr, TAD_ERR_OUT_OF_MEMORY, "out of host memory");
  e->device = dev;
  if (hipSetDevice(dev) != hipSuccess) { delete e; return fail(nullptr, TAD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", dev); }
  e->user_stream = opts ? static_cast<hipStream_t>(opts->stream) : nullptr;
  // a caller's stream orders the engine's work with the caller's own: one context, on that stream
  e->max_ctx = e->user_stream ? 1 : ((opts && opts->max_jobs_in_flight) ? opts->max_jobs_in_flight : kDefaultJobsInFlight);
  {
    int least = 0, greatest = 0;   // numerically lower = higher priority
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    e->prio_low = least;
    e->prio_normal = greatest < least ? least - 1 : least;   // one step above the lowest: ordinary (default) priority where the range has three levels
    if (e->prio_normal < greatest) e->prio_normal = greatest;
    e->prio_high = greatest;
  }
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  e->ws_limit = (opts && opts->workspace_limit) ? opts->workspace_limit : (uint64_t)(free_b / 4 * 3);
  if (opts) e->plan = opts->plan;
  // the pause word and the stream its writes go through (highest priority: a 4-byte fill must not queue behind anything)
  if (hipMalloc(reinterpret_cast<void **>(&e->pause_dev), 256) != hipSuccess || hipMemset(e->pause_dev, 0, 256) != hipSuccess ||
      hipStreamCreateWithPriority(&e->signal_stream, hipStreamNonBlocking, e->prio_high) != hipSuccess) {
    (void)hipGetLastError();     // (without it ARIMA fits never yield: the behaviour of ABI <= 11)
    if (e->pause_dev) hipFree(e->pause_dev);
    e->pause_dev = nullptr;
    e->signal_stream = nullptr;
  }
  JobCtx *c0 = ctx_create(e, true);
  if (!c0) {
    if (e->signal_stream) hipStreamDestroy(e->signal_stream);
    if (e->pause_dev) hipFree(e->pause_dev);
    delete e;
    return fail(nullptr, TAD_ERR_OUT_OF_MEMORY, "stream / pinned host allocation failed");
  }
  e->ctxs.push_back(c0);
  // The code objects of the library load lazily, on the first launch out of each translation unit: ~3.5 ms of the first job of a process
  // (profiles/r6_a1_cold_hip_api_stats.csv: 1.5 ms inside hipLaunchKernel, 1.9 ms inside hipFuncSetAttribute).  Touch one kernel of every
  // unit here, where the ~100 ms of runtime initialisation are being paid anyway.
  preload_code_objects();
  (void)hipGetLastError();
  *out = e;
  return TAD_OK;
}

void tad_engine_destroy(tad_engine *e) {
  if (!e) return;
  hipSetDevice(e->device);
  {
    std::unique_lock<std::mutex> lk(e->mu);   // (destroying an engine with jobs in flight is a caller bug; wait for them rather than crash)
    e->cv.wait(lk, [&] { for (JobCtx *c : e->ctxs) if (c->busy) return false; return true; });
  }
  for (JobCtx *c : e->ctxs) ctx_destroy(c);
  for (auto &fb : e->free_blocks) hipFree(fb.p);
  if (e->signal_stream) { hipStreamSynchronize(e->signal_stream); hipStreamDestroy(e->signal_stream); }
  if (e->pause_dev) hipFree(e->pause_dev);
  delete e;
}

int tad_engine_set_plan(tad_engine *e, const tad_plan *plan) {
  if (!e) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_engine_set_plan: engine is NULL");
  tad_plan p{};
  if (plan) p = *plan;
  if (!plan_ok(p)) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_engine_set_plan: a tad_plan field is out of range");
  std::lock_guard<std::mutex> lk(e->mu);   // jobs admitted from now on see it; jobs in flight keep the plan they were admitted with
  e->plan = p;
  return TAD_OK;
}

int tad_progress(tad_engine *e, int32_t *done, int32_t *total) {
  if (!e) return TAD_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(e->mu);
  int32_t d = 0, t = 0;
  bool any = false;
  for (JobCtx *c : e->ctxs)
    if (c->busy && c->total.load() != 0) { d += c->done.load(); t += c->total.load(); any = true; }
  if (!any) { d = e->last_done; t = e->last_total; }
  if (done) *done = d;
  if (total) *total = t;
  return TAD_OK;
}

int tad_job_progress(tad_engine *e, const char *id, int32_t *done, int32_t *total) {
  if (!e || !id) return TAD_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(e->mu);
  for (JobCtx *c : e->ctxs)
    if (c->busy && c->total.load() != 0 && strncmp(c->id, id, sizeof c->id) == 0) {
      if (done) *done = c->done.load();
      if (total) *total = c->total.load();
      return TAD_OK;
    }
  if (done) *done = 0;     // no job with this id is in flight (finished, or not started yet): total = 0
  if (total) *total = 0;
  return TAD_OK;
}

int tad_jobs_in_flight(tad_engine *e) {
  if (!e) return 0;
  std::lock_guard<std::mutex> lk(e->mu);
  int n = 0;
  for (JobCtx *c : e->ctxs) n += c->busy ? 1 : 0;
  return n;
}

void tad_result_free(tad_engine *e, tad_result *r) {
  if (!r) return;
  ResultPriv *rp = reinterpret_cast<ResultPriv *>(r);
  if (rp->block) {
    if (r->memory == TAD_MEM_DEVICE && e) release_block(e, rp->block, rp->block_cap);
    else if (r->memory == TAD_MEM_DEVICE) hipFree(rp->block);
    else free(rp->block);
  }
  delete rp;
}

// ------------------------------------------------------------------------------------------------
// the detector pipeline over a filled grid (shared by tad_run and the tad_series_* entry points)
// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

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

void emit_rows(JobCtx *e, Grid g, Lattice L, const JobParams &jp, OutRows out, uint64_t rows = 0) {
  const int kind = jp.algo == TAD_ALGO_EWMA ? 0 : (jp.algo == TAD_ALGO_ARIMA ? 1 : (jp.algo == TAD_ALGO_DROP ? 3 : (jp.lazy_sigma ? 4 : 2)));
  // DBSCAN job: only keys of the detector's work list (still in e->aux) can have rows
  if (kind == 4 && !jp.all_points &&
      launch_emit_dbscan_list(e->stream, g, L, e->aux.p, static_cast<const uint32_t *>(e->n_anom.p), static_cast<const unsigned long long *>(e->off.p), out))
    return;
  launch_emit(e->stream, g, L, kind, jp.all_points, jp.alpha, static_cast<const double *>(e->sigma.p),
              static_cast<const uint32_t *>(e->n_pts.p), static_cast<const double *>(kind == 3 ? e->key_mean.p : e->calc.p),
              static_cast<const unsigned long long *>(e->off.p), out, rows, e->plan.ewma_emit, e->plan.ewma_emit_rows);
}

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

struct PointsPriv {  // tad_points + its storage
  tad_points pub;
  void *block;
  size_t block_cap;
};

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
  bool sparse_lsd = plan.sparse_sort == 1;   // set when the partition + LDS-sort form of the sparse Stage 0 met a heavy key bin or a value too wide for its records
  // what the context's last job learnt about a table of this shape: skip the attempt that is known to fail
  {
    const JobCtx::Learnt &lt = e->learnt;
    if (depth == 0 && lt.valid && lt.n == n && lt.K == K && lt.has2 == has2 && lt.algo == (int)job->algo && lt.op == (int)op_max) {
      if (lt.exact_hist) force_exact_hist = true;
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
    if (v2) {
      // pass A: lattice partials + per-workgroup key-bin histogram in one read of the key/time columns
      if ((rc = ensure(e, e->binhist, (size_t)pl.G * pl.nbins * 4)) != TAD_OK) return rc;
      hist_sampled = launch_meta_hist(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts, n, K, rf,
                                      pl, static_cast<MetaPartial *>(e->meta.p), static_cast<uint32_t *>(e->binhist.p), ctr,
                                      // small regions (many keys: C4 has ~50 records per workgroup and 128-key block) make pass C's walk
                                      // over the regions cost more than the sampled pass A saves: sample only when a region of a
                                      // 128-key block is expected to hold a few hundred records
                                      // (lat_mode 2 re-derives the lattice with k_meta, which reuses the partials buffer the sampling ratios live in)
                                      !force_exact_hist && lat_mode != 2 && sampled_slots_bound(n * (has2 ? 2 : 1), pl) < (1ull << 32) &&
                                          (plan.histogram == 2 ||      // (A/B: sampled wherever it is possible at all)
                                           n * (has2 ? 2 : 1) / ((uint64_t)pl.G * ((K >> kSampleBlockShift) ? (K >> kSampleBlockShift) : 1)) >= 384));
      meta_blocks = pl.G;
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
        launch_part_offsets(s, static_cast<const uint32_t *>(e->binhist.p), spl, offs32, static_cast<uint32_t *>(e->part_total.p), part_start, false,
                            static_cast<const MetaPartial *>(e->meta.p), n, slots, e->slices.p, Grid{});
        // (no overflow list: a value that does not fit the record raises DEV_ERR_OVERFLOW_LIST and the LSD sort redoes the job)
        launch_partition(s, (const uint64_t *)d_key, (const uint64_t *)d_key2, (const int64_t *)d_te, (const int64_t *)d_ts, (const uint64_t *)d_val, n, K,
                         rf, L, spl, offs32, part_start, e->recs.p, nullptr, dev_ovf_count(e), 0, ctr, nullptr, nullptr);
        launch_sparse_sort(s, e->recs.p, part_start, static_cast<const uint32_t *>(e->binhist.p), spl, K, L.step, op_max,
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
      launch_part_offsets(s, static_cast<const uint32_t *>(e->binhist.p), pl, offs32, static_cast<uint32_t *>(e->part_total.p), part_start,
                          hist_sampled, static_cast<const MetaPartial *>(e->meta.p), n, slots, e->slices.p, g);
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
      st.hist_sampled = (v2 && hist_sampled) ? 1 : 0;
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
    st.hist_sampled = (v2 && hist_sampled) ? 1 : 0;
    st.host_syncs = (hinted || empty) ? 2 : 3;
    st.job_context = e->index;
    st.arima_relaunches = e->arima_relaunches;
    hipEventElapsedTime(&st.ms_total, e->ev[0], e->ev[4]);
    if (depth == 0 && !points_mode && !stream) {
      JobCtx::Learnt &w = e->learnt;
      w.valid = true; w.n = n; w.K = K; w.has2 = has2; w.algo = (int)job->algo; w.op = (int)op_max;
      w.exact_hist = force_exact_hist && plan.histogram != 1;
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

int tad_factorize(tad_engine *eng, const tad_key_columns *kc, uint64_t *key_id, uint64_t *key_id2, uint64_t *first_row, uint64_t first_row_cap,
                  uint64_t *num_keys) {
  if (!eng) return fail(nullptr, TAD_ERR_INVALID_ARGUMENT, "tad_factorize: engine is NULL");
  if (!kc || !num_keys || kc->n_cols < 1 || kc->n_cols > kFzMaxCols || !kc->cols_a || (kc->n_rows && !key_id) || (kc->cols_b && kc->n_rows && !key_id2) ||
      (first_row_cap && !first_row))
    return fail(eng, TAD_ERR_INVALID_ARGUMENT, "tad_factorize: bad arguments (1..%d key columns, key_id / key_id2 / first_row buffers)", kFzMaxCols);
  const uint64_t n = kc->n_rows;
  const uint32_t sides = kc->cols_b ? 2 : 1;
  *num_keys = 0;
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
    launch_factorize(s, ca, ka, sides == 2 ? cb : nullptr, kb, n, kc->n_cols, slots, base, d_key, d_key2, d_fr, first_row_cap, nk_dev, &flags_dev);
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
    return TAD_OK;
  }
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

void tad_points_free(tad_engine *e, tad_points *p) {
  if (!p) return;
  PointsPriv *pp = reinterpret_cast<PointsPriv *>(p);
  if (pp->block) {
    if (p->memory == TAD_MEM_DEVICE && e) release_block(e, pp->block, pp->block_cap);
    else if (p->memory == TAD_MEM_DEVICE) hipFree(pp->block);
    else free(pp->block);
  }
  delete pp;
}

// ------------------------------------------------------------------------------------------------
// per-series entry points (a one-key table; same kernels)
// ------------------------------------------------------------------------------------------------
}  // extern "C"

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

int tad_host_alloc(tad_engine *e, uint64_t bytes, void **ptr) {
  if (!e || !ptr) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_host_alloc: bad arguments");
  HIP_TRY(e, hipSetDevice(e->device));
  hipError_t r = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
  if (r != hipSuccess) { (void)hipGetLastError(); return fail(e, TAD_ERR_OUT_OF_MEMORY, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(r)); }
  return TAD_OK;
}

int tad_host_free(tad_engine *e, void *ptr) {
  if (!e) return TAD_ERR_INVALID_ARGUMENT;
  HIP_TRY(e, hipSetDevice(e->device));
  if (ptr) HIP_TRY(e, hipHostFree(ptr));
  return TAD_OK;
}

int tad_device_alloc(tad_engine *e, uint64_t bytes, void **ptr) {
  if (!e || !ptr) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_device_alloc: bad arguments");
  HIP_TRY(e, hipSetDevice(e->device));
  hipError_t r = hipMalloc(ptr, bytes ? bytes : 1);
  if (r != hipSuccess) return fail(e, TAD_ERR_OUT_OF_MEMORY, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(r));
  return TAD_OK;
}

int tad_device_free(tad_engine *e, void *ptr) {
  if (!e) return TAD_ERR_INVALID_ARGUMENT;
  HIP_TRY(e, hipSetDevice(e->device));
  if (ptr) HIP_TRY(e, hipFree(ptr));     // (every entry point synchronises its stream before it returns: nothing of the engine's is pending on caller memory)
  return TAD_OK;
}

int tad_copy_to_device(tad_engine *e, void *dst, const void *src, uint64_t bytes) {
  if (!e || (bytes && (!dst || !src))) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_copy_to_device: bad arguments");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return TAD_OK;
}

int tad_copy_to_host(tad_engine *e, void *dst, const void *src, uint64_t bytes) {
  if (!e || (bytes && (!dst || !src))) return fail(e, TAD_ERR_INVALID_ARGUMENT, "tad_copy_to_host: bad arguments");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return TAD_OK;
}

}  // extern "C"
