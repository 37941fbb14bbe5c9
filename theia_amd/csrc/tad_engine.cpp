// tad_engine.cpp — the engine's life cycle, its pool of job contexts, device memory and the small helpers of include/tad.h (see tad_engine.h).
#include "tad_engine.h"

using namespace tad;
using namespace tadh;

namespace tadh {

thread_local std::string g_static_err;

bool plan_ok(const tad_plan &p) {
  return p.stage0 >= 0 && p.stage0 <= 2 && p.partition_pass >= 0 && p.partition_pass <= 3 && p.histogram >= 0 && p.histogram <= 2 && p.sparse >= 0 &&
         p.sparse <= 2 && p.sparse_classes >= 0 && p.sparse_classes <= 1 && p.ewma_emit >= 0 && p.ewma_emit <= 1 && p.ewma_emit_rows <= 4096 && p.tile_cells >= 0 && p.tile_cells <= 1 && p.sparse_sort >= 0 && p.sparse_sort <= 2 && p.reserved0 == 0 && p.reserved1 == 0;
}

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

}  // namespace tadh

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

extern "C" {

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
