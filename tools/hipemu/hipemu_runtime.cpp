// tools/hipemu runtime — DEVELOPMENT AID (see include/hip/hip_runtime.h).  Executes one workgroup at a time; every
// work-item is a fiber (own stack, hand-written x86-64 context switch: no system call per switch) that runs until it reaches a
// barrier, a wavefront operation or its end.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
#include <ucontext.h>   // (signal handler only)

#include <chrono>
#include <cstdio>
#include <map>
#include <vector>

struct hipemu_stream_s { int unused; };
struct hipemu_event_s { std::chrono::steady_clock::time_point t; };

// Save the callee-saved registers on the current stack, store the stack pointer in *from, continue on *to.
extern "C" void hipemu_switch(void **from, void **to);
asm(R"(
  .text
  .globl hipemu_switch
  .type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
  .size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {
namespace {

enum State { RUNNABLE, AT_BARRIER, AT_WAVE, DONE };
constexpr size_t kStack = 256 * 1024;
constexpr unsigned kWave = 64;

struct Fiber {
  void *sp;
  State state;
  Idx tid;
  // pending wavefront operation
  WaveOp op;
  uint64_t bits, result;
  int arg;
  uintptr_t site;
};

struct Exec {   // one per OS thread
  void *sched = nullptr;
  std::vector<Fiber> fibers;
  char *stacks = nullptr;
  size_t n_stacks = 0;
  Fiber *cur = nullptr;
  Idx bidx{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
  const std::function<void()> *body = nullptr;
  unsigned long long inactive_reads = 0;
};
thread_local Exec X;
const Idx kZero{0, 0, 0};

void fiber_main() {
  (*X.body)();
  X.cur->state = DONE;
  hipemu_switch(&X.cur->sp, &X.sched);
  __builtin_trap();   // a finished fiber is never resumed
}

void yield(State st) {
  Fiber *f = X.cur;
  f->state = st;
  hipemu_switch(&f->sp, &X.sched);
}

[[noreturn]] void die(const char *what) {
  std::fprintf(stderr, "hipemu: %s (workgroup %u,%u,%u of %u,%u,%u; %zu work-items)\n", what, X.bidx.x, X.bidx.y, X.bidx.z, X.gdim.x, X.gdim.y,
               X.gdim.z, X.fibers.size());
  for (size_t i = 0; i < X.fibers.size() && i < 1024; ++i) {
    const Fiber &f = X.fibers[i];
    if (f.state == AT_WAVE) std::fprintf(stderr, "  work-item %zu waits at wavefront op %d, site %lx\n", i, (int)f.op, (unsigned long)f.site);
    else if (f.state == AT_BARRIER && i % 64 == 0) std::fprintf(stderr, "  work-item %zu at the barrier\n", i);
  }
  std::abort();
}

// resolve the groups of one wavefront: lanes [w0, w1) that wait at the same site form a group
bool resolve_wave(size_t w0, size_t w1) {
  bool any = false;
  bool handled[kWave] = {false};
  for (size_t a = w0; a < w1; ++a) {
    Fiber &fa = X.fibers[a];
    if (fa.state != AT_WAVE || handled[a - w0]) continue;
    uint64_t members = 0;
    for (size_t b = a; b < w1; ++b)
      if (X.fibers[b].state == AT_WAVE && X.fibers[b].site == fa.site && X.fibers[b].op == fa.op) members |= 1ull << (b - w0);
    uint64_t ballot = 0;
    if (fa.op == OP_BALLOT)
      for (size_t b = a; b < w1; ++b)
        if ((members >> (b - w0)) & 1 && X.fibers[b].bits) ballot |= 1ull << (b - w0);
    for (size_t b = a; b < w1; ++b) {
      if (!((members >> (b - w0)) & 1)) continue;
      Fiber &fb = X.fibers[b];
      const int lane = (int)(b - w0);
      int src = lane;
      switch (fb.op) {
        case OP_BALLOT: break;
        case OP_SHFL: src = fb.arg & 63; break;
        case OP_SHFL_DOWN: src = lane + fb.arg; break;
        case OP_SHFL_UP: src = lane - fb.arg; break;
        case OP_SHFL_XOR: src = lane ^ fb.arg; break;
      }
      if (fb.op == OP_BALLOT) {
        fb.result = ballot;
      } else if (src < 0 || src >= (int)kWave) {
        fb.result = fb.bits;                     // out of range: the lane's own value (hardware behaviour)
      } else if (!((members >> src) & 1)) {
        fb.result = fb.bits;                     // source lane inactive: undefined on hardware — counted, see hipemu_inactive_reads()
        X.inactive_reads++;
      } else {
        fb.result = X.fibers[w0 + src].bits;
      }
      handled[lane] = true;
    }
    for (size_t b = a; b < w1; ++b)
      if ((members >> (b - w0)) & 1) X.fibers[b].state = RUNNABLE;
    any = true;
  }
  return any;
}

void run_block() {
  const size_t n = X.fibers.size();
  for (size_t i = 0; i < n; ++i) {
    Fiber &f = X.fibers[i];
    // initial frame: six zeroed callee-saved registers, fiber_main as the address `ret` jumps to (on a 16-byte boundary, so
    // that the stack is aligned as after a call), a null return address above it
    void **top = reinterpret_cast<void **>(X.stacks + (i + 1) * kStack);
    top[-1] = nullptr;
    top[-2] = reinterpret_cast<void *>(&fiber_main);
    for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
    f.sp = top - 8;
    f.state = RUNNABLE;
  }
  for (;;) {
    bool ran = false;
    size_t live = 0;
    for (size_t i = 0; i < n; ++i) {
      Fiber &f = X.fibers[i];
      if (f.state == RUNNABLE) {
        X.cur = &f;
        hipemu_switch(&X.sched, &f.sp);
        ran = true;
      }
      if (f.state != DONE) live++;
    }
    if (live == 0) break;
    // every live work-item now waits at a barrier or a wavefront operation
    bool resolved = false;
    for (size_t w0 = 0; w0 < n; w0 += kWave) resolved |= resolve_wave(w0, std::min(n, w0 + kWave));
    if (resolved) continue;
    bool all_barrier = true;
    for (size_t i = 0; i < n; ++i)
      if (X.fibers[i].state != DONE && X.fibers[i].state != AT_BARRIER) all_barrier = false;
    if (all_barrier) {
      for (size_t i = 0; i < n; ++i)
        if (X.fibers[i].state == AT_BARRIER) X.fibers[i].state = RUNNABLE;
      continue;
    }
    if (!ran) die("deadlock");
  }
  X.cur = nullptr;
}

}  // namespace

const Idx &thread_idx() { return X.cur ? X.cur->tid : kZero; }
const Idx &block_idx() { return X.bidx; }
const Idx &block_dim() { return X.bdim; }
const Idx &grid_dim() { return X.gdim; }

void barrier() {
  if (X.cur == nullptr) die("__syncthreads outside a kernel");
  yield(AT_BARRIER);
}

uint64_t wave_op(WaveOp op, uint64_t bits, int arg, uintptr_t site) {
  Fiber *f = X.cur;
  if (f == nullptr) die("wavefront operation outside a kernel");
  f->op = op;
  f->bits = bits;
  f->arg = arg;
  f->site = site;
  yield(AT_WAVE);
  return f->result;
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
  if (X.cur != nullptr) die("nested launch");
  const size_t n = (size_t)block.x * block.y * block.z;
  if (n == 0 || (size_t)grid.x * grid.y * grid.z == 0) return;
  if (n > 1024) die("more than 1024 work-items per workgroup");
  if (lds_bytes > 160 * 1024) die("more than 160 KB of dynamic LDS");
  if (n > X.n_stacks) {
    if (X.stacks) munmap(X.stacks, X.n_stacks * kStack);
    X.stacks = static_cast<char *>(mmap(nullptr, n * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (X.stacks == MAP_FAILED) die("cannot map fiber stacks");
    X.n_stacks = n;
  }
  X.fibers.resize(n);
  for (size_t i = 0; i < n; ++i) X.fibers[i].tid = Idx{(unsigned)(i % block.x), (unsigned)(i / block.x % block.y), (unsigned)(i / ((size_t)block.x * block.y))};
  X.bdim = Idx{block.x, block.y, block.z};
  X.gdim = Idx{grid.x, grid.y, grid.z};
  X.body = &body;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        X.bidx = Idx{x, y, z};
        run_block();
      }
  X.body = nullptr;
}

double frexp_mant(double x) {   // v_frexp_mant_f64: +-0, inf, NaN come back unchanged
  if (x == 0.0 || std::isinf(x) || std::isnan(x)) return x;
  int e;
  return std::frexp(x, &e);
}
int frexp_exp(double x) {       // v_frexp_exp_i32_f64: 0 for +-0, inf, NaN
  if (x == 0.0 || std::isinf(x) || std::isnan(x)) return 0;
  int e;
  std::frexp(x, &e);
  return e;
}

}  // namespace hipemu

namespace {
// a fault inside a kernel: say which work-item it was and where (addresses: addr2line -e libtad_hipemu.so <offset>)
void on_fault(int sig, siginfo_t *si, void *uc_) {
  char msg[512];
  const hipemu::Idx t = hipemu::thread_idx(), b = hipemu::block_idx();
  const ucontext_t *uc = static_cast<const ucontext_t *>(uc_);
  void *ip = reinterpret_cast<void *>(uc->uc_mcontext.gregs[REG_RIP]);
  if (ip == nullptr) ip = *reinterpret_cast<void **>(uc->uc_mcontext.gregs[REG_RSP]);   // call through a null pointer: report the caller
  Dl_info di{};
  dladdr(ip, &di);
  const int len = std::snprintf(msg, sizeof msg,
                                "hipemu: signal %d (address %p) in work-item %u of workgroup %u (grid %u x %u) at %s+0x%lx [%s]\n"
                                "        addr2line -Cfe %s 0x%lx\n",
                                sig, si->si_addr, t.x, b.x, hipemu::grid_dim().x, hipemu::block_dim().x, di.dli_fname ? di.dli_fname : "?",
                                (unsigned long)((char *)ip - (char *)di.dli_fbase), di.dli_sname ? di.dli_sname : "?", di.dli_fname ? di.dli_fname : "?",
                                (unsigned long)((char *)ip - (char *)di.dli_fbase));
  if (write(2, msg, (size_t)len) < 0) {}
  _exit(139);
}
struct InstallHandler {
  InstallHandler() {
    static char alt[64 * 1024];
    stack_t ss{};
    ss.ss_sp = alt;
    ss.ss_size = sizeof alt;
    sigaltstack(&ss, nullptr);
    struct sigaction sa{};
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_ONSTACK | SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
    sigaction(SIGFPE, &sa, nullptr);
  }
} install_handler;
}  // namespace

extern "C" unsigned long long hipemu_inactive_reads() { return hipemu::X.inactive_reads; }

// ---- host API ----
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = *total_b = (size_t)8 << 30; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) {
  *p = nullptr;
  if (posix_memalign(p, 256, bytes ? bytes : 256) != 0) return hipErrorOutOfMemory;
  std::memset(*p, 0xA5, bytes);   // device memory is not zeroed: make reads of unwritten memory visible
  return hipSuccess;
}
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
  *p = nullptr;
  return posix_memalign(p, 256, bytes ? bytes : 256) == 0 ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) { std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t) { std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemset(void *dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new hipemu_stream_s{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 1; *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event_s{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : (e == hipErrorOutOfMemory ? "out of memory" : "error"); }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes *attr, const void *) { attr->numRegs = 0; attr->sharedSizeBytes = 0; return hipSuccess; }
