// tools/hipemu — DEVELOPMENT AID, not product code and not a fallback.
//
// A stand-in for <hip/hip_runtime.h> that lets the product's HIP sources be compiled with g++ and their kernels be
// EXECUTED ON THE HOST, one workgroup at a time, every work-item a fiber: __syncthreads, wavefront-64 __ballot / __shfl*,
// LDS (static and dynamic) and atomics behave as on the device, including divergence (lanes that wait at different
// call sites are resolved as separate groups, lanes that returned are inactive).  Purpose: find indexing, barrier and
// divergent-shuffle bugs in a new kernel BEFORE spending GPU minutes on it (the GPU box is metered), by running the
// `-m gpu` parity tests against tools/hipemu/_build/libtad_hipemu.so here (see tools/hipemu/README.md).
// It models no timing, no memory model and no races; it is never built by __graft_entry__.build(), never shipped to the
// GPU box (.gpurunignore) and nothing under theia_amd/, tests/ or bench.py refers to it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>

#define __HIPEMU__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local   // one workgroup at a time per OS thread: block-scope thread_local == the workgroup's LDS

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipemu_stream_s *hipStream_t;
typedef struct hipemu_event_s *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
struct ulonglong2 { unsigned long long x, y; };
struct longlong2 { long long x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
static inline longlong2 make_longlong2(long long x, long long y) { return longlong2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
const Idx &thread_idx();
const Idx &block_idx();
const Idx &block_dim();
const Idx &grid_dim();
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
void barrier();
enum WaveOp { OP_BALLOT, OP_SHFL, OP_SHFL_DOWN, OP_SHFL_UP, OP_SHFL_XOR };
// the calling lane posts `bits`, waits for the lanes of its wavefront that reach the same call site, and gets the result
uint64_t wave_op(WaveOp op, uint64_t bits, int arg, uintptr_t site);
double frexp_mant(double x);
int frexp_exp(double x);
}  // namespace hipemu

#define threadIdx (hipemu::thread_idx())
#define blockIdx (hipemu::block_idx())
#define blockDim (hipemu::block_dim())
#define gridDim (hipemu::grid_dim())
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (size_t)(lds), [&]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::barrier(); }
static inline void __threadfence_block() {}
static inline void __threadfence() {}

namespace hipemu {
template <typename T> inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "wave ops on values of at most 8 bytes");
  uint64_t b = 0;
  std::memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T> inline T from_bits(uint64_t b) {
  T v;
  std::memcpy(&v, &b, sizeof(T));
  return v;
}
}  // namespace hipemu

// The call site (file, line) identifies the instruction the lanes meet at — robust against whatever the host compiler does
// to the code (unrolling, versioning), unlike a return address.
#define HIPEMU_SITE const char *file_ = __builtin_FILE(), int line_ = __builtin_LINE()
#define HIPEMU_KEY (reinterpret_cast<uintptr_t>(file_) * 1000003u + (unsigned)line_)
template <typename P> inline unsigned long long __ballot(P pred, HIPEMU_SITE) { return hipemu::wave_op(hipemu::OP_BALLOT, pred ? 1u : 0u, 0, HIPEMU_KEY); }
template <typename T> inline T __shfl(T v, int src, HIPEMU_SITE) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL, hipemu::to_bits(v), src, HIPEMU_KEY)); }
template <typename T> inline T __shfl_down(T v, unsigned d, HIPEMU_SITE) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL_DOWN, hipemu::to_bits(v), (int)d, HIPEMU_KEY)); }
template <typename T> inline T __shfl_up(T v, unsigned d, HIPEMU_SITE) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL_UP, hipemu::to_bits(v), (int)d, HIPEMU_KEY)); }
template <typename T> inline T __shfl_xor(T v, int m, HIPEMU_SITE) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL_XOR, hipemu::to_bits(v), m, HIPEMU_KEY)); }

// __any / __all over the lanes that reach the call site (the active lanes)
template <typename P> inline int __any(P pred, HIPEMU_SITE) { return hipemu::wave_op(hipemu::OP_BALLOT, pred ? 1u : 0u, 0, HIPEMU_KEY) != 0; }
template <typename P> inline int __all(P pred, HIPEMU_SITE) { return hipemu::wave_op(hipemu::OP_BALLOT, pred ? 0u : 1u, 0, HIPEMU_KEY) == 0; }

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline unsigned int __umul24(unsigned int a, unsigned int b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
// compiler-level wavefront barrier on the device (no instruction); here the lanes really have to meet
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::wave_op(hipemu::OP_BALLOT, 1u, 0, reinterpret_cast<uintptr_t>(__FILE__) * 1000003u + (unsigned)__LINE__))
#define __builtin_amdgcn_readlane(v, l) __shfl((int)(v), (int)(l))
// value of the lowest active lane (the lanes that reach the call site); lanes below this one within a ballot mask
template <typename T> inline T hipemu_readfirstlane(T v, HIPEMU_SITE) {
  const unsigned long long m = hipemu::wave_op(hipemu::OP_BALLOT, 1u, 0, HIPEMU_KEY);
  return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL, hipemu::to_bits(v), __builtin_ctzll(m), HIPEMU_KEY ^ 0x9E3779B9u));
}
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
#define __builtin_amdgcn_mbcnt_lo(mask, acc) ((acc) + (unsigned)__builtin_popcount((unsigned)(mask) & (unsigned)(((threadIdx.x & 63u) >= 32u) ? 0xFFFFFFFFu : ((1u << (threadIdx.x & 31u)) - 1u))))
#define __builtin_amdgcn_mbcnt_hi(mask, acc) ((acc) + (unsigned)__builtin_popcount((unsigned)(mask) & (unsigned)(((threadIdx.x & 63u) < 32u) ? 0u : ((1u << (threadIdx.x & 31u)) - 1u))))
static inline long long __double_as_longlong(double v) { long long r; __builtin_memcpy(&r, &v, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; __builtin_memcpy(&r, &v, 8); return r; }
#define __builtin_amdgcn_frexp_mant(x) hipemu::frexp_mant(x)
#define __builtin_amdgcn_frexp_exp(x) hipemu::frexp_exp(x)

// atomics: workgroups run one after another on one OS thread, so plain read-modify-write would do; the builtins keep the
// door open for running workgroups on several threads
// (x86 performs misaligned atomics; gfx950 raises a memory fault on them — a 64-bit cursor behind a 4-byte-per-cell array once did
// exactly that on the GPU only — so the emulator checks the natural alignment of every atomic's address)
template <typename T> inline void hipemu_check_atomic(const T *p) {
  if (reinterpret_cast<uintptr_t>(p) % sizeof(T) != 0) { std::fprintf(stderr, "hipemu: %zu-byte atomic on misaligned address %p\n", sizeof(T), (const void *)p); std::abort(); }
}
template <typename T, typename U> inline T atomicAdd(T *p, U v) { hipemu_check_atomic(p); return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> inline T atomicSub(T *p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> inline T atomicMax(T *p, U v) {
  hipemu_check_atomic(p);
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T, typename U> inline T atomicMin(T *p, U v) {
  hipemu_check_atomic(p);
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T, typename U> inline T atomicCAS(T *p, U cmp, U v) {
  T expected = (T)cmp;
  __atomic_compare_exchange_n(p, &expected, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return expected;
}
template <typename T, typename U> inline T __hip_atomic_fetch_add(T *p, U v, int, int) { return atomicAdd(p, v); }
template <typename T, typename U> inline T __hip_atomic_fetch_max(T *p, U v, int, int) { return atomicMax(p, v); }
template <typename T, typename U> inline T __hip_atomic_fetch_min(T *p, U v, int, int) { return atomicMin(p, v); }
template <typename T> inline T __hip_atomic_load(const T *p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
template <typename T, typename U> inline void __hip_atomic_store(T *p, U v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_RELAXED); }

using std::max;
using std::min;

// ---- host API: the "device" is host memory, every call is synchronous ----
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t hipMemset(void *dst, int value, size_t bytes);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipFuncSetAttribute(const void *fn, hipFuncAttribute attr, int value);
struct hipFuncAttributes { int numRegs; size_t sharedSizeBytes; };
hipError_t hipFuncGetAttributes(hipFuncAttributes *attr, const void *fn);
template <typename T> inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc(reinterpret_cast<void **>(p), bytes); }
template <typename T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags) { return hipHostMalloc(reinterpret_cast<void **>(p), bytes, flags); }
