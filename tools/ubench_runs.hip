// ubench_runs.hip — design probe for pass B of Stage 0 (not part of the product library).
// Question: does the LAYOUT of the record buffer matter for the partition pass's write pattern?  256 workgroups x
// 1024 threads stream 24 B/row (three u64 columns) and write one 8-byte record per row in the pattern of an LDS-sorted
// tile: 10240 records per tile in F runs of ~7 records, run p of workgroup w going to
//   A  partition-major   recs[p][w][...]   (what k_partition does: partition p is one contiguous region for pass C)
//   B  workgroup-major   recs[w][p][...]   (every workgroup writes inside its own contiguous window)
//   C  contiguous        recs[w][tile][idx] (no scatter at all: the ceiling for 24 B read + 8 B written per row)
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_runs.hip -o /tmp/ubr && /tmp/ubr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kThreads = 1024, RPT = 10, TILE = kThreads * RPT, G = 256;

// X: the 32 workgroups of an XCD share one cursor per partition (emulated: lockstep interleave (tile * 32 + rank in XCD)),
// so that the short runs of different workgroups abut and a cache line is completed within ~a microsecond by writers
// that share the SAME L2 — does the L2 then merge them into full-line writes?
__global__ __launch_bounds__(kThreads) void k_runs_x(const ulonglong2* __restrict__ a, const ulonglong2* __restrict__ b,
                                                     const ulonglong2* __restrict__ c, unsigned long long* __restrict__ recs,
                                                     uint64_t rows_per_wg, uint32_t F, uint64_t part_size, unsigned int* xcc_count, int use_hw_id) {
  __shared__ uint32_t s_xcc, s_rank;
  if (threadIdx.x == 0) {
    uint32_t xcc = use_hw_id ? (__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u) : (blockIdx.x & 7u);
    s_xcc = xcc;
    s_rank = atomicAdd(&xcc_count[xcc], 1u);
  }
  __syncthreads();
  const uint32_t xcc = s_xcc, rank = s_rank;
  const uint32_t w = blockIdx.x;
  const uint64_t lo = (uint64_t)w * rows_per_wg;
  const uint64_t ntiles = rows_per_wg / TILE;
  const uint64_t xsub = part_size / 8;
  for (uint64_t tile = 0; tile < ntiles; ++tile) {
    const uint64_t base = lo + tile * TILE;
    ulonglong2 x[RPT / 2], y[RPT / 2], z[RPT / 2];
#pragma unroll
    for (int j = 0; j < RPT / 2; ++j) {
      const uint64_t i = (base >> 1) + (uint64_t)j * kThreads + threadIdx.x;
      x[j] = a[i]; y[j] = b[i]; z[j] = c[i];
    }
#pragma unroll
    for (int j = 0; j < RPT / 2; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t idx = (uint32_t)(j * 2 + h) * kThreads + threadIdx.x;
        const unsigned long long v = h ? (x[j].y + y[j].y + z[j].y) : (x[j].x + y[j].x + z[j].x);
        const uint32_t p = (uint32_t)(((uint64_t)idx * F) / TILE);
        const uint32_t start = (uint32_t)(((uint64_t)p * TILE + F - 1) / F);
        const uint32_t next = (uint32_t)(((uint64_t)(p + 1) * TILE + F - 1) / F);
        const uint64_t cur = (tile * 32 + rank) * (next - start) + (idx - start);
        recs[(uint64_t)p * part_size + (uint64_t)xcc * xsub + cur] = v;
      }
    }
  }
}

template <int LAYOUT, int NT = 0>
__global__ __launch_bounds__(kThreads) void k_runs(const ulonglong2* __restrict__ a, const ulonglong2* __restrict__ b,
                                                   const ulonglong2* __restrict__ c, unsigned long long* __restrict__ recs,
                                                   uint64_t rows_per_wg, uint32_t F, uint64_t part_size, uint64_t sub, uint32_t skew) {
  const uint32_t w = blockIdx.x;
  const uint64_t lo = (uint64_t)w * rows_per_wg;
  const uint64_t ntiles = rows_per_wg / TILE;
  for (uint64_t tile = 0; tile < ntiles; ++tile) {
    const uint64_t base = lo + tile * TILE;
    ulonglong2 x[RPT / 2], y[RPT / 2], z[RPT / 2];
#pragma unroll
    for (int j = 0; j < RPT / 2; ++j) {
      const uint64_t i = (base >> 1) + (uint64_t)j * kThreads + threadIdx.x;
      if (NT & 1) {
        x[j].x = __builtin_nontemporal_load(&a[i].x); x[j].y = __builtin_nontemporal_load(&a[i].y);
        y[j].x = __builtin_nontemporal_load(&b[i].x); y[j].y = __builtin_nontemporal_load(&b[i].y);
        z[j].x = __builtin_nontemporal_load(&c[i].x); z[j].y = __builtin_nontemporal_load(&c[i].y);
      } else {
        x[j] = a[i]; y[j] = b[i]; z[j] = c[i];
      }
    }
#pragma unroll
    for (int j = 0; j < RPT / 2; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t idx = (uint32_t)(j * 2 + h) * kThreads + threadIdx.x;   // consecutive lanes -> consecutive records
        const unsigned long long v = h ? (x[j].y + y[j].y + z[j].y) : (x[j].x + y[j].x + z[j].x);
        uint64_t dst;
        if (LAYOUT == 2) {
          dst = base + idx;
        } else {
          const uint32_t p = (uint32_t)(((uint64_t)idx * F) / TILE);
          const uint32_t start = (uint32_t)(((uint64_t)p * TILE + F - 1) / F);
          const uint32_t next = (uint32_t)(((uint64_t)(p + 1) * TILE + F - 1) / F);
          const uint64_t cur = tile * (next - start) + (idx - start);
          dst = (LAYOUT == 0 ? (uint64_t)p * part_size + (uint64_t)w * sub + cur : (uint64_t)w * (sub * F) + (uint64_t)p * sub + cur) + skew;
        }
        if (NT & 2) __builtin_nontemporal_store(v, &recs[dst]);
        else recs[dst] = v;
      }
    }
  }
}

template <int LAYOUT, int NT = 0>
static void run(const char* name, const uint64_t* a, const uint64_t* b, const uint64_t* c, unsigned long long* recs, uint64_t n, uint32_t F, uint32_t skew = 0) {
  const uint64_t rows_per_wg = (n / G) / TILE * TILE;
  const uint64_t ntiles = rows_per_wg / TILE;
  const uint64_t sub = (ntiles * ((TILE + F - 1) / F) + 15) & ~15ull;   // >= records of (w, p), 128-byte multiple
  const uint64_t part_size = sub * G;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0;
  for (int it = 0; it < 7; ++it) {
    CK(hipEventRecord(e0));
    k_runs<LAYOUT, NT><<<G, kThreads>>>((const ulonglong2*)a, (const ulonglong2*)b, (const ulonglong2*)c, recs, rows_per_wg, F, part_size, sub, skew);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  const double rows = (double)rows_per_wg * G;
  printf("  F=%4u %-18s avg %.3f ms  min %.3f ms  %.1f Grows/s  (read+write %.0f GB/s)\n", F, name, sum / 5, best, rows / (sum / 5) / 1e6,
         rows * 32 / (sum / 5) / 1e6);
}

static void run_x(const char* name, const uint64_t* a, const uint64_t* b, const uint64_t* c, unsigned long long* recs, uint64_t n, uint32_t F, int use_hw_id) {
  const uint64_t rows_per_wg = (n / G) / TILE * TILE;
  const uint64_t ntiles = rows_per_wg / TILE;
  const uint64_t sub = (ntiles * ((TILE + F - 1) / F) + 15) & ~15ull;
  const uint64_t part_size = sub * G;
  unsigned int* xc; CK(hipMalloc(&xc, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0;
  unsigned int h[8];
  for (int it = 0; it < 7; ++it) {
    CK(hipMemset(xc, 0, 64));
    CK(hipEventRecord(e0));
    k_runs_x<<<G, kThreads>>>((const ulonglong2*)a, (const ulonglong2*)b, (const ulonglong2*)c, recs, rows_per_wg, F, part_size, xc, use_hw_id);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  CK(hipMemcpy(h, xc, 32, hipMemcpyDeviceToHost));
  const double rows = (double)rows_per_wg * G;
  printf("  F=%4u %-18s avg %.3f ms  min %.3f ms  %.1f Grows/s  (read+write %.0f GB/s)  wgs/xcc %u %u %u %u %u %u %u %u\n", F, name, sum / 5, best,
         rows / (sum / 5) / 1e6, rows * 32 / (sum / 5) / 1e6, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  CK(hipFree(xc));
}

int main() {
  const uint64_t n = 100000000ull;
  uint64_t *a, *b, *c; unsigned long long* recs;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8));
  CK(hipMalloc(&recs, (size_t)3 << 30));   // sub * F * G * 8 B <= 38 * 8 * 2048 * 256 * 8 = 1.3 GB
  CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 2, n * 8)); CK(hipMemset(c, 3, n * 8)); CK(hipMemset(recs, 0, (size_t)3 << 30));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d\n", p.name, p.multiProcessorCount);
  for (int rep = 0; rep < 2; ++rep) {
    for (uint32_t F : {1470u, 368u, 2048u}) {
      run<0>("A partition-major", a, b, c, recs, n, F);
      run<1>("B workgroup-major", a, b, c, recs, n, F);
      run<2>("C contiguous", a, b, c, recs, n, F);
    }
    // run length exactly 8 / 16 / 32 records (TILE / F), starts aligned to the run length vs skewed by 3 records
    run<0>("A len 8 aligned", a, b, c, recs, n, 1280);
    run<0>("A len 8 skew 3", a, b, c, recs, n, 1280, 3);
    run<0>("A len 16 aligned", a, b, c, recs, n, 640);
    run<0>("A len 16 skew 3", a, b, c, recs, n, 640, 3);
    run<0>("A len 32 aligned", a, b, c, recs, n, 320);
    run<0>("A len 32 skew 3", a, b, c, recs, n, 320, 3);
    run<0>("A len 4 aligned", a, b, c, recs, n, 2560);
    run<0, 1>("A nt loads", a, b, c, recs, n, 1470);
    run<0, 2>("A nt stores", a, b, c, recs, n, 1470);
    run<0, 3>("A nt loads+stores", a, b, c, recs, n, 1470);
    run<2, 3>("C nt loads+stores", a, b, c, recs, n, 1470);
    run_x("X shared/XCD hw id", a, b, c, recs, n, 1470, 1);
    run_x("X shared/XCD blk&7", a, b, c, recs, n, 1470, 0);
    run_x("X shared/XCD F2048", a, b, c, recs, n, 2048, 1);
  }
  return 0;
}
