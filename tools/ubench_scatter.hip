// ubench_scatter.hip — design probe for the Stage-0 group-by (not part of the product library).
// Question: what does a random u64 atomic scatter into a K*T grid cost on gfx950 next to the
// 24 B/row column stream?  Variants: stream only, agent-scope atomic add (no return), + presence
// byte store, workgroup-scope atomic (NOT correct across XCDs, informative), plain RMW (incorrect,
// informative), atomic max.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void k_fill(uint64_t* key, uint64_t* bucket, uint64_t* val, uint64_t n, uint64_t K, uint64_t T, int sorted) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t h1 = mix64(0x7AD05EEDull + 0x9E3779B97F4A7C15ull * (i * 8 + 1));
    uint64_t h2 = mix64(0x7AD05EEDull + 0x9E3779B97F4A7C15ull * (i * 8 + 2));
    key[i] = sorted ? (i * K / n) : (h1 % K);
    bucket[i] = h2 % T;
    val[i] = 1000000000ull + (h1 >> 40);
  }
}

enum { V_STREAM = 0, V_ATOMIC_AGENT, V_ATOMIC_AGENT_PRES, V_ATOMIC_WG, V_PLAIN_RMW, V_ATOMIC_MAX_PRES, V_PRES_ONLY, V_ATOMIC_AGENT_PRES_KM };

template <int V>
__global__ __launch_bounds__(256) void k_scatter(const ulonglong2* __restrict__ key, const ulonglong2* __restrict__ bucket,
                                                 const ulonglong2* __restrict__ val, uint64_t n2, uint64_t K, uint64_t T,
                                                 unsigned long long* grid, uint8_t* pres, unsigned long long* sink) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (; i < n2; i += stride) {
    ulonglong2 k = key[i], b = bucket[i], v = val[i];
    uint64_t c0, c1;
    if (V == V_ATOMIC_AGENT_PRES_KM) { c0 = k.x * T + b.x; c1 = k.y * T + b.y; }  // key-major
    else { c0 = b.x * K + k.x; c1 = b.y * K + k.y; }                               // time-major
    if (V == V_STREAM) { acc += c0 + c1 + v.x + v.y; }
    if (V == V_ATOMIC_AGENT || V == V_ATOMIC_AGENT_PRES || V == V_ATOMIC_AGENT_PRES_KM) {
      __hip_atomic_fetch_add(grid + c0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(grid + c1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (V == V_ATOMIC_MAX_PRES) {
      __hip_atomic_fetch_max(grid + c0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_max(grid + c1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (V == V_ATOMIC_WG) {
      __hip_atomic_fetch_add(grid + c0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(grid + c1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (V == V_PLAIN_RMW) { grid[c0] += v.x; grid[c1] += v.y; }
    if (V == V_ATOMIC_AGENT_PRES || V == V_ATOMIC_MAX_PRES || V == V_PRES_ONLY || V == V_ATOMIC_AGENT_PRES_KM) { pres[c0] = 1; pres[c1] = 1; }
    if (V == V_PRES_ONLY) acc += v.x + v.y;
  }
  if (V == V_STREAM || V == V_PRES_ONLY) { if (acc == 0x1234567ull) sink[0] = acc; }
}

template <int V>
float run(const char* name, const uint64_t* key, const uint64_t* bucket, const uint64_t* val, uint64_t n, uint64_t K, uint64_t T,
          unsigned long long* grid, uint8_t* pres, unsigned long long* sink, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    CK(hipMemsetAsync(grid, 0, K * T * 8)); CK(hipMemsetAsync(pres, 0, K * T));
    CK(hipEventRecord(e0));
    k_scatter<V><<<blocks, 256>>>((const ulonglong2*)key, (const ulonglong2*)bucket, (const ulonglong2*)val, n / 2, K, T, grid, pres, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 0 && ms < best) best = ms;
  }
  printf("  %-28s blocks=%5d  %8.3f ms  %7.2f Grows/s  stream %7.1f GB/s\n", name, blocks, best, n / best / 1e6, 24.0 * n / best / 1e6);
  fflush(stdout);
  return best;
}

int main(int argc, char** argv) {
  uint64_t n = 100000000ull;
  struct Cfg { uint64_t K, T; int sorted; } cfgs[] = {{100000, 250, 0}, {1000000, 100, 0}, {100000, 250, 1}, {10000, 250, 0}};
  uint64_t *key, *bucket, *val; unsigned long long *grid, *sink; uint8_t* pres;
  CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&bucket, n * 8)); CK(hipMalloc(&val, n * 8));
  CK(hipMalloc(&grid, 100000000ull * 8)); CK(hipMalloc(&pres, 100000000ull)); CK(hipMalloc(&sink, 8));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d MHz L2=%d\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.l2CacheSize);
  for (auto c : cfgs) {
    printf("N=%llu K=%llu T=%llu cells=%llu (%.0f MB grid) sorted=%d\n", (unsigned long long)n, (unsigned long long)c.K, (unsigned long long)c.T,
           (unsigned long long)(c.K * c.T), c.K * c.T * 8 / 1e6, c.sorted);
    k_fill<<<4096, 256>>>(key, bucket, val, n, c.K, c.T, c.sorted); CK(hipDeviceSynchronize());
    for (int blocks : {2048, 8192}) {
      run<V_STREAM>("stream-only", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_ATOMIC_AGENT>("atomic-add agent", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_ATOMIC_AGENT_PRES>("atomic-add agent + pres", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_ATOMIC_AGENT_PRES_KM>("same, key-major grid", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_ATOMIC_MAX_PRES>("atomic-max agent + pres", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_ATOMIC_WG>("atomic-add workgroup-scope", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_PLAIN_RMW>("plain RMW (racy)", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
      run<V_PRES_ONLY>("presence store only", key, bucket, val, n, c.K, c.T, grid, pres, sink, blocks);
    }
  }
  return 0;
}
