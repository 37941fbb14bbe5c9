// fetch_calibrate.hip — what does FETCH_SIZE report for streams of 4 / 8 / 16 bytes per lane?  (measurement aid, not product)
//
// bench.py's `traffic` doubles rocprofv3's FETCH_SIZE (MI355X_MICROARCH.md: gfx950 tallies 64 B per 128-B request).  That factor was
// calibrated on 16 B/lane streams (pass B's column loads); the per-key walkers (k_key_sigma, k_emit_staged) load 8 B per lane and the round-5
// review asked whether their figures are overstated.  Each kernel below reads a 1 GiB buffer exactly once, coalesced, with one load width;
// run under `rocprofv3 --pmc FETCH_SIZE` the per-kernel counter against 1 GiB is the factor for that width.  `strided`: 8 B per lane with the
// lanes of a wavefront on consecutive 8-byte words but consecutive wavefront-steps K words apart — the time-major grid walk of the per-key kernels.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/fetch_calibrate tools/probes/fetch_calibrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ p, size_t n, unsigned long long *out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = p[i];
    const unsigned int *w = reinterpret_cast<const unsigned int *>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += w[k];
  }
  if (acc == 0x123456789abcdefull) *out = acc;   // never true: keeps the loads
}

// lane = key, step = bucket: word (b * K + k), the walk of k_key_sigma over a K x T grid of 8-byte cells
__global__ __launch_bounds__(256) void k_walk8(const unsigned long long *__restrict__ p, size_t K, size_t T, unsigned long long *out) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  unsigned long long acc = 0;
  for (size_t b = 0; b < T; ++b) acc += p[b * K + k];
  if (acc == 0x123456789abcdefull) *out = acc;
}

struct alignas(16) W16 { unsigned int w[4]; };
struct alignas(8) W8 { unsigned int w[2]; };

int main() {
  const size_t bytes = (size_t)1 << 30;
  void *buf = nullptr;
  unsigned long long *out = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto timed = [&](const char *name, auto launch) {
    launch();                       // warm
    hipEventRecord(a, 0);
    launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-10s 1 GiB read in %.3f ms = %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  };
  timed("k_read<u32>", [&] { hipLaunchKernelGGL((k_read<unsigned int>), dim3(4096), dim3(256), 0, 0, (const unsigned int *)buf, bytes / 4, out); });
  timed("k_read<W8>", [&] { hipLaunchKernelGGL((k_read<W8>), dim3(4096), dim3(256), 0, 0, (const W8 *)buf, bytes / 8, out); });
  timed("k_read<W16>", [&] { hipLaunchKernelGGL((k_read<W16>), dim3(4096), dim3(256), 0, 0, (const W16 *)buf, bytes / 16, out); });
  const size_t K = 100000, T = bytes / 8 / K;       // 1342 steps of 1e5 keys: 1.07e9 bytes, the shape of C2's grid walk (T = 250) stretched to ~1 GiB
  timed("k_walk8", [&] { hipLaunchKernelGGL(k_walk8, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, 0, (const unsigned long long *)buf, K, T, out); });
  printf("k_walk8 reads %zu bytes (K = %zu keys x T = %zu steps x 8 B)\n", K * T * 8, K, T);
  return 0;
}
