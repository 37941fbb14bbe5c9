// Measurement aid (not part of the library): how many cycles does one wavefront need per FP64 instruction on gfx950
//   (a) in ONE dependent chain, (b) in 2 / 4 / 8 independent chains interleaved, for fma and for an IEEE division?
// Decides whether k_arima_fit's likelihood loop (one wavefront per SIMD) is bound by issue or by dependent-instruction latency.
//   hipcc -O3 --offload-arch=gfx950 -o fp64_issue_probe tools/probes/fp64_issue_probe.hip && ./fp64_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, bool DIV>
__global__ __launch_bounds__(64) void k_probe(double *out, unsigned long long *cyc, double a, double b, int iters) {
  double x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = 1.0 + 1e-3 * (threadIdx.x + c);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = DIV ? a / x[c] + b : __builtin_fma(x[c], a, b);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS, bool DIV>
static void run(const char *name, int blocks) {
  double *out; unsigned long long *cyc;
  hipMalloc(&out, (size_t)blocks * 64 * 8); hipMalloc(&cyc, (size_t)blocks * 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<CHAINS, DIV>), dim3(blocks), dim3(64), 0, 0, out, cyc, 0.999, 1e-3, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_probe<CHAINS, DIV>), dim3(blocks), dim3(64), 0, 0, out, cyc, 0.999, 1e-3, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const double ops = (double)iters * 16.0 * CHAINS;   // FP64 operations (fma, or division + add) per wavefront
  printf("%-18s blocks %5d: %7.3f ticks per op in a wavefront | kernel %8.3f ms = %7.3f ns per op per SIMD (1024 SIMDs) | tick = %.3f ns\n", name, blocks,
         (double)h[0] / ops, ms, ms * 1e6 / (ops * blocks / 1024.0), ms * 1e6 / (double)h[0] * (blocks <= 1024 ? 1.0 : 0.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1024, 2048, 4096, 8192}) {   // 1, 2, 4, 8 wavefronts per SIMD on 256 CUs x 4 SIMDs
    run<1, false>("fma  1 chain", blocks);
    run<4, false>("fma  4 chains", blocks);
    run<8, false>("fma  8 chains", blocks);
    run<1, true>("div+add 1 chain", blocks);
    run<4, true>("div+add 4 chains", blocks);
  }
  return 0;
}
