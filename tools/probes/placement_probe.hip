// Measurement aid (not part of the library): WHY does pass B of the same job run 0.60 ms on some allocations of its record buffer and 0.67 ms
// on others of the same process (tad_capi.cpp:place_recs, profiles/r4_v37_record_buffer_placement.log)?
//
// A stand-alone process (no Python: starts in a second) that allocates the three columns of a C2-sized table and then NC candidate record
// buffers, all held at once (as place_recs does), and times on every candidate
//   pm    pass B's memory pattern as the engine lays it out — PARTITION-major: region (workgroup w, partition p) at p * R + w * r
//   wm    the same lines in a WORKGROUP-major layout: region (w, p) at w * (nparts * r) + p * r — a workgroup's open lines lie within a few MB
//   seq   the same reads, one contiguous 8-byte-per-row write stream per workgroup (no scatter at all)
//   chase a single wavefront's dependent loads across the candidate at strides of 4 KB / 64 KB / 2 MB (TLB reach: ns per access)
// Every launch is its own dispatch, in a fixed order that the program prints, so that `rocprofv3 --kernel-trace --pmc <counters>` of the very
// same command lines its counters up with the times:   tools/gpu_placement_r5.sh
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/placement_probe tools/probes/placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static constexpr int kThreads = 1024;

__global__ __launch_bounds__(kThreads) void k_fill(ulonglong2 *p, uint64_t n2, uint64_t seed) {
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * kThreads)
    p[i] = make_ulonglong2(i * 0x9E3779B97F4A7C15ull + seed, (i + 1) * 0xBF58476D1CE4E5B9ull ^ seed);
}

// MODE 0: partition-major, 1: workgroup-major, 2: sequential.  n2 = row pairs; 8 lanes (16 rows) fill one 128-byte line
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_pattern(const ulonglong2 *__restrict__ key, const ulonglong2 *__restrict__ te,
                                                      const ulonglong2 *__restrict__ val, uint64_t n2, unsigned long long *__restrict__ recs,
                                                      uint64_t slots, uint32_t nparts) {
  const uint64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
  const uint64_t region_slots = (slots / nparts) & ~15ull;              // R: a partition's slots
  const uint64_t share = region_slots / gridDim.x & ~15ull;             // r: this workgroup's slots inside every partition
  if (share < 16) return;
  const uint32_t lane8 = threadIdx.x & 7u, grp = threadIdx.x >> 3;
  for (uint64_t i = lo + threadIdx.x, it = 0; i < hi; i += kThreads, ++it) {
    const ulonglong2 k = key[i], t = te[i], v = val[i];
    const uint64_t g = it * (kThreads / 8) + grp;                       // the workgroup's g-th line
    const uint32_t part = (uint32_t)(g % nparts);
    const uint64_t line = (g / nparts) % (share / 16);
    unsigned long long *dst;
    if (MODE == 0) dst = recs + (uint64_t)part * region_slots + (uint64_t)blockIdx.x * share + line * 16 + lane8 * 2;
    else if (MODE == 1) dst = recs + ((uint64_t)blockIdx.x * nparts + part) * share + line * 16 + lane8 * 2;
    else dst = recs + (uint64_t)blockIdx.x * (slots / gridDim.x & ~15ull) + (g % ((slots / gridDim.x & ~15ull) / 16)) * 16 + lane8 * 2;
    reinterpret_cast<ulonglong2 *>(dst)[0] = make_ulonglong2(k.x ^ t.x ^ v.x, k.y ^ t.y ^ v.y);
  }
}

// PHASED: a workgroup alternates a READ phase (B rows of the three columns -> one 8-byte record each, parked in LDS) and a WRITE phase (the B
// records out, whole 128-byte lines; MODE 0: round robin over the partitions' regions as k_pattern<0>, MODE 2: one contiguous stream).  The 256
// workgroups start together and do equal work, so their phases stay roughly aligned chip-wide: does separating reads from writes in TIME remove
// the dependence on where the buffer lies relative to the columns?
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_phased(const ulonglong2 *__restrict__ key, const ulonglong2 *__restrict__ te,
                                                     const ulonglong2 *__restrict__ val, uint64_t n2, unsigned long long *__restrict__ recs,
                                                     uint64_t slots, uint32_t nparts, uint32_t batch_pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ulonglong2 *buf = reinterpret_cast<ulonglong2 *>(smem);
  const uint64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
  const uint64_t region_slots = (slots / nparts) & ~15ull;
  const uint64_t share = region_slots / gridDim.x & ~15ull;
  const uint64_t wg_slots = slots / gridDim.x & ~15ull;
  if (share < 16) return;
  const uint32_t lane8 = threadIdx.x & 7u, grp = threadIdx.x >> 3;
  uint64_t line0 = 0;
  for (uint64_t b0 = lo; b0 < hi; b0 += batch_pairs) {
    const uint64_t b1 = b0 + batch_pairs < hi ? b0 + batch_pairs : hi;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += kThreads) {          // read phase
      const ulonglong2 k = key[i], t = te[i], v = val[i];
      buf[i - b0] = make_ulonglong2(k.x ^ t.x ^ v.x, k.y ^ t.y ^ v.y);
    }
    __syncthreads();
    for (uint64_t i = b0 + threadIdx.x, it = 0; i < b1; i += kThreads, ++it) {   // write phase
      const uint64_t g = line0 + it * (kThreads / 8) + grp;
      unsigned long long *dst;
      if (MODE == 0) dst = recs + (uint64_t)(g % nparts) * region_slots + (uint64_t)blockIdx.x * share + ((g / nparts) % (share / 16)) * 16 + lane8 * 2;
      else dst = recs + (uint64_t)blockIdx.x * wg_slots + (g % (wg_slots / 16)) * 16 + lane8 * 2;
      reinterpret_cast<ulonglong2 *>(dst)[0] = buf[i - b0];
    }
    line0 += (b1 - b0 + 7) / 8;
    __syncthreads();
  }
}

// write-only / read-only streams over a candidate (no column reads): is the class a property of the buffer alone?
__global__ __launch_bounds__(kThreads) void k_wo(ulonglong2 *__restrict__ p, uint64_t n2) {
  const uint64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kThreads) p[i] = make_ulonglong2(i, ~i);
}
__global__ __launch_bounds__(kThreads) void k_ro(const ulonglong2 *__restrict__ p, uint64_t n2, unsigned long long *out) {
  const uint64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
  unsigned long long acc = 0;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kThreads) { const ulonglong2 v = p[i]; acc += v.x ^ v.y; }
  if (acc == 0x123456789abcdefull) out[0] = acc;
}

// one wavefront, lane 0: `count` dependent 8-byte loads, `stride` bytes apart, wrapping inside `bytes`
__global__ __launch_bounds__(64) void k_chase(const unsigned long long *__restrict__ p, uint64_t bytes, uint64_t stride, uint32_t count,
                                              unsigned long long *out) {
  if (threadIdx.x != 0) return;
  uint64_t off = 0, acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (uint32_t i = 0; i < count; ++i) {
    const unsigned long long v = __builtin_nontemporal_load(p + off / 8);
    acc += v;
    off = (off + stride + (v & 8ull)) % bytes;        // the next address depends on the loaded value
  }
  const unsigned long long t1 = wall_clock64();
  out[0] = t1 - t0;
  out[1] = acc;
}

int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const uint32_t nparts = argc > 2 ? (uint32_t)atoi(argv[2]) : 782;
  const int nc = argc > 3 ? atoi(argv[3]) : 12;
  const uint64_t slots = argc > 4 ? strtoull(argv[4], nullptr, 10) : 240000000ull;      // C2's sampled-region buffer: ~2.4x the records
  const int reps = argc > 5 ? atoi(argv[5]) : 3;
  const int G = 256;
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  ulonglong2 *col[3];
  for (int c = 0; c < 3; ++c) {
    CK(hipMalloc(&col[c], n * 8));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(kThreads), 0, s, col[c], n / 2, (uint64_t)(c + 1) * 0x1234567ull);
  }
  std::vector<unsigned long long *> cand(nc);
  for (int i = 0; i < nc; ++i) {
    CK(hipMalloc(&cand[i], slots * 8));
    CK(hipMemsetAsync(cand[i], 0, slots * 8, s));
  }
  unsigned long long *d_out, h_out[2];
  CK(hipMalloc(&d_out, 16));
  CK(hipStreamSynchronize(s));
  size_t free_b = 0, total_b = 0;
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("placement_probe: %llu rows, %u partitions, %d candidates of %.2f GB held together; free %.1f of %.1f GB\n", (unsigned long long)n, nparts, nc,
         slots * 8 / 1e9, free_b / 1e9, total_b / 1e9);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timed = [&](auto launch) {
    launch();   // warm
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(a, s));
      launch();
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      float t = 0; CK(hipEventElapsedTime(&t, a, b));
      best = t < best ? t : best;
    }
    return best;
  };
  printf("dispatch order per candidate: (1 + %d) x k_pattern<0>, (1 + %d) x k_pattern<1>, (1 + %d) x k_pattern<2>, 3 x k_chase (4 KB, 64 KB, 2 MB)\n", reps, reps, reps);
  printf("%-4s %-18s %9s %9s %9s | chase ns/access: %8s %8s %8s\n", "cand", "address", "pm ms", "wm ms", "seq ms", "4KB", "64KB", "2MB");
  for (int i = 0; i < nc; ++i) {
    float ms[3];
    ms[0] = timed([&] { hipLaunchKernelGGL(k_pattern<0>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, cand[i], slots, nparts); });
    ms[1] = timed([&] { hipLaunchKernelGGL(k_pattern<1>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, cand[i], slots, nparts); });
    ms[2] = timed([&] { hipLaunchKernelGGL(k_pattern<2>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, cand[i], slots, nparts); });
    double ns[3];
    const uint64_t strides[3] = {4096, 65536, 2097152};
    for (int k = 0; k < 3; ++k) {
      const uint32_t count = 2048;
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s, cand[i], slots * 8, strides[k] + 64, count, d_out);
      CK(hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      ns[k] = (double)h_out[0] * 10.0 / count;      // wall_clock64: the constant 100 MHz counter
    }
    printf("%-4d %-18p %9.4f %9.4f %9.4f | %26.0f %8.0f %8.0f\n", i, (void *)cand[i], ms[0], ms[1], ms[2], ns[0], ns[1], ns[2]);
    fflush(stdout);
  }
  if (argc > 6 && atoi(argv[6]) == 0) return 0;      // (the counter passes stop here)
  // ---- phase 1c: does the SIZE of the allocation matter?  (round 5: a C4 job found only slow buffers among candidates of a bigger table's size and a
  // fast one among candidates of its own size.)  The same pattern (the first `slots` slots) on allocations of other sizes, held next to the twelve.
  {
    const double gb[] = {0.96, 1.5, 1.92, 2.5, 3.0, 4.0, 6.0, 8.0, 1.92, 1.92};
    printf("sizes: %-8s %-18s %9s %9s\n", "GB", "address", "pm ms", "seq ms");
    std::vector<unsigned long long *> extra;
    for (double g : gb) {
      const uint64_t bytes = (uint64_t)(g * 1e9) & ~4095ull;
      const uint64_t use_slots = bytes / 8 < slots ? bytes / 8 : slots;       // (a smaller buffer gets the pattern over all of it)
      unsigned long long *q = nullptr;
      if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); printf("sizes: %.2f GB: allocation failed\n", g); continue; }
      extra.push_back(q);
      CK(hipMemsetAsync(q, 0, bytes, s));
      const float pm = timed([&] { hipLaunchKernelGGL(k_pattern<0>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, q, use_slots, nparts); });
      const float sq = timed([&] { hipLaunchKernelGGL(k_pattern<2>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, q, use_slots, nparts); });
      printf("sizes: %-8.2f %-18p %9.4f %9.4f\n", g, (void *)q, pm, sq);
      fflush(stdout);
    }
  }
  // ---- phase 1b: reads and writes separated in time ----
  printf("phased: %-4s | pm: batch %6s %6s %6s rows | seq: batch %6s %6s %6s rows\n", "cand", "2048", "8192", "16384", "2048", "8192", "16384");
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_phased<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_phased<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < nc; ++i) {
    float t[6];
    const uint32_t bp[3] = {1024, 4096, 8192};        // row PAIRS per batch
    for (int k = 0; k < 3; ++k) {
      t[k] = timed([&] { hipLaunchKernelGGL(k_phased<0>, dim3(G), dim3(kThreads), (size_t)bp[k] * 16, s, col[0], col[1], col[2], n / 2, cand[i], slots, nparts, bp[k]); });
      t[3 + k] = timed([&] { hipLaunchKernelGGL(k_phased<2>, dim3(G), dim3(kThreads), (size_t)bp[k] * 16, s, col[0], col[1], col[2], n / 2, cand[i], slots, nparts, bp[k]); });
    }
    printf("phased: %-4d | %16.4f %6.4f %6.4f      | %17.4f %6.4f %6.4f\n", i, t[0], t[1], t[2], t[3], t[4], t[5]);
    fflush(stdout);
  }
  // ---- phase 2: what is the class a property of? ----
  // (a) the buffer alone: write-only and read-only streams over the first 0.8 GB of every candidate;
  // (b) the base inside the allocation: the partition-major pattern into the same candidate shifted by 256 KB ... 64 MB (slots shrink accordingly);
  // (c) the pairing with the read streams: the pattern with candidate (i + 1) % nc standing in for the three columns (its first 2.4 GB ... 1.92: two thirds);
  // (d) stability: the partition-major pattern once more, after everything else
  printf("phase 2: %-4s %9s %9s | pm shifted by %8s %8s %8s %8s %8s | %12s %9s\n", "cand", "wo ms", "ro ms", "256KB", "1MB", "4MB", "16MB", "64MB", "pm, other src", "pm again");
  const uint64_t wo_n2 = n / 2;      // 0.8 GB = what pass B writes
  for (int i = 0; i < nc; ++i) {
    const float wo = timed([&] { hipLaunchKernelGGL(k_wo, dim3(G), dim3(kThreads), 0, s, reinterpret_cast<ulonglong2 *>(cand[i]), wo_n2); });
    const float ro = timed([&] { hipLaunchKernelGGL(k_ro, dim3(G), dim3(kThreads), 0, s, reinterpret_cast<const ulonglong2 *>(cand[i]), wo_n2, d_out); });
    float sh[5];
    const uint64_t shifts[5] = {256ull << 10, 1ull << 20, 4ull << 20, 16ull << 20, 64ull << 20};
    for (int k = 0; k < 5; ++k)
      sh[k] = timed([&] { hipLaunchKernelGGL(k_pattern<0>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, cand[i] + shifts[k] / 8, slots - (128ull << 20) / 8, nparts); });
    // the next candidate as the source of all three read streams (n rows of 8 bytes each fit three times into 2/3 of it only when slots * 8 >= 3 * n * 8: use n_src rows)
    const uint64_t n_src = (slots / 3) & ~1ull;
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(cand[(i + 1) % nc]);
    const float other = timed([&] { hipLaunchKernelGGL(k_pattern<0>, dim3(G), dim3(kThreads), 0, s, src, src + n_src / 2, src + n_src, n_src / 2, cand[i], slots, nparts); });
    const float again = timed([&] { hipLaunchKernelGGL(k_pattern<0>, dim3(G), dim3(kThreads), 0, s, col[0], col[1], col[2], n / 2, cand[i], slots, nparts); });
    printf("phase 2: %-4d %9.4f %9.4f | %21.4f %8.4f %8.4f %8.4f %8.4f | %12.4f %9.4f   (source rows %llu of %llu)\n", i, wo, ro, sh[0], sh[1], sh[2], sh[3], sh[4], other, again,
           (unsigned long long)n_src, (unsigned long long)n);
    fflush(stdout);
  }
  return 0;
}
