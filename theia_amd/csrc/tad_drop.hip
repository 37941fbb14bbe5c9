// tad_drop.hip — abnormal-traffic-drop detector: the same kernel shape as the TAD detectors
// (GROUP BY + per-key moments + threshold) for the reference's Snowflake UDF
//   /root/reference/snowflake/udfs/udfs/drop_detection/drop_detection_udf.py:42-56 (DropDetection.end_partition):
//     fewer than 3 samples -> nothing;  mean = Series.mean(), std = Series.std() (ddof 1);
//     anomaly <=> x > mean + 3 std  or  x < mean - 3 std;  one row per anomalous day with (mean, std).
// Stage 0 is the engine's integer GROUP BY: SUM(dropNumber) per (endpoint+direction key, date)
// (snowflake/cmd/dropDetection.go:151-162).
//
// Bit-exactness: pandas computes mean = sum / n and std = sqrt(sum((mean - x)^2) / (n - 1)) with numpy's float64
// add-reduce, which sums PAIRWISE (8 interleaved accumulators up to 128 elements, recursive halving with the left
// half rounded down to a multiple of 8 above that).  pairwise_sum below follows that order step by step, so the
// kernel returns the bits the reference returns (oracle/drop_oracle.py is pinned to the reference UDF's outputs).
#include "tad_internal.h"

namespace tad {

static constexpr int kDropBlock = 256;

// numpy pairwise_sum_DOUBLE over a[0], a[stride], ..., n elements
__device__ double pairwise_sum(const double *a, size_t stride, uint32_t n) {
  if (n < 8) {
    double r = 0.0;
    for (uint32_t i = 0; i < n; ++i) r += a[i * stride];
    return r;
  }
  if (n <= 128) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
    uint32_t i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i * stride];
    return res;
  }
  uint32_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum(a, stride, n2) + pairwise_sum(a + (size_t)n2 * stride, stride, n - n2);
}

// one lane = one key.  ws[T][K]: the key's present values compacted in time order (position-major, so that lanes
// stay coalesced), then reused for the squared deviations.
__global__ __launch_bounds__(kDropBlock) void k_drop_detect(Grid g, double n_sigma, uint32_t min_samples, double *__restrict__ ws,
                                                            double *__restrict__ sigma, uint32_t *__restrict__ n_pts,
                                                            double *__restrict__ key_mean, double *__restrict__ key_m2,
                                                            DevCounters *ctr) {
  const uint64_t k = (uint64_t)blockIdx.x * kDropBlock + threadIdx.x;
  unsigned long long my_pts = 0;
  unsigned my_key = 0, my_skip = 0;
  if (k < g.K) {
    const size_t st = g.K;
    double *col = ws + k;
    uint32_t n = 0;
    for (uint64_t t = 0; t < g.T; ++t) {
      const uint64_t c = t * g.K + k;
      if (g.flag[c] & FLAG_PRESENT) { col[(size_t)n * st] = (double)g.val[c]; n++; }
    }
    n_pts[k] = n;
    my_pts = n;
    my_key = n > 0;
    double mean = 0.0, std = 0.0, m2 = 0.0;
    if (n >= min_samples && n >= 2) {
      mean = pairwise_sum(col, st, n) / (double)n;
      for (uint32_t i = 0; i < n; ++i) { const double d = mean - col[(size_t)i * st]; col[(size_t)i * st] = d * d; }
      m2 = pairwise_sum(col, st, n);
      std = sqrt(m2 / (double)(n - 1));
      const double upper = mean + n_sigma * std, lower = mean - n_sigma * std;
      for (uint64_t t = 0; t < g.T; ++t) {
        const uint64_t c = t * g.K + k;
        const uint8_t fl = g.flag[c];
        if (fl & FLAG_PRESENT) {
          const double x = (double)g.val[c];
          if (x > upper || x < lower) g.flag[c] = fl | FLAG_ANOMALY;
        }
      }
    } else if (n > 0) {
      // too few samples: the UDF yields nothing for this partition (drop_detection_udf.py:44-45)
      for (uint64_t t = 0; t < g.T; ++t) g.flag[t * g.K + k] = 0;
      my_skip = 1;
    }
    sigma[k] = std;
    key_mean[k] = mean;
    key_m2[k] = m2;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    my_pts += __shfl_down(my_pts, d);
    my_key += __shfl_down(my_key, d);
    my_skip += __shfl_down(my_skip, d);
  }
  if ((threadIdx.x & 63) == 0 && my_key) {
    atomicAdd(&ctr->n_points, my_pts);
    atomicAdd(&ctr->n_keys, (unsigned long long)my_key);
    if (my_skip) atomicAdd(&ctr->keys_no_result, (unsigned long long)my_skip);
  }
}

void launch_drop(hipStream_t s, Grid g, double n_sigma, int min_samples, double *ws, double *sigma, uint32_t *n_pts,
                 double *key_mean, double *key_m2, DevCounters *ctr) {
  if (g.K == 0) return;
  const int blocks = (int)((g.K + kDropBlock - 1) / kDropBlock);
  hipLaunchKernelGGL(k_drop_detect, dim3(blocks), dim3(kDropBlock), 0, s, g, n_sigma, (uint32_t)(min_samples < 0 ? 0 : min_samples), ws,
                     sigma, n_pts, key_mean, key_m2, ctr);
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_drop() { return reinterpret_cast<const void *>(&k_drop_detect); }

}  // namespace tad
