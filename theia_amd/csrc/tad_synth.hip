// tad_synth.hip — deterministic synthetic flow table, generated straight into HBM.
// Bit-for-bit the same table as oracle/tad_oracle.py:synth_rows (tests/test_synth.py checks it):
//   h(s, i)   = mix64(seed + GOLDEN * (8 i + s))                 (splitmix64 finaliser, wrapping)
//   key_id    = h(1, i) mod K          bucket = h(2, i) mod T
//   flow_end_s= 1660202814 + 60 * bucket   (2022-08-11T07:26:54Z + 1 min steps — the e2e fixture's
//               timeline, test/e2e/throughputanomalydetection_test.go:402-403,440)
//   base_k    = 1e9 + mix64(seed ^ GOLDEN * (key_id + 3)) mod 3e9 ;  J = base_k / 1000
//   value     = base_k + (h(4, i) mod (2J + 1)) - J
//   r = h(5, i): (r & 8191) == 0 -> value *= 2 + (r >> 20) mod 10   (spike, p = 2^-13)
//                else ((r >> 13) & 16383) == 0 -> value /= 2 + (r >> 40) mod 18   (dip, p = 2^-14)
#include "tad_internal.h"

namespace tad {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_synth(uint64_t seed, uint64_t first_row, uint64_t n_rows,
                                               uint64_t K, uint64_t T, uint64_t *__restrict__ key_id,
                                               int64_t *__restrict__ flow_end_s,
                                               uint64_t *__restrict__ value) {
  const uint64_t golden = 0x9E3779B97F4A7C15ull;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_rows; j += stride) {
    const uint64_t i = first_row + j;
    const uint64_t key = mix64(seed + golden * (i * 8 + 1)) % K;
    const uint64_t bucket = mix64(seed + golden * (i * 8 + 2)) % T;
    const uint64_t base = 1000000000ull + mix64(seed ^ (golden * (key + 3))) % 3000000000ull;
    const uint64_t jit = base / 1000;
    uint64_t v = base + mix64(seed + golden * (i * 8 + 4)) % (2 * jit + 1) - jit;
    const uint64_t r = mix64(seed + golden * (i * 8 + 5));
    if ((r & 8191) == 0) v *= 2 + (r >> 20) % 10;
    else if (((r >> 13) & 16383) == 0) v /= 2 + (r >> 40) % 18;
    key_id[j] = key;
    flow_end_s[j] = 1660202814ll + 60ll * (int64_t)bucket;
    value[j] = v;
  }
}

void launch_synth(hipStream_t s, uint64_t seed, uint64_t first_row, uint64_t n_rows, uint64_t num_keys,
                  uint64_t n_buckets, uint64_t *key_id, int64_t *flow_end_s, uint64_t *value) {
  if (n_rows == 0) return;
  uint64_t blocks = (n_rows + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, s, seed, first_row, n_rows, num_keys,
                     n_buckets, key_id, flow_end_s, value);
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_synth() { return reinterpret_cast<const void *>(&k_synth); }

}  // namespace tad
