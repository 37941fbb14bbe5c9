// tools/hipemu — DEVELOPMENT AID (see ../hip/hip_runtime.h): the two hipCUB device algorithms the product calls, on the host.
#pragma once
#include <hip/hip_runtime.h>

#include <numeric>
#include <vector>

namespace hipcub {

struct DeviceRadixSort {
  template <typename K, typename V>
  static hipError_t SortPairs(void *temp, size_t &temp_bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n,
                              int begin_bit = 0, int end_bit = sizeof(K) * 8, hipStream_t = nullptr) {
    if (temp == nullptr) { temp_bytes = 256; return hipSuccess; }
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    const K mask = end_bit - begin_bit >= (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << (end_bit - begin_bit)) - 1);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ((keys_in[a] >> begin_bit) & mask) < ((keys_in[b] >> begin_bit) & mask); });
    for (size_t i = 0; i < n; ++i) { keys_out[i] = keys_in[order[i]]; vals_out[i] = vals_in[order[i]]; }
    return hipSuccess;
  }
};

struct DeviceReduce {
  template <typename K, typename V, typename N, typename Op>
  static hipError_t ReduceByKey(void *temp, size_t &temp_bytes, const K *keys_in, K *unique_out, const V *vals_in, V *aggr_out, N *num_runs,
                                Op op, size_t n, hipStream_t = nullptr) {
    if (temp == nullptr) { temp_bytes = 256; return hipSuccess; }
    size_t runs = 0;
    for (size_t i = 0; i < n;) {
      K k = keys_in[i];
      V a = vals_in[i];
      size_t j = i + 1;
      for (; j < n && keys_in[j] == k; ++j) a = op(a, vals_in[j]);
      unique_out[runs] = k;
      aggr_out[runs] = a;
      runs++;
      i = j;
    }
    *num_runs = (N)runs;
    return hipSuccess;
  }
};

}  // namespace hipcub
