// Device-wide primitives: rocPRIM called directly (scans, stable LSD radix sorts).  Count-then-run protocol as in
// rocPRIM: call with tmp == nullptr to get `bytes`.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <iterator>

template <class In, class Out>
static inline hipError_t phx_exclusive_sum(void *tmp, size_t &bytes, In in, Out out, size_t n, hipStream_t st) {
  using T = typename std::iterator_traits<Out>::value_type;
  return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}

// stable; sorts on the key bits [bit0, bit1)
template <class K, class V>
static inline hipError_t phx_sort_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout,
                                        size_t n, unsigned bit0, unsigned bit1, hipStream_t st) {
  return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, bit0, bit1, st);
}
