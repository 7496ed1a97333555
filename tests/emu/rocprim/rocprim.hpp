// tests/emu/rocprim/rocprim.hpp -- TEST INFRASTRUCTURE: the two rocPRIM calls the product makes (csrc/hop_prim.h), stated sequentially for the
// functional HIP model (tests/emu/hip/hip_runtime.h).  Two-phase protocol as rocPRIM defines it.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include <hip/hip_runtime.h>

namespace rocprim {
template <class T>
struct plus {
  T operator()(const T& a, const T& b) const { return a + b; }
};
template <class In, class Out, class T, class Op>
inline hipError_t exclusive_scan(void* tmp, size_t& tmp_bytes, In in, Out out, T init, size_t n, Op op, hipStream_t = nullptr) {
  if (!tmp) {
    tmp_bytes = 16;
    return hipSuccess;
  }
  T acc = init;
  for (size_t i = 0; i < n; ++i) {
    const T v = (T)in[i];  // (in and out may alias)
    out[i] = acc;
    acc = op(acc, v);
  }
  return hipSuccess;
}
template <class K, class V>
inline hipError_t radix_sort_pairs(void* tmp, size_t& tmp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, size_t n, unsigned begin_bit,
                                   unsigned end_bit, hipStream_t = nullptr) {
  if (!tmp) {
    tmp_bytes = 16;
    return hipSuccess;
  }
  const unsigned bits = end_bit - begin_bit;
  const K mask = bits >= sizeof(K) * 8 ? ~K(0) : (K)(((K)1 << bits) - 1);
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), (size_t)0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ((keys_in[a] >> begin_bit) & mask) < ((keys_in[b] >> begin_bit) & mask); });
  std::vector<K> k(n);
  std::vector<V> v(n);
  for (size_t i = 0; i < n; ++i) k[i] = keys_in[order[i]], v[i] = vals_in[order[i]];
  std::copy(k.begin(), k.end(), keys_out);
  std::copy(v.begin(), v.end(), vals_out);
  return hipSuccess;
}
}  // namespace rocprim
