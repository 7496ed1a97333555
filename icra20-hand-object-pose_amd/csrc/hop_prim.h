// Device-wide scans and radix sorts: rocPRIM called directly (utility passes of the list builders, the canonical hypothesis order
// and pcl::VoxelGrid's bucket sort).  Two-phase calls as rocPRIM defines them: temporary_storage == nullptr returns the size.
#pragma once
#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

namespace hop {

template <class T>
inline hipError_t prim_exclusive_sum(void* tmp, size_t& tmp_bytes, const T* in, T* out, size_t n, hipStream_t s) {
  return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}
// stable ascending sort of (key, value) pairs on the key bits [begin_bit, end_bit)
template <class K, class V>
inline hipError_t prim_sort_pairs(void* tmp, size_t& tmp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, size_t n, unsigned begin_bit,
                                  unsigned end_bit, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s);
}

}  // namespace hop
