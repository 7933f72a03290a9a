// sort.hip -- the (bucket key, point index) radix sort of the MSM pipeline: rocPRIM, in its own TU.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "runtime.hpp"

namespace nmx {
// rocPRIM switches from merge sort to onesweep at 2^20 items by default; on gfx950 onesweep already wins from ~10^5
// pairs (bench/sort_test.hip: 2.4e5 pairs 0.094 -> 0.069 ms, 1.0e6 pairs 0.160 -> 0.075 ms), which is the range of
// the 10 k - 200 k-pair MSMs of prove_step.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                              (size_t)1 << 17>;
void device_sort_pairs(void* tmp, size_t& tmp_bytes, uint32_t* k_in, uint32_t* k_out, uint32_t* v_in,
                       uint32_t* v_out, size_t total, uint32_t bits, hipStream_t stream) {
  HIPCHK(rocprim::radix_sort_pairs<SortConfig>(tmp, tmp_bytes, k_in, k_out, v_in, v_out, total, 0u, bits, stream));
}
}  // namespace nmx
