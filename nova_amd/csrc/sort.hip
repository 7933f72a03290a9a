// sort.hip -- the (bucket key, point index) radix sort of the MSM pipeline: rocPRIM, in its own TU.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "runtime.hpp"

namespace nmx {
void device_sort_pairs(void* tmp, size_t& tmp_bytes, uint32_t* k_in, uint32_t* k_out, uint32_t* v_in,
                       uint32_t* v_out, size_t total, uint32_t bits, hipStream_t stream) {
  HIPCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, total, 0u, bits, stream));
}
}  // namespace nmx
