// sortlib.hip — the rocPRIM calls of the build (meshes above the sample sort's range) and of the generic query path, in a translation unit of
// their own: rocPRIM's radix sort and select instantiate ~2 MB of device code per call site, and the HIP runtime loads a translation unit's
// code object — all of it — at the first launch of any of its kernels.  A grid call over a mesh the sample sort handles launches nothing
// from here, so its first call in a process no longer loads this code (bvh.o 2.0 -> 0.18 MB, distance.o 3.5 -> 1.05 MB: rocPRIM was three quarters of the 5.6 MB a
// first grid call used to load; profiles/r06_first_call_v3.txt).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace m2s {

// the build's sort of meshes above the sample sort's range (> 229 376 triangles)
hipError_t sort_pairs_u64(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                          unsigned begin_bit, unsigned end_bit, hipStream_t st) {
  return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st);
}

// m2s_warmup: one empty kernel per translation unit (the runtime loads a unit's code object at the first launch out of it)
__global__ void k_warm_sortlib() {}
void warm_sortlib(hipStream_t st) { hipLaunchKernelGGL(k_warm_sortlib, dim3(1), dim3(64), 0, st); }

}  // namespace m2s
