// pt_sort.hip -- key/index sort for the LBVH build.
//
// The reference's 16-pass 2-bit LSD radix sort (radix-sort.js:162-500) has one observable effect:
// a stable sort of the u32 keys that carries the permutation idx[i] = original position.  That is a
// plain library operation, so it is rocPRIM's device radix sort (stable, 32 key bits), kept in its
// own translation unit because the rocPRIM headers dominate compile time.
#include <cstring>
#include <cstdlib>
#include "pt_internal.h"
#include <rocprim/device/device_radix_sort.hpp>

hipError_t pt_sort_pairs_u32(void* tmp, size_t* tmp_bytes, const uint32_t* kin, uint32_t* kout,
                             const int32_t* vin, int32_t* vout, size_t n, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, 0, 32, s);
}
