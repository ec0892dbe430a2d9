// mfx_sort.hip -- stable radix sort of (owner rank -> position index) pairs for the
// sharded-index query exchange (hipCUB on rocPRIM; 8-bit keys, one pass).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "mfx_internal.h"

int mfx_sort_by_owner(void *tmp, size_t &tmp_bytes, const uint8_t *kin, uint8_t *kout, const uint32_t *vin, uint32_t *vout,
                      uint64_t n, hipStream_t st) {
  // hipCUB counts its items in an int; the values sorted here are 32-bit position indexes of one routed chunk
  // (mfx_route_tiles: max_tiles * MFX_TILE positions), so a chunk beyond 2^31 - 1 keys is refused, never truncated
  if (n > (uint64_t)INT32_MAX) return mfx_fail(MFX_E_INVAL, "mfx_sort_by_owner: %lu keys in one routed chunk (at most %d)", (unsigned long)n, INT32_MAX);
  if (tmp == nullptr) {
    size_t b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint8_t *)nullptr, (uint8_t *)nullptr, (const uint32_t *)nullptr,
                                       (uint32_t *)nullptr, (int)n, 0, 8, st);
    tmp_bytes = b;
    return MFX_OK;
  }
  if (n == 0) return MFX_OK;
  size_t b = tmp_bytes;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, b, kin, kout, vin, vout, (int)n, 0, 8, st);
  if (e != hipSuccess) return mfx_fail(MFX_E_HIP, "radix sort failed: %s", hipGetErrorString(e));
  return MFX_OK;
}
