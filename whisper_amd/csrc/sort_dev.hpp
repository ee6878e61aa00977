// Shared between the testing-id sort (sort.hip) and the C-ABI layer (api.hip).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace wh {

constexpr uint32_t SORT_TILE = 1024;       // elements per wavefront-sized workgroup and pass
inline uint64_t sort_blocks(uint64_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
// scratch: hist[256 * sort_blocks(n) + 256] (tile counts bin-major, then the bin totals) and tmp[n] (the second permutation buffer); returns a hipError_t as int
int launch_sort_by_id(const uint8_t *d_ids, uint32_t n, uint32_t *d_order, uint32_t *d_tmp, uint32_t *d_hist, hipStream_t st);
// keep[i] = (ids[i] not in bad), bad = n_bad pairs (lo, hi) ascending as u128
int launch_keep_mask(const uint8_t *d_ids, uint32_t n, const uint64_t *d_bad, uint32_t n_bad, uint8_t *d_keep, hipStream_t st);
int launch_sort_gather(const uint8_t *d_rows, uint32_t row_bytes, const uint32_t *d_order, uint32_t n, uint8_t *d_out, hipStream_t st);

}  // namespace wh
