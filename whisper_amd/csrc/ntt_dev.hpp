// Shared between the stand-alone transform kernel (ntt.hip) and the C-ABI layer (api.hip).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "f128.hpp"

namespace wh {

constexpr uint32_t NTT_MAX_LOG2 = 12;      // 4096 points: vector + twiddles = 96 KiB of LDS
// n vectors of n_in elements each (row stride n_in) -> n vectors of 2^lg elements; w = root_of_unity(lg), size_inv = 1 / 2^lg
int launch_ntt_f128(const F128 *d_in, uint32_t n_in, uint32_t lg, bool inverse, const F128 &w, const F128 &size_inv, uint32_t n,
                    F128 *d_out, hipStream_t stream);

}  // namespace wh
