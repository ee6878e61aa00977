// Shared between the prover kernel (prove.hip) and the C-ABI layer (api.hip).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "f128.hpp"

namespace wh {

struct ProveDev {
    uint32_t input_len, chunk, calls, m, log2m, arity, gpoly_len, proof_len, jr_len;
    const F128 *omega2m;  // [2m]  w_2m^k, w_2m = root_of_unity(log2 m + 1)
    const F128 *twist;    // [m]   w_2m^bitrev(pos) / m
    F128 inv2m;           // 1 / 2m
};

constexpr uint32_t PROVE_MAX_M = 512;
size_t flp_prove_lds_bytes(uint32_t m);
int launch_flp_prove(const ProveDev &D, bool pairs, const uint8_t *d_input, const uint8_t *d_prove_rand, const uint8_t *d_joint_rand,
                     uint32_t n, uint8_t *d_proof, int32_t *d_status, hipStream_t stream);

}  // namespace wh
