// Shared between the GPU sharder's kernels (shard.hip) and the C-ABI layer (api.hip).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "f128.hpp"

namespace wh {

struct ShardDev {
    uint32_t circuit, bits, len, input_len, proof_len, verifier_len, arity, jr_len, alg_id, leader_len, helper_len, meas_words;
    F128 bound;                 // sampler bound (p)
};
// device buffers of one sub-batch of n clients
struct ShardBuf {
    const uint64_t *meas;       // [n][meas_words]
    const uint8_t *nonces;      // [n][16]
    const uint8_t *rands;       // [n][144]  the seeds shard_with_random_new consumes (prio3.rs:559-883)
    uint64_t first_id;          // client ids first_id + i
    uint8_t *x, *hm;            // [n][input_len]   encoded measurement; Helper's measurement share
    uint8_t *hp[2], *pr[2];     // [n][proof_len] Helper proof shares; [n][arity] prove randomness
    uint8_t *proof[2];          // [n][proof_len]
    uint8_t *jr[2], *jrf;       // [n][jr_len] per run; [n][2 jr_len] fused
    uint8_t *qr[2];             // [n][1]
    uint8_t *pub[2];            // [n][verifier_len] public proofs
    uint8_t *parts;             // [n][4]  joint-rand parts (Leader, Helper), query-rand parts (Leader, Helper)
    uint8_t *lrep, *hrep;       // [n][leader_len], [n][helper_len] encoded reports
};

void launch_shard_stage1(const ShardDev &D, const ShardBuf &B, uint32_t n, hipStream_t st);   // encode, expansions, Leader share, joint randomness
void launch_shard_stage2(const ShardDev &D, const ShardBuf &B, uint32_t n, hipStream_t st);   // Leader proof shares, query randomness
void launch_shard_stage3(const ShardDev &D, const ShardBuf &B, uint32_t n, hipStream_t st);   // report assembly

}  // namespace wh
