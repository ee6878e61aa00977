// Prio3::shard_batched (shard_with_random_new, batched-flpcp-rs/src/vdaf/prio3.rs:537-991) + VdafBatchedKey::encode
// (vdaf.rs:148-162) on gfx950 for a batch of clients: everything around the two proofs (prove.hip) and the two public
// proofs (the query kernels): measurement encoding, Helper share and prove-randomness expansion, Leader shares, the
// joint-randomness and query-randomness Fiat-Shamir parts, and the two encoded reports per client.  It exists to make
// large batches of DISTINCT valid reports for the verifier (the reference's meta-client does this on the CPU,
// client-batch-prio3/src/main.rs:67-153) and is checked byte for byte against the CPU client (client.cpp).
#include <hip/hip_runtime.h>
#include "keccak.hpp"
#include "prio3_dev.hpp"
#include "shard_dev.hpp"

namespace wh {
namespace {

__device__ __forceinline__ void st128u(uint8_t *p, const F128 &v) { __builtin_memcpy(p, &v, 16); }

// Type::encode_measurement (types.rs:582-606, 388-399, 246-253): one thread per element
__global__ __launch_bounds__(256) void k_shard_encode(ShardDev D, ShardBuf B, uint32_t n) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)n * D.input_len) return;
    const uint64_t i = g / D.input_len;
    const uint32_t e = (uint32_t)(g - i * D.input_len);
    const uint64_t *m = B.meas + i * D.meas_words;
    uint64_t bit;
    if (D.circuit == CIRCUIT_SUMVEC) { const uint32_t l = e % D.bits; bit = l < 64 ? (m[e / D.bits] >> l) & 1 : 0; }
    else if (D.circuit == CIRCUIT_HISTOGRAM) bit = m[0] == e;
    else bit = e < 64 ? (m[0] >> e) & 1 : 0;
    st128u(B.x + 16 * g, F128{bit, 0});
}

// XofShake128::seed_stream(seed, dst(usage), binder).into_field_vec: the Helper's measurement share (one lane per client)
__global__ __launch_bounds__(64) void k_shard_expand_meas(ShardDev D, ShardBuf B, uint32_t n) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    SpongeG s;
    s.init_xof(D.alg_id, DST_MEAS, B.rands + 144ull * i);
    s.absorb_byte(1);
    s.finalize();
    xof_emit(s, B.hm + 16ull * D.input_len * i, D.input_len, D.bound);
}
// the four short streams of a client: Helper proof shares (seeds 1, 2; binder [1]) and prove randomness (seeds 5, 6)
__global__ __launch_bounds__(64) void k_shard_expand_short(ShardDev D, ShardBuf B, uint32_t n) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= 4 * n) return;
    const uint32_t i = g >> 2, k = g & 3, t = k & 1;
    const uint8_t *r = B.rands + 144ull * i;
    SpongeG s;
    if (k < 2) {
        s.init_xof(D.alg_id, DST_PROOF, r + 16 + 16 * t);
        s.absorb_byte(1);
        s.finalize();
        xof_emit(s, B.hp[t] + 16ull * D.proof_len * i, D.proof_len, D.bound);
    } else {
        s.init_xof(D.alg_id, DST_PROVE_RAND, r + 80 + 16 * t);
        s.finalize();
        xof_emit(s, B.pr[t] + 16ull * D.arity * i, D.arity, D.bound);
    }
}

// out[i] = a[i] - b[i] element-wise into the Leader report: its measurement share (which = 0), proof share 1 or 2
__global__ __launch_bounds__(256) void k_shard_leader_share(ShardDev D, ShardBuf B, uint32_t n, uint32_t which) {
    const uint32_t len = which == 0 ? D.input_len : D.proof_len;
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)n * len) return;
    const uint64_t i = g / len;
    const uint32_t e = (uint32_t)(g - i * len);
    const uint8_t *a = which == 0 ? B.x : B.proof[which - 1], *b = which == 0 ? B.hm : B.hp[which - 1];
    // Leader report: id 16 | joint-rand parts 32 | agg id 1 | meas | proof 1 | joint-rand blind 16 | proof 2 | ...
    const uint64_t off = 49 + (which >= 1 ? 16ull * D.input_len : 0) + (which == 2 ? 16ull * D.proof_len + 16 : 0);
    st128u(B.lrep + i * D.leader_len + off + 16ull * e, f_sub(ld128(a + 16 * g), ld128(b + 16 * g)));
}

__device__ __forceinline__ F128 shfl_f(const F128 &v, int src) {
    F128 o;
    o.lo = __shfl(v.lo, src, 64); o.hi = __shfl(v.hi, src, 64);
    return o;
}
__device__ __forceinline__ F128 derive_seed(uint32_t alg, const F128 &p0, const F128 &p1) {   // prio3.rs:482-493
    Sponge s;
    s.init_xof(alg, DST_JR_SEED, 0ull, 0ull);
    s.absorb_f(p0); s.absorb_f(p1);
    return s.into_seed();
}

// joint-randomness parts (prio3.rs:589-700): lane pair per client, even lane = Leader (blind 4, its share in the report),
// odd lane = Helper (blind 3, the expanded share); then the seed over both parts and the fused joint randomness
__global__ __launch_bounds__(64) void k_shard_joint_rand(ShardDev D, ShardBuf B, uint32_t n) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x, i = g >> 1, agg = g & 1;
    const bool live = i < n;
    F128 part = f_zero();
    if (live) {
        const uint8_t *r = B.rands + 144ull * i;
        SpongeG s;
        s.init_xof(D.alg_id, DST_JR_PART, r + (agg ? 48 : 64));
        s.absorb_byte((uint8_t)agg);
        s.absorb(B.nonces + 16ull * i, 16);
        s.absorb(agg ? B.hm + 16ull * D.input_len * i : B.lrep + (uint64_t)D.leader_len * i + 49, 16 * D.input_len);
        part = s.into_seed();
        st128u(B.parts + 64ull * i + 16 * agg, part);
    }
    const F128 other = shfl_f(part, (int)((threadIdx.x ^ 1)));
    if (live && agg == 0) {
        const F128 seed = derive_seed(D.alg_id, part, other);
        Sponge s;
        s.init_xof(D.alg_id, DST_JOINT_RAND, seed.lo, seed.hi);
        s.finalize();
        F128 jr[4];
        xof_emit(s, reinterpret_cast<uint8_t *>(jr), 2 * D.jr_len, D.bound);
        for (uint32_t q = 0; q < 2 * D.jr_len; q++) {
            st128u(B.jrf + 16ull * (2 * D.jr_len * (uint64_t)i + q), jr[q]);
            st128u(B.jr[q / D.jr_len] + 16ull * (D.jr_len * (uint64_t)i + q % D.jr_len), jr[q]);
        }
    }
}

// query-randomness parts (prio3.rs:826-949): blind 8 / 7, the fused joint randomness, both proof shares
__global__ __launch_bounds__(64) void k_shard_query_rand(ShardDev D, ShardBuf B, uint32_t n) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x, i = g >> 1, agg = g & 1;
    const bool live = i < n;
    F128 part = f_zero();
    if (live) {
        const uint8_t *r = B.rands + 144ull * i;
        const uint8_t *lr = B.lrep + (uint64_t)D.leader_len * i + 49 + 16ull * D.input_len;
        SpongeG s;
        s.init_xof(D.alg_id, DST_QR_PART, r + (agg ? 112 : 128));
        s.absorb(B.jrf + 32ull * D.jr_len * i, 32 * D.jr_len);
        s.absorb(agg ? B.hp[0] + 16ull * D.proof_len * i : lr, 16 * D.proof_len);
        s.absorb(agg ? B.hp[1] + 16ull * D.proof_len * i : lr + 16ull * D.proof_len + 16, 16 * D.proof_len);
        part = s.into_seed();
        st128u(B.parts + 64ull * i + 32 + 16 * agg, part);
    }
    const F128 other = shfl_f(part, (int)((threadIdx.x ^ 1)));
    if (live && agg == 0) {
        const F128 seed = derive_seed(D.alg_id, part, other);     // DST_JOINT_RAND_SEED again (prio3.rs:933)
        Sponge s;
        s.init_xof(D.alg_id, DST_QUERY_RAND, seed.lo, seed.hi);
        s.finalize();
        F128 qr[2];
        xof_emit(s, reinterpret_cast<uint8_t *>(qr), 2, D.bound);
        st128u(B.qr[0] + 16ull * i, qr[0]);
        st128u(B.qr[1] + 16ull * i, qr[1]);
    }
}

// VdafBatchedKey::encode (vdaf.rs:148-162): everything of the two reports that is not a Leader share
__global__ __launch_bounds__(256) void k_shard_assemble(ShardDev D, ShardBuf B, uint32_t n) {
    const uint64_t i = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    uint8_t *L = B.lrep + i * D.leader_len, *H = B.hrep + i * D.helper_len;
    const uint8_t *r = B.rands + 144 * i, *parts = B.parts + 64 * i, *nonce = B.nonces + 16 * i;
    const uint64_t id = B.first_id + i;
    const uint64_t l_tail = 49 + 16ull * D.input_len + 32ull * D.proof_len + 16, h_tail = 49 + 64;
    const uint32_t vb = 16 * D.verifier_len;
    if (tid < 16) { const uint8_t b = tid < 8 ? 0 : (uint8_t)(id >> (8 * (15 - tid))); L[tid] = b; H[tid] = b; }     // u128 big endian
    if (tid < 32) { L[16 + tid] = parts[tid]; H[16 + tid] = parts[tid]; }
    if (tid == 0) { L[48] = 0; H[48] = 1; }
    if (tid < 16) {
        L[49 + 16ull * D.input_len + 16ull * D.proof_len + tid] = r[64 + tid];                                     // Leader joint-rand blind
        H[49 + tid] = r[tid]; H[65 + tid] = r[16 + tid]; H[81 + tid] = r[48 + tid]; H[97 + tid] = r[32 + tid];     // seeds: meas, proof 1, blind, proof 2
    }
    if (tid < 32) { L[l_tail + tid] = parts[32 + tid]; H[h_tail + tid] = parts[32 + tid]; }
    if (tid < 8) { const uint8_t b = (uint8_t)((uint64_t)D.verifier_len >> (56 - 8 * tid)); L[l_tail + 32 + tid] = b; H[h_tail + 32 + tid] = b; }
    for (uint32_t k = tid; k < 2 * vb; k += 256) {
        const uint8_t b = k < vb ? B.pub[0][(uint64_t)vb * i + k] : B.pub[1][(uint64_t)vb * i + k - vb];
        L[l_tail + 40 + k] = b; H[h_tail + 40 + k] = b;
    }
    if (tid < 16) {
        L[l_tail + 40 + 2 * vb + tid] = r[128 + tid]; H[h_tail + 40 + 2 * vb + tid] = r[112 + tid];               // query-rand blinds
        L[l_tail + 56 + 2 * vb + tid] = nonce[tid]; H[h_tail + 56 + 2 * vb + tid] = nonce[tid];
    }
}

}  // namespace

void launch_shard_stage1(const ShardDev &D, const ShardBuf &B, uint32_t n, hipStream_t st) {
    const uint64_t el = (uint64_t)n * D.input_len;
    hipLaunchKernelGGL(k_shard_encode, dim3((uint32_t)((el + 255) / 256)), dim3(256), 0, st, D, B, n);
    hipLaunchKernelGGL(k_shard_expand_meas, dim3((n + 63) / 64), dim3(64), 0, st, D, B, n);
    hipLaunchKernelGGL(k_shard_expand_short, dim3((4 * n + 63) / 64), dim3(64), 0, st, D, B, n);
    hipLaunchKernelGGL(k_shard_leader_share, dim3((uint32_t)((el + 255) / 256)), dim3(256), 0, st, D, B, n, 0u);
    hipLaunchKernelGGL(k_shard_joint_rand, dim3((2 * n + 63) / 64), dim3(64), 0, st, D, B, n);
}
void launch_shard_stage2(const ShardDev &D, const ShardBuf &B, uint32_t n, hipStream_t st) {
    const uint64_t el = (uint64_t)n * D.proof_len;
    hipLaunchKernelGGL(k_shard_leader_share, dim3((uint32_t)((el + 255) / 256)), dim3(256), 0, st, D, B, n, 1u);
    hipLaunchKernelGGL(k_shard_leader_share, dim3((uint32_t)((el + 255) / 256)), dim3(256), 0, st, D, B, n, 2u);
    hipLaunchKernelGGL(k_shard_query_rand, dim3((2 * n + 63) / 64), dim3(64), 0, st, D, B, n);
}
void launch_shard_stage3(const ShardDev &D, const ShardBuf &B, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_shard_assemble, dim3(n), dim3(256), 0, st, D, B, n);
}

}  // namespace wh
