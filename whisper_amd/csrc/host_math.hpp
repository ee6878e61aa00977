// Host-side helpers shared by the C-ABI layer (constant tables) and the CPU client/report
// generator: circuit sizes, roots of unity and a radix-2 NTT over Field128 residues.
// Reference: flp/types.rs size formulas, fp.rs:391-421 (generator of the 2^66 subgroup),
// fft.rs:31-115 (what the transform computes; the butterfly order here is our own).
#pragma once
#include <vector>
#include "../../include/whisper_gfx950.h"
#include "f128.hpp"
#include "prio3_dev.hpp"

namespace wh {

inline uint32_t next_pow2_u32(uint32_t x) { uint32_t r = 1; while (r < x) r <<= 1; return r; }
inline uint32_t log2_u32(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) l++; return l; }

struct HostSizes {
    uint32_t input_len, proof_len, verifier_len, output_len, jr_len, calls, m, log2m, arity, gpoly_len;
    uint32_t alg_id, leader_len, helper_len;
};

inline int32_t host_sizes(const wh_prio3_params &p, HostSizes &s) {
    if (p.num_aggregators < 2 || p.num_aggregators > 254) return WH_ERR_PARAM;   // check_num_aggregators, prio3.rs:2945-2957 (1 is legal there: no sharing, refused here)
    switch (p.circuit) {
    case WH_SUMVEC:
        if (!p.bits || !p.len || !p.chunk || p.bits > 127 || (uint64_t)p.bits * p.len > (1u << 24)) return WH_ERR_PARAM;
        s.input_len = p.bits * p.len; s.calls = (s.input_len + p.chunk - 1) / p.chunk; s.arity = 2 * p.chunk;
        s.output_len = p.len; s.jr_len = 1; s.alg_id = 2;
        break;
    case WH_HISTOGRAM:
        if (!p.len || !p.chunk || p.len > (1u << 24)) return WH_ERR_PARAM;
        s.input_len = p.len; s.calls = (p.len + p.chunk - 1) / p.chunk; s.arity = 2 * p.chunk;
        s.output_len = p.len; s.jr_len = 2; s.alg_id = 3;
        break;
    case WH_AVERAGE:
        if (!p.bits || p.bits > 127) return WH_ERR_PARAM;
        s.input_len = p.bits; s.calls = p.bits; s.arity = 1; s.output_len = 1; s.jr_len = 1; s.alg_id = 0xFFFF0000u;
        break;
    default: return WH_ERR_PARAM;
    }
    if (p.alg_id) s.alg_id = p.alg_id;
    s.m = next_pow2_u32(1 + s.calls);
    s.log2m = log2_u32(s.m);
    if (s.log2m + 1 > 20) return WH_ERR_PARAM;  // fft.rs:42-44 MAX_ROOTS
    s.gpoly_len = 2 * (s.m - 1) + 1;
    s.proof_len = s.arity + s.gpoly_len;
    s.verifier_len = 2 + s.arity;
    // VdafBatchedKey (vdaf.rs:148-162): client id | parts | agg id | share | parts | num_queries | 2 public proofs | blind | nonce
    uint32_t common = 16 + 16 * p.num_aggregators + 1 + 16 * p.num_aggregators + 8 + 2 * 16 * s.verifier_len + 16 + 16;
    s.leader_len = common + 16 * (s.input_len + 2 * s.proof_len) + 16;
    s.helper_len = common + 64;
    return WH_OK;
}

// primitive 2^l-th root of unity: g^(2^(66-l)), g = 7^((p-1)/2^66)
inline F128 root_of_unity(uint32_t l) {
    static const F128 g = f_pow(f_small(7), (f_p() - 1) >> 66);
    return f_sqr_n(g, 66 - l);
}

// in-place NTT: a[i] <- sum_j a[j] w^(ij), w = root_of_unity(log2 n); inverse divides by n
inline void ntt(std::vector<F128> &a, bool inverse) {
    const uint32_t n = (uint32_t)a.size(), lg = log2_u32(n);
    for (uint32_t i = 1, j = 0; i < n; i++) {
        uint32_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { F128 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (uint32_t l = 1; l <= lg; l++) {
        F128 wl = root_of_unity(l);
        if (inverse) wl = f_inv(wl);
        uint32_t half = 1u << (l - 1);
        for (uint32_t s = 0; s < n; s += 2 * half) {
            F128 w = f_one();
            for (uint32_t k = 0; k < half; k++) {
                F128 u = a[s + k], v = f_mul(a[s + k + half], w);
                a[s + k] = f_add(u, v);
                a[s + k + half] = f_sub(u, v);
                w = f_mul(w, wl);
            }
        }
    }
    if (inverse) {
        F128 ninv = f_inv(f_small(n));
        for (auto &x : a) x = f_mul(x, ninv);
    }
}

inline F128 horner(const F128 *c, uint32_t n, const F128 &x) {
    F128 r = f_zero();
    for (uint32_t i = n; i-- > 0;) r = f_add(f_mul(r, x), c[i]);
    return r;
}

}  // namespace wh
