// Per-report device helpers shared by the hash / chain kernels (kernels.hip, chain.hip): the role lists, the barycentric
// weights and power tables of the query kernels, and the AES-128 testing id.
#pragma once
#include "keccak.hpp"
#include "prio3_dev.hpp"

namespace wh {

// x^e for a small per-lane exponent
__device__ __forceinline__ F128 f_pow_u32(F128 x, uint32_t e) {
    F128 r = f_one();
    while (e) {
        if (e & 1) r = f_mul(r, x);
        e >>= 1;
        if (e) x = f_sqr(x);
    }
    return r;
}

// RLC coefficients of a batched report (prio3.rs:2859-2871): verifier_len differences, 2 decide flags, 2 part checks and
// num_aggregators - 1 shares of zero made from the hash of the public parts
__host__ __device__ __forceinline__ uint32_t rlc_len(const Prio3Dev &P) { return P.verifier_len + 4 + (P.num_aggregators - 1); }
// The query kernels multiply ONE coefficient, rlc[verifier_len + 4], by -H (Leader) or +H (Helper).  With more than two
// aggregators the Leader's share of zero is -H at every one of the num_aggregators - 1 positions and Helper a's is +H at
// position a - 1 (prio3.rs:2844-2855): the lane that emits the coefficients folds that into the one slot.
__device__ __forceinline__ void fold_hash_coefficient(const Prio3Dev &P, F128 *rlc, uint32_t role) {
    if (P.num_aggregators == 2) return;
    F128 *c = rlc + P.verifier_len + 4;
    if (role) { c[0] = c[role - 1]; return; }
    F128 acc = c[0];
    for (uint32_t j = 1; j + 1 < P.num_aggregators; j++) acc = f_add(acc, c[j]);
    c[0] = acc;
}

// derive_joint_rand_seed over the parts with the own one substituted (prio3.rs:482-493, 2564-2578), three parts per pass
// through the caller's sponge loop: parts a0, a0 + 1, a0 + 2 as 16-byte segments from per-lane pointers, so that lanes of
// different aggregator ids stay in phase (keccak.hpp: xor_word).  Returns true when these were the last parts.
__device__ __forceinline__ bool seed_parts(const uint8_t *parts, const uint8_t *own, uint32_t role, uint32_t na, uint32_t a0,
                                           const uint8_t *&p1, uint32_t &l1, const uint8_t *&p2, uint32_t &l2,
                                           const uint8_t *&p3, uint32_t &l3) {
    p1 = a0 == role ? own : parts + 16 * a0; l1 = 16;
    if (a0 + 1 < na) { p2 = a0 + 1 == role ? own : parts + 16 * (a0 + 1); l2 = 16; }
    if (a0 + 2 < na) { p3 = a0 + 2 == role ? own : parts + 16 * (a0 + 2); l3 = 16; }
    return a0 + 3 >= na;
}

// Report handled by slot `slot` of a per-report kernel: the well-formed Leader-role reports
// first, then the Helper-role ones, so that a wavefront (almost always) holds one role only.
__device__ __forceinline__ bool report_of_slot(const Workspace &W, uint32_t slot, uint32_t &i) {
    const uint32_t nl = W.counters[1], nh = W.counters[0];
    if (slot >= nl + nh) return false;
    i = slot < nl ? W.llist[slot] : W.hlist[slot - nl];
    return true;
}

// Per-report constants of the query kernels, computed by the lane that derived the randomness:
// barycentric weights L_j(r) = (r^m - 1)/m * w^j / (r - w^j), j = 0..calls, w = omega_m (one
// inversion per report: prefix products of (r - w^j) are parked in the weight row), and for the
// ParallelSum circuits the weights as limbs and small power tables of rho and r.
__device__ __forceinline__ void weights_and_tables(const Prio3Dev &P, const Workspace &W, uint32_t i, const F128 &r,
                                                   const F128 &rm, const F128 &rho) {
    // Barycentric weights for run 2 (run 1's verifier never reaches the tag, prio3.rs:2713):
    //   L_j(r) = (r^m - 1)/m * w^j / (r - w^j),  j = 0..calls,  w = omega_m.
    // One inversion per report: prefix products of (r - w^j) are parked in the weight row.  The roots are fetched a step
    // ahead of their use, the parked products two.
    F128 *L = W.weights + (uint64_t)i * P.n_weights;
    F128 acc = f_one();
    {
        F128 wn = P.omega[0];
        for (uint32_t j = 0; j < P.n_weights; j++) {
            const F128 w = wn;
            if (j + 1 < P.n_weights) wn = P.omega[j + 1];
            L[j] = acc;
            acc = f_mul(acc, f_sub(r, w));
        }
    }
    // the common factor (r^m - 1)/m rides on the running inverse: three multiplications per weight
    F128 inv = f_mul(f_inv(acc), f_mul(f_sub(rm, f_one()), P.m_inv));
    {
        // prefix products and roots two steps ahead of their use: a step is three multiplications, a row read back after the
        // inversion comes from beyond L2
        const uint32_t nw = P.n_weights;
        F128 p0 = L[nw - 1], p1 = nw > 1 ? L[nw - 2] : f_zero();
        F128 w0 = P.omega[nw - 1], w1 = nw > 1 ? P.omega[nw - 2] : f_zero();
        for (uint32_t j = nw; j-- > 0;) {
            const F128 pre = p0, w = w0;
            p0 = p1; w0 = w1;
            if (j >= 2) { p1 = L[j - 2]; w1 = P.omega[j - 2]; }
            F128 dinv = f_mul(inv, pre);           // (r^m - 1)/m / (r - w^j)
            inv = f_mul(inv, f_sub(r, w));
            L[j] = f_mul(w, dinv);
        }
    }
    // what the query kernels would otherwise compute with a whole wavefront per value: small power tables of rho and r
    RepPow *pw = &W.pows[i];
    F128 t = rho;
    for (int q = 0; q < 8; q++) { pw->rho_lo[q] = t; if (q < 7) t = f_mul(t, rho); }
    F128 u = f_one();
    for (int q = 0; q < 8; q++) { pw->rho_hi[q] = u; u = f_mul(u, t); }     // t = rho^8
    const F128 u2 = f_sqr(u);                                               // u = rho^64
    pw->rho64[0] = f_one(); pw->rho64[1] = u; pw->rho64[2] = u2; pw->rho64[3] = f_mul(u2, u); pw->rho64[4] = f_sqr(u2);
    t = f_one();
    for (int q = 0; q < 8; q++) { pw->r_lo[q] = t; t = f_mul(t, r); }
    u = f_one();
    for (int q = 0; q < 8; q++) { pw->r_hi[q] = u; u = f_mul(u, t); }       // t = r^8
    pw->r64 = u;
    pw->L0 = L[0];
    if (P.circuit != CIRCUIT_AVERAGE) {
        // ParallelSum circuits: the weights as limbs, with M_j = L_j rho^(chunk (j-1)) next to L_j
        const F128 rc = f_pow_u32(rho, P.chunk);
        uint4 *wl = reinterpret_cast<uint4 *>(W.wlimbs + (uint64_t)i * 12 * P.calls);
        F128 pj = f_one(), sumL = f_zero();
        F128 lnext = L[1], lnext2 = P.calls > 1 ? L[2] : f_zero();      // two steps ahead, as above
        for (uint32_t j = 1; j <= P.calls; j++) {
            const F128 l = lnext;
            lnext = lnext2;
            if (j + 1 < P.calls) lnext2 = L[j + 2];
            const F128 m = f_mul(l, pj);
            sumL = f_add(sumL, l);
            uint32_t tl[5], tm[5];
            f_split26(l, tl);
            f_split26(m, tm);
            wl[3 * (j - 1)] = make_uint4(tl[0], tl[1], tl[2], tl[3]);
            wl[3 * (j - 1) + 1] = make_uint4(tl[4], tm[0], tm[1], tm[2]);
            wl[3 * (j - 1) + 2] = make_uint4(tm[3], tm[4], 0, 0);
            pj = f_mul(pj, rc);
        }
        pw->h_sumL = f_mul(P.half, sumL);
    }
}

// testing id: AES128_vk(client_id.to_le_bytes()) (server-batch-prio3/src/main.rs:51, common/src/prg.rs:650-655); aes_rk = 176 B
// of round keys followed by the 256-byte S-box; id_be = the report's first 16 bytes (u128 big endian)
__device__ __forceinline__ void aes_testing_id(const uint8_t *aes_rk, const uint8_t *id_be, uint8_t *out) {
    const uint8_t *sbox = aes_rk + 176;
    uint8_t st[16], t[16];
#pragma unroll
    for (int b = 0; b < 16; b++) st[b] = id_be[15 - b] ^ aes_rk[b];
#pragma unroll 1
    for (int round = 1; round <= 10; round++) {
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) t[4 * c + r] = sbox[st[4 * ((c + r) & 3) + r]];
        if (round < 10) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint8_t b0 = t[4 * c], b1 = t[4 * c + 1], b2 = t[4 * c + 2], b3 = t[4 * c + 3];
#define XT(x) ((uint8_t)(((x) << 1) ^ ((((x) >> 7) & 1) * 0x1b)))
                t[4 * c] = XT(b0) ^ XT(b1) ^ b1 ^ b2 ^ b3;
                t[4 * c + 1] = b0 ^ XT(b1) ^ XT(b2) ^ b2 ^ b3;
                t[4 * c + 2] = b0 ^ b1 ^ XT(b2) ^ XT(b3) ^ b3;
                t[4 * c + 3] = XT(b0) ^ b0 ^ b1 ^ b2 ^ XT(b3);
#undef XT
            }
        }
#pragma unroll
        for (int b = 0; b < 16; b++) st[b] = t[b] ^ aes_rk[16 * round + b];
    }
#pragma unroll
    for (int b = 0; b < 16; b++) out[b] = st[b];
}

}  // namespace wh
