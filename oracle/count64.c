/*
 * ORACLE (test infrastructure).  Prio3Count = Prio3<Count<Field64>, XofShake128, 16>, the reference's Field64
 * instantiation, restated from
 *   batched-flpcp-rs/src/fp.rs:359-389          (FP64: p = 2^64 - 2^32 + 1, generator 7)
 *   batched-flpcp-rs/src/field.rs:598-607, 649-675 (little-endian encoding, ModulusOverflow, try_from_random)
 *   batched-flpcp-rs/src/prng.rs:73-99          (rejection sampler over ENCODED_SIZE = 8 byte chunks)
 *   batched-flpcp-rs/src/flp/types.rs:39-105    (Count: one Mul gadget, valid = g(x, x) - x; lengths 1 / 5 / 4 / 1)
 *   batched-flpcp-rs/src/flp.rs:231-347, 359-479, 483-512, 677-724 (prove, query with the shim gadget, decide)
 *   batched-flpcp-rs/src/vdaf/prio3.rs:81, 1316-1494 (shard without joint randomness), 2216-2345 (prepare_init),
 *                                     2347-2442 (prepare_shares_to_prepare_message, prepare_next), 2445-2456 (aggregate)
 *   batched-flpcp-rs/src/vdaf.rs:209-233        (VdafKey: agg_id | public share (empty here) | input share | nonce)
 * Pinned by the reference's own vectors test_vec/07/Prio3Count_{0,1}.json (tests/test_oracle_count64.py).
 * Residues are kept canonical (the reference keeps Montgomery form; every encoded value is the same).
 */
#include "oracle_internal.h"
#include <string.h>

#define P64 0xffffffff00000001ull
enum { DST_MEAS = 1, DST_PROOF = 2, DST_PROVE_RAND = 4, DST_QUERY_RAND = 5 };

static uint64_t a64(uint64_t a, uint64_t b) { u128 s = (u128)a + b; return (uint64_t)(s >= P64 ? s - P64 : s); }
static uint64_t s64(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P64 - b); }
static uint64_t m64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P64); }
static uint64_t pow64(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = m64(r, a); a = m64(a, a); e >>= 1; } return r; }
static uint64_t inv64(uint64_t a) { return pow64(a, P64 - 2); }
static uint64_t ld64(const uint8_t *b) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | b[i]; return v; }
static void st64(uint8_t *b, uint64_t v) { for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); }
static int dec64(const uint8_t *b, uint64_t *out) { uint64_t v = ld64(b); if (v >= P64) return WO_ERR_MODULUS; *out = v; return 0; }

/* Prng::get over XOF(seed, dst) . binder for Field64 (prng.rs:73-99): consecutive 8-byte chunks, chunks >= p skipped */
static void expand64(const uint8_t seed[16], uint16_t usage, const uint8_t *binder, size_t binder_len, uint64_t *out, size_t n) {
    /* vdaf.rs:339-346: dst = version 7, algorithm class 0, algorithm id (Prio3Count = 0) big endian, usage big endian */
    const uint8_t dst[8] = {7, 0, 0, 0, 0, 0, (uint8_t)(usage >> 8), (uint8_t)usage};
    shake_t s;
    xof_init(&s, seed, dst, 8);
    if (binder_len) shake_absorb(&s, binder, binder_len);
    shake_finalize(&s);
    size_t got = 0;
    while (got < n) {
        uint8_t b[8];
        shake_squeeze(&s, b, 8);
        uint64_t v = ld64(b);
        if (v < P64) out[got++] = v;
    }
}

static uint64_t poly_eval64(const uint64_t *c, size_t n, uint64_t x) {
    uint64_t r = 0;
    for (size_t i = n; i-- > 0;) r = a64(m64(r, x), c[i]);
    return r;
}

/* Type::prove for Count (flp.rs:231-347): wire polynomials through (1, seed_w), (-1, x); gadget polynomial = their product */
static void prove64(uint64_t x, const uint64_t prove_rand[2], uint64_t proof[5]) {
    const uint64_t half = inv64(2);
    uint64_t f[2][2];
    for (int w = 0; w < 2; w++) {
        f[w][0] = m64(a64(prove_rand[w], x), half);      /* size-2 inverse transform: (v0 + v1) / 2, (v0 - v1) / 2 */
        f[w][1] = m64(s64(prove_rand[w], x), half);
        proof[w] = prove_rand[w];
    }
    proof[2] = m64(f[0][0], f[1][0]);
    proof[3] = a64(m64(f[0][0], f[1][1]), m64(f[0][1], f[1][0]));
    proof[4] = m64(f[0][1], f[1][1]);
}

/* Type::query for Count (flp.rs:359-479 with QueryShimGadget :677-724): verifier = [G(-1) - x, f_0(r), f_1(r), G(r)] */
static int query64(uint64_t x, const uint64_t proof[5], uint64_t r, uint64_t verifier[4]) {
    if (m64(r, r) == 1) return WO_ERR_QUERY_RAND;        /* r^m == 1 with m = 2 (flp.rs:413-419) */
    const uint64_t half = inv64(2), minus1 = P64 - 1;
    verifier[0] = s64(poly_eval64(proof + 2, 3, minus1), x);   /* p_vals[ct * step] = G(w^1), w = -1 */
    for (int w = 0; w < 2; w++) {
        const uint64_t c0 = m64(a64(proof[w], x), half), c1 = m64(s64(proof[w], x), half);
        verifier[1 + w] = a64(c0, m64(c1, r));
    }
    verifier[3] = poly_eval64(proof + 2, 3, r);
    return 0;
}

size_t wo_count_leader_share_len(void) { return 48; }

/* Prio3::shard_with_random without joint randomness (prio3.rs:1316-1494): rand = per Helper (meas seed, proof seed), then the
 * prove-randomness seed.  leader_share: enc(meas share) | enc(proof share), 48 bytes; helper_shares: 32 bytes each. */
int wo_count_shard(uint32_t num_aggregators, uint64_t meas, const uint8_t nonce[16], const uint8_t *rand, uint8_t *leader_share,
                   uint8_t *helper_shares) {
    (void)nonce;
    if (num_aggregators < 2 || num_aggregators > 254 || meas > 1) return WO_ERR_PARAM;   /* check_num_aggregators, prio3.rs:2945-2957 */
    uint64_t x = meas, leader_meas = x, proof[5], prove_rand[2], t[5];
    const uint8_t *rnd = rand;
    for (uint32_t a = 1; a < num_aggregators; a++) {
        uint8_t id = (uint8_t)a;
        memcpy(helper_shares + 32 * (a - 1), rnd, 32);
        expand64(rnd, DST_MEAS, &id, 1, t, 1);
        leader_meas = s64(leader_meas, t[0]);
        rnd += 32;
    }
    expand64(rnd, DST_PROVE_RAND, NULL, 0, prove_rand, 2);
    prove64(x, prove_rand, proof);
    for (uint32_t a = 1; a < num_aggregators; a++) {
        uint8_t id = (uint8_t)a;
        expand64(helper_shares + 32 * (a - 1) + 16, DST_PROOF, &id, 1, t, 5);
        for (int i = 0; i < 5; i++) proof[i] = s64(proof[i], t[i]);
    }
    st64(leader_share, leader_meas);
    for (int i = 0; i < 5; i++) st64(leader_share + 8 + 8 * i, proof[i]);
    return 0;
}

/* Prio3::prepare_init for Count (prio3.rs:2216-2345; no joint randomness, so the prepare share is the verifier share alone):
 * prep_share 32 bytes, out_share 8 bytes (truncate is the identity, types.rs:73-76). */
int wo_count_prepare_init(uint32_t num_aggregators, const uint8_t verify_key[16], uint32_t agg_id, const uint8_t nonce[16],
                          const uint8_t *input_share, size_t input_share_len, uint8_t prep_share[32], uint8_t out_share[8]) {
    if (agg_id >= num_aggregators) return WO_ERR_AGG_ID;
    uint64_t x, proof[5], r, v[4];
    int rc;
    if (agg_id == 0) {
        if (input_share_len < 48) return WO_ERR_SHORT;
        if ((rc = dec64(input_share, &x))) return rc;
        for (int i = 0; i < 5; i++) if ((rc = dec64(input_share + 8 + 8 * i, &proof[i]))) return rc;
    } else {
        if (input_share_len < 32) return WO_ERR_SHORT;
        uint8_t id = (uint8_t)agg_id;
        expand64(input_share, DST_MEAS, &id, 1, &x, 1);
        expand64(input_share + 16, DST_PROOF, &id, 1, proof, 5);
    }
    expand64(verify_key, DST_QUERY_RAND, nonce, 16, &r, 1);       /* prio3.rs:2232-2239 */
    if ((rc = query64(x, proof, r, v))) return rc;
    for (int i = 0; i < 4; i++) st64(prep_share + 8 * i, v[i]);
    st64(out_share, x);
    return 0;
}

/* VdafKey::decode_with_param + prepare_init on an encoded key (vdaf.rs:209-233; server-base-prio3/src/main.rs:41-61) */
int wo_count_prepare_init_encoded(uint32_t num_aggregators, const uint8_t verify_key[16], const uint8_t *key, size_t key_len,
                                  uint8_t prep_share[32], uint8_t out_share[8]) {
    if (key_len < 1) return WO_ERR_SHORT;
    const uint32_t agg_id = key[0];
    if (agg_id >= num_aggregators) return WO_ERR_AGG_ID;
    const size_t share_len = agg_id == 0 ? 48 : 32;
    if (key_len < 1 + share_len + 16) return WO_ERR_SHORT;
    return wo_count_prepare_init(num_aggregators, verify_key, agg_id, key + 1 + share_len, key + 1, share_len, prep_share, out_share);
}

/* prepare_shares_to_prepare_message + decide (prio3.rs:2347-2405, flp.rs:483-512): 0 = accept, 1 = reject, < 0 decode error.
 * The prepare message is empty for Count (no joint-randomness seed). */
int wo_count_prepare_finish(uint32_t num_aggregators, const uint8_t *prep_shares) {
    uint64_t v[4] = {0, 0, 0, 0};
    for (uint32_t a = 0; a < num_aggregators; a++)
        for (int i = 0; i < 4; i++) {
            uint64_t t;
            int rc = dec64(prep_shares + 32 * a + 8 * i, &t);
            if (rc) return rc;
            v[i] = a64(v[i], t);
        }
    return (v[0] == 0 && m64(v[1], v[2]) == v[3]) ? 0 : 1;
}

/* Aggregator::aggregate (prio3.rs:2445-2456, vdaf.rs:658-676) over the kept out shares */
void wo_count_aggregate(const uint8_t *out_shares, uint64_t n, const uint8_t *keep, uint8_t agg[8]) {
    uint64_t acc = 0;
    for (uint64_t i = 0; i < n; i++) if (!keep || keep[i]) acc = a64(acc, ld64(out_shares + 8 * i));
    st64(agg, acc);
}

/* Field64 sampler on its own, for the GPU's test hook */
void wo_f64_expand(const uint8_t seed[16], uint32_t usage, const uint8_t *binder, size_t binder_len, uint64_t *out, size_t n) {
    expand64(seed, (uint16_t)usage, binder, binder_len, out, n);
}
