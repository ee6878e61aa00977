/*
 * ORACLE (test infrastructure).  Prio3 sharding / preparation, un-batched (draft-07,
 * pinned by the reference's JSON vectors) and whisper's batched variant, restated from
 *   batched-flpcp-rs/src/vdaf/prio3.rs:69-78      (usage constants)
 *                                       :482-493   (derive_joint_rand_seed)
 *                                       :537-991   (shard_with_random_new)
 *                                       :1316-1494 (shard_with_random)
 *                                       :1549-1875 (wire structs)
 *                                       :2216-2345 (prepare_init)
 *                                       :2445-2456 (aggregate)
 *                                       :2466-2879 (prepare_batched)
 *   batched-flpcp-rs/src/vdaf.rs:148-192 (VdafBatchedKey codec), 339-346 (dst), 658-676 (merge)
 *   batched-flpcp-rs/src/codec.rs:237-267 (big-endian integers)
 *   server-batch-prio3/src/main.rs:33-60 (prepare_encoded_key), 108-118 (aggregation loop)
 *   common/src/prg.rs:639-655 (Prf)
 */
#include "oracle_internal.h"
#include <pthread.h>
#include <string.h>

enum { DST_MEAS = 1, DST_PROOF = 2, DST_JOINT_RAND = 3, DST_PROVE_RAND = 4, DST_QUERY_RAND = 5,
       DST_JR_SEED = 6, DST_JR_PART = 7, DST_QR_PART = 8, DST_RLC = 9, DST_HASH = 10 };

/* vdaf.rs:339-346 */
static void make_dst(const wo_params *p, uint16_t usage, uint8_t dst[8]) {
    uint32_t id = wo_alg_id(p);
    dst[0] = 7; dst[1] = 0;
    dst[2] = id >> 24; dst[3] = id >> 16; dst[4] = id >> 8; dst[5] = id;
    dst[6] = usage >> 8; dst[7] = usage;
}
static void xof_init_usage(shake_t *s, const wo_params *p, const uint8_t seed[16], uint16_t usage) {
    uint8_t dst[8];
    make_dst(p, usage, dst);
    xof_init(s, seed, dst, 8);
}
static void expand_usage(const wo_params *p, const uint8_t seed[16], uint16_t usage, const uint8_t *binder,
                         size_t binder_len, u128 *out, size_t n) {
    uint8_t dst[8];
    make_dst(p, usage, dst);
    xof_expand(seed, dst, 8, binder, binder_len, out, n);
}
static void absorb_vec(shake_t *s, const u128 *v, size_t n) {
    uint8_t b[16];
    for (size_t i = 0; i < n; i++) { f_encode(v[i], b); shake_absorb(s, b, 16); }
}
/* prio3.rs:482-493 */
static void derive_joint_rand_seed(const wo_params *p, const uint8_t *parts, uint32_t nparts, uint8_t out[16]) {
    static const uint8_t zero[16] = {0};
    shake_t s;
    xof_init_usage(&s, p, zero, DST_JR_SEED);
    shake_absorb(&s, parts, 16 * (size_t)nparts);
    xof_into_seed(&s, out);
}
/* joint-rand part: XOF(blind, dst 7) . [agg_id] . nonce . enc(meas share)  (prio3.rs:2543-2555) */
static void joint_rand_part(const wo_params *p, const uint8_t blind[16], uint8_t agg_id, const uint8_t nonce[16],
                            const u128 *meas, size_t n, uint8_t out[16]) {
    shake_t s;
    xof_init_usage(&s, p, blind, DST_JR_PART);
    shake_absorb(&s, &agg_id, 1);
    shake_absorb(&s, nonce, 16);
    absorb_vec(&s, meas, n);
    xof_into_seed(&s, out);
}
/* query-rand part: XOF(blind, dst 8) . enc(joint_rand_fused) . enc(proof1) . enc(proof2)  (prio3.rs:2610-2642) */
static void query_rand_part(const wo_params *p, const uint8_t blind[16], const u128 *jr_fused, size_t njr,
                            const u128 *proof1, const u128 *proof2, size_t proof_len, uint8_t out[16]) {
    shake_t s;
    xof_init_usage(&s, p, blind, DST_QR_PART);
    absorb_vec(&s, jr_fused, njr);
    absorb_vec(&s, proof1, proof_len);
    absorb_vec(&s, proof2, proof_len);
    xof_into_seed(&s, out);
}
static void enc_vec(const u128 *v, size_t n, uint8_t *out) { for (size_t i = 0; i < n; i++) f_encode(v[i], out + 16 * i); }
static int dec_vec(const uint8_t *in, size_t n, u128 *out) {
    for (size_t i = 0; i < n; i++) { int rc = f_decode(in + 16 * i, &out[i]); if (rc) return rc; }
    return 0;
}

/* ------------------------------------------------------- un-batched draft-07 ---- */
int wo_prio3_shard(const wo_params *p, const uint64_t *meas, const uint8_t nonce[16], const uint8_t *rand,
                   uint8_t *public_share, uint8_t *leader_share, uint8_t *helper_shares) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    uint32_t na = p->num_aggregators;
    const uint8_t *rnd = rand;
    u128 *enc = wo_alloc(sz.input_len * sizeof(u128)), *leader_meas = wo_alloc(sz.input_len * sizeof(u128));
    u128 *tmp = wo_alloc((sz.input_len > sz.proof_len ? sz.input_len : sz.proof_len) * sizeof(u128));
    rc = flp_encode(p, meas, enc);
    if (rc) goto done;
    memcpy(leader_meas, enc, sz.input_len * sizeof(u128));
    /* helpers: seeds in the order meas, proof, joint-rand blind (prio3.rs:1351-1360) */
    for (uint32_t a = 1; a < na; a++) {
        const uint8_t *meas_seed = rnd, *proof_seed = rnd + 16, *blind = rnd + 32;
        rnd += 48;
        uint8_t id = (uint8_t)a;
        expand_usage(p, meas_seed, DST_MEAS, &id, 1, tmp, sz.input_len);
        for (uint32_t i = 0; i < sz.input_len; i++) leader_meas[i] = f_sub(leader_meas[i], tmp[i]);
        joint_rand_part(p, blind, id, nonce, tmp, sz.input_len, public_share + 16 * a);
        uint8_t *hs = helper_shares + 48 * (a - 1);
        memcpy(hs, meas_seed, 16); memcpy(hs + 16, proof_seed, 16); memcpy(hs + 32, blind, 16);
    }
    const uint8_t *leader_blind = rnd; rnd += 16;
    joint_rand_part(p, leader_blind, 0, nonce, leader_meas, sz.input_len, public_share);
    uint8_t jr_seed[16];
    derive_joint_rand_seed(p, public_share, na, jr_seed);
    u128 jr[2];
    expand_usage(p, jr_seed, DST_JOINT_RAND, NULL, 0, jr, sz.joint_rand_len);
    const uint8_t *prove_seed = rnd; rnd += 16;
    u128 *prove_rand = wo_alloc(sz.prove_rand_len * sizeof(u128)), *proof = wo_alloc(sz.proof_len * sizeof(u128));
    expand_usage(p, prove_seed, DST_PROVE_RAND, NULL, 0, prove_rand, sz.prove_rand_len);
    rc = flp_prove(p, enc, prove_rand, jr, proof);
    for (uint32_t a = 1; a < na && !rc; a++) {
        uint8_t id = (uint8_t)a;
        expand_usage(p, helper_shares + 48 * (a - 1) + 16, DST_PROOF, &id, 1, tmp, sz.proof_len);
        for (uint32_t i = 0; i < sz.proof_len; i++) proof[i] = f_sub(proof[i], tmp[i]);
    }
    enc_vec(leader_meas, sz.input_len, leader_share);
    enc_vec(proof, sz.proof_len, leader_share + 16 * (size_t)sz.input_len);
    memcpy(leader_share + 16 * (size_t)(sz.input_len + sz.proof_len), leader_blind, 16);
    wo_free(prove_rand); wo_free(proof);
done:
    wo_free(enc); wo_free(leader_meas); wo_free(tmp);
    return rc;
}

int wo_prio3_prepare_init(const wo_params *p, const uint8_t verify_key[16], uint32_t agg_id, const uint8_t nonce[16],
                          const uint8_t *public_share, const uint8_t *input_share, size_t input_share_len,
                          uint8_t *prep_share, uint8_t *out_share) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    uint32_t na = p->num_aggregators;
    if (agg_id >= na) return WO_ERR_AGG_ID;
    u128 *meas = wo_alloc(sz.input_len * sizeof(u128)), *proof = wo_alloc(sz.proof_len * sizeof(u128));
    u128 *verifier = wo_alloc(sz.verifier_len * sizeof(u128)), *out = wo_alloc(sz.output_len * sizeof(u128));
    uint8_t id = (uint8_t)agg_id;
    const uint8_t *blind;
    if (agg_id == 0) {
        if (input_share_len < 16 * (size_t)(sz.input_len + sz.proof_len) + 16) { rc = WO_ERR_SHORT; goto done; }
        if ((rc = dec_vec(input_share, sz.input_len, meas))) goto done;
        if ((rc = dec_vec(input_share + 16 * (size_t)sz.input_len, sz.proof_len, proof))) goto done;
        blind = input_share + 16 * (size_t)(sz.input_len + sz.proof_len);
    } else {
        if (input_share_len < 48) { rc = WO_ERR_SHORT; goto done; }
        expand_usage(p, input_share, DST_MEAS, &id, 1, meas, sz.input_len);
        expand_usage(p, input_share + 16, DST_PROOF, &id, 1, proof, sz.proof_len);
        blind = input_share + 32;
    }
    u128 query_rand[1], jr[2];
    expand_usage(p, verify_key, DST_QUERY_RAND, nonce, 16, query_rand, 1); /* prio3.rs:2232-2239 */
    uint8_t parts[16 * 256], own[16], jr_seed[16];
    joint_rand_part(p, blind, id, nonce, meas, sz.input_len, own);
    memcpy(parts, public_share, 16 * (size_t)na);
    memcpy(parts + 16 * agg_id, own, 16);
    derive_joint_rand_seed(p, parts, na, jr_seed);
    expand_usage(p, jr_seed, DST_JOINT_RAND, NULL, 0, jr, sz.joint_rand_len);
    rc = flp_query(p, meas, proof, query_rand, jr, na, verifier);
    if (rc) goto done;
    enc_vec(verifier, sz.verifier_len, prep_share);
    memcpy(prep_share + 16 * (size_t)sz.verifier_len, own, 16);
    flp_truncate(p, meas, out); /* prepare_next, prio3.rs:2426-2439 */
    enc_vec(out, sz.output_len, out_share);
done:
    wo_free(meas); wo_free(proof); wo_free(verifier); wo_free(out);
    return rc;
}

/* -------------------------------------------------------------- batched shard ---- */
static void put_be64(uint8_t *b, uint64_t v) { for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (56 - 8 * i)); }

int wo_prio3_shard_batched(const wo_params *p, const uint64_t *meas, const uint8_t nonce[16], const uint8_t *rand,
                           const uint8_t client_id_be[16], uint8_t *leader_report, uint8_t *helper_report) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    /* Alice and Bob are two aggregators (server-batch-prio3); shard_batched itself is generic in their number and its own
     * test runs three (prio3.rs:3077-3085).  rand: 16 (5 na - 1) bytes (random_size_double_field, prio3.rs:515-533);
     * helper_report: na - 1 reports of helper_report_len bytes, aggregator 1 first. */
    const uint32_t na = p->num_aggregators, nh = na - 1;
    /* random_seeds order, prio3.rs:579-581,589 (per Helper), 647, 705, 753, 826 (per Helper), 883 */
    const uint8_t *l_jr_blind = rand + 64 * (size_t)nh, *prove_seed[2] = {l_jr_blind + 16, l_jr_blind + 32},
                  *h_qr_blinds = l_jr_blind + 48, *l_qr_blind = h_qr_blinds + 16 * (size_t)nh;
#define H_MEAS_SEED(a) (rand + 64 * (size_t)((a) - 1))
#define H_PROOF_SEED(a, t) (rand + 64 * (size_t)((a) - 1) + 16 + 16 * (t))
#define H_JR_BLIND(a) (rand + 64 * (size_t)((a) - 1) + 48)
    uint32_t njr = 2 * sz.joint_rand_len;
    u128 *enc = wo_alloc(sz.input_len * sizeof(u128)), *h_meas = wo_alloc(sz.input_len * sizeof(u128));
    u128 *l_meas = wo_alloc(sz.input_len * sizeof(u128));
    u128 *prove_rand = wo_alloc(sz.prove_rand_len * sizeof(u128));
    u128 *proof[2], *l_proof[2], *h_proof[2], *pub_proof[2];
    for (int t = 0; t < 2; t++) {
        proof[t] = wo_alloc(sz.proof_len * sizeof(u128)); l_proof[t] = wo_alloc(sz.proof_len * sizeof(u128));
        h_proof[t] = wo_alloc(sz.proof_len * sizeof(u128)); pub_proof[t] = wo_alloc(sz.verifier_len * sizeof(u128));
    }
    uint8_t jr_parts[16 * 256], qr_parts[16 * 256], seed[16];
    u128 jr_fused[4], qr_fused[2];
    rc = flp_encode(p, meas, enc);
    if (rc) goto done;
    memcpy(l_meas, enc, sz.input_len * sizeof(u128));
    for (uint32_t a = 1; a < na; a++) {                                      /* prio3.rs:574-638 */
        uint8_t id = (uint8_t)a;
        expand_usage(p, H_MEAS_SEED(a), DST_MEAS, &id, 1, h_meas, sz.input_len);
        for (uint32_t i = 0; i < sz.input_len; i++) l_meas[i] = f_sub(l_meas[i], h_meas[i]);
        joint_rand_part(p, H_JR_BLIND(a), id, nonce, h_meas, sz.input_len, jr_parts + 16 * a);
    }
    joint_rand_part(p, l_jr_blind, 0, nonce, l_meas, sz.input_len, jr_parts);
    derive_joint_rand_seed(p, jr_parts, na, seed);
    expand_usage(p, seed, DST_JOINT_RAND, NULL, 0, jr_fused, njr);           /* prio3.rs:675-688 */
    for (int t = 0; t < 2; t++) {                                            /* prio3.rs:703-782 */
        expand_usage(p, prove_seed[t], DST_PROVE_RAND, NULL, 0, prove_rand, sz.prove_rand_len);
        if ((rc = flp_prove(p, enc, prove_rand, jr_fused + t * sz.joint_rand_len, proof[t]))) goto done;
        memcpy(l_proof[t], proof[t], sz.proof_len * sizeof(u128));
        for (uint32_t a = 1; a < na; a++) {
            uint8_t id = (uint8_t)a;
            expand_usage(p, H_PROOF_SEED(a, t), DST_PROOF, &id, 1, h_proof[t], sz.proof_len);
            for (uint32_t i = 0; i < sz.proof_len; i++) l_proof[t][i] = f_sub(l_proof[t][i], h_proof[t][i]);
        }
    }
    for (uint32_t a = 1; a < na; a++) {                                      /* prio3.rs:806-874 */
        uint8_t id = (uint8_t)a;
        for (int t = 0; t < 2; t++) expand_usage(p, H_PROOF_SEED(a, t), DST_PROOF, &id, 1, h_proof[t], sz.proof_len);
        query_rand_part(p, h_qr_blinds + 16 * (size_t)(a - 1), jr_fused, njr, h_proof[0], h_proof[1], sz.proof_len, qr_parts + 16 * a);
    }
    query_rand_part(p, l_qr_blind, jr_fused, njr, l_proof[0], l_proof[1], sz.proof_len, qr_parts);
    derive_joint_rand_seed(p, qr_parts, na, seed);                           /* dst 6 again, prio3.rs:933 */
    expand_usage(p, seed, DST_QUERY_RAND, NULL, 0, qr_fused, 2);
    for (int t = 0; t < 2; t++)                                              /* prio3.rs:951-969, num_shares = 1 */
        if ((rc = flp_query(p, enc, proof[t], qr_fused + t, jr_fused + t * sz.joint_rand_len, 1, pub_proof[t]))) goto done;

    /* VdafBatchedKey::encode, vdaf.rs:148-162 */
    for (uint32_t a = 0; a < na; a++) {
        uint8_t *o = a ? helper_report + (size_t)(a - 1) * sz.helper_report_len : leader_report;
        memcpy(o, client_id_be, 16); o += 16;
        memcpy(o, jr_parts, 16 * (size_t)na); o += 16 * (size_t)na;
        *o++ = (uint8_t)a;
        if (a == 0) {
            enc_vec(l_meas, sz.input_len, o); o += 16 * (size_t)sz.input_len;
            enc_vec(l_proof[0], sz.proof_len, o); o += 16 * (size_t)sz.proof_len;
            memcpy(o, l_jr_blind, 16); o += 16;
            enc_vec(l_proof[1], sz.proof_len, o); o += 16 * (size_t)sz.proof_len;
        } else {
            memcpy(o, H_MEAS_SEED(a), 16); memcpy(o + 16, H_PROOF_SEED(a, 0), 16); memcpy(o + 32, H_JR_BLIND(a), 16);
            memcpy(o + 48, H_PROOF_SEED(a, 1), 16); o += 64;
        }
        memcpy(o, qr_parts, 16 * (size_t)na); o += 16 * (size_t)na;
        put_be64(o, sz.verifier_len); o += 8;   /* num_queries, client-batch-prio3/src/main.rs:111-112 */
        enc_vec(pub_proof[0], sz.verifier_len, o); o += 16 * (size_t)sz.verifier_len;
        enc_vec(pub_proof[1], sz.verifier_len, o); o += 16 * (size_t)sz.verifier_len;
        memcpy(o, a ? h_qr_blinds + 16 * (size_t)(a - 1) : l_qr_blind, 16); o += 16;
        memcpy(o, nonce, 16);
    }
#undef H_MEAS_SEED
#undef H_PROOF_SEED
#undef H_JR_BLIND
done:
    for (int t = 0; t < 2; t++) { wo_free(proof[t]); wo_free(l_proof[t]); wo_free(h_proof[t]); wo_free(pub_proof[t]); }
    wo_free(enc); wo_free(h_meas); wo_free(l_meas); wo_free(prove_rand);
    return rc;
}

/* ------------------------------------------------------------ batched prepare ---- */
int wo_prio3_prepare_encoded_key(const wo_params *p, const uint8_t verify_key[16], const uint8_t *rep, size_t rep_len,
                                 uint8_t tag_out[16], uint8_t testing_id[16], uint8_t *out_share) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    const uint32_t na = p->num_aggregators;   /* the servers run 2; prepare_batched is generic (prio3.rs:2843-2855) */
    const size_t psb = 16 * (size_t)na;       /* bytes of a public share: one part per aggregator */
    uint32_t njr = 2 * sz.joint_rand_len;
    u128 *meas = wo_alloc(sz.input_len * sizeof(u128)), *out = wo_alloc(sz.output_len * sizeof(u128));
    u128 *proof[2] = {wo_alloc(sz.proof_len * sizeof(u128)), wo_alloc(sz.proof_len * sizeof(u128))};
    u128 *pp[2] = {wo_alloc(sz.verifier_len * sizeof(u128)), wo_alloc(sz.verifier_len * sizeof(u128))};
    u128 *ver[2] = {wo_alloc(sz.verifier_len * sizeof(u128)), wo_alloc(sz.verifier_len * sizeof(u128))};
    u128 *rlc = NULL;

    /* --- VdafBatchedKey::decode_with_param (vdaf.rs:164-192); the server calls it on a Cursor
     * and does not look at trailing bytes (server-batch-prio3/src/main.rs:46-47).  Every
     * failure below is an unwrap()/assert abort in the reference; the codes only name the
     * cause.  Structure is checked before values so that the code does not depend on the
     * decoder's read order: SHORT / AGG_ID / NUM_QUERIES first, then MODULUS. --- */
    const size_t share_len[2] = {16 * (size_t)(sz.input_len + 2 * sz.proof_len) + 16, 64};
    if (rep_len < 17 + psb) { rc = WO_ERR_SHORT; goto done; }
    const uint8_t *client_id = rep, *ps_parts = rep + 16;
    uint8_t agg_id = rep[16 + psb];
    if (agg_id >= na) { rc = WO_ERR_AGG_ID; goto done; }   /* role_try_from via decode, prio3.rs:1780-1782 */
    size_t off = 17 + psb, share_off = 17 + psb;
    off += share_len[agg_id != 0];
    if (rep_len < off + psb + 8) { rc = WO_ERR_SHORT; goto done; }
    const uint8_t *ps2_parts = rep + off; off += psb;
    uint64_t num_queries = 0;
    for (int i = 0; i < 8; i++) num_queries = (num_queries << 8) | rep[off + i];
    off += 8;
    if (num_queries != sz.verifier_len) {
        /* the decoder reads num_queries answers per public proof, then prepare_batched
         * asserts the length (prio3.rs:2712) */
        rc = (num_queries > (rep_len - off) / 32 || rep_len - off - 32 * (size_t)num_queries < 32) ? WO_ERR_SHORT : WO_ERR_NUM_QUERIES;
        goto done;
    }
    if (rep_len < off + 32 * (size_t)sz.verifier_len + 32) { rc = WO_ERR_SHORT; goto done; }
    const uint8_t *pp_bytes = rep + off; off += 32 * (size_t)sz.verifier_len;
    const uint8_t *qr_blind = rep + off, *nonce = rep + off + 16, *jr_blind;
    if (agg_id == 0) {
        const uint8_t *q = rep + share_off;
        if ((rc = dec_vec(q, sz.input_len, meas))) goto done;
        q += 16 * (size_t)sz.input_len;
        if ((rc = dec_vec(q, sz.proof_len, proof[0]))) goto done;
        q += 16 * (size_t)sz.proof_len;
        jr_blind = q;
        q += 16;
        if ((rc = dec_vec(q, sz.proof_len, proof[1]))) goto done;
    } else {
        /* prio3.rs:2480-2536 */
        const uint8_t *q = rep + share_off;
        expand_usage(p, q, DST_MEAS, &agg_id, 1, meas, sz.input_len);
        expand_usage(p, q + 16, DST_PROOF, &agg_id, 1, proof[0], sz.proof_len);
        jr_blind = q + 32;
        expand_usage(p, q + 48, DST_PROOF, &agg_id, 1, proof[1], sz.proof_len);
    }
    if ((rc = dec_vec(pp_bytes, sz.verifier_len, pp[0]))) goto done;
    if ((rc = dec_vec(pp_bytes + 16 * (size_t)sz.verifier_len, sz.verifier_len, pp[1]))) goto done;

    /* --- prepare_batched --- */
    flp_truncate(p, meas, out);                                              /* prio3.rs:2496 */
    uint8_t own_jr[16], own_qr[16], parts[16 * 256], seed[16];
    u128 jr_fused[4], qr_fused[2];
    joint_rand_part(p, jr_blind, agg_id, nonce, meas, sz.input_len, own_jr); /* :2543-2555 */
    memcpy(parts, ps_parts, psb); memcpy(parts + 16 * agg_id, own_jr, 16);   /* :2564-2576 */
    derive_joint_rand_seed(p, parts, na, seed);
    expand_usage(p, seed, DST_JOINT_RAND, NULL, 0, jr_fused, njr);           /* :2580-2585 */
    query_rand_part(p, qr_blind, jr_fused, njr, proof[0], proof[1], sz.proof_len, own_qr); /* :2610-2642 */
    memcpy(parts, ps2_parts, psb); memcpy(parts + 16 * agg_id, own_qr, 16);
    derive_joint_rand_seed(p, parts, na, seed);                              /* :2665 (dst 6) */
    expand_usage(p, seed, DST_QUERY_RAND, NULL, 0, qr_fused, 2);             /* :2667-2672 */
    for (int t = 0; t < 2; t++)                                              /* :2687-2701 */
        if ((rc = flp_query(p, meas, proof[t], qr_fused + t, jr_fused + t * sz.joint_rand_len, na, ver[t]))) goto done;

    /* :2713-2739 -- run 1's difference is computed and dropped; run 2's is kept */
    u128 *diff = ver[1];
    if (agg_id == 0) for (uint32_t i = 0; i < sz.verifier_len; i++) diff[i] = f_sub(ver[1][i], pp[1][i]);
    u128 one = f_from_int(1);
    u128 decide_zero[2] = {flp_decide(p, pp[0]) ? 0 : one, flp_decide(p, pp[1]) ? 0 : one}; /* :2762-2786 */
    u128 rand_diffs[2] = {memcmp(ps_parts + 16 * agg_id, own_jr, 16) ? one : 0,            /* :2790-2805 */
                          memcmp(ps2_parts + 16 * agg_id, own_qr, 16) ? one : 0};
    /* :2808-2840 hash of the public parts */
    static const uint8_t zero[16] = {0};
    shake_t hs;
    xof_init_usage(&hs, p, zero, DST_HASH);
    shake_absorb(&hs, ps_parts, psb);
    shake_absorb(&hs, ps2_parts, psb);
    shake_absorb(&hs, pp_bytes, 32 * (size_t)sz.verifier_len); /* canonical bytes: decode checked < p */
    xof_into_seed(&hs, seed);
    u128 hash;
    expand_usage(p, seed, DST_HASH, NULL, 0, &hash, 1);
    /* :2859-2862 */
    uint32_t check_count = 2 * sz.verifier_len + 2 + 2 + (na - 1);
    rlc = wo_alloc(check_count * sizeof(u128));
    expand_usage(p, verify_key, DST_RLC, nonce, 16, rlc, check_count);
    /* :2864-2871 -- zip stops after verifier_len + 4 + (na - 1) terms */
    u128 tag = 0;
    uint32_t k = 0;
    for (uint32_t i = 0; i < sz.verifier_len; i++) tag = f_add(tag, f_mul(diff[i], rlc[k++]));
    for (int i = 0; i < 2; i++) tag = f_add(tag, f_mul(decide_zero[i], rlc[k++]));
    for (int i = 0; i < 2; i++) tag = f_add(tag, f_mul(rand_diffs[i], rlc[k++]));
    for (uint32_t j = 0; j + 1 < na; j++) {                                  /* :2844-2855: Leader -H everywhere, Helper a +H at a - 1 */
        u128 hash_share = agg_id == 0 ? f_neg(hash) : (j == (uint32_t)agg_id - 1 ? hash : 0);
        tag = f_add(tag, f_mul(hash_share, rlc[k++]));
    }

    /* prepare_encoded_key, main.rs:51-59 */
    uint8_t id_le[16];
    for (int i = 0; i < 16; i++) id_le[i] = client_id[15 - i]; /* u128 BE on the wire -> to_le_bytes() */
    wo_aes128_encrypt_block(verify_key, id_le, testing_id);    /* read back as u128 LE */
    if (agg_id != 0) tag = f_mul(tag, f_neg(one));
    f_encode(tag, tag_out);
    enc_vec(out, sz.output_len, out_share);
done:
    wo_free(meas); wo_free(out); wo_free(rlc);
    for (int t = 0; t < 2; t++) { wo_free(proof[t]); wo_free(pp[t]); wo_free(ver[t]); }
    return rc;
}

typedef struct {
    const wo_params *p; const uint8_t *vk, *blob; const uint64_t *offsets; uint64_t n;
    uint8_t *tags, *ids, *out_shares; int32_t *status; uint32_t output_len;
    uint64_t next; pthread_mutex_t mu;
} batch_t;
static void *batch_worker(void *arg) {
    batch_t *b = arg;
    for (;;) {
        pthread_mutex_lock(&b->mu);
        uint64_t lo = b->next, hi = lo + 16 < b->n ? lo + 16 : b->n;
        b->next = hi;
        pthread_mutex_unlock(&b->mu);
        if (lo >= hi) return NULL;
        for (uint64_t i = lo; i < hi; i++)
            b->status[i] = wo_prio3_prepare_encoded_key(b->p, b->vk, b->blob + b->offsets[i], b->offsets[i + 1] - b->offsets[i],
                                                        b->tags + 16 * i, b->ids + 16 * i, b->out_shares + 16 * (size_t)b->output_len * i);
    }
}
int wo_prio3_prepare_batch(const wo_params *p, const uint8_t verify_key[16], const uint8_t *blob, const uint64_t *offsets,
                           uint64_t n, unsigned nthreads, uint8_t *tags, uint8_t *testing_ids, int32_t *status, uint8_t *out_shares) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    batch_t b = {p, verify_key, blob, offsets, n, tags, testing_ids, out_shares, status, sz.output_len, 0, PTHREAD_MUTEX_INITIALIZER};
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (unsigned t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &b);
    batch_worker(&b);
    for (unsigned t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}

/* vdaf.rs:658-676 merge; main.rs:108-118 */
void wo_prio3_aggregate(const uint8_t *out_shares, uint64_t n, uint32_t output_len, const uint8_t *keep, uint8_t *agg) {
    wo_init();
    u128 *acc = wo_alloc(output_len * sizeof(u128));
    for (uint64_t i = 0; i < n; i++) {
        if (keep && !keep[i]) continue;
        for (uint32_t e = 0; e < output_len; e++) {
            u128 v;
            f_decode(out_shares + 16 * ((size_t)i * output_len + e), &v);
            acc[e] = f_add(acc[e], v);
        }
    }
    enc_vec(acc, output_len, agg);
    wo_free(acc);
}
