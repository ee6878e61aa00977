/*
 * wh_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's (ucbsky/whisper @ 2024_08_07) server hot
 * path: Prio3 batched FLPCP verification (`prepare_batched`) + share aggregation, plus
 * the client-side sharding needed to make valid inputs and the un-batched draft-07
 * `prepare_init` that the reference's golden vectors pin.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Nothing under whisper_amd/ links, imports or calls it.
 *
 * Parity status: field / XOF / FFT / query / truncate / prove / aggregate are PINNED by
 * the reference's own vectors (vdaf/test_vec/07/{XofShake128,Prio3SumVec_*,
 * Prio3Histogram_*,Prio3Sum_*}.json, copied as data to tests/golden/).  The
 * batched-only glue (two fused runs, query-rand Fiat-Shamir parts, public-parts hash,
 * RLC order, tag sign) has no known-answer vector anywhere in the reference and the
 * reference cannot be built here (no Rust toolchain): for the final tag value the
 * oracle is "parity unpinned"; it is checked through the reference's own property
 * test instead (sum of both aggregators' tags == 0, vdaf.rs:889-897).
 *
 * All field elements cross this API as 16-byte little-endian canonical residues
 * (field.rs:598-607), seeds as raw 16 bytes, exactly the reference's wire encoding.
 */
#ifndef WH_ORACLE_H
#define WH_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { WO_SUMVEC = 0, WO_HISTOGRAM = 1, WO_AVERAGE = 2 };

/* Circuit parameters (flp/types.rs).  `id` is the Prio3 algorithm id used in the
 * domain-separation tag (vdaf.rs:339-346); 0 selects the circuit's own id
 * (SumVec 2, Histogram 3, Average 0xFFFF0000).  Passing id=1 with WO_AVERAGE gives
 * the draft-07 Prio3Sum instance (identical circuit, types.rs:142-208 vs 241-313). */
typedef struct {
    uint32_t circuit;
    uint32_t bits;      /* SumVec / Average */
    uint32_t len;       /* SumVec vector length / Histogram buckets */
    uint32_t chunk;     /* SumVec / Histogram chunk_length */
    uint32_t num_aggregators;
    uint32_t id;
} wo_params;

typedef struct {
    uint32_t input_len, proof_len, verifier_len, output_len;
    uint32_t joint_rand_len, prove_rand_len, query_rand_len;
    uint32_t calls, m, arity;
    uint32_t leader_report_len, helper_report_len; /* encoded VdafBatchedKey sizes */
} wo_sizes;

/* status codes shared with include/whisper_gfx950.h (same numeric values) */
enum {
    WO_OK = 0,
    WO_ERR_PARAM = -1,
    WO_ERR_SHORT = -2,        /* CodecError: unexpected end of input */
    WO_ERR_MODULUS = -3,      /* FieldError::ModulusOverflow while decoding */
    WO_ERR_AGG_ID = -4,       /* role_try_from: unexpected aggregator id */
    WO_ERR_NUM_QUERIES = -5,  /* assert_eq!(public_proof.len, verifier.len) would panic */
    WO_ERR_QUERY_RAND = -6,   /* FlpError::Query: query randomness is a root of unity */
    WO_ERR_KEY_FORMAT = -9    /* heavy-hitter key chain: bincode would fail, or a length disagrees with the shape */
};

int wo_sizes_of(const wo_params *p, wo_sizes *out);

/* ---- field / xof / fft primitives (exposed for KATs and kernel unit parity) ---- */
void wo_f128_mul(const uint8_t a[16], const uint8_t b[16], uint8_t out[16]);
void wo_f128_add(const uint8_t a[16], const uint8_t b[16], uint8_t out[16]);
void wo_f128_sub(const uint8_t a[16], const uint8_t b[16], uint8_t out[16]);
void wo_f128_inv(const uint8_t a[16], uint8_t out[16]);
/* Montgomery constants as the reference tabulates them (fp.rs:391-421), recomputed */
void wo_f128_constants(uint8_t p[16], uint8_t r2[16], uint8_t g_mont[16], uint64_t *mu);
/* root(l) as a canonical residue */
void wo_f128_root(unsigned l, uint8_t out[16]);

void wo_shake128(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
/* XofShake128::seed_stream(seed,dst,binder).into_field_vec(n)  (xof.rs:127-139,197-218; prng.rs:73-99) */
void wo_xof_expand_f128(const uint8_t seed[16], const uint8_t *dst, size_t dst_len,
                        const uint8_t *binder, size_t binder_len, size_t n, uint8_t *out);
/* XofShake128 init/update(binder)/into_seed */
/* test hook: rejection bound of the sampler (NULL or anything > p restores p) */
void wo_set_test_sampler_bound(const uint8_t bound[16]);
void wo_xof_derive_seed(const uint8_t seed[16], const uint8_t *dst, size_t dst_len,
                        const uint8_t *binder, size_t binder_len, uint8_t out[16]);
/* discrete_fourier_transform (fft.rs:31-85) on n_in<=size inputs, canonical residues */
int wo_fft_f128(const uint8_t *in, size_t n_in, size_t size, uint8_t *out);
/* inverse = forward + inv_finish (fft.rs:102-115) */
int wo_ifft_f128(const uint8_t *in, size_t n_in, size_t size, uint8_t *out);

/* ---- FLP (flp.rs) on canonical residues ---- */
int wo_flp_prove(const wo_params *p, const uint8_t *input, const uint8_t *prove_rand,
                 const uint8_t *joint_rand, uint8_t *proof);
int wo_flp_query(const wo_params *p, const uint8_t *input, const uint8_t *proof,
                 const uint8_t *query_rand, const uint8_t *joint_rand, uint32_t num_shares,
                 uint8_t *verifier);
/* returns 1 accept, 0 reject */
int wo_flp_decide(const wo_params *p, const uint8_t *verifier);
int wo_flp_truncate(const wo_params *p, const uint8_t *input, uint8_t *out);
/* encode_measurement: SumVec -> meas[len] values; Histogram -> meas[0] index; Average -> meas[0] */
int wo_flp_encode(const wo_params *p, const uint64_t *meas, uint8_t *input);

/* ---- draft-07 un-batched Prio3 (pins the shared machinery) ---- */
/* shard_with_random (prio3.rs:1316-1494): rand = random_size() bytes.
 * input_shares: leader share bytes then each helper's 48 bytes, concatenated. */
int wo_prio3_shard(const wo_params *p, const uint64_t *meas, const uint8_t nonce[16],
                   const uint8_t *rand, uint8_t *public_share, uint8_t *leader_share,
                   uint8_t *helper_shares);
/* prepare_init (prio3.rs:2216-2345). prep_share = verifier || joint_rand_part. */
int wo_prio3_prepare_init(const wo_params *p, const uint8_t verify_key[16], uint32_t agg_id,
                          const uint8_t nonce[16], const uint8_t *public_share,
                          const uint8_t *input_share, size_t input_share_len,
                          uint8_t *prep_share, uint8_t *out_share);

/* ---- whisper's batched Prio3 ---- */
/* shard_with_random_new (prio3.rs:537-991) + VdafBatchedKey::encode (vdaf.rs:148-162)
 * for agg_id 0 and 1.  rand = 144 bytes for 2 aggregators. */
int wo_prio3_shard_batched(const wo_params *p, const uint64_t *meas, const uint8_t nonce[16],
                           const uint8_t *rand, const uint8_t client_id_be[16],
                           uint8_t *leader_report, uint8_t *helper_report);
/* prepare_encoded_key (server-batch-prio3/src/main.rs:33-60): decode, prepare_batched
 * (prio3.rs:2466-2879), PRF testing id, negate tag for agg_id != 0. */
int wo_prio3_prepare_encoded_key(const wo_params *p, const uint8_t verify_key[16],
                                 const uint8_t *report, size_t report_len, uint8_t tag[16],
                                 uint8_t testing_id[16], uint8_t *out_share);
/* the same over a batch, `nthreads` host threads (rayon stand-in, main.rs:138-156) */
int wo_prio3_prepare_batch(const wo_params *p, const uint8_t verify_key[16],
                           const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                           unsigned nthreads, uint8_t *tags, uint8_t *testing_ids,
                           int32_t *status, uint8_t *out_shares);
/* Aggregatable::merge over kept reports (vdaf.rs:658-676; main.rs:108-118) */
void wo_prio3_aggregate(const uint8_t *out_shares, uint64_t n, uint32_t output_len,
                        const uint8_t *keep, uint8_t *agg);

/* ---- misc host pieces on the path's edge ---- */
void wo_aes128_encrypt_block(const uint8_t key[16], const uint8_t in[16], uint8_t out[16]);
/* heavy hitters: sum of kept clients' bucket vectors, u32 wrapping (bucket.rs:48-55) */
void wo_hh_bucket_sum_u32(const uint32_t *buckets, const uint8_t *keep, uint64_t n,
                          uint64_t words, uint32_t *acc);


/* ---- heavy-hitter key chains (hhcore/src/{vdpf,dpf,protocol}.rs, common/src/prg.rs); see vdpf.c ---- */
/* bytes of one bincode-encoded KeyChain<u32> whose main key has `depth` levels (num_buckets = 2^depth) */
uint64_t wo_hh_key_chain_len(uint32_t depth);
/* gen_key_chain (protocol.rs:31-76) for IntModN<2^depth>; bucket = 256 data + sign_data + ctr, entries +-1;
 * thread_rng is replaced by SHAKE128(rng_seed).  Writes both servers' encoded key chains. */
int wo_hh_gen_key_chain(uint32_t depth, const uint8_t id_le[16], uint32_t alpha, const uint32_t bucket[258],
                        const uint8_t rng_seed[32], uint8_t *kc0, uint8_t *kc1);
/* eval_and_check (protocol.rs:78-176): dat[2^depth][258] (may be NULL), tag = 64 lower-case hex characters */
int wo_hh_eval_and_check(uint32_t depth, const uint8_t *kc, uint64_t kc_len, uint32_t *dat, uint8_t tag_hex[64]);
/* the per-key loop of server-hh (main.rs:255-265) with the PRF testing id, `nthreads` host threads */
int wo_hh_eval_and_check_batch(uint32_t depth, const uint8_t verify_key[16], const uint8_t *keys, uint64_t stride,
                               uint64_t n, unsigned nthreads, uint32_t *dat, uint8_t *tags_hex, uint8_t *testing_ids,
                               int32_t *status);
void wo_sha256(const uint8_t *in, size_t n, uint8_t out[32]);
/* AES_k(in) ^ in under the fixed key [key_byte; 16] (key_byte 0: FixedKeyPrgStream, 1: HasherStream) */
void wo_mmo_fixed(int key_byte, const uint8_t in[16], uint8_t out[16]);
int wo_aes_backend(void); /* 1 = AES-NI, 0 = portable */

#ifdef __cplusplus
}
#endif
/* ---- Prio3Count over Field64 (count64.c; prio3.rs:81, flp/types.rs:39-105) ---- */
int wo_count_shard(uint32_t num_aggregators, uint64_t meas, const uint8_t nonce[16], const uint8_t *rand, uint8_t *leader_share,
                   uint8_t *helper_shares);
int wo_count_prepare_init(uint32_t num_aggregators, const uint8_t verify_key[16], uint32_t agg_id, const uint8_t nonce[16],
                          const uint8_t *input_share, size_t input_share_len, uint8_t prep_share[32], uint8_t out_share[8]);
int wo_count_prepare_init_encoded(uint32_t num_aggregators, const uint8_t verify_key[16], const uint8_t *key, size_t key_len,
                                  uint8_t prep_share[32], uint8_t out_share[8]);
int wo_count_prepare_finish(uint32_t num_aggregators, const uint8_t *prep_shares);
void wo_count_aggregate(const uint8_t *out_shares, uint64_t n, const uint8_t *keep, uint8_t agg[8]);
void wo_f64_expand(const uint8_t seed[16], uint32_t usage, const uint8_t *binder, size_t binder_len, uint64_t *out, size_t n);

#endif
