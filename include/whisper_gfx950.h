/*
 * whisper_gfx950.h -- C ABI of libwhisper_gfx950.so: the MI355X (gfx950) implementation of
 * ucbsky/whisper's server hot path (batched FLPCP verification + share aggregation).
 *
 * This is the drop-in boundary.  The reference has no FFI seam of its own: the path sits behind
 * Rust traits, so each entry point below names the reference interface it replaces.  All
 * functions return an int32 status (0 = ok, negative = error), never throw, and take plain
 * pointers and sizes.  One context per GPU (or a few); threads may share a context: every entry point
 * takes the context's lock while it works (wh_ctx_lock below for multi-call sequences).
 *
 * Encodings are the reference's own: reports are `VdafBatchedKey::get_encoded()` bytes
 * (batched-flpcp-rs/src/vdaf.rs:148-162), field elements are 16-byte little-endian canonical
 * residues (field.rs:598-607), testing ids are u128 little-endian.
 */
#ifndef WHISPER_GFX950_H
#define WHISPER_GFX950_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wh_ctx wh_ctx;

/* Circuit selector: bin-utils/src/lib.rs:18-32 `AggFunc` / `Prio3Gadgets`. */
enum { WH_SUMVEC = 0, WH_HISTOGRAM = 1, WH_AVERAGE = 2 };

/* Instance parameters: the arguments of Prio3::new_sum_vec_256 / new_histogram_256 /
 * new_average_256 (vdaf/prio3.rs:150-165, 300-313, 360-380) as the servers call them
 * (server-batch-prio3/src/main.rs:249-257).  The servers run two aggregators (Alice and Bob); the batched path, like
 * Prio3::prepare_batched (prio3.rs:2843-2855; its own test runs three, :3077-3085), and the draft-07 prepare_init path
 * take 2..254 (check_num_aggregators, vdaf/prio3.rs:2945-2957; a single aggregator, legal there, is refused:
 * WH_ERR_PARAM).  Aggregator 0 is the Leader (explicit shares), 1.. are Helpers (seeds).  The GPU sharder
 * (wh_client_shard_many_gpu) is built for two. */
typedef struct {
    uint32_t circuit;         /* WH_SUMVEC | WH_HISTOGRAM | WH_AVERAGE */
    uint32_t bits;            /* SumVec: 16 in the servers; Average: AVG_BITS = 64 */
    uint32_t len;             /* SumVec vec_size / Histogram length */
    uint32_t chunk;           /* chunk_size */
    uint32_t num_aggregators; /* 2 in the servers */
    uint32_t alg_id;          /* 0 = the circuit's own algorithm id (types.rs:577,383,242); otherwise the
                               * `Type::ID` to bind into every XOF, e.g. 1 to run Average's circuit as
                               * draft-07 Prio3Sum (types.rs:142-208) */
} wh_prio3_params;

/* Derived sizes (flp/types.rs:650-676, 444-470, 286-312; report sizes from vdaf.rs:148-162). */
typedef struct {
    uint32_t input_len, proof_len, verifier_len, output_len;
    uint32_t joint_rand_len, calls, m, arity;
    uint32_t leader_report_len, helper_report_len;
} wh_prio3_sizes;

/* Per-report status = the reference's failure classes.  Every non-zero value is an
 * unwrap()/assert abort in server-batch-prio3 (main.rs:46-49); here the report is skipped. */
enum {
    WH_OK = 0,
    WH_ERR_PARAM = -1,
    WH_ERR_SHORT = -2,        /* CodecError: input ends early (codec.rs) */
    WH_ERR_MODULUS = -3,      /* FieldError::ModulusOverflow (field.rs:417-419) */
    WH_ERR_AGG_ID = -4,       /* "unexpected aggregator id" (prio3.rs:1496-1501) */
    WH_ERR_NUM_QUERIES = -5,  /* public proof length != verifier length (prio3.rs:2712) */
    WH_ERR_QUERY_RAND = -6,   /* FlpError::Query: r is an m-th root of unity (flp.rs:413-419) */
    WH_ERR_DECIDE = -7,       /* "proof verifier check failed" (prio3.rs:2372-2378) */
    WH_ERR_JOINT_RAND = -8,   /* "joint randomness mismatch" (prio3.rs:2414-2420) */
    WH_ERR_KEY_FORMAT = -9,   /* heavy-hitter key chain: bincode would fail on it, or a length disagrees with the configured
                               * shape (index panic in eval_all, assert at hhcore/src/protocol.rs:95) */
    /* library-level */
    WH_ERR_HIP = -100,
    WH_ERR_NO_DEVICE = -101,
    WH_ERR_STATE = -102,
    WH_ERR_UNSUPPORTED = -103,
    WH_ERR_COMM = -104        /* RCCL reported an error (wh_last_error has its text) */
};

/* ---- context ---- */
/* replaces: Prio3::new_* + Prf::new (common/src/prg.rs:644-648).  Binds `device_id`. */
int32_t wh_ctx_create(int32_t device_id, const wh_prio3_params *params, wh_ctx **out);
void wh_ctx_destroy(wh_ctx *ctx);
int32_t wh_prio3_sizes_of(const wh_prio3_params *params, wh_prio3_sizes *out);
/* the context's last error text, copied into a buffer of the calling thread (valid until that thread's next wh_last_error).
 * With several threads on one context it may be another thread's failure: bracket the failing call and this query with
 * wh_ctx_lock / wh_ctx_unlock to get the text of your own. */
const char *wh_last_error(const wh_ctx *ctx);
/* the HIP stream (hipStream_t) the context launches on, for callers that time or order work */
void *wh_ctx_stream(wh_ctx *ctx);
/* replaces: nothing in the reference, where `Prio3: Clone + Send + Sync` is called from 32 tokio tasks and a rayon pool at
 * once (server-batch-prio3/src/main.rs:282-335, 138-156).  Every entry point takes the context's (recursive) lock while it
 * works, so threads may share a context; a task that needs a sequence undisturbed -- prepare, the host's group test, then
 * aggregate of THAT batch -- brackets it with wh_ctx_lock / wh_ctx_unlock and the other tasks queue behind it.  A server gives
 * each GPU a few contexts (each owns a workspace) and lets its tasks pick one: tests/cpp/funnel_gpu.cpp runs 32 threads
 * over two contexts. */
int32_t wh_ctx_lock(wh_ctx *ctx);
int32_t wh_ctx_unlock(wh_ctx *ctx);

/* ---- batched verification ---- */
/* replaces: the per-report loop of run_vdaf_prepare{,_rayon} over prepare_encoded_key
 * (server-batch-prio3/src/main.rs:33-60, 75-91, 138-156) =
 *   VdafBatchedKey::decode_with_param + Prio3::prepare_batched (vdaf/prio3.rs:2466-2879)
 *   + Prf::compute_prf + tag negation for agg_id != 0.
 * blob/offsets: n encoded reports back to back, offsets[n+1] (bytes).  All pointers are DEVICE
 * pointers; nothing is copied to the host and the call returns after enqueueing on the context
 * stream (wh_ctx_synchronize to wait).  Precondition: d_blob holds at least max(offsets) bytes.  A
 * pair offsets[i+1] < offsets[i] is an empty report (status WH_ERR_SHORT), never a read outside the blob.
 * The status array may be freed or reused once the call has been waited for or once later work has been
 * enqueued on the context stream: the context keeps its own copy for wh_prio3_aggregate*.
 *   tags        n*16   proof tags (already negated for Helper-role reports)
 *   testing_ids n*16   u128 LE
 *   status      n      int32 per report
 * A report that does not decode (status != 0 from the parse: short, bad aggregator id, wrong num_queries) gets a zero tag, a
 * zero testing id and a zero out share, whatever the buffers held before.
 * The out shares (truncated measurement shares) stay in the context for wh_prio3_aggregate*.
 * Limit (all verifying entry points: this one, prepare_init, wh_flp_query_batch): the query kernel keeps 48 B per gadget call
 * in LDS, at most 160 KiB with the truncate staging of its wavefronts -- 3 228 calls at chunk_length <= 64, 3 047 up to 128,
 * 2 684 above; a circuit with more (a chunk length far below sqrt(input_len)) returns WH_ERR_UNSUPPORTED. */
int32_t wh_prio3_prepare_batch_dev(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *d_blob,
                                   const uint64_t *d_offsets, uint64_t n, uint8_t *d_tags,
                                   uint8_t *d_testing_ids, int32_t *d_status);
/* replaces: run_vdaf_prepare up to the group test (server-batch-prio3/src/main.rs:75-94) = the map over prepare_encoded_key AND
 * `tags_and_shares.sort_by(testing_id)`: wh_prio3_prepare_batch_dev followed by wh_sort_by_testing_id_dev (below) in one call,
 * with the same outputs as the two calls -- d_order[n]: position k of the sorted batch holds report d_order[k] (stable);
 * d_sorted_ids[n][16], d_sorted_tags[n][16] (either may be NULL) in that order, ready for general_binary_split_test.  The
 * testing ids exist once the batch is parsed, so the sort of the permutation runs on a side stream beside the sponge kernels
 * and costs the step nothing; only the gather of the sorted tags waits for the tags.  n <= 2^28. */
int32_t wh_prio3_prepare_sorted_batch_dev(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *d_blob,
                                          const uint64_t *d_offsets, uint64_t n, uint8_t *d_tags, uint8_t *d_testing_ids,
                                          int32_t *d_status, uint32_t *d_order, uint8_t *d_sorted_ids, uint8_t *d_sorted_tags);
/* Same with HOST buffers (what a Rust caller holding Vec<Vec<u8>> would use); copies in and out
 * and synchronises.  The batch crosses PCIe in chunks (2 GiB; WH_HOST_CHUNK_MB) through two device buffers, the copy of
 * chunk k+1 under the verification of chunk k.  Pageable input is staged through pinned context memory by several host
 * threads (WH_HOST_COPY_THREADS, at most 8); input in pinned memory (wh_host_alloc, hipHostRegister) goes straight to the
 * DMA engine. */
int32_t wh_prio3_prepare_batch(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *blob,
                               const uint64_t *offsets, uint64_t n, uint8_t *tags, uint8_t *testing_ids,
                               int32_t *status);
/* No counterpart in the reference (its reports arrive in Vec<u8>s from tokio sockets, bridge/src/client_server.rs): pinned host
 * memory for a receive path that wants the DMA engine to read the reports where the sockets put them. */
int32_t wh_host_alloc(uint64_t bytes, void **out);
int32_t wh_host_free(void *p);
/* replaces: Prio3::prepare_batched's second return value (OutputShare).  Copies the n*output_len
 * out-share elements of the last batch to the host.  WH_ERR_STATE unless the last call that used the
 * context's workspace was a prepare call or wh_flp_query_batch (wh_client_shard_many_gpu leaves nothing). */
int32_t wh_prio3_out_shares(wh_ctx *ctx, uint8_t *out_shares);

/* ---- un-batched Prio3 (draft-07), the paper's baseline server ---- */
/* replaces: the first map of server-base-prio3's run_vdaf_prepare (server-base-prio3/src/main.rs:41-61) =
 *   VdafKey::decode_with_param (vdaf.rs:219-233) + Prio3::prepare_init (vdaf/prio3.rs:2216-2345)
 *   + Prio3PrepareShare::get_encoded (prio3.rs:1953-1970).
 * blob/offsets: n encoded VdafKeys: agg_id u8 | num_aggregators joint-rand parts | input share | nonce[16].
 *   prep_shares  n * (verifier_len + 1) * 16   verifier share | own joint-rand part, per report
 *   status       n
 * The prepare state (out share = truncated measurement share, joint-rand seed) stays in the context. */
int32_t wh_prio3_prepare_init_batch_dev(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *d_blob,
                                        const uint64_t *d_offsets, uint64_t n, uint8_t *d_prep_shares,
                                        int32_t *d_status);
int32_t wh_prio3_prepare_init_batch(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *blob,
                                    const uint64_t *offsets, uint64_t n, uint8_t *prep_shares, int32_t *status);
/* replaces: the second map of run_vdaf_prepare (main.rs:68-100) =
 *   prepare_shares_to_prepare_message (prio3.rs:2347-2394: sum of the verifier shares, Type::decide,
 *   joint-rand seed from the parts) + prepare_next (:2396-2442: seed check; the out share is the state's).
 * prep_shares[a] = aggregator a's n prepare shares in agg_id order (one of them is this context's own output
 * of the last prepare_init batch).  prep_msgs: n * 16 (the joint-rand seed).  status: per report, WH_ERR_DECIDE /
 * WH_ERR_JOINT_RAND for rejected reports; reports that already failed keep their status.  Afterwards
 * wh_prio3_out_shares / wh_prio3_aggregate* see only the accepted reports. */
int32_t wh_prio3_prepare_finish_batch(wh_ctx *ctx, const uint8_t *const *prep_shares, uint64_t n, uint8_t *prep_msgs,
                                      int32_t *status);

/* ---- Prio3Count = Prio3<Count<Field64>, XofShake128, 16> (batched-flpcp-rs/src/vdaf/prio3.rs:81; flp/types.rs:39-105; Field64:
 * fp.rs:359-389): the reference's Field64 instantiation, un-batched draft-07 preparation for a whole batch.  Any context will do
 * (wh_hh_ctx_create is the cheapest); the state lives beside the context's other batches.  Elements are 8 bytes, little endian.
 * replaces: both maps of server-base-prio3's run_vdaf_prepare (src/main.rs:41-61, 68-100) and its merge loop, for a server
 * built with Prio3::new_count(num_aggregators).
 *   blob / offsets: n encoded VdafKey<Prio3Count> (vdaf.rs:209-233): agg_id u8 | input share | nonce[16]; input share:
 *     aggregator 0: meas[1] | proof[5] (48 bytes); others: measurement-share seed | proof-share seed (32 bytes).  There is no
 *     public share and no blind (joint_rand_len = 0).
 *   prep_shares: n * 32 bytes, the verifier share (Prio3PrepareShare without a joint-randomness part, prio3.rs:1953-1970).
 *   status: WH_ERR_SHORT / WH_ERR_AGG_ID / WH_ERR_MODULUS / WH_ERR_QUERY_RAND where the reference's unwrap() would panic.
 * finish: prep_shares[a] = aggregator a's n prepare shares; sums them, Type::decide (flp.rs:483-512); the prepare message is
 *   empty for Count, so there is no output besides status (WH_ERR_DECIDE for rejected reports, WH_ERR_MODULUS for a
 *   non-canonical share element).  aggregate: the mod-p sum of the out shares of accepted, kept reports (8 bytes). */
int32_t wh_prio3_count_prepare_init_batch(wh_ctx *ctx, uint32_t num_aggregators, const uint8_t verify_key[16], const uint8_t *blob,
                                          const uint64_t *offsets, uint64_t n, uint8_t *prep_shares, int32_t *status);
int32_t wh_prio3_count_prepare_finish_batch(wh_ctx *ctx, uint32_t num_aggregators, const uint8_t *const *prep_shares, uint64_t n,
                                            int32_t *status);
int32_t wh_prio3_count_out_shares(wh_ctx *ctx, uint8_t *out_shares);            /* n * 8 bytes */
int32_t wh_prio3_count_aggregate(wh_ctx *ctx, const uint8_t *keep, uint8_t agg[8]);
/* test hooks: the Field64 sampler (prng.rs:73-99 on 8-byte chunks) with any bound; out: n rows of 5 words */
int32_t wh_test_f64_xof_expand(wh_ctx *ctx, const uint8_t *seeds, uint32_t n, uint32_t usage, const uint8_t *binder, uint32_t binder_len,
                               uint32_t n_elems, uint64_t bound, uint64_t *out);
int32_t wh_prio3_count_set_test_sampler_bound(wh_ctx *ctx, uint64_t bound);     /* 0 = p */

/* ---- the flp::Type surface (batched-flpcp-rs/src/flp.rs:186-196) ---- */
/* replaces: Type::query (flp.rs:359-479) for n independent (input, proof, query_rand, joint_rand) tuples, all of this
 * context's circuit; HOST buffers of n rows each: input[input_len], proof[proof_len], query_rand[1],
 * joint_rand[joint_rand_len] elements.  verifier: n * verifier_len elements.  status per tuple: WH_ERR_QUERY_RAND
 * when query_rand is an m-th root of unity (FlpError::Query), WH_ERR_MODULUS for non-canonical elements.
 * Type::truncate (types.rs:641-648, 439-442, 280-284) of every input is left in the context: wh_prio3_out_shares. */
int32_t wh_flp_query_batch(wh_ctx *ctx, const uint8_t *input, const uint8_t *proof, const uint8_t *query_rand,
                           const uint8_t *joint_rand, uint32_t num_shares, uint64_t n, uint8_t *verifier, int32_t *status);
/* replaces: Type::decide (flp.rs:483-512) for n verifiers: decision[i] = 1 accept, 0 reject, WH_ERR_MODULUS. */
int32_t wh_flp_decide_batch(wh_ctx *ctx, const uint8_t *verifier, uint64_t n, int32_t *decision);
/* replaces: Type::prove (flp.rs:231-347) with num_shares = 1 for n independent tuples on the GPU (the client-side half of
 * the proof system; the meta-client of the reference proves on the CPU, client-batch-prio3/src/main.rs:67-153).  HOST
 * buffers of n rows: input[input_len], prove_rand[arity], joint_rand[joint_rand_len] elements; proof: n * proof_len
 * elements.  status per tuple: WH_ERR_MODULUS for a non-canonical element (the proof row is then zero).  Circuits with more
 * than 511 gadget calls: WH_ERR_UNSUPPORTED. */
int32_t wh_flp_prove_batch(wh_ctx *ctx, const uint8_t *input, const uint8_t *prove_rand, const uint8_t *joint_rand, uint64_t n,
                           uint8_t *proof, int32_t *status);
/* replaces (CPU, client side): Type::encode_measurement (types.rs:582-606, 388-399, 246-253) and Type::prove
 * (flp.rs:231-347) with num_shares = 1; prove_rand: arity elements, joint_rand: joint_rand_len elements. */
int32_t wh_flp_encode(const wh_prio3_params *params, const uint64_t *measurement, uint8_t *input);
int32_t wh_flp_prove(const wh_prio3_params *params, const uint8_t *input, const uint8_t *prove_rand,
                     const uint8_t *joint_rand, uint8_t *proof);

/* replaces: discrete_fourier_transform + discrete_fourier_transform_inv_finish (batched-flpcp-rs/src/fft.rs:31-85, 102-115), the
 * transform under poly_fft / poly_interpret_eval (polynomial.rs:187-231) that Type::prove and Type::query call per wire, for n
 * independent vectors: in[n][n_in] Field128 residues, zero-padded to `size` (a power of two, n_in <= size), out[n][size] with
 * out[i] = sum_j in[j] w^(ij), w = the reference's primitive size-th root (fp.rs:391-421); inverse != 0: the same transform
 * followed by inv_finish (scaled by 1/size, index i <-> size - i), i.e. interpolation.  Host buffers, synchronous.  An element
 * >= p: WH_ERR_MODULUS; size > 2^20 (the reference's MAX_ROOTS): WH_ERR_PARAM; size > 4096: WH_ERR_UNSUPPORTED (one
 * workgroup holds a vector and its twiddles in LDS).  The product path does not call it (the prover fuses its transforms,
 * the verifier evaluates in barycentric form): it is the stand-alone surface for parity against fft.rs. */
int32_t wh_ntt_f128_batch(wh_ctx *ctx, const uint8_t *in, uint64_t n_in, uint64_t size, int32_t inverse, uint64_t n, uint8_t *out);

/* ---- aggregation ---- */
/* replaces: the merge loop over reports whose testing id is not in the bad set
 * (server-batch-prio3/src/main.rs:108-118, 173-183; Aggregatable::merge vdaf.rs:658-676;
 * Prio3::aggregate prio3.rs:2445-2456).  keep[n]: 1 = aggregate, 0 = skip (NULL = keep all).
 * Reports with non-zero status are never aggregated.  WH_ERR_STATE unless the context's last workspace
 * user was wh_prio3_prepare_batch[_dev] or wh_prio3_prepare_init_batch[_dev] (+ finish): wh_flp_query_batch and
 * wh_client_shard_many_gpu overwrite the out shares.  After wh_prio3_prepare_init_batch[_dev] also WH_ERR_STATE until
 * wh_prio3_prepare_finish_batch has decided the batch (prepare_next, prio3.rs:2396-2442): before it the statuses are
 * prepare_init's and no report has been accepted.  (wh_prio3_out_shares stays available there: it is the prepare state's
 * truncated measurement share, Prio3PrepareState, which the test vectors list per report.) */
int32_t wh_prio3_aggregate_dev(wh_ctx *ctx, const uint8_t *d_keep, uint8_t *d_agg);
int32_t wh_prio3_aggregate(wh_ctx *ctx, const uint8_t *keep, uint8_t *agg);
/* replaces: AggregateShare::merge across partial aggregates (main.rs:327-335); with one process
 * per GPU the rows are the RCCL all-gathered per-GPU partials.  d_rows[n_rows][len] -> d_out[len]. */
int32_t wh_f128_sum_rows_dev(wh_ctx *ctx, const uint8_t *d_rows, uint64_t n_rows, uint32_t len,
                             const uint8_t *d_keep, uint8_t *d_out);

/* ---- several GPUs in one server: one context per GPU, combined over RCCL / xGMI ---- */
/* replaces: the merge of the per-task partial aggregates (server-batch-prio3/src/main.rs:327-335; AggregateShare::merge,
 * vdaf.rs:658-676) once the batch is sharded over the GPUs of a node -- reports shard by index with no data-path
 * collective, each GPU verifies and aggregates its shard (wh_prio3_aggregate_dev), then the output_len*16-byte partials
 * are all-gathered as raw limbs (RCCL has no mod-p reduction) and summed mod p on every GPU.
 * wh_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks over any channel the
 * server already has.  wh_comm_init binds the context to rank `rank` of `world` (ncclCommInitRank on the context's GPU;
 * collective: all ranks call it).  id == NULL with world == 1 is a lone GPU without RCCL; a non-NULL id with world == 1 runs
 * the same collectives through RCCL.  RCCL is loaded at the first of these calls (WH_ERR_UNSUPPORTED if it is absent). */
#define WH_COMM_ID_BYTES 128
int32_t wh_comm_unique_id(uint8_t id[WH_COMM_ID_BYTES]);
int32_t wh_comm_init(wh_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world);
int32_t wh_comm_destroy(wh_ctx *ctx);
int32_t wh_comm_rank(wh_ctx *ctx, int32_t *rank, int32_t *world);
/* why RCCL could not be bound (the loader's message), "" if it could or if no call has tried yet; the library named by
 * WH_RCCL_LIBRARY (read at the first attempt) is tried instead of the default names when it is set */
const char *wh_comm_why(void);
/* d_partial[output_len] (this GPU's aggregate share, device) -> d_agg[output_len] = sum over ranks mod p, on every rank;
 * asynchronous on the context stream.  The _f128_ form takes the row length (any Field128 row). */
int32_t wh_prio3_allgather_reduce_dev(wh_ctx *ctx, const uint8_t *d_partial, uint8_t *d_agg);
int32_t wh_f128_allgather_reduce_dev(wh_ctx *ctx, const uint8_t *d_partial, uint32_t len, uint8_t *d_out);
/* wh_prio3_aggregate (host keep mask, host result) of the last batch on this GPU followed by the combine: what a Rust
 * task holding `keep` calls instead of the loop at main.rs:108-118 + the merge at :327-335. */
int32_t wh_prio3_allgather_reduce(wh_ctx *ctx, const uint8_t *keep, uint8_t *agg);
/* replaces: the element-wise sum of sketches held by several GPUs of one server (the same Bucket::add / CountSketch::insert
 * arithmetic as server-hh/src/main.rs:72-104, 192-197, 374-394): in-place all-reduce of d_sketch[words] u32 counters with
 * wrapping addition (ncclSum on ncclUint32 is exact in Z/2^32). */
int32_t wh_hh_allreduce_u32_dev(wh_ctx *ctx, uint32_t *d_sketch, uint64_t words);

/* ---- heavy hitters ---- */
/* replaces: aggregate_hhs' rayon reduce of Vec<Bucket<u32>> (server-hh/src/main.rs:72-104;
 * Bucket::add hhcore/src/bucket.rs:48-55; u32 wrapping add common/src/group.rs:88-91).
 * buckets[n][words] with words = num_buckets * 258 (256 data + sign + ctr); acc[words]. */
int32_t wh_hh_bucket_sum_u32_dev(wh_ctx *ctx, const uint32_t *d_buckets, const uint8_t *d_keep, uint64_t n,
                                 uint64_t words, uint32_t *d_acc);
/* replaces: the `streaming` loop of server-hh (server-hh/src/main.rs:113-216), which feeds `batch_size` clients at a
 * time and adds each slice's bucket vector into the running sketch (CountSketch::insert, countsketch.rs:22-28):
 * same sum, but d_acc is added to instead of being overwritten. */
int32_t wh_hh_bucket_accumulate_u32_dev(wh_ctx *ctx, const uint32_t *d_buckets, const uint8_t *d_keep, uint64_t n,
                                        uint64_t words, uint32_t *d_acc);

/* Extension with no counterpart in the reference: the same bucket sums over Field128 counters (16-byte little-endian
 * residues; non-canonical inputs are reduced), for a deployment that swaps AggRing (hhcore/src/lib.rs:13 fixes it to u32;
 * eval_and_check only converts to 16/32/64-bit rings, protocol.rs:157-173) for the Prio3 field -- BASELINE.json's streaming
 * configuration names Field128.  buckets[n][words][16], keep[n] (NULL = all), acc[words][16]; _accumulate_ adds into acc
 * (CountSketch::insert over batch_size slices).  Only the sum is built: no Field128 key-chain expansion exists. */
int32_t wh_hh_bucket_sum_f128_dev(wh_ctx *ctx, const uint8_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words,
                                  uint8_t *d_acc);
int32_t wh_hh_bucket_accumulate_f128_dev(wh_ctx *ctx, const uint8_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words,
                                         uint8_t *d_acc);

/* replaces: the expansion loop of server-hh (server-hh/src/main.rs:163-174, 255-265) =
 *   Prf::compute_prf(key.id.to_le_bytes()) + BCSketch::eval_and_check::<IntModN<NumBuckets>> (hhcore/src/protocol.rs:78-176)
 *   = VDPFKey::eval_all of the main key and of the 257 support keys (hhcore/src/vdpf.rs:123-214; PRG and MMO hashes
 *   common/src/prg.rs:133-241, 280-475, 599-631), consistency folding and the SHA-256 proof string.
 * keys: n key chains, `stride` bytes apart, each the bincode 1.3.3 bytes of a KeyChain<u32> exactly as the servers receive
 * them (serialize/src/lib.rs:72-80): wh_hh_key_chain_len(depth) bytes, num_buckets = 2^depth, depth 1..11.
 *   dat         n * 2^depth * 258 u32   Vec<Bucket<u32>> per key chain (256 data, sign_data, ctr); NULL keeps it in the context
 *   tags        n * 64                  the proof `String`: 64 lower-case hex characters
 *   testing_ids n * 16                  u128 LE
 *   status      n                       WH_ERR_KEY_FORMAT where the reference would abort; such a key chain gets a zero tag and
 *                                       an all-zero bucket vector
 * The caller sorts by testing id and runs general_binary_split_test over the tags, as server-hh does. */
uint64_t wh_hh_key_chain_len(uint32_t depth);
/* a context for a server that runs only the heavy-hitter path (no Prio3 instance to describe) */
int32_t wh_hh_ctx_create(int32_t device_id, wh_ctx **out);
int32_t wh_hh_eval_and_check_batch_dev(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint64_t n,
                                       uint32_t depth, uint32_t *d_dat, uint8_t *d_tags, uint8_t *d_testing_ids, int32_t *d_status);
/* replaces: the same expansion TOGETHER with aggregate_hhs' sum (server-hh/src/main.rs:72-104; Bucket::add, hhcore/src/bucket.rs:
 * 48-55): the caller holds no Vec<Bucket<u32>> (528 KB per key chain at 512 buckets); a sub-batch of bucket vectors lives in
 * context scratch and is summed into the sketches on a side stream under the proof-string kernels of the same sub-batch.
 * Key chain i goes into sketch i / group_size (one sketch per repetition when the batch is
 * repetition-major with group_size clients each): d_sketch[ceil(n / group_size)][2^depth * 258] u32, overwritten unless
 * `accumulate` (the streaming server's CountSketch::insert, main.rs:192-197).  ALL well-formed key chains are summed, the
 * group test has not run yet; afterwards wh_hh_sketch_subtract_dev takes the rejected ones out again ("sum all, subtract the
 * bad": exact in Z/2^32).  Malformed key chains (status != 0) contribute nothing in either call.
 * bad_index: n_bad indices into d_keys (HOST memory), the key chains whose testing ids the group test returned. */
int32_t wh_hh_eval_and_sketch_batch_dev(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint64_t n,
                                        uint32_t depth, uint64_t group_size, uint32_t *d_sketch, int32_t accumulate, uint8_t *d_tags,
                                        uint8_t *d_testing_ids, int32_t *d_status);
int32_t wh_hh_sketch_subtract_dev(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint32_t depth,
                                  const uint64_t *bad_index, uint64_t n_bad, uint64_t group_size, uint32_t *d_sketch);
/* Same with HOST buffers; the bucket vectors stay in the context for wh_hh_aggregate* / wh_hh_buckets. */
int32_t wh_hh_eval_and_check_batch(wh_ctx *ctx, const uint8_t verify_key[16], const uint8_t *keys, uint64_t stride, uint64_t n,
                                   uint32_t depth, uint8_t *tags, uint8_t *testing_ids, int32_t *status);
/* eval_and_check's first return value for the last batch, copied to the host (n * 2^depth * 258 u32) */
int32_t wh_hh_buckets(wh_ctx *ctx, uint32_t *dat);
/* replaces: aggregate_hhs' reduce over the last batch's bucket vectors (server-hh/src/main.rs:72-104): keep[n] 1 = add,
 * 0 = skip (NULL = all); acc = 2^depth * 258 u32.  The _dev form adds into d_acc when `accumulate` is non-zero
 * (CountSketch::insert in the streaming server, main.rs:192-197). */
int32_t wh_hh_aggregate(wh_ctx *ctx, const uint8_t *keep, uint32_t *acc);
int32_t wh_hh_aggregate_dev(wh_ctx *ctx, const uint8_t *d_keep, uint32_t *d_acc, int32_t accumulate);

/* ---- ordering for the group test ---- */
/* replaces: `tags_and_shares.sort_by(|a, b| a.0.testing_id.cmp(&b.0.testing_id))` and its par_sort_by forms
 * (server-batch-prio3/src/main.rs:93, 158; server-hh/src/main.rs:175, 266): the STABLE sort of a batch by its u128 testing
 * ids, on the ids the prepare / eval_and_check call left in HBM.  d_ids[n][16] little endian.  Writes d_order[n]: position k
 * of the sorted batch holds report d_order[k] (equal ids keep their input order).  If non-NULL, d_sorted_ids[n][16] and
 * d_sorted_tags[n][tag_bytes] receive the ids and the tags (16-byte Prio3 tags, 64-byte heavy-hitter proof strings; tag_bytes
 * a multiple of 16) in that order, ready for the host's general_binary_split_test.  n <= 2^28.  Asynchronous on the context
 * stream. */
int32_t wh_sort_by_testing_id_dev(wh_ctx *ctx, const uint8_t *d_ids, uint64_t n, uint32_t *d_order, const uint8_t *d_tags,
                                  uint32_t tag_bytes, uint8_t *d_sorted_ids, uint8_t *d_sorted_tags);
/* No counterpart in the reference: wh_sort_by_testing_id_dev for callers that cannot allocate -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same results, asynchronous on the context
 * stream, but the sort's scratch (second permutation buffer + count table, 5.0 MiB) is a file-scope __device__ array that
 * every GPU the library runs on holds until the process ends, so the call never allocates or frees anything.
 * n <= WH_SORT_NOALLOC_MAX_N (2^20), larger batches: WH_ERR_PARAM.  All calls on one GPU share that array; the entry point
 * orders them itself: each waits, on its own stream, for a per-device event the call before it recorded (created with the
 * scratch address at the first wh_ctx_create on the device), so two contexts on one GPU take turns.  Work that touches the
 * array by other means is not covered.  WH_ERR_STATE if the array's address could not be resolved on this device. */
#define WH_SORT_NOALLOC_MAX_N (1u << 20)
int32_t wh_sort_by_testing_id_noalloc_dev(wh_ctx *ctx, const uint8_t *d_ids, uint64_t n, uint32_t *d_order,
                                          const uint8_t *d_tags, uint32_t tag_bytes, uint8_t *d_sorted_ids,
                                          uint8_t *d_sorted_tags);
/* the same permutation for ids in host memory (ids[n][16] in, order[n] out); synchronous */
int32_t wh_sort_by_testing_id(wh_ctx *ctx, const uint8_t *ids, uint64_t n, uint32_t *order);
/* replaces: the `if !bad_set.contains(&tag.testing_id)` filter of the merge loops (server-batch-prio3/src/main.rs:108-118,
 * 173-183; server-hh/src/main.rs:79-103): d_keep[i] = 0 iff the testing id of report i (d_ids[n][16], device, 16-byte aligned)
 * is one of the n_bad ids the group test returned (bad_ids[n_bad][16], little endian, HOST memory, any order), else 1.
 * The mask then goes to wh_prio3_aggregate_dev / wh_hh_aggregate_dev.  Asynchronous: the ids are copied into pinned
 * staging owned by the context before the call returns, so bad_ids may be reused at once. */
int32_t wh_keep_mask_dev(wh_ctx *ctx, const uint8_t *d_ids, uint64_t n, const uint8_t *bad_ids, uint64_t n_bad, uint8_t *d_keep);
/* replaces: ClientProofTag::batch_proof / batch_proof_shake128 / batch_proof_sha256 (common/src/grouptest.rs:43-99) for k slices
 * of ONE sorted tag list at a time -- the two calls per mismatching range of a group-test round (grouptest.rs:306-318).
 * d_ids[n][16] (little-endian u128, ascending: the output of wh_sort_by_testing_id_dev) and d_tags[n][tag_bytes] (tag_bytes a
 * multiple of 8: 16 for the Prio3 tags, 64 for the heavy-hitter proof strings) are device memory.  Slice i is the items
 * [starts[i], ends[i]) (host arrays, ends[i] <= n); its proof covers the leading items with id <= uppers[i] (host, k x 16 bytes,
 * little endian): counts[i] = their number (batch_proof's split index), proofs[i] = SHAKE128(verify_key || tags)[0..16] when
 * proof_size = 16, SHA-256(verify_key || tags) when proof_size = 32 (host, k x proof_size bytes).  One GPU lane per slice: worth
 * calling for tens of slices and more; one long slice (the complete proof of grouptest.rs:143) is faster on a host core.
 * k <= 2^24 slices per call.  Synchronous. */
int32_t wh_grouptest_batch_proofs_dev(wh_ctx *ctx, const uint8_t *d_ids, const uint8_t *d_tags, uint64_t n, uint32_t tag_bytes,
                                      const uint8_t *verify_key, const uint64_t *starts, const uint64_t *ends, const uint8_t *uppers,
                                      uint32_t k, uint32_t proof_size, uint8_t *proofs, uint64_t *counts);
/* replaces: the loop over the d initial ranges (common/src/grouptest.rs:182-208), whose slice i starts where slice i - 1
 * stopped: proofs[i], counts[i] as above for the items behind every id <= max(uppers[0..i-1]) with id <= uppers[i];
 * firsts[i] = index of the slice's first item (the reference's `slice_so_far`).  All d slices are found (binary search) and
 * hashed side by side.  Synchronous. */
int32_t wh_grouptest_split_proofs_dev(wh_ctx *ctx, const uint8_t *d_ids, const uint8_t *d_tags, uint64_t n, uint32_t tag_bytes,
                                      const uint8_t *verify_key, const uint8_t *uppers, uint32_t d, uint32_t proof_size, uint8_t *proofs,
                                      uint64_t *firsts, uint64_t *counts);

int32_t wh_ctx_synchronize(wh_ctx *ctx);

/* ---- tuning and test knobs ---- */
/* No counterpart in the reference.  Every knob lives in the context; the environment variable named beside it is read ONCE,
 * by wh_ctx_create, as the knob's initial value, and never again: no entry point's behaviour depends on the environment of
 * the moment it is called.  Results never depend on a knob, only the launch shapes do. */
enum {
    WH_OPT_MAX_SUBBATCH = 0,        /* WH_MAX_SUBBATCH: reports per verification sub-batch (>= 64; 0 = the library's choice) */
    WH_OPT_HOST_CHUNK_MB = 1,       /* WH_HOST_CHUNK_MB: MiB per H2D chunk of wh_prio3_prepare_batch (default 2048) */
    WH_OPT_HOST_COPY_THREADS = 2,   /* WH_HOST_COPY_THREADS: host threads staging pageable input (0 = all cores, at most 8) */
    WH_OPT_HH_SUBBATCH = 3,         /* WH_HH_SUBBATCH: key chains per heavy-hitter sub-batch (0 = the library's choice) */
    WH_OPT_HH_SUM_LDS = 4,          /* WH_HH_SUM_LDS: bytes of LDS a side-stream sketch-sum workgroup reserves (default 120 KiB) */
    WH_OPT_SIDE_STREAM = 5,         /* 0 if WH_NO_SIDE_STREAM is set: proof-share expansion on the main stream (long-share path) */
    WH_OPT_HASH_B_SIDE = 6,         /* 0 if WH_HASH_B_MAIN is set: public-parts hash / RLC / testing ids in the main hash launch */
    WH_OPT_THIRD_STREAM = 7,        /* 0 if WH_NO_THIRD_STREAM is set */
    WH_OPT_BASE_PAIRED = 8,         /* 0 if WH_BASE_NO_MEAS is set: prepare_init expands, then absorbs, in two passes */
    WH_OPT_CHAIN_MAX_INPUT_LEN = 9, /* WH_CHAIN_MAX_INPUT_LEN: circuits with input_len up to this run all their sponges in one
                                     * launch (chain.hip) instead of the k_meas pipeline */
    WH_OPT_CHAIN_PREFETCH = 10,     /* WH_CHAIN_PREFETCH: 1 (default) = that launch keeps the next rate block's loads in flight (3 waves/SIMD), 0 = 4 waves/SIMD without */
    WH_OPT_QUERY_TINY = 11,         /* WH_QUERY_TINY: 0 = circuits of at most two gadget calls of at most 64 slots take the general
                                     * query kernel instead of the one-wavefront-per-report one */
    WH_NUM_OPTS = 12
};
#define WH_CHAIN_MAX_INPUT_LEN_DEFAULT 2048u
int32_t wh_ctx_set_option(wh_ctx *ctx, int32_t option, uint64_t value);
int32_t wh_ctx_get_option(wh_ctx *ctx, int32_t option, uint64_t *value);

/* ---- measurement ---- */
/* no counterpart in the reference (it prints wall-clock timers, server-batch-prio3/src/main.rs:
 * 276,340-341,414-421).  With profiling on, every kernel launch is bracketed by HIP events on
 * the context stream; wh_ctx_kernel_times synchronises and returns accumulated milliseconds and
 * launch counts per kernel (indices WH_K_*), optionally resetting them. */
enum { WH_K_PARSE = 0, WH_K_EXPAND = 1, WH_K_HASH = 2, WH_K_QUERY = 3, WH_K_AGG = 4, WH_K_HH = 5, WH_K_MEAS = 6,
       WH_K_HH_MAIN = 7, WH_K_HH_SUPPORT = 8, WH_K_HH_CHAIN = 9, WH_K_HH_FINAL = 10, WH_K_PROVE = 11, WH_K_SORT = 12,
       WH_K_CHAIN = 13, WH_NUM_KERNELS = 14 };
int32_t wh_ctx_set_profiling(wh_ctx *ctx, int32_t on);
int32_t wh_ctx_kernel_times(wh_ctx *ctx, double *ms, uint64_t *launches, int32_t reset);

/* ---- client side (CPU): input generator ---- */
/* replaces: Prio3::shard_batched + VdafBatchedKey::get_encoded as client-batch-prio3 uses them
 * (client-batch-prio3/src/main.rs:67-153; vdaf/prio3.rs:537-991).  `rand` is the random string
 * shard_with_random_new consumes, 16 (5 num_aggregators - 1) bytes (144 for Alice and Bob); writes the Leader's
 * encoded report and the num_aggregators - 1 Helpers' (helper_report_len bytes each, aggregator 1 first). */
int32_t wh_client_shard_batched(const wh_prio3_params *params, const uint64_t *measurement,
                                const uint8_t nonce[16], const uint8_t *rand,
                                const uint8_t client_id_be[16], uint8_t *leader_report,
                                uint8_t *helper_report);
/* many reports on `nthreads` host threads; measurements[n][meas_words], nonces[n][16],
 * rands[n][16 (5 num_aggregators - 1)], client ids = first_client_id + i; reports written at leader_reports + i*leader_len
 * and helper_reports + (i*(num_aggregators - 1) + a - 1)*helper_len */
int32_t wh_client_shard_many(const wh_prio3_params *params, const uint64_t *measurements, uint32_t meas_words,
                             const uint8_t *nonces, const uint8_t *rands, uint64_t first_client_id, uint64_t n,
                             uint32_t nthreads, uint8_t *leader_reports, uint8_t *helper_reports);

/* replaces: the same Prio3::shard_batched + VdafBatchedKey::get_encoded for n clients ON THE GPU (vdaf/prio3.rs:537-991,
 * vdaf.rs:148-162): measurement encoding, share / prove-randomness expansion, both proofs (Type::prove), the joint- and
 * query-randomness Fiat-Shamir parts, both public proofs (Type::query with num_shares = 1) and the report encoding.  Same
 * arguments and byte-identical reports as wh_client_shard_many; for a meta-client that must make millions of DISTINCT
 * valid reports (client-batch-prio3/src/main.rs:67-153).  Circuits with at most 511 gadget calls.  Uses the context's
 * verification workspace: results of an earlier prepare call on `ctx` (out shares, status) are gone afterwards.
 * measurements, nonces and rands are host buffers; leader_reports and helper_reports may be host or device memory
 * (a verifier bench keeps them in HBM). */
int32_t wh_client_shard_many_gpu(wh_ctx *ctx, const uint64_t *measurements, uint32_t meas_words, const uint8_t *nonces,
                                 const uint8_t *rands, uint64_t first_client_id, uint64_t n, uint8_t *leader_reports,
                                 uint8_t *helper_reports);

/* replaces (CPU, client side): BCSketch::gen_key_chain (hhcore/src/protocol.rs:31-76) for IntModN<2^depth> -- VDPFKey::gen
 * (hhcore/src/vdpf.rs:54-113) of the main key and of the 257 support keys -- and bincode serialisation.  bucket: 256 data
 * + sign_data (each 1 or -1 mod 2^32) + ctr; thread_rng is replaced by SHAKE128(rng_seed).  Writes both servers' key chains. */
int32_t wh_hh_gen_key_chain(uint32_t depth, const uint8_t id_le[16], uint32_t alpha, const uint32_t bucket[258],
                            const uint8_t rng_seed[32], uint8_t *key_chain_0, uint8_t *key_chain_1);

/* ---- unit-parity surface (mirrors field.rs / xof.rs for kernel-level tests) ---- */
/* op: 0 add, 1 sub, 2 mul, 3 inv(a).  Host buffers of n 16-byte elements. */
int32_t wh_test_f128_op(wh_ctx *ctx, int32_t op, const uint8_t *a, const uint8_t *b, uint8_t *out, uint64_t n);
/* Field64 (fp.rs:359-389), canonical u64 residues: op 0 add, 1 sub, 2 mul, 3 inv.  Arithmetic layer only: no server of
 * the reference instantiates a Field64 type. */
int32_t wh_test_f64_op(wh_ctx *ctx, int32_t op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n);
/* per seed i: XofShake128::seed_stream(seed_i, dst(alg_id, usage), binder).into_field_vec(n_elems) */
int32_t wh_test_xof_expand(wh_ctx *ctx, const uint8_t *seeds, uint64_t n, uint32_t alg_id, uint32_t usage,
                           const uint8_t *binder, uint32_t binder_len, uint32_t n_elems, uint8_t *out);

/* Test hook, no counterpart in the reference: the rejection sampler (prng.rs:73-99) skips 16-byte chunks >= p,
 * which SHAKE output does with probability 2^-59 per element.  A bound below p (16 bytes, little endian) makes
 * that branch run in every XOF of the protocol; NULL or a value >= p restores p.  The client generator has its
 * own (process-wide) setting so that reports and verifier agree. */
int32_t wh_ctx_set_test_sampler_bound(wh_ctx *ctx, const uint8_t bound[16]);
void wh_client_set_test_sampler_bound(const uint8_t bound[16]);

#ifdef __cplusplus
}
#endif
#endif
