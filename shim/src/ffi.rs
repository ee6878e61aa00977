//! Raw declarations of `libwhisper_gfx950.so`: one `extern "C"` item per entry point of `include/whisper_gfx950.h`, same
//! order, same argument names.  `tests/test_shim_matches_header.py` parses both files and fails on a missing export, a
//! different arity or a different integer width / pointer constness, so the two cannot drift apart unnoticed.
//!
//! Nothing here is safe to call directly; `lib.rs` wraps the calls a server needs.
#![allow(non_camel_case_types, clippy::too_many_arguments)]
use std::os::raw::{c_char, c_void};

/// Opaque context: one GPU, one HIP stream, the workspace of one batch (`wh_ctx`).
#[repr(C)]
pub struct WhCtx {
    _private: [u8; 0],
}

/// `wh_prio3_params`: the arguments of `Prio3::new_sum_vec_256 / new_histogram_256 / new_average_256`
/// (batched-flpcp-rs/src/vdaf/prio3.rs:150-165, 300-313, 360-380).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct WhPrio3Params {
    pub circuit: u32,
    pub bits: u32,
    pub len: u32,
    pub chunk: u32,
    pub num_aggregators: u32,
    pub alg_id: u32,
}

/// `wh_prio3_sizes` (flp/types.rs:650-676, 444-470, 286-312; report sizes from vdaf.rs:148-162).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct WhPrio3Sizes {
    pub input_len: u32,
    pub proof_len: u32,
    pub verifier_len: u32,
    pub output_len: u32,
    pub joint_rand_len: u32,
    pub calls: u32,
    pub m: u32,
    pub arity: u32,
    pub leader_report_len: u32,
    pub helper_report_len: u32,
}

pub const WH_SUMVEC: u32 = 0;
pub const WH_HISTOGRAM: u32 = 1;
pub const WH_AVERAGE: u32 = 2;

pub const WH_OK: i32 = 0;
pub const WH_ERR_PARAM: i32 = -1;
pub const WH_ERR_SHORT: i32 = -2;
pub const WH_ERR_MODULUS: i32 = -3;
pub const WH_ERR_AGG_ID: i32 = -4;
pub const WH_ERR_NUM_QUERIES: i32 = -5;
pub const WH_ERR_QUERY_RAND: i32 = -6;
pub const WH_ERR_DECIDE: i32 = -7;
pub const WH_ERR_JOINT_RAND: i32 = -8;
pub const WH_ERR_KEY_FORMAT: i32 = -9;
pub const WH_ERR_HIP: i32 = -100;
pub const WH_ERR_NO_DEVICE: i32 = -101;
pub const WH_ERR_STATE: i32 = -102;
pub const WH_ERR_UNSUPPORTED: i32 = -103;
pub const WH_ERR_COMM: i32 = -104;

pub const WH_COMM_ID_BYTES: usize = 128;
pub const WH_SORT_NOALLOC_MAX_N: u64 = 1 << 20;
pub const WH_NUM_KERNELS: usize = 14;
pub const WH_NUM_OPTS: i32 = 12;

extern "C" {
    pub fn wh_ctx_create(device_id: i32, params: *const WhPrio3Params, out: *mut *mut WhCtx) -> i32;
    pub fn wh_ctx_destroy(ctx: *mut WhCtx);
    pub fn wh_prio3_sizes_of(params: *const WhPrio3Params, out: *mut WhPrio3Sizes) -> i32;
    pub fn wh_last_error(ctx: *const WhCtx) -> *const c_char;
    pub fn wh_ctx_stream(ctx: *mut WhCtx) -> *mut c_void;
    pub fn wh_ctx_lock(ctx: *mut WhCtx) -> i32;
    pub fn wh_ctx_unlock(ctx: *mut WhCtx) -> i32;
    pub fn wh_prio3_prepare_batch_dev(ctx: *mut WhCtx, verify_key: *const u8, d_blob: *const u8, d_offsets: *const u64, n: u64, d_tags: *mut u8, d_testing_ids: *mut u8, d_status: *mut i32) -> i32;
    pub fn wh_prio3_prepare_sorted_batch_dev(ctx: *mut WhCtx, verify_key: *const u8, d_blob: *const u8, d_offsets: *const u64, n: u64, d_tags: *mut u8, d_testing_ids: *mut u8, d_status: *mut i32, d_order: *mut u32, d_sorted_ids: *mut u8, d_sorted_tags: *mut u8) -> i32;
    pub fn wh_prio3_prepare_batch(ctx: *mut WhCtx, verify_key: *const u8, blob: *const u8, offsets: *const u64, n: u64, tags: *mut u8, testing_ids: *mut u8, status: *mut i32) -> i32;
    pub fn wh_host_alloc(bytes: u64, out: *mut *mut c_void) -> i32;
    pub fn wh_host_free(p: *mut c_void) -> i32;
    pub fn wh_prio3_out_shares(ctx: *mut WhCtx, out_shares: *mut u8) -> i32;
    pub fn wh_prio3_prepare_init_batch_dev(ctx: *mut WhCtx, verify_key: *const u8, d_blob: *const u8, d_offsets: *const u64, n: u64, d_prep_shares: *mut u8, d_status: *mut i32) -> i32;
    pub fn wh_prio3_prepare_init_batch(ctx: *mut WhCtx, verify_key: *const u8, blob: *const u8, offsets: *const u64, n: u64, prep_shares: *mut u8, status: *mut i32) -> i32;
    pub fn wh_prio3_prepare_finish_batch(ctx: *mut WhCtx, prep_shares: *const *const u8, n: u64, prep_msgs: *mut u8, status: *mut i32) -> i32;
    pub fn wh_prio3_count_prepare_init_batch(ctx: *mut WhCtx, num_aggregators: u32, verify_key: *const u8, blob: *const u8, offsets: *const u64, n: u64, prep_shares: *mut u8, status: *mut i32) -> i32;
    pub fn wh_prio3_count_prepare_finish_batch(ctx: *mut WhCtx, num_aggregators: u32, prep_shares: *const *const u8, n: u64, status: *mut i32) -> i32;
    pub fn wh_prio3_count_out_shares(ctx: *mut WhCtx, out_shares: *mut u8) -> i32;
    pub fn wh_prio3_count_aggregate(ctx: *mut WhCtx, keep: *const u8, agg: *mut u8) -> i32;
    pub fn wh_test_f64_xof_expand(ctx: *mut WhCtx, seeds: *const u8, n: u32, usage: u32, binder: *const u8, binder_len: u32, n_elems: u32, bound: u64, out: *mut u64) -> i32;
    pub fn wh_prio3_count_set_test_sampler_bound(ctx: *mut WhCtx, bound: u64) -> i32;
    pub fn wh_flp_query_batch(ctx: *mut WhCtx, input: *const u8, proof: *const u8, query_rand: *const u8, joint_rand: *const u8, num_shares: u32, n: u64, verifier: *mut u8, status: *mut i32) -> i32;
    pub fn wh_flp_decide_batch(ctx: *mut WhCtx, verifier: *const u8, n: u64, decision: *mut i32) -> i32;
    pub fn wh_flp_prove_batch(ctx: *mut WhCtx, input: *const u8, prove_rand: *const u8, joint_rand: *const u8, n: u64, proof: *mut u8, status: *mut i32) -> i32;
    pub fn wh_flp_encode(params: *const WhPrio3Params, measurement: *const u64, input: *mut u8) -> i32;
    pub fn wh_flp_prove(params: *const WhPrio3Params, input: *const u8, prove_rand: *const u8, joint_rand: *const u8, proof: *mut u8) -> i32;
    pub fn wh_ntt_f128_batch(ctx: *mut WhCtx, in_: *const u8, n_in: u64, size: u64, inverse: i32, n: u64, out: *mut u8) -> i32;
    pub fn wh_prio3_aggregate_dev(ctx: *mut WhCtx, d_keep: *const u8, d_agg: *mut u8) -> i32;
    pub fn wh_prio3_aggregate(ctx: *mut WhCtx, keep: *const u8, agg: *mut u8) -> i32;
    pub fn wh_f128_sum_rows_dev(ctx: *mut WhCtx, d_rows: *const u8, n_rows: u64, len: u32, d_keep: *const u8, d_out: *mut u8) -> i32;
    pub fn wh_comm_unique_id(id: *mut u8) -> i32;
    pub fn wh_comm_init(ctx: *mut WhCtx, id: *const u8, rank: i32, world: i32) -> i32;
    pub fn wh_comm_destroy(ctx: *mut WhCtx) -> i32;
    pub fn wh_comm_rank(ctx: *mut WhCtx, rank: *mut i32, world: *mut i32) -> i32;
    pub fn wh_comm_why() -> *const c_char;
    pub fn wh_prio3_allgather_reduce_dev(ctx: *mut WhCtx, d_partial: *const u8, d_agg: *mut u8) -> i32;
    pub fn wh_f128_allgather_reduce_dev(ctx: *mut WhCtx, d_partial: *const u8, len: u32, d_out: *mut u8) -> i32;
    pub fn wh_prio3_allgather_reduce(ctx: *mut WhCtx, keep: *const u8, agg: *mut u8) -> i32;
    pub fn wh_hh_allreduce_u32_dev(ctx: *mut WhCtx, d_sketch: *mut u32, words: u64) -> i32;
    pub fn wh_hh_bucket_sum_u32_dev(ctx: *mut WhCtx, d_buckets: *const u32, d_keep: *const u8, n: u64, words: u64, d_acc: *mut u32) -> i32;
    pub fn wh_hh_bucket_accumulate_u32_dev(ctx: *mut WhCtx, d_buckets: *const u32, d_keep: *const u8, n: u64, words: u64, d_acc: *mut u32) -> i32;
    pub fn wh_hh_bucket_sum_f128_dev(ctx: *mut WhCtx, d_buckets: *const u8, d_keep: *const u8, n: u64, words: u64, d_acc: *mut u8) -> i32;
    pub fn wh_hh_bucket_accumulate_f128_dev(ctx: *mut WhCtx, d_buckets: *const u8, d_keep: *const u8, n: u64, words: u64, d_acc: *mut u8) -> i32;
    pub fn wh_hh_key_chain_len(depth: u32) -> u64;
    pub fn wh_hh_ctx_create(device_id: i32, out: *mut *mut WhCtx) -> i32;
    pub fn wh_hh_eval_and_check_batch_dev(ctx: *mut WhCtx, verify_key: *const u8, d_keys: *const u8, stride: u64, n: u64, depth: u32, d_dat: *mut u32, d_tags: *mut u8, d_testing_ids: *mut u8, d_status: *mut i32) -> i32;
    pub fn wh_hh_eval_and_sketch_batch_dev(ctx: *mut WhCtx, verify_key: *const u8, d_keys: *const u8, stride: u64, n: u64, depth: u32, group_size: u64, d_sketch: *mut u32, accumulate: i32, d_tags: *mut u8, d_testing_ids: *mut u8, d_status: *mut i32) -> i32;
    pub fn wh_hh_sketch_subtract_dev(ctx: *mut WhCtx, verify_key: *const u8, d_keys: *const u8, stride: u64, depth: u32, bad_index: *const u64, n_bad: u64, group_size: u64, d_sketch: *mut u32) -> i32;
    pub fn wh_hh_eval_and_check_batch(ctx: *mut WhCtx, verify_key: *const u8, keys: *const u8, stride: u64, n: u64, depth: u32, tags: *mut u8, testing_ids: *mut u8, status: *mut i32) -> i32;
    pub fn wh_hh_buckets(ctx: *mut WhCtx, dat: *mut u32) -> i32;
    pub fn wh_hh_aggregate(ctx: *mut WhCtx, keep: *const u8, acc: *mut u32) -> i32;
    pub fn wh_hh_aggregate_dev(ctx: *mut WhCtx, d_keep: *const u8, d_acc: *mut u32, accumulate: i32) -> i32;
    pub fn wh_sort_by_testing_id_dev(ctx: *mut WhCtx, d_ids: *const u8, n: u64, d_order: *mut u32, d_tags: *const u8, tag_bytes: u32, d_sorted_ids: *mut u8, d_sorted_tags: *mut u8) -> i32;
    pub fn wh_sort_by_testing_id_noalloc_dev(ctx: *mut WhCtx, d_ids: *const u8, n: u64, d_order: *mut u32, d_tags: *const u8, tag_bytes: u32, d_sorted_ids: *mut u8, d_sorted_tags: *mut u8) -> i32;
    pub fn wh_sort_by_testing_id(ctx: *mut WhCtx, ids: *const u8, n: u64, order: *mut u32) -> i32;
    pub fn wh_keep_mask_dev(ctx: *mut WhCtx, d_ids: *const u8, n: u64, bad_ids: *const u8, n_bad: u64, d_keep: *mut u8) -> i32;
    pub fn wh_grouptest_batch_proofs_dev(ctx: *mut WhCtx, d_ids: *const u8, d_tags: *const u8, n: u64, tag_bytes: u32, verify_key: *const u8, starts: *const u64, ends: *const u64, uppers: *const u8, k: u32, proof_size: u32, proofs: *mut u8, counts: *mut u64) -> i32;
    pub fn wh_grouptest_split_proofs_dev(ctx: *mut WhCtx, d_ids: *const u8, d_tags: *const u8, n: u64, tag_bytes: u32, verify_key: *const u8, uppers: *const u8, d: u32, proof_size: u32, proofs: *mut u8, firsts: *mut u64, counts: *mut u64) -> i32;
    pub fn wh_ctx_synchronize(ctx: *mut WhCtx) -> i32;
    pub fn wh_ctx_set_option(ctx: *mut WhCtx, option: i32, value: u64) -> i32;
    pub fn wh_ctx_get_option(ctx: *mut WhCtx, option: i32, value: *mut u64) -> i32;
    pub fn wh_ctx_set_profiling(ctx: *mut WhCtx, on: i32) -> i32;
    pub fn wh_ctx_kernel_times(ctx: *mut WhCtx, ms: *mut f64, launches: *mut u64, reset: i32) -> i32;
    pub fn wh_client_shard_batched(params: *const WhPrio3Params, measurement: *const u64, nonce: *const u8, rand: *const u8, client_id_be: *const u8, leader_report: *mut u8, helper_report: *mut u8) -> i32;
    pub fn wh_client_shard_many(params: *const WhPrio3Params, measurements: *const u64, meas_words: u32, nonces: *const u8, rands: *const u8, first_client_id: u64, n: u64, nthreads: u32, leader_reports: *mut u8, helper_reports: *mut u8) -> i32;
    pub fn wh_client_shard_many_gpu(ctx: *mut WhCtx, measurements: *const u64, meas_words: u32, nonces: *const u8, rands: *const u8, first_client_id: u64, n: u64, leader_reports: *mut u8, helper_reports: *mut u8) -> i32;
    pub fn wh_hh_gen_key_chain(depth: u32, id_le: *const u8, alpha: u32, bucket: *const u32, rng_seed: *const u8, key_chain_0: *mut u8, key_chain_1: *mut u8) -> i32;
    pub fn wh_test_f128_op(ctx: *mut WhCtx, op: i32, a: *const u8, b: *const u8, out: *mut u8, n: u64) -> i32;
    pub fn wh_test_f64_op(ctx: *mut WhCtx, op: i32, a: *const u64, b: *const u64, out: *mut u64, n: u64) -> i32;
    pub fn wh_test_xof_expand(ctx: *mut WhCtx, seeds: *const u8, n: u64, alg_id: u32, usage: u32, binder: *const u8, binder_len: u32, n_elems: u32, out: *mut u8) -> i32;
    pub fn wh_ctx_set_test_sampler_bound(ctx: *mut WhCtx, bound: *const u8) -> i32;
    pub fn wh_client_set_test_sampler_bound(bound: *const u8);
}
