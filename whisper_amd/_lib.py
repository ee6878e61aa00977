"""ctypes binding of libwhisper_gfx950.so (include/whisper_gfx950.h)."""
import ctypes as C
import os

# torch supplies device memory, streams and torch.distributed (RCCL) around this library.  It ships
# its own libamdhip64 under the same SONAME; loading torch first makes this library bind to that one
# copy instead of bringing a second HIP runtime into the process (the second one sees no GPU).
import torch  # noqa: F401

# WHISPER_GFX950_LIB selects another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("WHISPER_GFX950_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwhisper_gfx950.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C whisper_amd/csrc` (or __graft_entry__.build()); "
        "there is no CPU fallback for the GPU path")
lib = C.CDLL(LIB_PATH)


class WhError(RuntimeError):
    def __init__(self, code, what=""):
        super().__init__(f"whisper_gfx950 error {code} {what}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("circuit", C.c_uint32), ("bits", C.c_uint32), ("len", C.c_uint32), ("chunk", C.c_uint32),
                ("num_aggregators", C.c_uint32), ("alg_id", C.c_uint32)]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "input_len", "proof_len", "verifier_len", "output_len", "joint_rand_len", "calls", "m", "arity",
        "leader_report_len", "helper_report_len")]


EXPORTS = [
    "wh_ctx_create", "wh_ctx_destroy", "wh_prio3_sizes_of", "wh_last_error", "wh_ctx_stream",
    "wh_prio3_prepare_batch_dev", "wh_prio3_prepare_batch", "wh_prio3_out_shares",
    "wh_prio3_aggregate_dev", "wh_prio3_aggregate", "wh_f128_sum_rows_dev", "wh_hh_bucket_sum_u32_dev",
    "wh_ctx_synchronize", "wh_client_shard_batched", "wh_client_shard_many", "wh_test_f128_op",
    "wh_test_xof_expand", "wh_ctx_set_profiling", "wh_ctx_kernel_times",
    "wh_ctx_set_test_sampler_bound", "wh_client_set_test_sampler_bound",
    "wh_test_f64_op", "wh_ntt_f128_batch", "wh_flp_query_batch", "wh_flp_decide_batch", "wh_flp_encode", "wh_flp_prove", "wh_hh_bucket_accumulate_u32_dev", "wh_prio3_prepare_init_batch_dev", "wh_prio3_prepare_init_batch", "wh_prio3_prepare_finish_batch",
    "wh_hh_key_chain_len", "wh_hh_ctx_create", "wh_hh_eval_and_check_batch_dev", "wh_hh_eval_and_check_batch", "wh_hh_buckets",
    "wh_hh_aggregate", "wh_hh_aggregate_dev", "wh_hh_gen_key_chain", "wh_flp_prove_batch", "wh_client_shard_many_gpu", "wh_sort_by_testing_id_dev", "wh_sort_by_testing_id", "wh_keep_mask_dev",
    "wh_sort_by_testing_id_noalloc_dev", "wh_grouptest_batch_proofs_dev", "wh_grouptest_split_proofs_dev", "wh_hh_bucket_sum_f128_dev", "wh_hh_bucket_accumulate_f128_dev", "wh_hh_eval_and_sketch_batch_dev", "wh_hh_sketch_subtract_dev", "wh_ctx_lock", "wh_ctx_unlock", "wh_host_alloc", "wh_host_free", "wh_comm_unique_id", "wh_comm_init", "wh_comm_destroy", "wh_comm_rank",
    "wh_prio3_allgather_reduce_dev", "wh_f128_allgather_reduce_dev", "wh_prio3_allgather_reduce", "wh_hh_allreduce_u32_dev",
    "wh_prio3_count_prepare_init_batch", "wh_prio3_count_prepare_finish_batch", "wh_prio3_count_out_shares", "wh_prio3_count_aggregate",
    "wh_test_f64_xof_expand", "wh_prio3_count_set_test_sampler_bound",
    "wh_ctx_set_option", "wh_ctx_get_option", "wh_comm_why", "wh_prio3_prepare_sorted_batch_dev",
]

vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32
lib.wh_ctx_create.argtypes = [i32, C.POINTER(Params), C.POINTER(vp)]
lib.wh_ctx_destroy.argtypes = [vp]
lib.wh_ctx_destroy.restype = None
lib.wh_prio3_sizes_of.argtypes = [C.POINTER(Params), C.POINTER(Sizes)]
lib.wh_last_error.argtypes = [vp]
lib.wh_last_error.restype = C.c_char_p
lib.wh_ctx_stream.argtypes = [vp]
lib.wh_ctx_stream.restype = vp
lib.wh_prio3_prepare_batch_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp]
lib.wh_prio3_prepare_batch.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp]
lib.wh_prio3_out_shares.argtypes = [vp, vp]
lib.wh_prio3_aggregate_dev.argtypes = [vp, vp, vp]
lib.wh_prio3_aggregate.argtypes = [vp, vp, vp]
lib.wh_f128_sum_rows_dev.argtypes = [vp, vp, u64, u32, vp, vp]
lib.wh_hh_bucket_sum_u32_dev.argtypes = [vp, vp, vp, u64, u64, vp]
lib.wh_hh_bucket_accumulate_u32_dev.argtypes = [vp, vp, vp, u64, u64, vp]
lib.wh_ctx_synchronize.argtypes = [vp]
lib.wh_client_shard_batched.argtypes = [C.POINTER(Params), vp, vp, vp, vp, vp, vp]
lib.wh_client_shard_many.argtypes = [C.POINTER(Params), vp, u32, vp, vp, u64, u64, u32, vp, vp]
lib.wh_ctx_set_profiling.argtypes = [vp, i32]
lib.wh_ctx_kernel_times.argtypes = [vp, vp, vp, i32]
lib.wh_test_f128_op.argtypes = [vp, i32, vp, vp, vp, u64]
lib.wh_test_xof_expand.argtypes = [vp, vp, u64, u32, u32, vp, u32, u32, vp]
lib.wh_prio3_prepare_init_batch_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp]
lib.wh_prio3_prepare_init_batch.argtypes = [vp, vp, vp, vp, u64, vp, vp]
lib.wh_prio3_prepare_finish_batch.argtypes = [vp, vp, u64, vp, vp]
lib.wh_test_f64_op.argtypes = [vp, i32, vp, vp, vp, u64]
lib.wh_prio3_count_prepare_init_batch.argtypes = [vp, u32, vp, vp, vp, u64, vp, vp]
lib.wh_prio3_count_prepare_finish_batch.argtypes = [vp, u32, vp, u64, vp]
lib.wh_prio3_count_out_shares.argtypes = [vp, vp]
lib.wh_prio3_count_aggregate.argtypes = [vp, vp, vp]
lib.wh_test_f64_xof_expand.argtypes = [vp, vp, u32, u32, vp, u32, u32, u64, vp]
lib.wh_prio3_count_set_test_sampler_bound.argtypes = [vp, u64]
lib.wh_flp_query_batch.argtypes = [vp, vp, vp, vp, vp, u32, u64, vp, vp]
lib.wh_flp_decide_batch.argtypes = [vp, vp, u64, vp]
lib.wh_ntt_f128_batch.argtypes = [vp, vp, u64, u64, i32, u64, vp]
lib.wh_flp_encode.argtypes = [C.POINTER(Params), vp, vp]
lib.wh_flp_prove.argtypes = [C.POINTER(Params), vp, vp, vp, vp]
lib.wh_ctx_set_test_sampler_bound.argtypes = [vp, vp]
lib.wh_client_set_test_sampler_bound.argtypes = [vp]
lib.wh_client_set_test_sampler_bound.restype = None
lib.wh_flp_prove_batch.argtypes = [vp, vp, vp, vp, u64, vp, vp]
lib.wh_client_shard_many_gpu.argtypes = [vp, vp, u32, vp, vp, u64, u64, vp, vp]
lib.wh_sort_by_testing_id_dev.argtypes = [vp, vp, u64, vp, vp, u32, vp, vp]
lib.wh_sort_by_testing_id.argtypes = [vp, vp, u64, vp]
lib.wh_sort_by_testing_id_noalloc_dev.argtypes = [vp, vp, u64, vp, vp, u32, vp, vp]
lib.wh_keep_mask_dev.argtypes = [vp, vp, u64, vp, u64, vp]
lib.wh_grouptest_batch_proofs_dev.argtypes = [vp, vp, vp, u64, u32, vp, vp, vp, vp, u32, u32, vp, vp]
lib.wh_grouptest_split_proofs_dev.argtypes = [vp, vp, vp, u64, u32, vp, vp, u32, u32, vp, vp, vp]
lib.wh_host_alloc.argtypes = [u64, C.POINTER(vp)]
lib.wh_host_free.argtypes = [vp]
lib.wh_ctx_lock.argtypes = [vp]
lib.wh_ctx_unlock.argtypes = [vp]
lib.wh_comm_unique_id.argtypes = [vp]
lib.wh_comm_init.argtypes = [vp, vp, i32, i32]
lib.wh_comm_destroy.argtypes = [vp]
lib.wh_comm_rank.argtypes = [vp, vp, vp]
lib.wh_prio3_allgather_reduce_dev.argtypes = [vp, vp, vp]
lib.wh_f128_allgather_reduce_dev.argtypes = [vp, vp, u32, vp]
lib.wh_prio3_allgather_reduce.argtypes = [vp, vp, vp]
lib.wh_hh_allreduce_u32_dev.argtypes = [vp, vp, u64]
lib.wh_hh_key_chain_len.argtypes = [u32]
lib.wh_hh_ctx_create.argtypes = [i32, C.POINTER(vp)]
lib.wh_hh_eval_and_check_batch_dev.argtypes = [vp, vp, vp, u64, u64, u32, vp, vp, vp, vp]
lib.wh_hh_eval_and_check_batch.argtypes = [vp, vp, vp, u64, u64, u32, vp, vp, vp]
lib.wh_hh_eval_and_sketch_batch_dev.argtypes = [vp, vp, vp, u64, u64, u32, u64, vp, i32, vp, vp, vp]
lib.wh_hh_sketch_subtract_dev.argtypes = [vp, vp, vp, u64, u32, vp, u64, u64, vp]
lib.wh_hh_bucket_sum_f128_dev.argtypes = [vp, vp, vp, u64, u64, vp]
lib.wh_hh_bucket_accumulate_f128_dev.argtypes = [vp, vp, vp, u64, u64, vp]
lib.wh_hh_buckets.argtypes = [vp, vp]
lib.wh_hh_aggregate.argtypes = [vp, vp, vp]
lib.wh_hh_aggregate_dev.argtypes = [vp, vp, vp, i32]
lib.wh_hh_gen_key_chain.argtypes = [u32, vp, u32, vp, vp, vp, vp]
lib.wh_prio3_prepare_sorted_batch_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, vp, vp, vp]
lib.wh_ctx_set_option.argtypes = [vp, i32, u64]
lib.wh_ctx_get_option.argtypes = [vp, i32, C.POINTER(u64)]
lib.wh_comm_why.argtypes = []
for _n in EXPORTS:
    getattr(lib, _n)  # AttributeError here = the library does not export what the header declares
    if _n not in ("wh_ctx_destroy", "wh_last_error", "wh_ctx_stream", "wh_client_set_test_sampler_bound", "wh_comm_why"):
        getattr(lib, _n).restype = i32
lib.wh_hh_key_chain_len.restype = u64
lib.wh_comm_why.restype = C.c_char_p

# WH_OPT_* of wh_ctx_set_option (include/whisper_gfx950.h)
OPTIONS = {"max_subbatch": 0, "host_chunk_mb": 1, "host_copy_threads": 2, "hh_subbatch": 3, "hh_sum_lds": 4, "side_stream": 5,
           "hash_b_side": 6, "third_stream": 7, "base_paired": 8, "chain_max_input_len": 9, "chain_prefetch": 10, "query_tiny": 11}
