"""Host-side mirror of the reference's Prio3 server surface for the batched path.

Names follow the Rust crate: `new_sum_vec_256` / `new_histogram_256` / `new_average_256`
(batched-flpcp-rs/src/vdaf/prio3.rs:150-165, 300-313, 360-380), `BatchAggregator::prepare_batched`
(vdaf.rs:508-520) -- here over a whole batch, as the GPU wants it -- `Aggregator::aggregate`
(prio3.rs:2445-2456) and `Collector::unshard` (prio3.rs:2909-2921).  The per-report loop of
server-batch-prio3 (main.rs:33-60, 75-91) is `Prio3.prepare_encoded_keys`.
"""
import ctypes as C

import numpy as np

from ._lib import lib, Params, Sizes, WhError, OPTIONS

SUMVEC, HISTOGRAM, AVERAGE = 0, 1, 2
P128 = 2**128 - 7 * 2**66 + 1
AVG_BITS = 64  # bin-utils/src/lib.rs:34


class Prio3Params:
    def __init__(self, circuit, bits=0, length=0, chunk=0, num_aggregators=2, alg_id=0):
        self.c = Params(circuit, bits, length, chunk, num_aggregators, alg_id)
        s = Sizes()
        rc = lib.wh_prio3_sizes_of(C.byref(self.c), C.byref(s))
        if rc:
            raise WhError(rc, "invalid Prio3 parameters")
        self.sizes = s

    def __getattr__(self, name):
        return getattr(self.sizes, name)


def _ptr(a):
    return None if a is None else a.ctypes.data


def client_set_test_sampler_bound(bound=None):
    """Test hook for the CPU client generator (process-wide), see Prio3.set_test_sampler_bound."""
    b = None if bound is None else np.frombuffer(int(bound).to_bytes(16, "little"), np.uint8).copy()
    lib.wh_client_set_test_sampler_bound(None if b is None else b.ctypes.data)


def flp_encode(params, measurement):
    """Type::encode_measurement -> bytes(input_len*16)."""
    m = np.ascontiguousarray(measurement, dtype=np.uint64)
    out = np.zeros(16 * params.sizes.input_len, np.uint8)
    rc = lib.wh_flp_encode(C.byref(params.c), m.ctypes.data, out.ctypes.data)
    if rc:
        raise WhError(rc, "flp_encode")
    return out


def flp_prove(params, inp, prove_rand, joint_rand):
    """Type::prove (CPU) -> uint8[proof_len*16]."""
    a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (inp, prove_rand, joint_rand)]
    out = np.zeros(16 * params.sizes.proof_len, np.uint8)
    rc = lib.wh_flp_prove(C.byref(params.c), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, out.ctypes.data)
    if rc:
        raise WhError(rc, "flp_prove")
    return out


def shard_batched_many(params, measurements, nonces, rands, first_client_id=0, nthreads=8):
    """CPU client: client-batch-prio3/src/main.rs:67-153 for n clients.

    measurements: uint64 [n, words]; nonces: uint8 [n,16]; rands: uint8 [n, 16 (5 num_aggregators - 1)] (144 for two).
    Returns (leader_reports uint8 [n, leader_len], helper_reports uint8 [n, helper_len]); with more than two aggregators
    helper_reports is [n, num_aggregators - 1, helper_len], aggregator 1 first."""
    meas = np.ascontiguousarray(measurements, dtype=np.uint64)
    if meas.ndim == 1:
        meas = meas.reshape(-1, 1)
    n = meas.shape[0]
    na = params.c.num_aggregators
    nonces = np.ascontiguousarray(nonces, dtype=np.uint8).reshape(n, 16)
    rands = np.ascontiguousarray(rands, dtype=np.uint8).reshape(n, 16 * (5 * na - 1))
    s = params.sizes
    lr = np.zeros((n, s.leader_report_len), np.uint8)
    hr = np.zeros((n, s.helper_report_len) if na == 2 else (n, na - 1, s.helper_report_len), np.uint8)
    rc = lib.wh_client_shard_many(C.byref(params.c), meas.ctypes.data, meas.shape[1], nonces.ctypes.data,
                                  rands.ctypes.data, first_client_id, n, nthreads, lr.ctypes.data, hr.ctypes.data)
    if rc:
        raise WhError(rc, "shard_batched")
    return lr, hr


# indices WH_K_* of wh_ctx_kernel_times
KERNELS = ("parse", "expand", "hash", "query", "aggregate", "hh_sum", "meas", "hh_main", "hh_support", "hh_chain", "hh_final", "prove", "sort", "chain")


class Prio3:
    """One Prio3 instance bound to one GPU (one wh_ctx)."""

    def __init__(self, params, device=0):
        self.params = params
        self.sizes = params.sizes
        self._ctx = C.c_void_p()
        rc = lib.wh_ctx_create(device, C.byref(params.c), C.byref(self._ctx))
        if rc:
            raise WhError(rc, "wh_ctx_create (is a GPU present? there is no CPU fallback)")
        self.n_last = 0

    def close(self):
        if self._ctx:
            lib.wh_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise WhError(rc, f"{what}: {lib.wh_last_error(self._ctx).decode()}")

    @property
    def stream(self):
        return lib.wh_ctx_stream(self._ctx)

    def synchronize(self):
        self._chk(lib.wh_ctx_synchronize(self._ctx), "synchronize")

    def set_option(self, name, value):
        """wh_ctx_set_option: a tuning / test knob of this context (names: whisper_amd._lib.OPTIONS)."""
        self._chk(lib.wh_ctx_set_option(self._ctx, OPTIONS[name], int(value)), f"set_option({name})")

    def get_option(self, name):
        v = C.c_uint64()
        self._chk(lib.wh_ctx_get_option(self._ctx, OPTIONS[name], C.byref(v)), f"get_option({name})")
        return v.value

    def output_len(self):
        return self.sizes.output_len

    def verifier_len(self):
        return self.sizes.verifier_len

    # ---- host-buffer path (what the Rust servers hold: Vec<Vec<u8>>) ----
    def prepare_encoded_keys(self, verify_key, reports):
        """run_vdaf_prepare's map over prepare_encoded_key (server-batch-prio3/src/main.rs:75-91).

        reports: list of bytes.  Returns (tags [n,16] u8, testing_ids [n,16] u8, status [n] i32)."""
        n = len(reports)
        offsets = np.zeros(n + 1, np.uint64)
        if n:
            offsets[1:] = np.cumsum([len(r) for r in reports])
        blob = np.frombuffer(b"".join(reports) + b"\0", dtype=np.uint8)
        return self.prepare_batch_host(verify_key, blob, offsets)

    def prepare_batch_host(self, verify_key, blob, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        n = len(offsets) - 1
        tags = np.zeros((n, 16), np.uint8)
        ids = np.zeros((n, 16), np.uint8)
        status = np.zeros(n, np.int32)
        vk = np.frombuffer(bytes(verify_key), dtype=np.uint8)
        self._chk(lib.wh_prio3_prepare_batch(self._ctx, vk.ctypes.data, blob.ctypes.data, offsets.ctypes.data, n,
                                             tags.ctypes.data, ids.ctypes.data, status.ctypes.data), "prepare_batch")
        self.n_last = n
        return tags, ids, status

    # ---- un-batched Prio3 (draft-07): server-base-prio3/src/main.rs:26-103 ----
    @staticmethod
    def encode_vdaf_key(agg_id, public_share, input_share, nonce):
        """VdafKey::encode (vdaf.rs:210-217)."""
        return bytes([agg_id]) + bytes(public_share) + bytes(input_share) + bytes(nonce)

    def prepare_init_encoded_keys(self, verify_key, keys):
        """The first map of run_vdaf_prepare: decode + Prio3::prepare_init per VdafKey.

        keys: list of bytes.  Returns (prep_shares [n, (verifier_len+1)*16] u8, status [n] i32)."""
        n = len(keys)
        offsets = np.zeros(n + 1, np.uint64)
        if n:
            offsets[1:] = np.cumsum([len(r) for r in keys])
        blob = np.frombuffer(b"".join(keys) + b"\0", dtype=np.uint8)
        prep = np.zeros((n, 16 * (self.sizes.verifier_len + 1)), np.uint8)
        status = np.zeros(n, np.int32)
        vk = np.frombuffer(bytes(verify_key), dtype=np.uint8)
        self._chk(lib.wh_prio3_prepare_init_batch(self._ctx, vk.ctypes.data, blob.ctypes.data, offsets.ctypes.data, n,
                                                  prep.ctypes.data, status.ctypes.data), "prepare_init_batch")
        self.n_last = n
        return prep, status

    def prepare_finish(self, prep_shares_by_aggregator):
        """prepare_shares_to_prepare_message + prepare_next for the last prepare_init batch.

        prep_shares_by_aggregator: one [n, (verifier_len+1)*16] array per aggregator, in agg_id order.
        Returns (prep_msgs [n,16] u8, status [n] i32)."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in prep_shares_by_aggregator]
        n = self.n_last
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        msgs = np.zeros((n, 16), np.uint8)
        status = np.zeros(n, np.int32)
        self._chk(lib.wh_prio3_prepare_finish_batch(self._ctx, ptrs, n, msgs.ctypes.data, status.ctypes.data), "prepare_finish_batch")
        return msgs, status

    # ---- flp::Type surface (flp.rs:186-196) ----
    def flp_query(self, inputs, proofs, query_rands, joint_rands, num_shares):
        """Type::query for n tuples: uint8 arrays [n, input_len*16], [n, proof_len*16], [n, 16], [n, jr_len*16].
        Returns (verifiers [n, verifier_len*16], status [n])."""
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (inputs, proofs, query_rands, joint_rands)]
        n = a[0].shape[0]
        ver = np.zeros((n, 16 * self.sizes.verifier_len), np.uint8)
        status = np.zeros(n, np.int32)
        self._chk(lib.wh_flp_query_batch(self._ctx, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                         num_shares, n, ver.ctypes.data, status.ctypes.data), "flp_query_batch")
        self.n_last = n
        return ver, status

    def flp_prove(self, inputs, prove_rands, joint_rands):
        """Type::prove on the GPU for n tuples: uint8 [n, input_len*16], [n, arity*16], [n, joint_rand_len*16]
        -> (proofs uint8 [n, proof_len*16], status int32 [n])."""
        s = self.sizes
        a = np.ascontiguousarray(inputs, dtype=np.uint8).reshape(-1, 16 * s.input_len)
        n = a.shape[0]
        pr = np.ascontiguousarray(prove_rands, dtype=np.uint8).reshape(n, 16 * s.arity)
        jr = np.ascontiguousarray(joint_rands, dtype=np.uint8).reshape(n, 16 * s.joint_rand_len)
        out = np.zeros((n, 16 * s.proof_len), np.uint8)
        status = np.zeros(n, np.int32)
        self._chk(lib.wh_flp_prove_batch(self._ctx, a.ctypes.data, pr.ctypes.data, jr.ctypes.data, n, out.ctypes.data,
                                         status.ctypes.data), "flp_prove_batch")
        return out, status

    def shard_batched_many(self, measurements, nonces, rands, first_client_id=0):
        """Prio3::shard_batched for n clients on the GPU: same arguments and byte-identical reports as the module-level
        shard_batched_many (the CPU client).  Overwrites the state of an earlier prepare call on this instance."""
        meas = np.ascontiguousarray(measurements, dtype=np.uint64)
        if meas.ndim == 1:
            meas = meas.reshape(-1, 1)
        n = meas.shape[0]
        nonces = np.ascontiguousarray(nonces, dtype=np.uint8).reshape(n, 16)
        rands = np.ascontiguousarray(rands, dtype=np.uint8).reshape(n, 144)
        s = self.sizes
        lr = np.zeros((n, s.leader_report_len), np.uint8)
        hr = np.zeros((n, s.helper_report_len), np.uint8)
        self._chk(lib.wh_client_shard_many_gpu(self._ctx, meas.ctypes.data, meas.shape[1], nonces.ctypes.data, rands.ctypes.data,
                                               first_client_id, n, lr.ctypes.data, hr.ctypes.data), "shard_batched_many (GPU)")
        self.n_last = 0
        return lr, hr

    def shard_batched_many_dev(self, measurements, nonces, rands, first_client_id, d_leader_reports, d_helper_reports):
        """The same, reports written to device buffers [n, leader_report_len] and [n, helper_report_len] (raw pointers)."""
        meas = np.ascontiguousarray(measurements, dtype=np.uint64)
        if meas.ndim == 1:
            meas = meas.reshape(-1, 1)
        n = meas.shape[0]
        nonces = np.ascontiguousarray(nonces, dtype=np.uint8).reshape(n, 16)
        rands = np.ascontiguousarray(rands, dtype=np.uint8).reshape(n, 144)
        self._chk(lib.wh_client_shard_many_gpu(self._ctx, meas.ctypes.data, meas.shape[1], nonces.ctypes.data, rands.ctypes.data,
                                               first_client_id, n, d_leader_reports, d_helper_reports), "shard_batched_many (GPU)")
        self.n_last = 0

    def ntt(self, vectors, size, inverse=False):
        """discrete_fourier_transform (fft.rs:31-85) of every row of `vectors` (uint8 [n, n_in, 16] residues), zero-padded to
        `size`; inverse=True adds inv_finish (fft.rs:102-115).  Returns uint8 [n, size, 16]."""
        v = np.ascontiguousarray(vectors, dtype=np.uint8)
        n, n_in = v.shape[0], v.shape[1]
        out = np.zeros((n, size, 16), np.uint8)
        self._chk(lib.wh_ntt_f128_batch(self._ctx, v.ctypes.data if v.size else None, n_in, size, 1 if inverse else 0, n, out.ctypes.data),
                  "ntt_f128_batch")
        return out

    def flp_decide(self, verifiers):
        v = np.ascontiguousarray(verifiers, dtype=np.uint8)
        n = v.shape[0]
        out = np.zeros(n, np.int32)
        self._chk(lib.wh_flp_decide_batch(self._ctx, v.ctypes.data, n, out.ctypes.data), "flp_decide_batch")
        return out

    def out_shares(self):
        out = np.zeros((self.n_last, 16 * self.sizes.output_len), np.uint8)
        self._chk(lib.wh_prio3_out_shares(self._ctx, out.ctypes.data), "out_shares")
        return out

    def aggregate(self, keep=None):
        """Aggregator::aggregate over the kept reports of the last batch -> bytes(output_len*16)."""
        if keep is not None:
            keep = np.ascontiguousarray(keep, dtype=np.uint8)
            assert len(keep) == self.n_last
        agg = np.zeros(16 * self.sizes.output_len, np.uint8)
        self._chk(lib.wh_prio3_aggregate(self._ctx, _ptr(keep), agg.ctypes.data), "aggregate")
        return agg

    # ---- device-pointer path (inputs resident in HBM; torch tensors supply the memory) ----
    def prepare_batch_dev(self, verify_key, d_blob, d_offsets, n, d_tags, d_ids, d_status):
        vk = np.frombuffer(bytes(verify_key), dtype=np.uint8)
        self._chk(lib.wh_prio3_prepare_batch_dev(self._ctx, vk.ctypes.data, d_blob, d_offsets, n, d_tags, d_ids,
                                                 d_status), "prepare_batch_dev")
        self.n_last = n

    def prepare_sorted_batch_dev(self, verify_key, d_blob, d_offsets, n, d_tags, d_ids, d_status, d_order, d_sorted_ids=None, d_sorted_tags=None):
        """prepare_batch_dev + sort_by_testing_id_dev in one call (run_vdaf_prepare up to the group test, main.rs:75-94): the sort
        runs beside the sponge kernels.  Same outputs as the two calls."""
        vk = np.frombuffer(bytes(verify_key), dtype=np.uint8)
        self._chk(lib.wh_prio3_prepare_sorted_batch_dev(self._ctx, vk.ctypes.data, d_blob, d_offsets, n, d_tags, d_ids, d_status, d_order,
                                                        d_sorted_ids, d_sorted_tags), "prepare_sorted_batch_dev")
        self.n_last = n

    def sort_by_testing_id_dev(self, d_ids, n, d_order, d_tags=None, tag_bytes=16, d_sorted_ids=None, d_sorted_tags=None):
        """The servers' stable sort by testing id (main.rs:93) on device buffers (raw pointers); asynchronous."""
        self._chk(lib.wh_sort_by_testing_id_dev(self._ctx, d_ids, n, d_order, d_tags, tag_bytes, d_sorted_ids, d_sorted_tags),
                  "sort_by_testing_id_dev")

    def sort_by_testing_id_noalloc_dev(self, d_ids, n, d_order, d_tags=None, tag_bytes=16, d_sorted_ids=None, d_sorted_tags=None):
        """The same sort through the entry point that allocates nothing (scratch in a file-scope device array, n <= 2^20)."""
        self._chk(lib.wh_sort_by_testing_id_noalloc_dev(self._ctx, d_ids, n, d_order, d_tags, tag_bytes, d_sorted_ids, d_sorted_tags),
                  "sort_by_testing_id_noalloc_dev")

    def keep_mask_dev(self, d_ids, n, bad_ids, d_keep):
        """keep[i] = testing id i not in bad_ids (iterable of ints, the group test's verdict); device buffers as raw pointers."""
        bad = np.frombuffer(b"".join(int(x).to_bytes(16, "little") for x in bad_ids), np.uint8) if len(bad_ids) else np.zeros(0, np.uint8)
        self._chk(lib.wh_keep_mask_dev(self._ctx, d_ids, n, bad.ctypes.data if len(bad) else None, len(bad) // 16, d_keep), "keep_mask_dev")

    def allgather_reduce(self, keep=None):
        """wh_prio3_allgather_reduce: the masked aggregate of the last batch on this GPU, combined over the ranks the context was
        bound to (whisper_amd.dist.comm_init) -> bytes(output_len*16) on every rank."""
        if keep is not None:
            keep = np.ascontiguousarray(keep, dtype=np.uint8)
        agg = np.zeros(16 * self.sizes.output_len, np.uint8)
        self._chk(lib.wh_prio3_allgather_reduce(self._ctx, _ptr(keep), agg.ctypes.data), "allgather_reduce")
        return agg

    def lock(self):
        self._chk(lib.wh_ctx_lock(self._ctx), "lock")

    def unlock(self):
        self._chk(lib.wh_ctx_unlock(self._ctx), "unlock")

    def aggregate_dev(self, d_keep, d_agg):
        self._chk(lib.wh_prio3_aggregate_dev(self._ctx, d_keep, d_agg), "aggregate_dev")

    def sum_rows_dev(self, d_rows, n_rows, length, d_keep, d_out):
        self._chk(lib.wh_f128_sum_rows_dev(self._ctx, d_rows, n_rows, length, d_keep, d_out), "sum_rows_dev")

    def hh_bucket_sum_u32_dev(self, d_buckets, d_keep, n, words, d_acc, accumulate=False):
        fn = lib.wh_hh_bucket_accumulate_u32_dev if accumulate else lib.wh_hh_bucket_sum_u32_dev
        self._chk(fn(self._ctx, d_buckets, d_keep, n, words, d_acc), "hh_bucket_sum")

    # ---- measurement ----
    KERNELS = KERNELS

    def set_profiling(self, on=True):
        self._chk(lib.wh_ctx_set_profiling(self._ctx, int(on)), "set_profiling")

    def kernel_times(self, reset=True):
        """{kernel: (total_ms, launches)} measured with HIP events on the context stream."""
        ms = np.zeros(len(self.KERNELS), np.float64)
        cnt = np.zeros(len(self.KERNELS), np.uint64)
        self._chk(lib.wh_ctx_kernel_times(self._ctx, ms.ctypes.data, cnt.ctypes.data, int(reset)), "kernel_times")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNELS)}

    # ---- unit-parity surface ----
    def set_test_sampler_bound(self, bound=None):
        """Test hook: lower the rejection bound of the sampler (int < p), None restores p."""
        b = None if bound is None else np.frombuffer(int(bound).to_bytes(16, "little"), np.uint8).copy()
        self._chk(lib.wh_ctx_set_test_sampler_bound(self._ctx, None if b is None else b.ctypes.data), "set_test_sampler_bound")

    def test_f128_op(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        n = a.size // 16
        out = np.zeros(a.size, np.uint8)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.uint8)
            bp = b.ctypes.data
        self._chk(lib.wh_test_f128_op(self._ctx, op, a.ctypes.data, bp, out.ctypes.data, n), "test_f128_op")
        return out

    def test_f64_op(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(a)
        self._chk(lib.wh_test_f64_op(self._ctx, op, a.ctypes.data, _ptr(bb), out.ctypes.data, a.size), "test_f64_op")
        return out

    def test_xof_expand(self, seeds, alg_id, usage, binder, n_elems):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint8)
        n = seeds.size // 16
        out = np.zeros((n, 16 * n_elems), np.uint8)
        b = np.frombuffer(bytes(binder) + b"\0", dtype=np.uint8)
        self._chk(lib.wh_test_xof_expand(self._ctx, seeds.ctypes.data, n, alg_id, usage, b.ctypes.data, len(binder),
                                         n_elems, out.ctypes.data), "test_xof_expand")
        return out

    @staticmethod
    def unshard(agg_shares):
        """Collector::unshard (prio3.rs:2909-2921): element-wise sum of the aggregate shares."""
        n = len(agg_shares[0]) // 16
        out = []
        for e in range(n):
            out.append(sum(int.from_bytes(bytes(a[16 * e:16 * e + 16]), "little") for a in agg_shares) % P128)
        return out


def new_sum_vec_256(num_aggregators, bits, length, chunk_length, device=0):
    return Prio3(Prio3Params(SUMVEC, bits, length, chunk_length, num_aggregators), device)


def new_histogram_256(num_aggregators, length, chunk_length, device=0):
    return Prio3(Prio3Params(HISTOGRAM, 0, length, chunk_length, num_aggregators), device)


def new_average_256(num_aggregators, bits=AVG_BITS, device=0):
    return Prio3(Prio3Params(AVERAGE, bits, 0, 0, num_aggregators), device)
