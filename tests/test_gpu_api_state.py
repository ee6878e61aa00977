"""C-ABI contract of the device-pointer entry points (include/whisper_gfx950.h): malformed offset arrays are a per-report
CodecError, never a wild read; the context only aggregates what a prepare call left; a caller may free its status array
between prepare and aggregate.  The results on the well-formed reports are the oracle's."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import whisper_amd as W
from common import VK, blob_of, make_reports, server_views

pytestmark = pytest.mark.gpu

CFG = (W.SUMVEC, 16, 16, 8)
WH_ERR_SHORT, WH_ERR_STATE = -2, -102


def device_inputs(reports, offsets=None):
    dev = torch.device("cuda:0")
    blob, off = blob_of(reports)
    if offsets is not None:
        off = offsets
    n = len(off) - 1
    d_blob = torch.from_numpy(np.concatenate([blob, np.zeros(1, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_tags = torch.full((16 * n,), 7, dtype=torch.uint8, device=dev)
    d_ids = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), 99, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    return d_blob, d_off, d_tags, d_ids, d_status


def test_descending_offset_pair_is_a_short_report_not_a_wild_read():
    p = W.Prio3Params(*CFG)
    n = 40
    lr, hr, _ = make_reports(p, n, seed=31)
    alice, _ = server_views(lr, hr)
    blob, off = blob_of(alice)
    bad_off = off.copy()
    # report 7 ends before it starts (its neighbours keep their true bounds: 6 ends at off[7], 8 starts at off[8])
    broken = bad_off.copy()
    broken[8] = broken[7] - 5
    # the kernel sees pairs (off[i], off[i + 1]): pair 7 is descending, pair 8 starts early and therefore holds garbage
    vdaf = W.Prio3(p)
    try:
        d_blob, d_off, d_tags, d_ids, d_status = device_inputs(alice, broken)
        d_tags.fill_(0xa5); d_ids.fill_(0x5a)             # what a reused buffer holds: a malformed report's tag and id are zero all the same
        vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
        vdaf.synchronize()
        status = d_status.cpu().numpy()
        tags = d_tags.cpu().numpy().reshape(n, 16)
        assert status[7] == WH_ERR_SHORT and not tags[7].any()
        assert not d_ids.cpu().numpy().reshape(n, 16)[7].any()
        good = [i for i in range(n) if i not in (7, 8)]
        o_tags, o_ids, o_status, _ = O.prio3_prepare_batch(O.params(*CFG), VK, blob, off, nthreads=2)
        assert not status[good].any()
        assert np.array_equal(tags[good], o_tags[good])
        assert np.array_equal(d_ids.cpu().numpy().reshape(n, 16)[good], o_ids[good])
        # the un-batched entry point parses the same way
        d_prep = torch.zeros(n * 16 * (p.verifier_len + 1), dtype=torch.uint8, device="cuda:0")
        d_status.fill_(99)
        torch.cuda.synchronize()
        rc = W.lib.wh_prio3_prepare_init_batch_dev(vdaf._ctx, np.frombuffer(VK, np.uint8).ctypes.data, d_blob.data_ptr(), d_off.data_ptr(), n,
                                                   d_prep.data_ptr(), d_status.data_ptr())
        assert rc == 0
        vdaf.synchronize()
        assert d_status.cpu().numpy()[7] == WH_ERR_SHORT
    finally:
        vdaf.close()


def test_aggregate_only_after_prepare_and_status_may_be_freed():
    p = W.Prio3Params(*CFG)
    n = 64
    lr, hr, meas = make_reports(p, n, seed=32)
    alice, _ = server_views(lr, hr)
    blob, off = blob_of(alice)
    _, _, _, o_outs = O.prio3_prepare_batch(O.params(*CFG), VK, blob, off, nthreads=2)
    want = O.prio3_aggregate(o_outs, p.output_len)
    dev = torch.device("cuda:0")
    vdaf = W.Prio3(p)
    try:
        d_agg = torch.zeros(16 * p.output_len, dtype=torch.uint8, device=dev)
        # nothing prepared yet
        with pytest.raises(W.WhError) as e:
            vdaf.aggregate_dev(None, d_agg.data_ptr())
        assert e.value.code == WH_ERR_STATE
        d_blob, d_off, d_tags, d_ids, d_status = device_inputs(alice)
        vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
        # the caller's status array goes away (and its memory is overwritten) before the aggregation: the context kept a copy
        vdaf.synchronize()
        d_status.fill_(-7)
        del d_status
        torch.cuda.empty_cache()
        junk = torch.full((1 << 20,), -3, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        vdaf.aggregate_dev(None, d_agg.data_ptr())
        vdaf.synchronize()
        assert np.array_equal(d_agg.cpu().numpy(), want)
        del junk
        # Type::query on the same context reuses the workspace: out shares yes (its documented by-product), aggregate no
        inp = np.zeros((3, 16 * p.input_len), np.uint8)
        proofs = np.zeros((3, 16 * p.proof_len), np.uint8)
        qr = np.ones((3, 16), np.uint8)
        jr = np.ones((3, 16 * p.joint_rand_len), np.uint8)
        vdaf.flp_query(inp, proofs, qr, jr, 2)
        assert vdaf.out_shares().shape == (3, 16 * p.output_len)
        with pytest.raises(W.WhError) as e:
            vdaf.aggregate(None)
        assert e.value.code == WH_ERR_STATE
        with pytest.raises(W.WhError) as e:
            vdaf.aggregate_dev(None, d_agg.data_ptr())
        assert e.value.code == WH_ERR_STATE
        # the sharder too: nothing to aggregate and no out shares afterwards
        vdaf.prepare_encoded_keys(VK, alice)
        assert np.array_equal(vdaf.aggregate(None), want)
        m = np.zeros((2, p.c.len), np.uint64)
        vdaf.shard_batched_many(m, np.zeros((2, 16), np.uint8), np.zeros((2, 144), np.uint8))
        for call in (lambda: vdaf.aggregate(None), lambda: vdaf.aggregate_dev(None, d_agg.data_ptr())):
            with pytest.raises(W.WhError) as e:
                call()
            assert e.value.code == WH_ERR_STATE
        # and a fresh prepare makes the context whole again
        vdaf.prepare_encoded_keys(VK, alice)
        assert np.array_equal(vdaf.aggregate(None), want)
    finally:
        vdaf.close()


def test_keep_mask_calls_back_to_back_stay_ordered():
    """wh_keep_mask_dev stages the sorted bad ids in context-owned pinned memory: two calls in a row with different bad sets, no
    synchronisation in between, give two correct masks."""
    n = 50_000
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    as_int = [int.from_bytes(ids[i].tobytes(), "little") for i in range(n)]
    dev = torch.device("cuda:0")
    d_ids = torch.from_numpy(ids).to(dev)
    d_keep1 = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    d_keep2 = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    bad1 = [as_int[i] for i in range(0, n, 7)]
    bad2 = [as_int[i] for i in range(3, n, 1001)]
    vdaf = W.Prio3(W.Prio3Params(*CFG))
    try:
        torch.cuda.synchronize()
        vdaf.keep_mask_dev(d_ids.data_ptr(), n, bad1, d_keep1.data_ptr())
        vdaf.keep_mask_dev(d_ids.data_ptr(), n, bad2, d_keep2.data_ptr())
        vdaf.synchronize()
        s1, s2 = set(bad1), set(bad2)
        assert np.array_equal(d_keep1.cpu().numpy(), np.array([v not in s1 for v in as_int], np.uint8))
        assert np.array_equal(d_keep2.cpu().numpy(), np.array([v not in s2 for v in as_int], np.uint8))
    finally:
        vdaf.close()


@pytest.mark.parametrize("through_rccl", [False, True])
def test_combine_behind_the_c_abi_with_one_rank(through_rccl):
    """wh_comm_init / wh_prio3_allgather_reduce[_dev] / wh_hh_allreduce_u32_dev with world = 1: a lone GPU without RCCL, and the
    same collectives through RCCL (ncclCommInitRank, ncclAllGather, ncclAllReduce with one rank).  More ranks need more GPUs:
    the driver's multi-GPU bench runs this path; tests/test_dist_gloo.py rehearses the sharding on the CPU."""
    from whisper_amd import dist as D
    from whisper_amd import hh
    p = W.Prio3Params(*CFG)
    n = 48
    lr, hr, _ = make_reports(p, n, seed=33)
    alice, _ = server_views(lr, hr)
    keep = np.ones(n, np.uint8)
    keep[5] = 0
    dev = torch.device("cuda:0")
    vdaf = W.Prio3(p)
    try:
        with pytest.raises(W.WhError) as e:                     # not bound to a rank yet
            vdaf.allgather_reduce(None)
        assert e.value.code == WH_ERR_STATE
        D.comm_init(vdaf, D.comm_unique_id() if through_rccl else None, 0, 1)
        assert D.comm_rank(vdaf) == (0, 1)
        vdaf.prepare_encoded_keys(VK, alice)
        want = vdaf.aggregate(keep)
        assert np.array_equal(vdaf.allgather_reduce(keep), want)
        d_part = torch.from_numpy(want.copy()).to(dev)
        d_out = torch.zeros_like(d_part)
        torch.cuda.synchronize()
        D.allgather_reduce_dev(vdaf, d_part.data_ptr(), d_out.data_ptr())
        vdaf.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want)
        # any Field128 row length, non-canonical limbs included: the sum of one row is that row reduced mod p
        rows = np.frombuffer(np.random.default_rng(1).bytes(16 * 300), np.uint8).copy()
        rows[-16:] = 255
        d_rows = torch.from_numpy(rows).to(dev)
        d_red = torch.zeros_like(d_rows)
        torch.cuda.synchronize()
        D.allgather_reduce_dev(vdaf, d_rows.data_ptr(), d_red.data_ptr(), length=300)
        vdaf.synchronize()
        got = d_red.cpu().numpy()
        for e_ in (0, 150, 299):
            assert int.from_bytes(got[16 * e_:16 * e_ + 16].tobytes(), "little") == int.from_bytes(rows[16 * e_:16 * e_ + 16].tobytes(), "little") % W.prio3.P128
    finally:
        vdaf.close()
    h = hh.HeavyHitters(64)
    try:
        D.comm_init(h, D.comm_unique_id() if through_rccl else None, 0, 1)
        sk = np.random.default_rng(2).integers(0, 2**32, 64 * 258, dtype=np.uint32)
        d_sk = torch.from_numpy(sk.view(np.int32).copy()).to(dev)
        torch.cuda.synchronize()
        D.hh_allreduce_u32_dev(h, d_sk.data_ptr(), sk.size)
        h.synchronize()
        assert np.array_equal(d_sk.cpu().numpy().view(np.uint32), sk)
    finally:
        h.close()


def test_host_buffer_path_in_chunks_pageable_and_pinned(monkeypatch):
    """wh_prio3_prepare_batch moves the batch in chunks through two device buffers (copy of chunk k + 1 under the verification of
    chunk k): with 1 MiB chunks a 300-report batch takes a dozen of them.  Same tags, ids, status and aggregate as one chunk, for
    pageable and for pinned (wh_host_alloc) input, with malformed reports at chunk seams; a sample against the oracle."""
    import ctypes as C
    cfg = (W.SUMVEC, 16, 128, 64)
    p = W.Prio3Params(*cfg)
    n = 300
    lr, hr, _ = make_reports(p, n, seed=34)
    alice, _ = server_views(lr, hr)
    alice[17] = alice[17][:100]                       # short
    alice[150] = alice[150][:48] + b"\x07" + alice[150][49:]   # bad aggregator id
    alice[299] = b""                                  # empty, last
    blob, off = blob_of(alice)
    blob = np.concatenate([blob, np.zeros(1, np.uint8)])
    keep = np.ones(n, np.uint8)
    keep[::7] = 0
    vdaf = W.Prio3(p)
    try:
        ref = vdaf.prepare_batch_host(VK, blob, off)
        ref_agg = vdaf.aggregate(keep)
        assert ref[2][17] == WH_ERR_SHORT and ref[2][150] == -4 and ref[2][299] == WH_ERR_SHORT
        assert int((ref[2] != 0).sum()) == 3
        sample = list(range(0, 40))
        sblob, soff = blob_of([alice[i] for i in sample])
        o_tags, o_ids, o_status, _ = O.prio3_prepare_batch(O.params(*cfg), VK, sblob, soff, nthreads=4)
        assert np.array_equal(ref[0][sample], o_tags) and np.array_equal(ref[1][sample], o_ids) and np.array_equal(ref[2][sample], o_status)
        vdaf.set_option("host_chunk_mb", 1)
        got = vdaf.prepare_batch_host(VK, blob, off)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        assert np.array_equal(vdaf.aggregate(keep), ref_agg)
        # pinned input: no staging copy
        ptr = C.c_void_p()
        assert W.lib.wh_host_alloc(blob.size, C.byref(ptr)) == 0
        try:
            pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(blob.size,))
            pinned[:] = blob
            got = vdaf.prepare_batch_host(VK, pinned, off)
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))
            assert np.array_equal(vdaf.aggregate(keep), ref_agg)
            vdaf.set_option("host_copy_threads", 1)
            got = vdaf.prepare_batch_host(VK, blob, off)       # pageable again, one copy thread
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        finally:
            del pinned
            assert W.lib.wh_host_free(ptr) == 0
    finally:
        vdaf.close()
