"""The servers' sort by testing id (server-batch-prio3/src/main.rs:93, 158; server-hh/src/main.rs:175, 266: a stable sort on
a u128) on the GPU, against numpy's stable lexicographic sort: permutation, sorted ids and gathered tags, bit for bit."""
import numpy as np
import pytest
import torch

import whisper_amd as W
from whisper_amd import hh

pytestmark = pytest.mark.gpu


def expected_order(ids):
    v = ids.view(np.uint64).reshape(-1, 2)          # little endian: column 0 = low half
    return np.lexsort((v[:, 0], v[:, 1])).astype(np.uint32)      # stable; last key is the primary one


def run(ctx, ids, tags):
    dev = torch.device("cuda:0")
    n = ids.shape[0]
    d_ids = torch.from_numpy(ids.copy()).to(dev)
    d_tags = torch.from_numpy(tags.copy()).to(dev)
    d_order = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    d_sids = torch.zeros_like(d_ids)
    d_stags = torch.zeros_like(d_tags)
    torch.cuda.synchronize()
    ctx.sort_by_testing_id_dev(d_ids.data_ptr(), n, d_order.data_ptr(), d_tags.data_ptr(), tags.shape[1], d_sids.data_ptr(), d_stags.data_ptr())
    ctx.synchronize()
    return d_order.cpu().numpy().view(np.uint32)[:n], d_sids.cpu().numpy(), d_stags.cpu().numpy()


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1023, 1024, 1025, 5000, 125_000, 1_000_003, 1_300_000])   # the last: more than 1 024 tiles per bin row
def test_sort_matches_stable_host_sort(n):
    rng = np.random.default_rng(n + 1)
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    if n >= 64:
        ids[n // 3: n // 3 + 20] = ids[0]                 # duplicates: the sort must keep their input order
        ids[n // 2: n // 2 + 9, 8:] = ids[1, 8:]          # equal high halves, random low halves
        ids[5, :] = 0
        ids[6, :] = 255
    tags = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    p = W.Prio3Params(W.SUMVEC, 2, 5, 3)
    vdaf = W.Prio3(p)
    try:
        order, sids, stags = run(vdaf, ids, tags)
        want = expected_order(ids)
        assert np.array_equal(order, want)
        assert np.array_equal(sids, ids[want]) and np.array_equal(stags, tags[want])
    finally:
        vdaf.close()


def test_sort_few_distinct_ids_and_hh_tags():
    """Heavy-hitter proof strings are 64 bytes; many equal ids exercise stability in every pass."""
    n = 20_000
    rng = np.random.default_rng(3)
    pool = rng.integers(0, 256, (7, 16), dtype=np.uint8)
    ids = pool[rng.integers(0, 7, n)]
    tags = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ctx = hh.HeavyHitters(512)
    try:
        order, sids, stags = run(ctx, ids, tags)
        want = expected_order(ids)
        assert np.array_equal(order, want)
        assert np.array_equal(sids, ids[want]) and np.array_equal(stags, tags[want])
        # the order alone (no gathers), and the documented argument checks
        dev = torch.device("cuda:0")
        d_ids = torch.from_numpy(ids.copy()).to(dev)
        d_order = torch.zeros(n, dtype=torch.int32, device=dev)
        ctx.sort_by_testing_id_dev(d_ids.data_ptr(), n, d_order.data_ptr(), None, 0, None, None)
        ctx.synchronize()
        assert np.array_equal(d_order.cpu().numpy().view(np.uint32), want)
        with pytest.raises(W.WhError):
            ctx.sort_by_testing_id_dev(d_ids.data_ptr(), n, d_order.data_ptr(), d_ids.data_ptr(), 24, None, d_ids.data_ptr())
    finally:
        ctx.close()


@pytest.mark.parametrize("n,n_bad", [(1, 0), (1, 1), (1000, 0), (1000, 17), (125_000, 125), (300_001, 5000)])
def test_keep_mask_from_bad_ids(n, n_bad):
    """The merge loops' `!bad_set.contains(testing_id)` (server-batch-prio3/src/main.rs:108-118) as a mask on the GPU."""
    rng = np.random.default_rng(7 * n + n_bad)
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    if n >= 1000:
        ids[10] = ids[11]                                  # a duplicated id is rejected in both places
        ids[20, 8:] = ids[21, 8:]                          # equal high halves only: distinct ids
    as_int = [int.from_bytes(ids[i].tobytes(), "little") for i in range(n)]
    chosen = rng.choice(n, n_bad, replace=False) if n_bad else np.zeros(0, np.int64)
    if n >= 1000 and n_bad:
        chosen[0], chosen[1] = 10, 20
    bad = [as_int[i] for i in chosen] + ([12345, (1 << 128) - 1] if n_bad else [])     # ids nobody has are ignored
    rng.shuffle(bad)
    dev = torch.device("cuda:0")
    d_ids = torch.from_numpy(ids.copy()).to(dev)
    d_keep = torch.full((n,), 7, dtype=torch.uint8, device=dev)
    vdaf = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    try:
        vdaf.keep_mask_dev(d_ids.data_ptr(), n, bad, d_keep.data_ptr())
        vdaf.synchronize()
        bad_set = set(bad)
        want = np.array([0 if v in bad_set else 1 for v in as_int], np.uint8)
        assert np.array_equal(d_keep.cpu().numpy(), want)
    finally:
        vdaf.close()


def test_device_resident_server_flow():
    """run_vdaf_prepare (server-batch-prio3/src/main.rs:65-125) with everything but the group test on the GPU: prepare,
    sort by testing id, (host: general_binary_split_test on the sorted tags), keep mask from the bad ids, masked aggregate.
    Same bad set and aggregate as the host-side flow of test_gpu_parity.py."""
    from common import VK, make_reports, server_views, blob_of
    p = W.Prio3Params(W.SUMVEC, 16, 16, 8)
    n, bad = 3000, {3, 17, 18, 1501, 2999}
    lr, hr, meas = make_reports(p, n, seed=99)
    dev = torch.device("cuda:0")
    vdaf = W.Prio3(p)
    try:
        sorted_views, keep_inputs = [], []
        for reports in server_views(lr, hr, bad):
            blob, off = blob_of(reports)
            d_blob = torch.from_numpy(np.concatenate([blob, np.zeros(1, np.uint8)])).to(dev)
            d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
            d_tags = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_ids = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_status = torch.zeros(n, dtype=torch.int32, device=dev)
            d_order = torch.zeros(n, dtype=torch.int32, device=dev)
            d_sids, d_stags = torch.zeros_like(d_ids), torch.zeros_like(d_tags)
            torch.cuda.synchronize()
            vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            vdaf.sort_by_testing_id_dev(d_ids.data_ptr(), n, d_order.data_ptr(), d_tags.data_ptr(), 16, d_sids.data_ptr(), d_stags.data_ptr())
            vdaf.synchronize()
            assert not d_status.cpu().numpy().any()
            sids, stags = d_sids.cpu().numpy().reshape(n, 16), d_stags.cpu().numpy().reshape(n, 16)
            proofs = [W.ClientProofTag(int.from_bytes(sids[k].tobytes(), "little"), stags[k].tobytes()) for k in range(n)]
            assert all(proofs[k].testing_id <= proofs[k + 1].testing_id for k in range(n - 1))
            sorted_views.append(proofs)
            keep_inputs.append((d_blob, d_off, d_ids.clone()))
        bad_a, bad_b, rounds = W.general_binary_split_test(sorted_views[0], sorted_views[1], VK, len(bad), 16)
        assert bad_a == bad_b and len(bad_a) == len(bad) and rounds > 1
        aggs = []
        for (d_blob, d_off, d_ids), bad_ids in zip(keep_inputs, (bad_a, bad_b)):
            d_tags = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_ids2 = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_status = torch.zeros(n, dtype=torch.int32, device=dev)
            d_keep = torch.zeros(n, dtype=torch.uint8, device=dev)
            d_agg = torch.zeros(16 * p.sizes.output_len, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, d_tags.data_ptr(), d_ids2.data_ptr(), d_status.data_ptr())
            vdaf.keep_mask_dev(d_ids.data_ptr(), n, sorted(bad_ids), d_keep.data_ptr())
            vdaf.aggregate_dev(d_keep.data_ptr(), d_agg.data_ptr())
            vdaf.synchronize()
            keep = d_keep.cpu().numpy()
            assert set(np.flatnonzero(keep == 0)) == bad
            aggs.append(d_agg.cpu().numpy().tobytes())
        assert W.Prio3.unshard(aggs) == [int(sum(int(meas[i][e]) for i in range(n) if i not in bad)) for e in range(16)]
    finally:
        vdaf.close()


@pytest.mark.parametrize("cfg,path,sub", [((W.HISTOGRAM, 0, 128, 64), "default", 0), ((W.SUMVEC, 16, 16, 8), "default", 64),
                                          ((W.SUMVEC, 16, 16, 8), "long_share", 64), ((W.AVERAGE, 9, 0, 0), "default", 0)],
                         ids=["histogram", "sumvec_subbatches", "sumvec_long_share_path", "average"])
def test_prepare_sorted_batch_equals_prepare_then_sort(cfg, path, sub):
    """wh_prio3_prepare_sorted_batch_dev (the sort runs on its own stream beside the sponge kernels) against
    wh_prio3_prepare_batch_dev followed by wh_sort_by_testing_id_dev: tags, ids, status, permutation, sorted ids and sorted tags
    bit for bit -- with malformed reports (zero ids: ties that must keep their input order), several sub-batches, both pipelines,
    and twice in a row on one context."""
    import torch
    from common import VK, blob_of, make_reports, server_views
    p = W.Prio3Params(*cfg)
    n = 300
    lr, hr, _ = make_reports(p, n, seed=31, nthreads=4)
    alice, _ = server_views(lr, hr, {5})
    alice = list(alice)
    for i in (7, 100, 101, 250):
        alice[i] = alice[i][:20]                                              # undecodable: testing id 0
    blob, off = blob_of(alice)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(np.concatenate([blob, np.zeros(1, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)

    def bufs():
        return dict(tags=torch.zeros(16 * n, dtype=torch.uint8, device=dev), ids=torch.full((16 * n,), 7, dtype=torch.uint8, device=dev),
                    status=torch.full((n,), -99, dtype=torch.int32, device=dev), order=torch.full((n,), -1, dtype=torch.int32, device=dev),
                    sids=torch.zeros(16 * n, dtype=torch.uint8, device=dev), stags=torch.zeros(16 * n, dtype=torch.uint8, device=dev))
    vdaf = W.Prio3(p)
    if path == "long_share":
        vdaf.set_option("chain_max_input_len", 0)
    if sub:
        vdaf.set_option("max_subbatch", sub)
    try:
        a = bufs()
        vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, a["tags"].data_ptr(), a["ids"].data_ptr(), a["status"].data_ptr())
        vdaf.sort_by_testing_id_dev(a["ids"].data_ptr(), n, a["order"].data_ptr(), a["tags"].data_ptr(), 16, a["sids"].data_ptr(), a["stags"].data_ptr())
        vdaf.synchronize()
        agg_a = vdaf.aggregate(None)
        for _ in range(2):
            b = bufs()
            vdaf.prepare_sorted_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, b["tags"].data_ptr(), b["ids"].data_ptr(), b["status"].data_ptr(),
                                          b["order"].data_ptr(), b["sids"].data_ptr(), b["stags"].data_ptr())
            vdaf.synchronize()
            for k in a:
                assert torch.equal(a[k], b[k]), k
            assert np.array_equal(vdaf.aggregate(None), agg_a)
        st = a["status"].cpu().numpy()
        assert list(np.flatnonzero(st)) == [7, 100, 101, 250]
        order = a["order"].cpu().numpy()
        assert list(order[:4]) == [7, 100, 101, 250]                       # the zero ids first, in input order
        ids = a["ids"].cpu().numpy().reshape(n, 16)
        v = ids.view(np.uint64).reshape(n, 2)
        assert np.array_equal(order.astype(np.int64), np.lexsort((np.arange(n), v[:, 0], v[:, 1])))
    finally:
        vdaf.close()
