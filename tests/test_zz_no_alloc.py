"""wh_sort_by_testing_id_noalloc_dev, the entry point that allocates nothing (its scratch is a file-scope device array of
the library), against wh_sort_by_testing_id_dev: same permutation, sorted ids and gathered tags, bit for bit, on two calls
in a row -- the second one smaller, so it runs on scratch the first call left behind.  No counterpart in the reference; the
sort itself replaces server-batch-prio3/src/main.rs:93, 158."""
import numpy as np
import pytest
import torch

import whisper_amd as W

pytestmark = pytest.mark.gpu


def expected_order(ids):
    v = ids.view(np.uint64).reshape(-1, 2)
    return np.lexsort((v[:, 0], v[:, 1])).astype(np.uint32)


def make_ids(n, seed, tag_bytes):
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    if n >= 64:
        ids[n // 3: n // 3 + 20] = ids[0]                 # duplicates keep their input order
        ids[n // 2: n // 2 + 9, 8:] = ids[1, 8:]
        ids[5, :] = 0
        ids[6, :] = 255
    return ids, rng.integers(0, 256, (n, tag_bytes), dtype=np.uint8)


def sort_with(fn, ids, tags):
    dev = torch.device("cuda:0")
    n = ids.shape[0]
    d_ids = torch.from_numpy(ids.copy()).to(dev)
    d_tags = torch.from_numpy(tags.copy()).to(dev)
    d_order = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    d_sids = torch.zeros_like(d_ids)
    d_stags = torch.zeros_like(d_tags)
    torch.cuda.synchronize()
    fn(d_ids.data_ptr(), n, d_order.data_ptr(), d_tags.data_ptr(), tags.shape[1], d_sids.data_ptr(), d_stags.data_ptr())
    torch.cuda.synchronize()
    return d_order.cpu().numpy().view(np.uint32)[:n], d_sids.cpu().numpy(), d_stags.cpu().numpy()


def test_noalloc_sort_matches_allocating_sort_twice_in_a_row():
    vdaf = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    try:
        # the second call is smaller than the first, the third has fewer tiles than a bin row of the second, the last is the maximum
        for n, tag_bytes, seed in ((300_000, 16, 1), (5_000, 64, 2), (1, 16, 3), (65, 16, 4), (W.SORT_NOALLOC_MAX_N, 16, 5)):
            ids, tags = make_ids(n, seed, tag_bytes)
            got = sort_with(vdaf.sort_by_testing_id_noalloc_dev, ids, tags)
            ref = sort_with(vdaf.sort_by_testing_id_dev, ids, tags)
            want = expected_order(ids)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[0], want), n
            assert np.array_equal(got[1], ref[1]) and np.array_equal(got[1], ids[want]), n
            assert np.array_equal(got[2], ref[2]) and np.array_equal(got[2], tags[want]), n
    finally:
        vdaf.close()


def test_noalloc_sort_limits_and_empty_input():
    vdaf = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    try:
        dev = torch.device("cuda:0")
        n = W.SORT_NOALLOC_MAX_N + 1
        d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
        d_order = torch.zeros(n, dtype=torch.int32, device=dev)
        with pytest.raises(W.WhError) as e:
            vdaf.sort_by_testing_id_noalloc_dev(d_ids.data_ptr(), n, d_order.data_ptr(), None, 0, None, None)
        assert e.value.code == -1 and "2^20" in str(e.value)
        vdaf.sort_by_testing_id_noalloc_dev(None, 0, None, None, 0, None, None)            # nothing to do, nothing touched
        with pytest.raises(W.WhError):                                                      # tag rows must be multiples of 16
            vdaf.sort_by_testing_id_noalloc_dev(d_ids.data_ptr(), 10, d_order.data_ptr(), d_ids.data_ptr(), 24, None, d_ids.data_ptr())
        # a second context on the same GPU shares the array (same address): sequential use is fine
        other = W.Prio3(W.Prio3Params(W.HISTOGRAM, 0, 4, 2))
        try:
            ids, tags = make_ids(2_000, 9, 16)
            got = sort_with(other.sort_by_testing_id_noalloc_dev, ids, tags)
            assert np.array_equal(got[0], expected_order(ids))
        finally:
            other.close()
    finally:
        vdaf.close()


def test_two_contexts_on_one_gpu_take_turns_on_the_scratch():
    """Two contexts (two streams) on one GPU call the allocation-free sort back to back with nothing awaited in between: the entry
    point orders them by its per-device event, so neither sort sees the other's permutation buffer or count table."""
    dev = torch.device("cuda:0")
    a = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    b = W.Prio3(W.Prio3Params(W.HISTOGRAM, 0, 4, 2))
    try:
        jobs = []
        for rnd, (ctx, n) in enumerate(((a, 400_000), (b, 350_000), (a, 90_000), (b, 400_000), (a, 1_000), (b, 777))):
            ids, _ = make_ids(n, 40 + rnd, 16)
            d_ids = torch.from_numpy(ids).to(dev)
            d_order = torch.full((n,), -1, dtype=torch.int32, device=dev)
            jobs.append((ctx, ids, d_ids, d_order))
        torch.cuda.synchronize()
        for ctx, ids, d_ids, d_order in jobs:               # enqueue everything first: the sorts of a and b overlap in time
            ctx.sort_by_testing_id_noalloc_dev(d_ids.data_ptr(), ids.shape[0], d_order.data_ptr(), None, 0, None, None)
        a.synchronize(); b.synchronize()
        for _, ids, _, d_order in jobs:
            assert np.array_equal(d_order.cpu().numpy().view(np.uint32), expected_order(ids)), ids.shape
    finally:
        a.close(); b.close()
