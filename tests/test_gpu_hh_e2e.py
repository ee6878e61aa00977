"""BASELINE.json config 4 / 5 end to end on the GPU: heavy hitters with malicious clients, batch and streaming.

Clients pick 32-byte strings (a few heavy ones, a long tail), put one signed bucket per repetition into a count sketch
(get_sign_and_bkt, bytes_to_bucket) and send each server a key chain; some clients are malformed as client-hh makes them
(two different key chains, client-hh/src/main.rs:84-118).  Both servers expand on the GPU, group-test the tags, sum the
kept clients' buckets per repetition, add the two sketches and read the heavy hitters back (server-hh/src/main.rs:236-402).
Checked against the same flow with the oracle's CPU expansion: identical sketches, identical rejected clients, identical
recovered heavy hitters -- and those are the true heavy strings."""
import hashlib

import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu

DEPTH, REPS, N_CLIENTS, THRESHOLD = 6, 4, 240, 0.1
SALT = bytes([1] * 32)
VK = bytes(range(30, 46))


def _population(k):
    # client-hh/src/main.rs:27-37
    return [hashlib.sha256(SALT + i.to_bytes(8, "big")).digest() for i in range(k)]


def _clients():
    """(alice[rep][client], bob[rep][client] key chains, strings, bad set)"""
    from whisper_amd import hh
    rng = np.random.default_rng(4242)
    pop = _population(60)
    # two true heavy hitters (30 % and 15 % of the clients), the rest spread over the tail
    choice = rng.choice(len(pop), N_CLIENTS, p=[0.30, 0.15] + [0.55 / 58] * 58)
    bad = set(int(i) for i in rng.choice(N_CLIENTS, 7, replace=False))
    ln = hh.key_chain_len(DEPTH)
    A = np.zeros((REPS, N_CLIENTS, ln), np.uint8); B = np.zeros_like(A)
    for c in range(N_CLIENTS):
        for r in range(REPS):
            s = pop[choice[c]]
            sign, idx = hh.get_sign_and_bkt(s, 1 << DEPTH, r)
            seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
            A[r, c], B[r, c] = hh.gen_key_chain(DEPTH, c, idx, hh.bytes_to_bucket(s, sign), seed)
            if c in bad:    # Bob gets the key chain of another string
                s2 = pop[(choice[c] + 1 + r) % len(pop)]
                sign2, idx2 = hh.get_sign_and_bkt(s2, 1 << DEPTH, r)
                _, B[r, c] = hh.gen_key_chain(DEPTH, c, idx2, hh.bytes_to_bucket(s2, sign2), rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
    return A, B, [pop[i] for i in choice], bad, pop


def _server_rep_gpu(h, keys):
    tags, ids, status = h.eval_and_check(VK, keys)
    assert not status.any()
    return tags, ids


def _server_rep_cpu(keys):
    dat, tags, ids, status = o.hh_eval_and_check_batch(DEPTH, VK, keys, nthreads=4)
    assert not status.any()
    return dat, [bytes(t).decode() for t in tags], [int.from_bytes(bytes(i), "little") for i in ids]


def _group_test(tags_a, ids_a, tags_b, ids_b, d):
    from whisper_amd.grouptest import ClientProofTag, general_binary_split_test
    pa = [ClientProofTag(ids_a[i], tags_a[i].encode()) for i in sorted(range(len(ids_a)), key=lambda i: ids_a[i])]
    pb = [ClientProofTag(ids_b[i], tags_b[i].encode()) for i in sorted(range(len(ids_b)), key=lambda i: ids_b[i])]
    bad_a, bad_b, _ = general_binary_split_test(pa, pb, VK, d, tag_size=32)
    assert bad_a == bad_b
    return bad_a


@pytest.mark.parametrize("streaming", [False, True], ids=["batch", "streaming"])
def test_heavy_hitters_end_to_end(streaming):
    from whisper_amd import hh
    A, B, strings, bad, pop = _clients()
    nb = 1 << DEPTH
    ha, hb = hh.HeavyHitters(nb), hh.HeavyHitters(nb)
    recovered_gpu, recovered_cpu = set(), set()
    threshold = int(N_CLIENTS * THRESHOLD)                       # server-hh/src/main.rs:382-383
    slices = [(0, N_CLIENTS)] if not streaming else [(lo, min(lo + 64, N_CLIENTS)) for lo in range(0, N_CLIENTS, 64)]
    for r in range(REPS):
        sk_gpu = np.zeros((2, nb, 258), np.uint32); sk_cpu = np.zeros((2, nb, 258), np.uint32)
        rejected = set()
        for lo, hi in slices:                                    # `batch_size` clients at a time in streaming mode
            ta, ia = _server_rep_gpu(ha, A[r, lo:hi]); tb, ib = _server_rep_gpu(hb, B[r, lo:hi])
            da, cta, cia = _server_rep_cpu(A[r, lo:hi]); db, ctb, cib = _server_rep_cpu(B[r, lo:hi])
            assert (ta, ia, tb, ib) == (cta, cia, ctb, cib)      # tags and testing ids: GPU == oracle
            n_bad_here = len([c for c in bad if lo <= c < hi])
            bad_ids = _group_test(ta, ia, tb, ib, max(n_bad_here, 2))
            assert bad_ids == {ia[c - lo] for c in bad if lo <= c < hi}
            rejected |= {lo + k for k in range(hi - lo) if ia[k] in bad_ids}
            keep = np.array([ia[k] not in bad_ids for k in range(hi - lo)], np.uint8)
            sk_gpu[0] += ha.aggregate_hhs(keep); sk_gpu[1] += hb.aggregate_hhs(keep)          # CountSketch::insert per slice
            sk_cpu[0] += o.hh_bucket_sum_u32(da, keep); sk_cpu[1] += o.hh_bucket_sum_u32(db, keep)
        assert rejected == bad
        assert (sk_gpu == sk_cpu).all()
        merged = sk_gpu[0] + sk_gpu[1]                           # my_sketch.insert(received_sketch), main.rs:386-391
        want = np.zeros((nb, 258), np.uint32)
        for c in range(N_CLIENTS):
            if c not in bad:
                sign, idx = hh.get_sign_and_bkt(strings[c], nb, r)
                want[idx] += hh.bytes_to_bucket(strings[c], sign)
        assert (merged == want).all()
        recovered_gpu |= hh.recover(merged, threshold, r)
        recovered_cpu |= hh.recover(sk_cpu[0] + sk_cpu[1], threshold, r)
    assert recovered_gpu == recovered_cpu
    assert {pop[0], pop[1]} <= recovered_gpu                     # the true heavy hitters are found
    assert recovered_gpu <= set(pop)                             # and nothing that no client sent
    ha.close(); hb.close()


def test_heavy_hitters_device_resident_repetition():
    """One repetition of batch_collect_and_aggregate_per_run (server-hh/src/main.rs:255-282) with everything but the group
    test on the GPU: expansion, sort by testing id (64-byte proof strings gathered in sorted order), keep mask from the
    rejected ids, masked sketch sum.  Same rejected clients and sketches as the host-buffer flow above."""
    import torch
    from whisper_amd import hh
    from whisper_amd.grouptest import ClientProofTag, general_binary_split_test
    A, B, strings, bad, pop = _clients()
    nb, n, r = 1 << DEPTH, N_CLIENTS, 1
    dev = torch.device("cuda:0")
    servers = [hh.HeavyHitters(nb), hh.HeavyHitters(nb)]
    try:
        state, views = [], []
        for h, keys in zip(servers, (A[r], B[r])):
            d_keys = torch.from_numpy(np.ascontiguousarray(keys)).to(dev)
            d_tags = torch.zeros(64 * n, dtype=torch.uint8, device=dev)
            d_ids = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_status = torch.zeros(n, dtype=torch.int32, device=dev)
            d_order = torch.zeros(n, dtype=torch.int32, device=dev)
            d_sids, d_stags = torch.zeros_like(d_ids), torch.zeros_like(d_tags)
            torch.cuda.synchronize()
            h.eval_and_check_dev(VK, d_keys.data_ptr(), keys.shape[1], n, None, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            h.sort_by_testing_id_dev(d_ids.data_ptr(), n, d_order.data_ptr(), d_tags.data_ptr(), 64, d_sids.data_ptr(), d_stags.data_ptr())
            h.synchronize()
            assert not d_status.cpu().numpy().any()
            sids, stags = d_sids.cpu().numpy().reshape(n, 16), d_stags.cpu().numpy().reshape(n, 64)
            views.append([ClientProofTag(int.from_bytes(sids[k].tobytes(), "little"), stags[k].tobytes()) for k in range(n)])
            state.append((d_ids, d_keys))
        bad_a, bad_b, _ = general_binary_split_test(views[0], views[1], VK, len(bad), tag_size=32)
        assert bad_a == bad_b and len(bad_a) == len(bad)
        sketches = []
        for h, (d_ids, _), keys in zip(servers, state, (A[r], B[r])):
            d_keep = torch.zeros(n, dtype=torch.uint8, device=dev)
            d_acc = torch.zeros(nb * 258, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            h.keep_mask_dev(d_ids.data_ptr(), n, sorted(bad_a), d_keep.data_ptr())
            h.aggregate_hhs_dev(d_keep.data_ptr(), d_acc.data_ptr())
            h.synchronize()
            keep = d_keep.cpu().numpy()
            assert set(int(i) for i in np.flatnonzero(keep == 0)) == bad
            sketches.append(d_acc.cpu().numpy().view(np.uint32).reshape(nb, 258))
            dat, _, _, _ = o.hh_eval_and_check_batch(DEPTH, VK, keys, nthreads=4)
            assert (sketches[-1] == o.hh_bucket_sum_u32(dat, keep)).all()
        want = np.zeros((nb, 258), np.uint32)
        for c in range(N_CLIENTS):
            if c not in bad:
                sign, idx = hh.get_sign_and_bkt(strings[c], nb, r)
                want[idx] += hh.bytes_to_bucket(strings[c], sign)
        assert (sketches[0] + sketches[1] == want).all()
    finally:
        for h in servers:
            h.close()
