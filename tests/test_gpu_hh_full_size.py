"""BASELINE.json's heavy-hitter configurations at their per-GPU sizes -- recovery threshold 0.01: 14 repetitions x 512 buckets,
1 000 clients; 0.001: 17 repetitions x 2 048 buckets, 250 clients (bin-utils/src/lib.rs:66-88) -- with malformed clients, both
servers on the GPU, the group test per repetition, per-repetition sketches built the way the bench builds them (all key chains
summed, the rejected ones taken out again), merge and read-out (server-hh/src/main.rs:236-402).  Size-independent checks:
tags and testing ids agree between the servers except for the malformed clients, the merged sketches equal the honest
clients' votes exactly, the planted heavy hitters are recovered and nothing that no client sent; a sample of key chains
(malformed ones included) agrees with the oracle bit for bit.  Plus the Field128 ring of the streaming configuration
(wh_hh_bucket_{sum,accumulate}_f128_dev, an extension: the reference's AggRing is u32) against Python integers."""
import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import oracle_lib as o

pytestmark = pytest.mark.gpu

SALT = bytes([1] * 32)
VK = bytes(range(60, 76))
P128 = 2**128 - 7 * 2**66 + 1


def _population(k):
    return [hashlib.sha256(SALT + i.to_bytes(8, "big")).digest() for i in range(k)]     # client-hh/src/main.rs:27-37


def _make_clients(depth, reps, clients, n_bad, heavy, seed):
    """Key chains repetition-major: A[rep * clients + c], B[...]; strings per client; the malformed clients (Bob gets the key
    chain of another string, client-hh/src/main.rs:84-118)."""
    from whisper_amd import hh
    rng = np.random.default_rng(seed)
    pop = _population(400)
    probs = np.array(list(heavy) + [(1.0 - sum(heavy)) / (len(pop) - len(heavy))] * (len(pop) - len(heavy)))
    choice = rng.choice(len(pop), clients, p=probs)
    bad = set(int(i) for i in rng.choice(clients, n_bad, replace=False))
    ln = hh.key_chain_len(depth)
    N = 1 << depth
    A = np.zeros((reps * clients, ln), np.uint8)
    B = np.zeros_like(A)
    seeds = rng.integers(0, 256, (reps * clients, 2, 32), dtype=np.uint8)

    def one(k):
        r, c = divmod(k, clients)
        s = pop[choice[c]]
        sign, idx = hh.get_sign_and_bkt(s, N, r)
        a, b = hh.gen_key_chain(depth, c, idx, hh.bytes_to_bucket(s, sign), seeds[k, 0].tobytes())
        if c in bad:
            s2 = pop[(choice[c] + 1 + r) % len(pop)]
            sign2, idx2 = hh.get_sign_and_bkt(s2, N, r)
            _, b = hh.gen_key_chain(depth, c, idx2, hh.bytes_to_bucket(s2, sign2), seeds[k, 1].tobytes())
        A[k], B[k] = a, b

    with ThreadPoolExecutor(8) as ex:
        list(ex.map(one, range(reps * clients)))
    return A, B, [pop[i] for i in choice], bad, pop


@pytest.mark.parametrize("depth,reps,clients,n_bad,threshold", [(9, 14, 1000, 10, 0.01), (11, 17, 250, 3, 0.001)],
                         ids=["t0.01_14x512_1000", "t0.001_17x2048_250"])
def test_heavy_hitters_full_size(depth, reps, clients, n_bad, threshold):
    from whisper_amd import hh
    from whisper_amd.grouptest import ClientProofTag, general_binary_split_test
    heavy = (0.20, 0.12, 0.06)
    A, B, strings, bad, pop = _make_clients(depth, reps, clients, n_bad, heavy, 9000 + depth)
    n, N, words = reps * clients, 1 << depth, (1 << depth) * 258
    dev = torch.device("cuda:0")
    servers = [hh.HeavyHitters(N), hh.HeavyHitters(N)]
    try:
        out = []
        for h, keys in zip(servers, (A, B)):
            d_keys = torch.from_numpy(keys).to(dev)
            d_sk = torch.zeros((reps, N, 258), dtype=torch.int32, device=dev)
            d_tags = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
            d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
            d_status = torch.full((n,), 99, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            # every key chain of every repetition in one call; one sketch per repetition
            h.eval_and_sketch_dev(VK, d_keys.data_ptr(), keys.shape[1], n, clients, d_sk.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            h.synchronize()
            assert not d_status.cpu().numpy().any()
            out.append((d_keys, d_sk, d_tags.cpu().numpy(), d_ids.cpu().numpy()))
        (ka, ska, ta, ia), (kb, skb, tb, ib) = out
        # testing ids: equal on both servers, the same for a client in every repetition, pairwise distinct
        assert np.array_equal(ia, ib)
        ids = [int.from_bytes(r.tobytes(), "little") for r in ia[:clients]]
        assert len(set(ids)) == clients
        assert np.array_equal(ia.reshape(reps, clients, 16), np.broadcast_to(ia[:clients], (reps, clients, 16)))
        # tags differ exactly for the malformed clients, in every repetition
        differ = (ta != tb).any(axis=1).reshape(reps, clients)
        assert all(set(np.flatnonzero(differ[r])) == bad for r in range(reps))
        # a sample against the oracle: tags, ids, bucket vectors (malformed clients included)
        sample = sorted({0, 1, clients - 1, n - 1, n // 2} | {r * clients + c for r in (0, reps - 1) for c in list(bad)[:2]})
        for keys, tags, idb in ((A, ta, ia), (B, tb, ib)):
            odat, otags, oids, ostatus = o.hh_eval_and_check_batch(depth, VK, keys[sample], nthreads=4)
            assert not ostatus.any()
            assert np.array_equal(tags[sample], np.array([np.frombuffer(bytes(t), np.uint8) for t in otags]))
            assert np.array_equal(idb[sample], np.array([np.frombuffer(bytes(i), np.uint8) for i in oids]))
        # the group test per repetition on the sorted tags (server-hh/src/main.rs:61-69, 266), then the rejected key chains leave the sums
        order = sorted(range(clients), key=lambda c: ids[c])
        rejected_idx = []
        for r in range(reps):
            pa = [ClientProofTag(ids[c], ta[r * clients + c].tobytes()) for c in order]
            pb = [ClientProofTag(ids[c], tb[r * clients + c].tobytes()) for c in order]
            bad_a, bad_b, _ = general_binary_split_test(pa, pb, VK, max(n_bad, 2), tag_size=32)
            assert bad_a == bad_b == {ids[c] for c in bad}
            rejected_idx += [r * clients + c for c in range(clients) if ids[c] in bad_a]
        for h, d_keys, d_sk, keys in ((servers[0], ka, ska, A), (servers[1], kb, skb, B)):
            h.sketch_subtract_dev(VK, d_keys.data_ptr(), keys.shape[1], rejected_idx, clients, d_sk.data_ptr())
            h.synchronize()
        merged = (ska.cpu().numpy().view(np.uint32) + skb.cpu().numpy().view(np.uint32))       # my_sketch.insert(received), main.rs:386-391
        # exactly the honest clients' votes
        recovered = set()
        for r in range(reps):
            want = np.zeros((N, 258), np.uint32)
            for c in range(clients):
                if c not in bad:
                    sign, idx = hh.get_sign_and_bkt(strings[c], N, r)
                    want[idx] += hh.bytes_to_bucket(strings[c], sign)
            assert np.array_equal(merged[r], want), r
            recovered |= hh.recover(merged[r], int(clients * threshold), r)
        assert {pop[0], pop[1], pop[2]} <= recovered
        assert recovered <= set(strings)
        # the first repetition once more through the materialised path with a keep mask: the same sketch
        keep = np.ones(clients, np.uint8)
        keep[list(bad)] = 0
        servers[0].eval_and_check(VK, A[:clients])
        assert np.array_equal(servers[0].aggregate_hhs(keep), ska[0].cpu().numpy().view(np.uint32))
    finally:
        for h in servers:
            h.close()


def _f128_rows(rng, n, words, noncanonical_at=()):
    v = np.frombuffer(rng.bytes(16 * n * words), np.uint8).reshape(n, words, 16).copy()
    v[..., 15] &= 0x7f                       # canonical: well below p
    for (r, w) in noncanonical_at:
        v[r, w, :] = 255                     # 2^128 - 1 >= p: reduced, not rejected
    return v


def _to_ints(a):
    lo = a.view(np.uint64)[..., 0].astype(object)
    hi = a.view(np.uint64)[..., 1].astype(object)
    return lo + (hi << 64)


@pytest.mark.parametrize("buckets,clients,batch", [(64, 37, 10), (512, 40, 16), (2048, 9, 4)])
def test_field128_sketch_sums_batch_and_streaming(buckets, clients, batch):
    """The streaming server's loop (server-hh/src/main.rs:113-216) over a Field128 ring: batch_size clients at a time, each
    slice's kept bucket vectors added into the running sketch; equal to the one-shot sum and to Python integers mod p."""
    from whisper_amd import hh
    rng = np.random.default_rng(buckets + clients)
    words = buckets * 258
    rows = _f128_rows(rng, clients, words, noncanonical_at=[(1, 5), (clients - 1, words - 1)])
    keep = np.ones(clients, np.uint8)
    keep[[2, clients - 2]] = 0
    dev = torch.device("cuda:0")
    h = hh.HeavyHitters(buckets)
    try:
        d_rows = torch.from_numpy(rows).to(dev)
        d_keep = torch.from_numpy(keep).to(dev)
        d_one = torch.full((words, 16), 3, dtype=torch.uint8, device=dev)
        d_run = torch.full((words, 16), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        h.bucket_sum_f128_dev(d_rows.data_ptr(), d_keep.data_ptr(), clients, words, d_one.data_ptr())
        for i, lo in enumerate(range(0, clients, batch)):
            m = min(batch, clients - lo)
            h.bucket_sum_f128_dev(d_rows.data_ptr() + lo * words * 16, d_keep.data_ptr() + lo, m, words, d_run.data_ptr(), accumulate=i > 0)
        h.synchronize()
        one, run = d_one.cpu().numpy(), d_run.cpu().numpy()
        assert np.array_equal(one, run)
        cols = np.unique(np.concatenate([[0, 5, words - 1], rng.integers(0, words, 300)]))
        ints = _to_ints(rows[:, cols, :])
        want = [(sum(int(ints[r, j]) for r in range(clients) if keep[r])) % P128 for j in range(len(cols))]
        got = [int(x) for x in _to_ints(one[cols])]
        assert got == want
        # NULL keep = all; zero rows = zero sketch; accumulate onto it = the sketch itself
        h.bucket_sum_f128_dev(d_rows.data_ptr(), None, clients, words, d_one.data_ptr())
        h.bucket_sum_f128_dev(d_rows.data_ptr(), None, 0, words, d_run.data_ptr())
        h.bucket_sum_f128_dev(d_rows.data_ptr(), None, clients, words, d_run.data_ptr(), accumulate=True)
        h.synchronize()
        assert np.array_equal(d_one.cpu().numpy(), d_run.cpu().numpy())
        assert [int(x) for x in _to_ints(d_one.cpu().numpy()[cols[:20]])] == [sum(int(ints[r, j]) for r in range(clients)) % P128 for j in range(20)]
    finally:
        h.close()
