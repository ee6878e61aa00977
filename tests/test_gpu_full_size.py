"""BASELINE.json's configurations at their full per-GPU sizes, every client distinct (made on the GPU by
wh_client_shard_many_gpu, itself byte-identical to the CPU client and the oracle in test_gpu_shard.py), through size-independent
properties of the protocol (vdaf.rs:889-897, server-batch-prio3/src/main.rs:65-125): Alice's and Bob's tags and testing ids
agree for every honest client and differ for exactly the tampered ones, the two aggregates over the kept clients reconstruct
the column sums of the measurements, and a sample of reports agrees with the oracle bit for bit."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import whisper_amd as W
from common import VK

pytestmark = pytest.mark.gpu

CASES = [
    ("sumvec_large_125k", (W.SUMVEC, 16, 1024, 128), 125_000),     # configs[2]: 1M clients over 8 GPUs
    ("sumvec_cfg1_500k", (W.SUMVEC, 16, 128, 64), 500_000),        # configs[0] sizes, bench.py --workload sumvec_cfg1
    ("histogram_100k", (W.HISTOGRAM, 0, 128, 64), 100_000),        # configs[1]
]


def views(d_l, d_h, n, L, H, dev):
    """Alice sees the Leader report of even clients and the Helper report of odd ones, Bob the others (main.rs:114-115)."""
    out = []
    pairs = n // 2
    assert n % 2 == 0
    idx = torch.arange(n + 1, device=dev, dtype=torch.int64)
    for first, second, a, b in ((d_l.view(n, L)[0::2], d_h.view(n, H)[1::2], L, H), (d_h.view(n, H)[0::2], d_l.view(n, L)[1::2], H, L)):
        blob = torch.empty(pairs * (a + b) + 1, dtype=torch.uint8, device=dev)
        v = blob[: pairs * (a + b)].view(pairs, a + b)
        v[:, :a] = first
        v[:, a:] = second
        out.append((blob, ((idx // 2) * (a + b) + (idx % 2) * a).contiguous()))
    return out


@pytest.mark.parametrize("name,cfg,n", CASES, ids=[c[0] for c in CASES])
def test_full_size_two_servers(name, cfg, n):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    s = p.sizes
    L, H = s.leader_report_len, s.helper_report_len
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(42)
    meas = rng.integers(0, 1 << bits, (n, ln), dtype=np.uint64) if circ == W.SUMVEC else rng.integers(0, ln, (n, 1), dtype=np.uint64)
    nonces = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    rands = rng.integers(0, 256, (n, 144), dtype=np.uint8)
    bad = np.sort(rng.choice(n, n // 1000, replace=False))
    vdaf = W.Prio3(p)
    try:
        d_l = torch.empty(n * L, dtype=torch.uint8, device=dev)
        d_h = torch.empty(n * H, dtype=torch.uint8, device=dev)
        vdaf.shard_batched_many_dev(meas, nonces, rands, 7 << 32, d_l.data_ptr(), d_h.data_ptr())
        (blob_a, off_a), (blob_b, off_b) = views(d_l, d_h, n, L, H, dev)
        del d_l, d_h
        torch.cuda.empty_cache()
        # the meta-client's malicious clients: Bob's copy differs in one nonce byte (main.rs:146-149)
        pos = off_b[torch.from_numpy(bad + 1).to(dev)] - 16
        blob_b[pos] += 1
        keep = np.ones(n, np.uint8)
        keep[bad] = 0
        d_keep = torch.from_numpy(keep).to(dev)
        res = []
        for blob, off in ((blob_a, off_a), (blob_b, off_b)):
            d_tags = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_ids = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            d_status = torch.full((n,), -99, dtype=torch.int32, device=dev)
            d_agg = torch.zeros(16 * s.output_len, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            vdaf.prepare_batch_dev(VK, blob.data_ptr(), off.data_ptr(), n, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            vdaf.aggregate_dev(d_keep.data_ptr(), d_agg.data_ptr())
            vdaf.synchronize()
            res.append((d_tags.view(n, 16).cpu().numpy(), d_ids.view(n, 16).cpu().numpy(), d_status.cpu().numpy(), d_agg.cpu().numpy().tobytes()))
        (tags_a, ids_a, st_a, agg_a), (tags_b, ids_b, st_b, agg_b) = res
        # a second pass over Alice's batch gives the same bytes (no dependence on scheduling or on workspace left-overs)
        d_tags = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
        d_ids = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
        d_status = torch.full((n,), -99, dtype=torch.int32, device=dev)
        d_agg = torch.zeros(16 * s.output_len, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        vdaf.prepare_batch_dev(VK, blob_a.data_ptr(), off_a.data_ptr(), n, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
        vdaf.aggregate_dev(d_keep.data_ptr(), d_agg.data_ptr())
        vdaf.synchronize()
        assert np.array_equal(d_tags.view(n, 16).cpu().numpy(), tags_a) and np.array_equal(d_ids.view(n, 16).cpu().numpy(), ids_a)
        assert d_agg.cpu().numpy().tobytes() == agg_a
        assert not st_a.any() and not st_b.any()
        assert np.array_equal(ids_a, ids_b)
        assert len(np.unique(ids_a.view(np.uint64).reshape(n, 2), axis=0)) == n          # distinct clients, distinct testing ids
        differ = np.flatnonzero((tags_a != tags_b).any(axis=1))
        assert np.array_equal(differ, bad)
        # the aggregate over the kept clients is the column sum of their measurements
        total = W.Prio3.unshard([agg_a, agg_b])
        if circ == W.SUMVEC:
            want = [int(x) for x in meas[keep == 1].sum(axis=0, dtype=np.uint64)]
        else:
            want = [int(x) for x in np.bincount(meas[keep == 1, 0].astype(np.int64), minlength=ln)]
        assert total == want
        # a sample against the oracle, both servers: 2 048 seeded-random clients per view plus every tampered one, their
        # neighbours and the batch's ends -- tags, testing ids and status; and for the view whose out shares the context
        # still holds (Alice's second pass) the out shares of the same sample, byte for byte
        srng = np.random.default_rng(4242)
        sample = np.unique(np.concatenate([srng.choice(n, 2048, replace=False), bad, np.minimum(bad + 1, n - 1), [0, 1, n - 2, n - 1]])).astype(np.int64)
        outs_a = vdaf.out_shares()[sample]
        for view, (blob, off, tags, ids, st) in enumerate(((blob_a, off_a, tags_a, ids_a, st_a), (blob_b, off_b, tags_b, ids_b, st_b))):
            off_h = off.cpu().numpy()
            reps = [blob[int(off_h[i]):int(off_h[i + 1])].cpu().numpy() for i in sample]
            o_off = np.zeros(len(reps) + 1, np.uint64)
            o_off[1:] = np.cumsum([len(r) for r in reps])
            o_tags, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, np.concatenate(reps), o_off, nthreads=16)
            assert np.array_equal(o_status, st[sample])
            assert np.array_equal(o_tags, tags[sample]) and np.array_equal(o_ids, ids[sample])
            if view == 0:
                assert np.array_equal(np.asarray(o_outs).reshape(len(sample), -1), outs_a)
            del reps
    finally:
        vdaf.close()
