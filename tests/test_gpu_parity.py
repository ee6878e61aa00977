"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: bit-exact (integer / byte work)."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import VK, blob_of, det, make_reports, server_views

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_ctx():
    v = W.new_sum_vec_256(2, 2, 5, 3)
    yield v
    v.close()


def test_field_ops_bit_exact(small_ctx):
    p = O.P128
    rnd = np.random.default_rng(11)
    edge = [0, 1, 2, p - 1, p - 2, 2**64 - 1, 2**64, 2**127, 7 * 2**66, p - 2**64, 2**128 - 2**66 * 7]
    vals = edge + [int.from_bytes(rnd.bytes(16), "little") % p for _ in range(500)]
    a = [vals[i % len(vals)] for i in range(2048)]
    b = [vals[(i * 7 + i // len(vals)) % len(vals)] for i in range(2048)]
    ab = np.frombuffer(b"".join(O.f_enc(x) for x in a), np.uint8)
    bb = np.frombuffer(b"".join(O.f_enc(x) for x in b), np.uint8)
    for op, fn in ((0, lambda x, y: (x + y) % p), (1, lambda x, y: (x - y) % p), (2, lambda x, y: x * y % p)):
        out = small_ctx.test_f128_op(op, ab, bb).tobytes()
        want = b"".join(O.f_enc(fn(x, y)) for x, y in zip(a, b))
        assert out == want, op
    out = small_ctx.test_f128_op(3, ab[:16 * 64]).tobytes()
    for i in range(64):
        got = O.f_dec(out[16 * i:16 * i + 16])
        assert got == (pow(a[i], p - 2, p) if a[i] else 0)
        assert got == O.f_inv(a[i])
    # the lazy dot product (f128.hpp): one term, and the maximum number of terms with the largest
    # limbs (non-canonical all-ones inputs included), must reduce to the same element
    ones = (2**128 - 1).to_bytes(16, "little")
    a2 = a[:2040] + [2**128 - 1] * 8
    b2 = b[:2040] + [2**128 - 1, p - 1, 2**104 - 1, 1, 0, 2**26 - 1, 2**52, 2**127]
    ab2 = np.frombuffer(b"".join(x.to_bytes(16, "little") for x in a2), np.uint8)
    bb2 = np.frombuffer(b"".join(x.to_bytes(16, "little") for x in b2), np.uint8)
    out = small_ctx.test_f128_op(4, ab2, bb2).tobytes()
    assert out == b"".join(O.f_enc(x * y % p) for x, y in zip(a2, b2))
    out = small_ctx.test_f128_op(5, ab2, bb2).tobytes()
    assert out == b"".join(O.f_enc(512 * x * y % p) for x, y in zip(a2, b2))
    out = small_ctx.test_f128_op(6, ab2).tobytes()
    assert out == b"".join(O.f_enc((2**32 - 1) * x % p) for x in a2)


def test_field64_ops_bit_exact(small_ctx):
    """Field64 = GF(2^64 - 2^32 + 1) (fp.rs:359-389): the device arithmetic against Python integers, with the
    edge values of the Goldilocks reduction (multiples of 2^32 - 1, p - 1, values whose product wraps)."""
    p = 2**64 - 2**32 + 1
    assert p == 18446744069414584321      # fp.rs:360
    rnd = np.random.default_rng(64)
    edge = [0, 1, 2, p - 1, p - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63, p - 2**32, 0xffffffff, 0xfffffffe00000001, 7]
    vals = edge + [int(x) % p for x in rnd.integers(0, 2**64, size=600, dtype=np.uint64)]
    a = np.array([vals[i % len(vals)] for i in range(4096)], np.uint64)
    b = np.array([vals[(i * 7 + i // len(vals)) % len(vals)] for i in range(4096)], np.uint64)
    for op, fn in ((0, lambda x, y: (x + y) % p), (1, lambda x, y: (x - y) % p), (2, lambda x, y: x * y % p)):
        out = small_ctx.test_f64_op(op, a, b)
        assert [int(v) for v in out] == [fn(int(x), int(y)) for x, y in zip(a, b)], op
    out = small_ctx.test_f64_op(3, a[:256])
    assert [int(v) for v in out] == [pow(int(x), p - 2, p) for x in a[:256]]


@pytest.mark.parametrize("n_elems,binder", [(1, b""), (2, b"\x01"), (10, b"x" * 16), (11, b""), (21, b"abc"), (22, b"q" * 143),
                                            (255, b"\x01"), (2048, b"\x01")])
def test_xof_expand_bit_exact(small_ctx, n_elems, binder):
    seeds = det(b"seeds", 16 * 70).reshape(70, 16)
    out = small_ctx.test_xof_expand(seeds, 2, 1, binder, n_elems)
    dst = bytes([7, 0, 0, 0, 0, 2, 0, 1])
    for i in (0, 1, 63, 64, 69):
        assert out[i].tobytes() == O.xof_expand(seeds[i].tobytes(), dst, binder, n_elems)


@pytest.mark.parametrize("path", ("default", "long_share"))
def test_sampler_rejection_branch_with_lowered_bound(small_ctx, path):
    """prng.rs:73-99 skips chunks >= p; SHAKE output gets there with probability 2^-59.  With the test
    bound at ~p/3 two of three chunks are skipped, in the XOF unit and in every XOF of a whole batch
    (client generator, oracle and GPU all sample with the same bound)."""
    bound = O.P128 // 3 + 12345
    seeds = det(b"rej", 16 * 70).reshape(70, 16)
    dst = bytes([7, 0, 0, 0, 0, 2, 0, 1])
    p = W.Prio3Params(W.SUMVEC, 16, 16, 8)
    op = O.params(W.SUMVEC, 16, 16, 8)
    vdaf = W.Prio3(p)
    if path == "long_share":
        vdaf.set_option("chain_max_input_len", 0)      # lane pairs in k_meas, every Helper on the dirty list, the general path
    try:
        O.set_test_sampler_bound(bound)
        small_ctx.set_test_sampler_bound(bound)
        vdaf.set_test_sampler_bound(bound)
        W.client_set_test_sampler_bound(bound)
        for n_elems in (1, 10, 21, 64, 300):
            out = small_ctx.test_xof_expand(seeds, 2, 1, b"\x01", n_elems)
            for i in (0, 5, 69):
                want = O.xof_expand(seeds[i].tobytes(), dst, b"\x01", n_elems)
                assert out[i].tobytes() == want
                assert all(int.from_bytes(want[16 * e:16 * e + 16], "little") < bound for e in range(n_elems))
        n = 12
        lr, hr, meas = make_reports(p, n, seed=77)
        alice, bob = server_views(lr, hr, {3})
        tags = []
        for reports in (alice, bob):
            t, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            outs = vdaf.out_shares()
            o_t, o_ids, o_status, o_outs = oracle_batch(op, reports)
            assert not status.any() and not o_status.any()
            assert np.array_equal(t, o_t) and np.array_equal(ids, o_ids) and np.array_equal(outs, o_outs)
            tags.append(t)
        for i in range(n):
            assert (tags[0][i].tobytes() != tags[1][i].tobytes()) == (i == 3)
    finally:
        O.set_test_sampler_bound(None)
        small_ctx.set_test_sampler_bound(None)
        W.client_set_test_sampler_bound(None)
        vdaf.close()


CASES = [
    ("sumvec_2_5_3", (W.SUMVEC, 2, 5, 3), 9),
    ("sumvec_ragged_16_3_7", (W.SUMVEC, 16, 3, 7), 8),
    ("sumvec_cfg1_16_128_64", (W.SUMVEC, 16, 128, 64), 70),       # configs/prio3-server-alice.json
    ("sumvec_chunk128", (W.SUMVEC, 16, 64, 128), 6),               # chunk wider than a wavefront
    ("sumvec_chunk5_len40", (W.SUMVEC, 8, 40, 5), 5),              # many calls: m = 128
    ("sumvec_chunk200", (W.SUMVEC, 4, 150, 200), 5),               # two slot blocks per wave
    ("sumvec_calls600", (W.SUMVEC, 1, 1200, 2), 4),                # lazy sums flushed mid-way
    ("sumvec_bits33", (W.SUMVEC, 33, 4, 12), 4),                   # truncate by doubling
    ("histogram_chunk130", (W.HISTOGRAM, 0, 300, 130), 5),
    ("histogram_4_2", (W.HISTOGRAM, 0, 4, 2), 8),
    ("histogram_11_3", (W.HISTOGRAM, 0, 11, 3), 8),
    ("histogram_128_64", (W.HISTOGRAM, 0, 128, 64), 20),
    ("average_64", (W.AVERAGE, 64, 0, 0), 20),                     # AVG_BITS, bin-utils/src/lib.rs:34
    ("average_8", (W.AVERAGE, 8, 0, 0), 6),
    ("average_3", (W.AVERAGE, 3, 0, 0), 6),                        # m = 4
    ("average_100", (W.AVERAGE, 100, 0, 0), 6),                    # more than 64 calls: two terms per lane
    ("sumvec_one_call", (W.SUMVEC, 8, 4, 32), 7),                  # one / two gadget calls of at most 64 slots: the one-wavefront
    ("sumvec_two_calls_ragged", (W.SUMVEC, 5, 9, 30), 7),          # query kernel, SumVec's truncate included
    ("histogram_two_calls_ragged", (W.HISTOGRAM, 0, 100, 64), 7),
]
# Every shape runs twice: the way the library would run it (short shares: all sponges in one launch, chain.hip; few gadget
# calls: the one-wavefront query kernel), and forced down the long-share pipeline (k_expand + k_meas lane pairs + k_hash +
# k_query_psum) that the large configurations take.
PATHS = ("default", "long_share")


def force_path(vdaf, path):
    if path == "long_share":
        vdaf.set_option("chain_max_input_len", 0)
        vdaf.set_option("query_tiny", 0)


def oracle_batch(op, reports):
    blob, offsets = blob_of(reports)
    return O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=8)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name,cfg,n", CASES, ids=[c[0] for c in CASES])
def test_prepare_and_aggregate_match_oracle(name, cfg, n, path):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    lr, hr, meas = make_reports(p, n, seed=len(name))
    bad = {1, n - 2}
    alice, bob = server_views(lr, hr, bad)
    # malformed reports on Alice's side (decode / structure errors of vdaf.rs:164-192)
    a_err = list(alice)
    a_err.append(alice[0][:-1])                                   # short
    a_err.append(alice[1][:30])                                   # short header
    x = bytearray(alice[0]); x[48] = 2; a_err.append(bytes(x))    # agg id
    x = bytearray(alice[0]); x[49:65] = (O.P128).to_bytes(16, "little"); a_err.append(bytes(x))   # meas >= p
    x = bytearray(alice[2]); x[-40:-32] = b"\xff" * 8; x[-48:-40] = b"\xff" * 8; a_err.append(bytes(x))  # pp >= p
    nq_off = 49 + 64 + 32
    x = bytearray(alice[1]); x[nq_off:nq_off + 8] = (p.verifier_len - 1).to_bytes(8, "big"); a_err.append(bytes(x))
    a_err.append(alice[0] + b"\x00" * 7)                          # trailing bytes are not looked at
    x = bytearray(alice[0]); x[60] ^= 0x40; a_err.append(bytes(x))  # still canonical, wrong share -> bad tag
    x = bytearray(alice[0]); x[49:65] = b"\x00" * 12 + b"\xff" * 4; a_err.append(bytes(x))   # top word all ones, < p

    vdaf = W.Prio3(p)
    force_path(vdaf, path)
    try:
        for reports in (a_err, bob):
            tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            outs = vdaf.out_shares()
            o_tags, o_ids, o_status, o_outs = oracle_batch(op, reports)
            assert np.array_equal(status, o_status), (status, o_status)
            ok = status == 0
            assert np.array_equal(tags[ok], o_tags[ok])
            assert np.array_equal(ids[ok], o_ids[ok])
            assert np.array_equal(outs[ok], o_outs[ok])
            keep = np.ones(len(reports), np.uint8)
            keep[[1, 3]] = 0
            agg = vdaf.aggregate(keep)
            assert np.array_equal(agg, O.prio3_aggregate(o_outs, p.output_len, keep & ok.astype(np.uint8)))
            assert np.array_equal(vdaf.aggregate(None), O.prio3_aggregate(o_outs, p.output_len, ok.astype(np.uint8)))
        # the two servers: tags agree exactly on honest clients and differ on tampered ones
        ta, _, sa = vdaf.prepare_encoded_keys(VK, alice)
        agg_a = vdaf.aggregate(np.array([i not in bad for i in range(n)], np.uint8))
        tb, _, sb = vdaf.prepare_encoded_keys(VK, bob)
        agg_b = vdaf.aggregate(np.array([i not in bad for i in range(n)], np.uint8))
        assert not sa.any() and not sb.any()
        for i in range(n):
            assert (ta[i].tobytes() != tb[i].tobytes()) == (i in bad)
        # unshard over the good clients gives the true aggregate
        total = W.Prio3.unshard([agg_a, agg_b])
        good = [i for i in range(n) if i not in bad]
        if circ == W.SUMVEC:
            want = [int(sum(int(meas[i][e]) for i in good)) for e in range(ln)]
        elif circ == W.HISTOGRAM:
            want = [sum(1 for i in good if int(meas[i][0]) == e) for e in range(ln)]
        else:
            want = [int(sum(int(meas[i][0]) for i in good))]
        assert total == want
    finally:
        vdaf.close()


def test_empty_and_all_malformed_batches():
    """Edge cases of the batch boundary: no reports; only malformed reports; one report."""
    p = W.Prio3Params(W.SUMVEC, 16, 16, 8)
    op = O.params(W.SUMVEC, 16, 16, 8)
    vdaf = W.Prio3(p)
    try:
        tags, ids, status = vdaf.prepare_encoded_keys(VK, [])
        assert tags.shape == (0, 16) and status.shape == (0,)
        assert vdaf.aggregate(None).tobytes() == bytes(16 * p.sizes.output_len)
        lr, hr, _ = make_reports(p, 2, seed=3)
        junk = [b"", b"\x00" * 48, lr[0].tobytes()[:100], hr[0].tobytes()[:-1]]
        tags, ids, status = vdaf.prepare_encoded_keys(VK, junk)
        o_tags, o_ids, o_status, o_outs = oracle_batch(op, junk)
        assert np.array_equal(status, o_status) and (status != 0).all()
        assert not tags.any() and not vdaf.out_shares().any()
        assert vdaf.aggregate(None).tobytes() == bytes(16 * p.sizes.output_len)
        one = [hr[1].tobytes()]
        tags, ids, status = vdaf.prepare_encoded_keys(VK, one)
        o_tags, o_ids, o_status, o_outs = oracle_batch(op, one)
        assert not status.any() and np.array_equal(tags, o_tags) and np.array_equal(ids, o_ids)
        assert np.array_equal(vdaf.aggregate(None), O.prio3_aggregate(o_outs, p.sizes.output_len, np.ones(1, np.uint8)))
    finally:
        vdaf.close()


def test_end_to_end_group_test_identifies_bad_clients():
    """C1 of BASELINE.md in miniature: Alice + Bob in one process, tags from the GPU, group testing
    on the host, aggregate over the clients that pass (server-batch-prio3/src/main.rs:65-125)."""
    p = W.Prio3Params(W.SUMVEC, 16, 16, 8)
    n, bad = 200, {3, 17, 18, 101, 199}
    lr, hr, meas = make_reports(p, n, seed=77)
    alice, bob = server_views(lr, hr, bad)
    vdaf = W.Prio3(p)
    try:
        views = []
        for reports in (alice, bob):
            tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            assert not status.any()
            tid = [int.from_bytes(ids[i].tobytes(), "little") for i in range(n)]
            order = sorted(range(n), key=lambda i: tid[i])                      # main.rs:93
            proofs = [W.ClientProofTag(tid[i], tags[i].tobytes()) for i in order]
            views.append((tid, proofs, vdaf))
            views[-1] = (tid, proofs, vdaf.out_shares())
        bad_a, bad_b, rounds = W.general_binary_split_test(views[0][1], views[1][1], VK, len(bad), 16)
        want_ids = {views[0][0][i] for i in bad}
        assert bad_a == want_ids and bad_b == want_ids and rounds > 1
        aggs = []
        for reports, (tid, _, _) in zip((alice, bob), views):
            vdaf.prepare_encoded_keys(VK, reports)
            keep = np.array([tid[i] not in bad_a for i in range(n)], np.uint8)
            aggs.append(vdaf.aggregate(keep))
        total = W.Prio3.unshard(aggs)
        assert total == [int(sum(int(meas[i][e]) for i in range(n) if i not in bad)) for e in range(16)]
    finally:
        vdaf.close()


def test_config_c1_thousand_clients_hundred_bad():
    """BASELINE configs[0] / SURVEY C1 at full size: SumVec bits 16, vec_size 128, chunk_size 64 (the reference's
    configs/prio3-server-{alice,bob}.json), 1000 clients of which 100 are tampered on Bob's copy; both servers'
    tags against the oracle, host group test, aggregate of the 900 honest clients."""
    p = W.Prio3Params(W.SUMVEC, 16, 128, 64)
    op = O.params(W.SUMVEC, 16, 128, 64)
    n = 1000
    rnd = np.random.default_rng(1)
    bad = set(int(x) for x in rnd.choice(n, 100, replace=False))
    lr, hr, meas = make_reports(p, n, seed=1234, nthreads=8)
    alice, bob = server_views(lr, hr, bad)
    vdaf = W.Prio3(p)
    try:
        views = []
        for reports in (alice, bob):
            tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            o_tags, o_ids, o_status, _ = oracle_batch(op, reports)
            assert not status.any() and np.array_equal(tags, o_tags) and np.array_equal(ids, o_ids)
            tid = [int.from_bytes(ids[i].tobytes(), "little") for i in range(n)]
            order = sorted(range(n), key=lambda i: tid[i])
            views.append((tid, [W.ClientProofTag(tid[i], tags[i].tobytes()) for i in order]))
        bad_a, bad_b, _ = W.general_binary_split_test(views[0][1], views[1][1], VK, len(bad), 16)
        want_ids = {views[0][0][i] for i in bad}
        assert bad_a == want_ids and bad_b == want_ids
        aggs = []
        for reports, (tid, _) in zip((alice, bob), views):
            vdaf.prepare_encoded_keys(VK, reports)
            aggs.append(vdaf.aggregate(np.array([tid[i] not in bad_a for i in range(n)], np.uint8)))
        good = [i for i in range(n) if i not in bad]
        assert W.Prio3.unshard(aggs) == [int(sum(int(meas[i][e]) for i in good)) for e in range(128)]
    finally:
        vdaf.close()


def test_device_pointer_path_and_large_batch_properties():
    """Inputs resident in HBM (torch tensors own the memory), sub-batching, and size-independent
    properties at a size the oracle is only sampled on: duplicated reports give identical tags,
    the aggregate of t copies is t times the aggregate of one (linearity), keep = 0 gives zero."""
    import torch
    p = W.Prio3Params(W.SUMVEC, 16, 128, 64)
    op = O.params(W.SUMVEC, 16, 128, 64)
    k, reps = 64, 40
    lr, hr, _ = make_reports(p, k, seed=3)
    alice, _ = server_views(lr, hr)
    blob1, off1 = blob_of(alice)
    blob = np.tile(blob1, reps)
    offsets = np.concatenate([off1[:-1] + np.uint64(r * int(off1[-1])) for r in range(reps)] + [np.array([reps * int(off1[-1])], np.uint64)])
    n = k * reps
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_tags = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_ids = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), -99, dtype=torch.int32, device=dev)
    d_agg = torch.zeros(16 * p.output_len, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    vdaf = W.Prio3(p)
    try:
        vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
        vdaf.aggregate_dev(None, d_agg.data_ptr())
        vdaf.synchronize()
        tags = d_tags.cpu().numpy().reshape(n, 16)
        assert not d_status.cpu().numpy().any()
        o_tags, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob1, off1, nthreads=8)
        for r in range(reps):
            assert np.array_equal(tags[r * k:(r + 1) * k], o_tags)
        assert np.array_equal(d_ids.cpu().numpy().reshape(n, 16)[:k], o_ids)
        one = O.prio3_aggregate(o_outs, p.output_len)
        want = b"".join(O.f_enc(O.f_dec(one[16 * e:16 * e + 16].tobytes()) * reps) for e in range(p.output_len))
        assert d_agg.cpu().numpy().tobytes() == want
        d_keep = torch.zeros(n, dtype=torch.uint8, device=dev)
        vdaf.aggregate_dev(d_keep.data_ptr(), d_agg.data_ptr())
        vdaf.synchronize()
        assert not d_agg.cpu().numpy().any()
    finally:
        vdaf.close()
    # the same batch in four sub-batches (workspace reused, side stream forked and joined per sub-batch)
    vdaf = W.Prio3(p)
    vdaf.set_option("max_subbatch", 700)
    try:
        d_tags2 = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_status.fill_(-99)
        vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, d_tags2.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
        vdaf.aggregate_dev(None, d_agg.data_ptr())
        vdaf.synchronize()
        assert not d_status.cpu().numpy().any()
        assert np.array_equal(d_tags2.cpu().numpy().reshape(n, 16), tags)
        assert d_agg.cpu().numpy().tobytes() == want
    finally:
        vdaf.close()


def test_sum_rows_and_hh_bucket_sum():
    import torch
    dev = torch.device("cuda:0")
    vdaf = W.new_sum_vec_256(2, 2, 5, 3)
    try:
        rng = np.random.default_rng(5)
        # partial aggregates of 8 GPUs, output_len 1024 (the all-gather + mod-p reduce step)
        rows = np.frombuffer(b"".join(O.f_enc(int.from_bytes(rng.bytes(16), "little")) for _ in range(8 * 1024)), np.uint8).reshape(8, -1)
        d_rows = torch.from_numpy(rows.copy()).to(dev)
        d_out = torch.zeros(16 * 1024, dtype=torch.uint8, device=dev)
        vdaf.sum_rows_dev(d_rows.data_ptr(), 8, 1024, None, d_out.data_ptr())
        vdaf.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), O.prio3_aggregate(rows, 1024))
        # heavy hitters: (reps folded into n) clients x 512 buckets x 258 u32, bad clients skipped
        for n, nb in ((37, 512), (5, 3), (300, 64)):
            words = nb * 258
            buckets = rng.integers(0, 2**32, size=(n, words), dtype=np.uint32)
            keep = (rng.random(n) > 0.2).astype(np.uint8)
            d_b = torch.from_numpy(buckets.view(np.int32)).to(dev)
            d_k = torch.from_numpy(keep).to(dev)
            d_acc = torch.zeros(words, dtype=torch.int32, device=dev)
            for kp, dk in ((keep, d_k.data_ptr()), (None, None)):
                vdaf.hh_bucket_sum_u32_dev(d_b.data_ptr(), dk, n, words, d_acc.data_ptr())
                vdaf.synchronize()
                assert np.array_equal(d_acc.cpu().numpy().view(np.uint32), O.hh_bucket_sum_u32(buckets, kp))
            # streaming (server-hh/src/main.rs:113-216, `batch_size` slices): the accumulator carries over
            step = max(1, n // 3)
            for lo in range(0, n, step):
                m = min(step, n - lo)
                vdaf.hh_bucket_sum_u32_dev(d_b[lo:lo + m].data_ptr(), d_k[lo:lo + m].data_ptr(), m, words, d_acc.data_ptr(),
                                           accumulate=lo > 0)
            vdaf.synchronize()
            assert np.array_equal(d_acc.cpu().numpy().view(np.uint32), O.hh_bucket_sum_u32(buckets, keep))
    finally:
        vdaf.close()


def test_prepare_matches_oracle_on_random_shapes():
    """Seeded random circuit shapes (ragged last chunks, chunk > input_len, odd bit widths): tags, testing ids, status, out
    shares and the masked aggregate of both servers against the oracle, bit for bit, with one tampered client per shape."""
    rng = np.random.default_rng(7020240807)
    shapes = []
    for _ in range(12):
        bits, ln = int(rng.integers(1, 17)), int(rng.integers(1, 41))
        shapes.append((W.SUMVEC, bits, ln, int(rng.integers(1, bits * ln + 3))))
    for _ in range(6):
        ln = int(rng.integers(2, 61))
        shapes.append((W.HISTOGRAM, 0, ln, int(rng.integers(1, ln + 3))))
    for bits in (1, 7, 33, 64, 65, 127):
        shapes.append((W.AVERAGE, bits, 0, 0))
    for k, (circ, bits, ln, ch) in enumerate(shapes):
        p = W.Prio3Params(circ, bits, ln, ch)
        op = O.params(circ, bits, ln, ch)
        n = 6
        lr, hr, _ = make_reports(p, n, seed=500 + k, nthreads=2)
        keep = np.ones(n, np.uint8)
        keep[2] = 0
        views = server_views(lr, hr, {2})
        want = [oracle_batch(op, reports) for reports in views]
        for path in PATHS:
            vdaf = W.Prio3(p)
            force_path(vdaf, path)
            try:
                for reports, (o_tags, o_ids, o_status, o_outs) in zip(views, want):
                    tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
                    assert np.array_equal(status, o_status) and not status.any(), (circ, bits, ln, ch, path)
                    assert np.array_equal(tags, o_tags) and np.array_equal(ids, o_ids), (circ, bits, ln, ch, path)
                    assert np.array_equal(np.asarray(vdaf.out_shares()).reshape(n, -1), np.asarray(o_outs).reshape(n, -1)), (circ, bits, ln, ch, path)
                    assert np.array_equal(vdaf.aggregate(keep), O.prio3_aggregate(o_outs, p.sizes.output_len, keep)), (circ, bits, ln, ch, path)
            finally:
                vdaf.close()


@pytest.mark.parametrize("cfg", [(W.SUMVEC, 16, 16, 8), (W.SUMVEC, 3, 11, 7), (W.HISTOGRAM, 0, 11, 3), (W.AVERAGE, 9, 0, 0)],
                         ids=["sumvec_16_16_8", "sumvec_ragged", "histogram_11_3", "average_9"])
def test_mutated_reports_differential_fuzz(cfg):
    """Seeded mutations of valid reports -- byte flips anywhere, truncations, extensions, elements overwritten with values
    at and above p, the all-ones word, length prefixes off by one -- through the GPU and the oracle: same status for every
    report, and the same tag, testing id and out share wherever the report still decodes."""
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    n = 48
    lr, hr, _ = make_reports(p, n, seed=4242 + ln, nthreads=4)
    alice, bob = server_views(lr, hr)
    rng = np.random.default_rng(99 + bits + ln)
    special = [O.P128, O.P128 - 1, O.P128 + 1, (1 << 128) - 1, (0xffffffff << 96), 0]
    vdaf = W.Prio3(p)
    try:
        for reports in (alice, bob):
            mutated = []
            for r in reports:
                x = bytearray(r)
                kind = int(rng.integers(0, 7))
                if kind == 0:                                   # one byte flipped anywhere
                    x[int(rng.integers(0, len(x)))] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:                                 # truncated
                    x = x[: int(rng.integers(0, len(x)))]
                elif kind == 2:                                 # extended (trailing bytes are ignored)
                    x += bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
                elif kind == 3 and len(x) >= 49 + 16:           # a 16-byte window overwritten with a boundary value
                    off = 49 + 16 * int(rng.integers(0, (len(x) - 49) // 16))
                    x[off:off + 16] = special[int(rng.integers(0, len(special)))].to_bytes(16, "little")
                elif kind == 4:                                 # a header byte replaced
                    x[int(rng.integers(0, 49))] = int(rng.integers(0, 256))
                elif kind == 5:                                 # several byte flips in the tail (query-rand parts, lengths, public proofs)
                    for _ in range(3):
                        x[len(x) - 1 - int(rng.integers(0, min(len(x), 200)))] ^= 0x80
                mutated.append(bytes(x))
            tags, ids, status = vdaf.prepare_encoded_keys(VK, mutated)
            outs = vdaf.out_shares()
            o_tags, o_ids, o_status, o_outs = oracle_batch(op, mutated)
            assert np.array_equal(status, o_status), (list(status), list(o_status))
            ok = status == 0
            assert ok.any() and (~ok).any()
            assert np.array_equal(tags[ok], o_tags[ok]) and np.array_equal(ids[ok], o_ids[ok])
            assert np.array_equal(np.asarray(outs)[ok], np.asarray(o_outs)[ok])
            assert np.array_equal(vdaf.aggregate(None), O.prio3_aggregate(o_outs, p.output_len, ok.astype(np.uint8)))
    finally:
        vdaf.close()


@pytest.mark.parametrize("length", [1, 3, 100, 255, 256, 300, 5000])
def test_masked_row_sums_every_row_shape(length):
    """wh_f128_sum_rows_dev (the aggregation kernels behind wh_prio3_aggregate*, the multi-GPU combine and the Field128 sketch sums)
    against Python integers for the three launch shapes -- short rows swept side by side (Average's output_len = 1), rows of at
    least 256 elements, long rows by columns -- with a keep mask, non-canonical and all-ones inputs, and row counts around the
    slab sizes."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + length)
    vdaf = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    try:
        for n in (1, 2, 257, 4099):
            rows = rng.integers(0, 1 << 32, (n, length, 4), dtype=np.uint64).astype(np.uint32)
            rows[:, :, 3] >>= 1                                  # below p ...
            rows[0, 0, :] = 0xffffffff                           # ... except an all-ones element (reduced mod p, as documented)
            keep = (rng.integers(0, 4, n) != 0).astype(np.uint8)
            keep[0] = 1
            d_rows = torch.from_numpy(rows.view(np.int32).copy()).to(dev)
            d_keep = torch.from_numpy(keep).to(dev)
            d_out = torch.zeros(length * 4, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            vdaf.sum_rows_dev(d_rows.data_ptr(), n, length, d_keep.data_ptr(), d_out.data_ptr())
            vdaf.synchronize()
            got = d_out.cpu().numpy().view(np.uint32).reshape(length, 4).astype(object)
            vals = rows.astype(object)
            ints = vals[:, :, 0] + (vals[:, :, 1] << 32) + (vals[:, :, 2] << 64) + (vals[:, :, 3] << 96)
            want = [int(sum(ints[r, e] for r in range(n) if keep[r]) % O.P128) for e in range(length)]
            assert [int(g[0]) | int(g[1]) << 32 | int(g[2]) << 64 | int(g[3]) << 96 for g in got] == want
    finally:
        vdaf.close()
