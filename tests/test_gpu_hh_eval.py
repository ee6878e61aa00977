"""GPU parity of the heavy-hitter expansion (wh_hh_eval_and_check_batch, vdpf.hip) against the oracle, through the C ABI.

Bar: bit-exact -- bucket vectors (u32), tags (64 hex characters), testing ids, status.  The oracle's VALUES are
"parity unpinned" w.r.t. the Rust binary (no known-answer vector in the reference); its primitives and properties
are pinned in tests/test_oracle_vdpf.py.
"""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu


def _bucket(rng):
    bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32)
    bk[257] = 1
    return bk


def _make(depth, n, seed, bad=()):
    """n clients: (alice keys [n, len], bob keys [n, len], alphas, buckets); clients in `bad` get two different chains."""
    from whisper_amd import hh
    rng = np.random.default_rng(seed)
    ln = hh.key_chain_len(depth)
    A = np.zeros((n, ln), np.uint8); B = np.zeros((n, ln), np.uint8)
    alphas = rng.integers(0, 1 << depth, n)
    bks = np.stack([_bucket(rng) for _ in range(n)])
    for i in range(n):
        s = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        A[i], B[i] = hh.gen_key_chain(depth, 1000 + i, int(alphas[i]), bks[i], s)
        if i in bad:
            s2 = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
            _, B[i] = hh.gen_key_chain(depth, 1000 + i, int((alphas[i] + 1) % (1 << depth)), _bucket(rng), s2)
    return A, B, alphas, bks


def _check_against_oracle(h, vk, keys, depth):
    tags, ids, status = h.eval_and_check(vk, keys)
    dat = h.buckets()
    odat, otags, oids, ostatus = o.hh_eval_and_check_batch(depth, vk, keys, nthreads=4)
    assert (status == ostatus).all()
    assert [t.encode() if s == 0 else bytes(64) for t, s in zip(tags, status)] == [bytes(t) for t in otags]
    assert ids == [int.from_bytes(bytes(i), "little") for i in oids]
    assert (dat == odat).all()
    return tags, ids, status, dat


@pytest.mark.parametrize("depth,n", [(1, 3), (2, 5), (5, 9), (9, 6), (11, 2)])
def test_eval_and_check_bit_exact(depth, n):
    from whisper_amd import hh
    vk = bytes(range(7, 23))
    A, B, alphas, bks = _make(depth, n, 100 + depth)
    h = hh.HeavyHitters(1 << depth)
    ta, ia, sa, da = _check_against_oracle(h, vk, A, depth)
    tb, ib, sb, db = _check_against_oracle(h, vk, B, depth)
    # the reference's properties on the GPU outputs (protocol.rs:219-230)
    assert ta == tb and ia == ib and not sa.any() and not sb.any()
    s = da + db
    for i in range(n):
        assert (s[i, alphas[i]] == bks[i]).all()
        s[i, alphas[i]] = 0
    assert not s.any()
    h.close()


def test_bad_clients_and_aggregate():
    """server-hh's flow: tags differ exactly for the malformed clients; the sum over the kept ones equals the oracle's
    and reconstructs the true sketch of the good clients."""
    from whisper_amd import hh
    depth, n, bad = 6, 40, {3, 17, 31}
    vk = bytes([9] * 16)
    A, B, alphas, bks = _make(depth, n, 77, bad)
    ha, hb = hh.HeavyHitters(1 << depth), hh.HeavyHitters(1 << depth)
    ta, _, sa = ha.eval_and_check(vk, A)
    tb, _, sb = hb.eval_and_check(vk, B)
    assert not sa.any() and not sb.any()
    assert {i for i in range(n) if ta[i] != tb[i]} == bad
    keep = np.array([i not in bad for i in range(n)], np.uint8)
    acc_a, acc_b = ha.aggregate_hhs(keep), hb.aggregate_hhs(keep)
    assert (acc_a == o.hh_bucket_sum_u32(ha.buckets(), keep)).all()
    want = np.zeros((1 << depth, 258), np.uint32)
    for i in range(n):
        if i not in bad:
            want[alphas[i]] += bks[i]
    assert ((acc_a + acc_b) == want).all()
    assert (ha.aggregate_hhs(None) == o.hh_bucket_sum_u32(ha.buckets())).all()
    ha.close(); hb.close()


def test_key_format_errors_and_padding():
    """Key chains the reference would abort on get WH_ERR_KEY_FORMAT, a zero tag and zero buckets; a stride larger than
    the key chain (trailing bytes) is accepted; an empty batch is a no-op."""
    from whisper_amd import hh
    depth, n = 3, 8
    vk = bytes(16)
    A, _, _, _ = _make(depth, n, 5)
    ln = A.shape[1]
    K = np.zeros((n, ln + 13), np.uint8)
    K[:, :ln] = A
    K[:, ln:] = 0xee
    K[1, 16] = 2                       # key_id is not a bool
    K[2, 33] = depth + 1               # correction-word count
    K[4, 41 + 18 * depth + 64] = 1     # Bucket.data length
    K[5, ln - 111 + 17] = 2            # a support key with two levels
    K[6, ln - 111 * 200] = 7           # a support key's key_id
    h = hh.HeavyHitters(1 << depth)
    tags, ids, status, dat = _check_against_oracle(h, vk, K, depth)
    assert [int(s) for s in status] == [0, -9, -9, 0, -9, -9, -9, 0]
    assert not dat[1].any() and tags[1] == ""
    t, i, s = h.eval_and_check(vk, np.zeros((0, ln), np.uint8))
    assert t == [] and i == [] and s.size == 0
    h.close()


def test_device_entry_overwrites_stale_outputs():
    """Caller-owned output buffers that held something else before the call (a reused allocation): every tag, testing id,
    status and bucket vector of the batch is written, for well-formed and malformed key chains alike, and equals the oracle's."""
    import torch
    from whisper_amd import hh
    depth, n = 3, 8
    vk = bytes(range(3, 19))
    A, _, _, _ = _make(depth, n, 17)
    ln = A.shape[1]
    K = A.copy()
    K[1, 16] = 2                       # key_id is not a bool
    K[4, 41 + 18 * depth + 64] = 1     # Bucket.data length
    K[6, ln - 111 * 200] = 7           # a support key's key_id
    h = hh.HeavyHitters(1 << depth)
    dev = torch.device("cuda:0")
    d_keys = torch.from_numpy(K).to(dev)
    d_dat = torch.full((n, 1 << depth, 258), 0x5a5a5a5a, dtype=torch.int32, device=dev)
    d_tags = torch.full((n, 64), 0xa5, dtype=torch.uint8, device=dev)
    d_ids = torch.full((n, 16), 0x3c, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), 77, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    h.eval_and_check_dev(vk, d_keys.data_ptr(), ln, n, d_dat.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
    h.synchronize()
    odat, otags, oids, ostatus = o.hh_eval_and_check_batch(depth, vk, K, nthreads=4)
    assert [int(x) for x in ostatus] == [0, -9, 0, 0, -9, 0, -9, 0]
    assert (d_status.cpu().numpy() == ostatus).all()
    assert (d_tags.cpu().numpy() == otags).all()
    assert (d_ids.cpu().numpy() == oids).all()
    assert (d_dat.cpu().numpy().view(np.uint32) == odat).all()
    h.close()


def test_sub_batches_and_device_entry(monkeypatch):
    """The device-pointer entry point with caller-owned buffers; WH_HH_SUBBATCH makes the 300 key chains take five
    sub-batches of scratch (the last one partial), more workgroups than CUs walk the persistent loop, and the sum runs in
    its accumulate form."""
    import torch
    from whisper_amd import hh
    monkeypatch.setenv("WH_HH_SUBBATCH", "64")
    depth, n = 4, 300
    vk = bytes(range(16))
    rng = np.random.default_rng(1)
    A0, _, _, _ = _make(depth, 12, 9)
    keys = A0[rng.integers(0, 12, n)].copy()
    keys[:, :8] = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)   # fresh client ids
    h = hh.HeavyHitters(1 << depth)
    dev = torch.device("cuda:0")
    d_keys = torch.from_numpy(keys).to(dev)
    d_dat = torch.zeros((n, 1 << depth, 258), dtype=torch.int32, device=dev)
    d_tags = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    h.eval_and_check_dev(vk, d_keys.data_ptr(), keys.shape[1], n, d_dat.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
    h.synchronize()
    odat, otags, oids, ostatus = o.hh_eval_and_check_batch(depth, vk, keys, nthreads=8)
    assert (d_status.cpu().numpy() == ostatus).all()
    assert (d_tags.cpu().numpy() == otags).all() and (d_ids.cpu().numpy() == oids).all()
    assert (d_dat.cpu().numpy().view(np.uint32) == odat).all()
    d_acc = torch.zeros((1 << depth, 258), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    h.aggregate_hhs_dev(None, d_acc.data_ptr(), accumulate=False)
    h.aggregate_hhs_dev(None, d_acc.data_ptr(), accumulate=True)
    h.synchronize()
    want = o.hh_bucket_sum_u32(odat)
    assert (d_acc.cpu().numpy().view(np.uint32) == (want + want)).all()
    h.close()


def test_server_hh_flow_with_group_test():
    """batch_collect_and_aggregate_per_run + aggregate_hhs (server-hh/src/main.rs:53-110, 255-282) for Alice and Bob:
    expansion on the GPU, sort by testing id, general_binary_split_test over the String tags with the 32-byte (SHA-256)
    batch proofs the heavy-hitter server asks for (main.rs:61-68), masked sketch sums, and the two servers' sketches
    inserted into each other (main.rs:386-391) give the honest clients' buckets."""
    from whisper_amd import hh
    from whisper_amd.grouptest import ClientProofTag, general_binary_split_test
    depth, n, bad = 5, 60, {0, 13, 14, 41, 59}
    vk = bytes(range(100, 116))
    A, B, alphas, bks = _make(depth, n, 2024, bad)
    servers = []
    for keys in (A, B):
        h = hh.HeavyHitters(1 << depth)
        tags, ids, status = h.eval_and_check(vk, keys)
        assert not status.any()
        order = sorted(range(n), key=lambda i: ids[i])                       # par_sort_by testing_id
        proofs = [ClientProofTag(ids[i], tags[i].encode()) for i in order]
        servers.append((h, ids, proofs))
    bad_a, bad_b, rounds = general_binary_split_test(servers[0][2], servers[1][2], vk, len(bad), tag_size=32)
    ids = servers[0][1]
    assert servers[1][1] == ids                                              # same PRF key, same client ids
    assert bad_a == bad_b == {ids[i] for i in bad} and rounds > 1
    sketches = []
    for (h, ids_, _), bad_ids in zip(servers, (bad_a, bad_b)):
        keep = np.array([ids_[i] not in bad_ids for i in range(n)], np.uint8)
        sketches.append(h.aggregate_hhs(keep))
        h.close()
    want = np.zeros((1 << depth, 258), np.uint32)
    for i in range(n):
        if i not in bad:
            want[alphas[i]] += bks[i]
    assert ((sketches[0] + sketches[1]) == want).all()


def test_many_key_chains_per_cu_and_mixed_validity():
    """More key chains than CUs, so every persistent workgroup takes several and its producer wavefront prepares key
    chain k + G while k is converted; malformed ones in between must not disturb their neighbours."""
    from whisper_amd import hh
    depth, n = 3, 700
    vk = bytes(range(50, 66))
    rng = np.random.default_rng(99)
    A0, _, _, _ = _make(depth, 16, 321)
    keys = A0[rng.integers(0, 16, n)].copy()
    keys[:, :8] = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    broken = rng.choice(n, 40, replace=False)
    for k, i in enumerate(broken):
        at = (16, 33, keys.shape[1] - 111 + 17, 41 + 18 * depth + 64)[k % 4]
        keys[i, at] = 9
    h = hh.HeavyHitters(1 << depth)
    tags, ids, status, dat = _check_against_oracle(h, vk, keys, depth)
    assert set(np.nonzero(status)[0]) == set(int(i) for i in broken)
    h.close()


@pytest.mark.parametrize("depth", [1, 4, 9])
def test_mutated_key_chains_differential_fuzz(depth):
    """Seeded mutations of valid key chains -- a byte flipped anywhere (seeds, correction words, control bits, bincode
    lengths, bool bytes of the support keys), windows overwritten with 0x00 / 0xff -- through the GPU and the oracle: the
    same status for every key chain and the same proof string, testing id and bucket vectors, bit for bit."""
    from whisper_amd import hh
    vk = bytes(range(60, 76))
    n = 40
    A, B, _, _ = _make(depth, n, seed=900 + depth)
    rng = np.random.default_rng(31 * depth)
    ln = A.shape[1]
    h = hh.HeavyHitters(1 << depth)
    try:
        for keys in (A, B):
            m = keys.copy()
            for i in range(n):
                kind = int(rng.integers(0, 5))
                if kind == 0:
                    m[i, int(rng.integers(0, ln))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                elif kind == 1:                               # in the header of the main key: key id, root seed, lengths
                    m[i, int(rng.integers(16, 49))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                elif kind == 2:
                    off = int(rng.integers(0, ln - 8))
                    m[i, off:off + 8] = 0xff if rng.integers(0, 2) else 0
                elif kind == 3:                               # inside the support keys
                    m[i, int(rng.integers(ln - 257 * 111, ln))] ^= np.uint8(1 << int(rng.integers(0, 8)))
            tags, ids, status, dat = _check_against_oracle(h, vk, m, depth)
            assert (status == 0).any()
    finally:
        h.close()


@pytest.mark.parametrize("depth,clients,reps,subbatch", [(5, 23, 3, None), (9, 12, 2, "7"), (9, 12, 2, "1"), (11, 3, 2, None), (1, 5, 1, None)])
def test_sketch_sum_and_subtract_match_materialised_sum(monkeypatch, depth, clients, reps, subbatch):
    """wh_hh_eval_and_sketch_batch_dev (the caller holds no bucket vectors: sub-batches are summed into the sketches on a side
    stream) + wh_hh_sketch_subtract_dev ("sum all, subtract the rejected": re-expansion with atomic subtraction) against the
    materialised path: same tags / ids / status, and per repetition the same sketch as aggregate_hhs over the kept key chains --
    with malformed key chains, sub-batches that straddle repetitions, and accumulation into a running sketch."""
    import torch
    from whisper_amd import hh
    if subbatch:
        monkeypatch.setenv("WH_HH_SUBBATCH", subbatch)
    vk = bytes(range(40, 56))
    n = clients * reps
    A, _, _, _ = _make(depth, n, 500 + depth)
    A = A.copy()
    A[1, 17 + 16] ^= 0xff            # key chain 1: broken bincode length -> WH_ERR_KEY_FORMAT, contributes nothing
    A[n - 1, 16] = 9                 # the last one: bad key id
    N, words = 1 << depth, (1 << depth) * 258
    dev = torch.device("cuda:0")
    h = hh.HeavyHitters(N)
    try:
        tags0, ids0, status0 = h.eval_and_check(vk, A)
        dat = h.buckets()
        assert status0[1] != 0 and status0[n - 1] != 0 and int((status0 != 0).sum()) == 2
        rejected = sorted({0, 1, clients // 2, n - 2, n - 1} & set(range(n)))        # the group test's verdict (by index)
        keep = np.ones(n, np.uint8)
        keep[rejected] = 0
        want_all = np.stack([dat[r * clients:(r + 1) * clients].sum(axis=0, dtype=np.uint32) for r in range(reps)])
        want_kept = np.stack([(dat[r * clients:(r + 1) * clients] * keep[r * clients:(r + 1) * clients, None, None]).sum(axis=0, dtype=np.uint32)
                              for r in range(reps)])
        d_keys = torch.from_numpy(A).to(dev)
        d_sk = torch.full((reps, N, 258), 12345, dtype=torch.int32, device=dev)
        d_tags = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
        d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
        d_status = torch.full((n,), 99, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for rnd in range(2):                                   # twice: the second call overwrites, it does not accumulate
            h.eval_and_sketch_dev(vk, d_keys.data_ptr(), A.shape[1], n, clients, d_sk.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            h.synchronize()
            assert np.array_equal(d_sk.cpu().numpy().view(np.uint32), want_all), rnd
        assert np.array_equal(d_status.cpu().numpy(), status0)
        got_tags = d_tags.cpu().numpy()
        for i in range(n):
            assert bytes(got_tags[i]) == (tags0[i].encode() if status0[i] == 0 else bytes(64))
        assert [int.from_bytes(bytes(r), "little") for r in d_ids.cpu().numpy()] == ids0
        h.sketch_subtract_dev(vk, d_keys.data_ptr(), A.shape[1], rejected, clients, d_sk.data_ptr())
        h.synchronize()
        assert np.array_equal(d_sk.cpu().numpy().view(np.uint32), want_kept)
        # streaming: a second slice accumulates into the running sketch (CountSketch::insert)
        h.eval_and_sketch_dev(vk, d_keys.data_ptr(), A.shape[1], n, clients, d_sk.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr(),
                              accumulate=True)
        h.synchronize()
        assert np.array_equal(d_sk.cpu().numpy().view(np.uint32), want_kept + want_all)
        # nothing was materialised by the fused call
        with pytest.raises(Exception):
            h.aggregate_hhs(None)
    finally:
        h.close()
