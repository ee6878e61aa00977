"""The group test's batch proofs on the GPU (common/src/grouptest.rs:43-99, 182-208, 306-318): wh_grouptest_batch_proofs_dev /
wh_grouptest_split_proofs_dev against hashlib's SHAKE128 and SHA-256 over the same slices, bit for bit, and the whole
general_binary_split_test with the GPU hasher against the host one.  hashlib is an independent implementation of both
hashes; the slice rule (`testing_id > upper` ends the proof) follows the reference's loop."""
import hashlib

import numpy as np
import pytest
import torch

import whisper_amd as W
from whisper_amd import hh
from whisper_amd.grouptest import ClientProofTag, DeviceHasher, HostHasher, U128_MAX, general_binary_split_test

pytestmark = pytest.mark.gpu
VK = bytes(range(100, 116))


def make_list(rng, n, tag_bytes, dup=True):
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    if n >= 16:
        ids[3] = 0
        ids[4] = 255
        if dup:
            ids[7:11] = ids[6]
    v = ids.view(np.uint64).reshape(-1, 2)
    ids = ids[np.lexsort((v[:, 0], v[:, 1]))]
    tags = rng.integers(0, 256, (n, tag_bytes), dtype=np.uint8)
    proofs = [ClientProofTag(int.from_bytes(ids[i].tobytes(), "little"), tags[i].tobytes()) for i in range(n)]
    return ids, tags, proofs


class OnDevice:
    def __init__(self, ctx, ids, tags, proofs, tag_size):
        dev = torch.device("cuda:0")
        self.d_ids = torch.from_numpy(ids.copy()).to(dev) if len(ids) else torch.zeros(16, dtype=torch.uint8, device=dev)
        self.d_tags = torch.from_numpy(tags.copy()).to(dev) if len(tags) else torch.zeros(16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        self.hasher = DeviceHasher(ctx, self.d_ids.data_ptr(), self.d_tags.data_ptr(), len(ids), tags.shape[1], VK, tag_size)
        self.host = HostHasher(proofs, VK, tag_size)


@pytest.fixture(scope="module")
def ctx():
    c = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    yield c
    c.close()


@pytest.mark.parametrize("tag_bytes,tag_size", [(16, 16), (64, 32), (8, 32), (24, 32), (40, 16), (8, 16)])
def test_slices_of_every_length(ctx, tag_bytes, tag_size):
    """Slices of 0 .. 200 items from the front (every block boundary of both hashes, every padding case) and a few long ones."""
    rng = np.random.default_rng(tag_bytes * 100 + tag_size)
    n = 6000
    ids, tags, proofs = make_list(rng, n, tag_bytes)
    o = OnDevice(ctx, ids, tags, proofs, tag_size)
    jobs = [(5, 5 + c, U128_MAX) for c in range(0, 201)]
    jobs += [(0, n, U128_MAX), (17, 4000, U128_MAX), (n, n, U128_MAX), (n - 1, n, 0)]
    got, want = o.hasher.batch(jobs), o.host.batch(jobs)
    assert got == want
    # the digests really are the hashes of key || tags
    h = hashlib.shake_128(VK + tags[5:5 + 21].tobytes()).digest(16) if tag_size == 16 else hashlib.sha256(VK + tags[5:5 + 21].tobytes()).digest()
    assert got[21] == (h, 21)


@pytest.mark.parametrize("tag_bytes,tag_size", [(16, 16), (64, 32)])
def test_upper_bound_ends_the_slice(ctx, tag_bytes, tag_size):
    rng = np.random.default_rng(7 + tag_size)
    n = 3000
    ids, tags, proofs = make_list(rng, n, tag_bytes)
    o = OnDevice(ctx, ids, tags, proofs, tag_size)
    jobs = []
    for _ in range(300):
        s = int(rng.integers(0, n))
        e = int(rng.integers(s, n + 1))
        kind = rng.integers(0, 4)
        if kind == 0:
            up = int.from_bytes(rng.bytes(16), "little")
        elif kind == 1:
            up = proofs[int(rng.integers(0, n))].testing_id                      # an id of the list: `<=` keeps it (and its duplicates)
        elif kind == 2:
            up = max(proofs[int(rng.integers(0, n))].testing_id - 1, 0)
        else:
            up = [0, U128_MAX, 1 << 64, (1 << 64) - 1][int(rng.integers(0, 4))]
        jobs.append((s, e, up))
    jobs.append((6, 12, proofs[6].testing_id))      # the run of equal ids
    assert o.hasher.batch(jobs) == o.host.batch(jobs)


@pytest.mark.parametrize("d", [2, 3, 7, 64, 1000, 5000])
@pytest.mark.parametrize("tag_bytes,tag_size", [(16, 16), (64, 32)])
def test_initial_split(ctx, d, tag_bytes, tag_size):
    """grouptest.rs:155-208: the d ranges of the first split, consecutive slices."""
    rng = np.random.default_rng(d)
    n = 20_000 if d < 1000 else 3000            # d > n: most slices empty or singletons
    ids, tags, proofs = make_list(rng, n, tag_bytes)
    o = OnDevice(ctx, ids, tags, proofs, tag_size)
    q, r = divmod(U128_MAX, d)
    uppers = [(q + 1) * (i + 1) if i < r else r * (q + 1) + (i + 1 - r) * q for i in range(d)]
    got, want = o.hasher.split(uppers), o.host.split(uppers)
    assert got == want
    assert sum(c for _, _, c in got) == n


def test_split_with_bounds_that_go_back(ctx):
    """A bound below an earlier one: the reference's slice is empty and the next slice starts where the last one stopped."""
    rng = np.random.default_rng(11)
    ids, tags, proofs = make_list(rng, 500, 16)
    o = OnDevice(ctx, ids, tags, proofs, 16)
    t = [p.testing_id for p in proofs]
    uppers = [t[100], t[50], t[100], t[99], t[300] - 1, 0, U128_MAX, 5]
    assert o.hasher.split(uppers) == o.host.split(uppers)


def test_empty_list_and_bad_arguments(ctx):
    ids, tags, proofs = np.zeros((0, 16), np.uint8), np.zeros((0, 16), np.uint8), []
    o = OnDevice(ctx, ids, tags, proofs, 16)
    o.hasher.n = 0
    assert o.hasher.split([5, U128_MAX]) == o.host.split([5, U128_MAX])
    assert o.hasher.batch([(0, 0, 9)]) == o.host.batch([(0, 0, 9)])
    assert o.hasher.batch([]) == []
    rng = np.random.default_rng(2)
    ids, tags, proofs = make_list(rng, 100, 16)
    o = OnDevice(ctx, ids, tags, proofs, 16)
    for bad in ([(3, 2, 0)], [(0, 101, 0)]):
        with pytest.raises(W.WhError):
            o.hasher.batch(bad)
    o.hasher.tag_bytes = 12
    with pytest.raises(W.WhError):
        o.hasher.batch([(0, 1, 0)])
    o.hasher.tag_bytes, o.hasher.tag_size = 16, 20
    with pytest.raises(W.WhError):
        o.hasher.batch([(0, 1, 0)])


@pytest.mark.parametrize("tag_bytes,tag_size,n,d", [(16, 16, 3000, 5), (64, 32, 1200, 3), (16, 16, 20_000, 40)])
def test_whole_group_test_with_the_gpu_hasher(tag_bytes, tag_size, n, d):
    """grouptest.rs:373-447 scaled up: Bob has corrupted and missing tags; both hashers must name the same clients."""
    rng = np.random.default_rng(n)
    ids, tags, proofs = make_list(rng, n, tag_bytes, dup=False)
    bad = sorted(int(x) for x in rng.choice(n, 2 * d, replace=False))
    corrupt, missing = bad[:d], set(bad[d:])
    tags_b = tags.copy()
    for i in corrupt:
        tags_b[i, 0] ^= 1
    keep = np.array([i not in missing for i in range(n)])
    ids_b, tags_b = ids[keep], tags_b[keep]
    proofs_b = [ClientProofTag(int.from_bytes(ids_b[i].tobytes(), "little"), tags_b[i].tobytes()) for i in range(len(ids_b))]
    want_a, want_b, want_rounds = general_binary_split_test(proofs, proofs_b, VK, d, tag_size)
    ca = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3)) if tag_size == 16 else hh.HeavyHitters(512)
    cb = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3)) if tag_size == 16 else hh.HeavyHitters(512)
    try:
        oa, ob = OnDevice(ca, ids, tags, proofs, tag_size), OnDevice(cb, ids_b, tags_b, proofs_b, tag_size)
        oa.hasher.host_proofs, ob.hasher.host_proofs = proofs, proofs_b
        got_a, got_b, rounds = general_binary_split_test(proofs, proofs_b, VK, d, tag_size, oa.hasher, ob.hasher)
    finally:
        ca.close()
        cb.close()
    assert (got_a, got_b, rounds) == (want_a, want_b, want_rounds)
    expect = {proofs[i].testing_id for i in bad}
    assert got_a == expect and got_b == expect - {proofs[i].testing_id for i in missing}
