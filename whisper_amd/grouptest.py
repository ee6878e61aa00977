"""Host-side group testing over proof tags, both parties in one process.

Mirror of common/src/grouptest.rs:123-371 (`general_binary_split_test`) with the TCP exchange
(`peer.exchange_message`) replaced by running Alice's and Bob's state machines in lock step, as
the reference's own test does with two tasks on localhost (grouptest.rs:373-447).  The control
flow stays on the host (it consumes 16-byte tags, not reports); only the tags come from the GPU.

The batch proofs of a round (grouptest.rs:43-99) come from a hasher object: `HostHasher` (hashlib, the default) or
`DeviceHasher`, which keeps the sorted ids and tags in HBM and hashes all slices of a round side by side through
wh_grouptest_split_proofs_dev / wh_grouptest_batch_proofs_dev (one GPU lane per slice).  The complete proof that opens the
test is one long slice and is hashed on the host either way.
"""
import hashlib
from dataclasses import dataclass

import numpy as np

U128_MAX = 2**128 - 1


@dataclass(frozen=True)
class ClientProofTag:
    testing_id: int
    tag: bytes


def _batch_proof(proofs, verify_key, upper, tag_size):
    """grouptest.rs:43-99: hash of verify_key || tags while testing_id <= upper; (digest, split_idx)."""
    h = hashlib.shake_128() if tag_size == 16 else hashlib.sha256()
    h.update(verify_key)
    split = len(proofs)
    for i, pf in enumerate(proofs):
        if pf.testing_id > upper:
            split = i
            break
        h.update(pf.tag)
    return (h.digest(16) if tag_size == 16 else h.digest()), split


class HostHasher:
    """Batch proofs with hashlib over the party's own list."""

    def __init__(self, proofs, verify_key, tag_size):
        self.proofs, self.vk, self.tag_size = proofs, bytes(verify_key), tag_size

    def complete(self):
        return _batch_proof(self.proofs, self.vk, U128_MAX, self.tag_size)[0]

    def split(self, uppers):
        """grouptest.rs:182-208: consecutive slices; [(digest, first, count)]."""
        out, first = [], 0
        for upper in uppers:
            pf, idx = _batch_proof(self.proofs[first:], self.vk, upper, self.tag_size)
            out.append((pf, first, idx))
            first += idx
        return out

    def batch(self, jobs):
        """jobs = [(start, end, upper)] -> [(digest, count)]."""
        return [_batch_proof(self.proofs[s:e], self.vk, up, self.tag_size) for s, e, up in jobs]


class DeviceHasher:
    """Batch proofs on the GPU.  `ctx` is a whisper_amd context object (Prio3, HeavyHitters ...: anything with `_ctx` and
    `_chk`); d_ids / d_tags are raw device pointers to the n SORTED ids (16 bytes, little endian) and tags (tag_bytes each)."""

    def __init__(self, ctx, d_ids, d_tags, n, tag_bytes, verify_key, tag_size, host_proofs=None):
        from ._lib import lib
        self._lib, self.ctx = lib, ctx
        self.d_ids, self.d_tags, self.n, self.tag_bytes = d_ids, d_tags, n, tag_bytes
        self.vk = np.frombuffer(bytes(verify_key), np.uint8).copy()
        self.tag_size = tag_size
        self.host_proofs = host_proofs

    def complete(self):
        if self.host_proofs is not None:      # one long stream: a host core is faster
            return _batch_proof(self.host_proofs, self.vk.tobytes(), U128_MAX, self.tag_size)[0]
        return self.batch([(0, self.n, U128_MAX)])[0][0]

    @staticmethod
    def _u128s(vals):
        return np.frombuffer(b"".join(int(v).to_bytes(16, "little") for v in vals), np.uint8).copy()

    def split(self, uppers):
        d = len(uppers)
        ups = self._u128s(uppers)
        proofs = np.zeros(d * self.tag_size, np.uint8)
        firsts, counts = np.zeros(d, np.uint64), np.zeros(d, np.uint64)
        self.ctx._chk(self._lib.wh_grouptest_split_proofs_dev(self.ctx._ctx, self.d_ids, self.d_tags, self.n, self.tag_bytes,
                                                              self.vk.ctypes.data, ups.ctypes.data, d, self.tag_size,
                                                              proofs.ctypes.data, firsts.ctypes.data, counts.ctypes.data),
                      "grouptest_split_proofs_dev")
        pb = proofs.tobytes()
        return [(pb[self.tag_size * i:self.tag_size * (i + 1)], int(firsts[i]), int(counts[i])) for i in range(d)]

    def batch(self, jobs):
        k = len(jobs)
        if k == 0:
            return []
        starts = np.array([j[0] for j in jobs], np.uint64)
        ends = np.array([j[1] for j in jobs], np.uint64)
        ups = self._u128s([j[2] for j in jobs])
        proofs = np.zeros(k * self.tag_size, np.uint8)
        counts = np.zeros(k, np.uint64)
        self.ctx._chk(self._lib.wh_grouptest_batch_proofs_dev(self.ctx._ctx, self.d_ids, self.d_tags, self.n, self.tag_bytes,
                                                              self.vk.ctypes.data, starts.ctypes.data, ends.ctypes.data,
                                                              ups.ctypes.data, k, self.tag_size, proofs.ctypes.data,
                                                              counts.ctypes.data), "grouptest_batch_proofs_dev")
        pb = proofs.tobytes()
        return [(pb[self.tag_size * i:self.tag_size * (i + 1)], int(counts[i])) for i in range(k)]


class _Party:
    """One server's state.  A slice of the sorted list is a (start, end) index pair."""

    def __init__(self, proofs, verify_key, d, tag_size, hasher=None):
        self.proofs, self.vk, self.tag_size = proofs, verify_key, tag_size
        self.hasher = hasher if hasher is not None else HostHasher(proofs, verify_key, tag_size)
        self.d = max(d, 2)
        self.result = set()
        self.msgs, self.slices, self.corrections = [], [], []

    def _msg(self, first, lower, upper, pf, idx):
        if idx == 0:
            return {"range": (lower, upper), "singleton": None, "batch": None}
        if idx == 1:
            return {"range": (lower, upper), "singleton": self.proofs[first], "batch": None}
        return {"range": (lower, upper), "singleton": None, "batch": pf}

    def first(self):
        return self.hasher.complete()

    def start(self):
        d, n = self.d, len(self.proofs)
        q, r = divmod(U128_MAX, d)
        uppers = [(q + 1) * (i + 1) if i < r else r * (q + 1) + (i + 1 - r) * q for i in range(d)]
        lower = 0
        for upper, (pf, first, idx) in zip(uppers, self.hasher.split(uppers)):
            self.msgs.append(self._msg(first, lower, upper, pf, idx))
            self.slices.append((first, n))           # `slice_so_far`: to the end of the list (grouptest.rs:205)
            lower = upper

    def done(self):
        return not self.msgs and not self.corrections

    def step(self, recv_msgs, recv_corrections):
        self.result.update(recv_corrections)
        new_corr, to_split = [], []
        for mine, other, (s, e) in zip(self.msgs, recv_msgs, self.slices):
            assert mine["range"] == other["range"]
            if mine["singleton"] is not None and mine["batch"] is None:
                if other["singleton"] is None and other["batch"] is None:
                    self.result.add(mine["singleton"].testing_id)
                elif other["singleton"] is not None and other["batch"] is None:
                    if other["singleton"] != mine["singleton"]:
                        self.result.add(mine["singleton"].testing_id)
            elif mine["singleton"] is None and mine["batch"] is not None:
                if other["singleton"] is not None and other["batch"] is None:
                    sg = other["singleton"]
                    for pf in self.proofs[s:e]:
                        if sg.testing_id == pf.testing_id:
                            if sg.tag == pf.tag:
                                continue
                            new_corr.append(pf.testing_id)
                        self.result.add(pf.testing_id)
                elif other["singleton"] is None and other["batch"] is None:
                    for pf in self.proofs[s:e]:
                        self.result.add(pf.testing_id)
                elif other["singleton"] is None and other["batch"] is not None:
                    if mine["batch"] != other["batch"]:
                        to_split.append((mine["range"], s, e))
        # the halves of every mismatching range (grouptest.rs:306-357): all left halves in one batch, then all right ones
        mids = [lo + ((hi - lo) >> 1) for (lo, hi), _, _ in to_split]
        lefts = self.hasher.batch([(s, e, mid) for (_, s, e), mid in zip(to_split, mids)])
        rights = self.hasher.batch([(s + lc, e, rng[1]) for (rng, s, e), (_, lc) in zip(to_split, lefts)])
        new_msgs, new_slices = [], []
        for (rng, s, e), mid, (lpf, lc), (rpf, rc) in zip(to_split, mids, lefts, rights):
            new_msgs += [self._msg(s, rng[0], mid, lpf, lc), self._msg(s + lc, mid, rng[1], rpf, rc)]
            new_slices += [(s, s + lc), (s + lc, e)]
        self.msgs, self.slices, self.corrections = new_msgs, new_slices, new_corr


def general_binary_split_test(proofs_alice, proofs_bob, verify_key, d, tag_size=16, hasher_alice=None, hasher_bob=None):
    """Both servers' runs of general_binary_split_test; proofs must be sorted by testing_id
    (server-batch-prio3/src/main.rs:93).  Returns (bad_set_alice, bad_set_bob, rounds).  A hasher (DeviceHasher) moves a
    party's batch proofs to the GPU; the verdicts are the same."""
    a = _Party(list(proofs_alice), bytes(verify_key), d, tag_size, hasher_alice)
    b = _Party(list(proofs_bob), bytes(verify_key), d, tag_size, hasher_bob)
    rounds = 1
    if a.first() == b.first():
        return set(), set(), rounds
    a.start()
    b.start()
    while not (a.done() and b.done()):
        rounds += 1
        am, ac, bm, bc = a.msgs, a.corrections, b.msgs, b.corrections
        a.step(bm, bc)
        b.step(am, ac)
    return a.result, b.result, rounds
