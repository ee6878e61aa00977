"""Host-side group testing over proof tags, both parties in one process.

Mirror of common/src/grouptest.rs:123-371 (`general_binary_split_test`) with the TCP exchange
(`peer.exchange_message`) replaced by running Alice's and Bob's state machines in lock step, as
the reference's own test does with two tasks on localhost (grouptest.rs:373-447).  The control
flow stays on the host (it consumes 16-byte tags, not reports); only the tags come from the GPU.
"""
import hashlib
from dataclasses import dataclass

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


def _msg(slice_, lower, upper, verify_key, tag_size):
    pf, idx = _batch_proof(slice_, verify_key, upper, tag_size)
    if idx == 0:
        return {"range": (lower, upper), "singleton": None, "batch": None}, idx
    if idx == 1:
        return {"range": (lower, upper), "singleton": slice_[0], "batch": None}, idx
    return {"range": (lower, upper), "singleton": None, "batch": pf}, idx


class _Party:
    def __init__(self, proofs, verify_key, d, tag_size):
        self.proofs, self.vk, self.tag_size = proofs, verify_key, tag_size
        self.d = max(d, 2)
        self.result = set()
        self.msgs, self.slices, self.corrections = [], [], []

    def first(self):
        return _batch_proof(self.proofs, self.vk, U128_MAX, self.tag_size)[0]

    def start(self):
        d = self.d
        q, r = divmod(U128_MAX, d)
        uppers = [(q + 1) * (i + 1) if i < r else r * (q + 1) + (i + 1 - r) * q for i in range(d)]
        rest, lower = self.proofs, 0
        for upper in uppers:
            m, idx = _msg(rest, lower, upper, self.vk, self.tag_size)
            self.msgs.append(m)
            self.slices.append(rest)
            rest = rest[idx:]
            lower = upper

    def done(self):
        return not self.msgs and not self.corrections

    def step(self, recv_msgs, recv_corrections):
        self.result.update(recv_corrections)
        new_msgs, new_slices, new_corr = [], [], []
        for mine, other, sl in zip(self.msgs, recv_msgs, self.slices):
            assert mine["range"] == other["range"]
            if mine["singleton"] is not None and mine["batch"] is None:
                if other["singleton"] is None and other["batch"] is None:
                    self.result.add(mine["singleton"].testing_id)
                elif other["singleton"] is not None and other["batch"] is None:
                    if other["singleton"] != mine["singleton"]:
                        self.result.add(mine["singleton"].testing_id)
            elif mine["singleton"] is None and mine["batch"] is not None:
                if other["singleton"] is not None and other["batch"] is None:
                    s = other["singleton"]
                    for pf in sl:
                        if s.testing_id == pf.testing_id:
                            if s.tag == pf.tag:
                                continue
                            new_corr.append(pf.testing_id)
                        self.result.add(pf.testing_id)
                elif other["singleton"] is None and other["batch"] is None:
                    for pf in sl:
                        self.result.add(pf.testing_id)
                elif other["singleton"] is None and other["batch"] is not None:
                    if mine["batch"] != other["batch"]:
                        lo, hi = mine["range"]
                        mid = lo + ((hi - lo) >> 1)
                        left, mid_idx = _msg(sl, lo, mid, self.vk, self.tag_size)
                        right, _ = _msg(sl[mid_idx:], mid, hi, self.vk, self.tag_size)
                        new_msgs += [left, right]
                        new_slices += [sl[:mid_idx], sl[mid_idx:]]
        self.msgs, self.slices, self.corrections = new_msgs, new_slices, new_corr


def general_binary_split_test(proofs_alice, proofs_bob, verify_key, d, tag_size=16):
    """Both servers' runs of general_binary_split_test; proofs must be sorted by testing_id
    (server-batch-prio3/src/main.rs:93).  Returns (bad_set_alice, bad_set_bob, rounds)."""
    a = _Party(list(proofs_alice), bytes(verify_key), d, tag_size)
    b = _Party(list(proofs_bob), bytes(verify_key), d, tag_size)
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
