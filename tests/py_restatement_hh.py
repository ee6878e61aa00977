"""A second, independent restatement of the heavy-hitter servers' per-key-chain work, in pure Python.

TEST INFRASTRUCTURE ONLY (see tests/py_restatement.py for why it exists): written from the Rust sources alone, without looking at
oracle/vdpf.c.  All paths are relative to /root/reference.

  server-hh/src/main.rs:163-174, 255-265        testing id = Prf(verify_key)(id.to_le_bytes()), eval_and_check::<IntModN<NumBuckets>>
  hhcore/src/protocol.rs:10-15, 78-176          KeyChain, BCSketch::eval_and_check
  hhcore/src/vdpf.rs:22-29, 115-214             VDPFKey, eval_init, eval_all
  hhcore/src/dpf.rs:5-15, 52-84                 CorrectionWord, OutputCorrection, TupleUtil::get
  hhcore/src/bucket.rs:5-10, 27-110             Bucket<G> group operations and from_rng
  common/src/prg.rs:128-241, 280-515, 517-631   get_lsb, expand_direction, convert, mmo_hash2to4, the fixed-key streams, mmo_hash4to4
  common/src/group.rs:77-122, 294-317, 371-413, 440-482   u32, bool, IntModN (whose one() has val 0)
  common/src/lib.rs:83-105, 158-170             as_bytes of u32 and bool
Key chains arrive as bincode (fixed-width little-endian integers, u64 lengths, bool = one byte).
"""
import hashlib

from py_restatement import _SBOX, _sbox, _xt  # noqa: F401  (the FIPS-197 pieces restated there)

M32 = 0xFFFFFFFF
STRING_SIZE = 256                                       # bucket.rs:12


# ---------------------------------------------------------------- AES-128 with a cached key schedule ----
_SCHED = {}


def _schedule(key):
    import py_restatement as R
    if R._SBOX is None:
        R._SBOX = _sbox()
    S = R._SBOX
    if key not in _SCHED:
        w = [list(key[4 * i:4 * i + 4]) for i in range(4)]
        rcon = 1
        for i in range(4, 44):
            t = list(w[i - 1])
            if i % 4 == 0:
                t = [S[t[1]] ^ rcon, S[t[2]], S[t[3]], S[t[0]]]
                rcon = _xt(rcon)
            w.append([a ^ b for a, b in zip(w[i - 4], t)])
        _SCHED[key] = [sum(w[4 * r:4 * r + 4], []) for r in range(11)]
    return S, _SCHED[key]


_XT = [_xt(b) for b in range(256)]
_SHIFT = [4 * ((c + r) & 3) + r for c in range(4) for r in range(4)]


def aes(key, block):
    S, rk = _schedule(bytes(key))
    st = [b ^ k for b, k in zip(block, rk[0])]
    for rnd in range(1, 11):
        st = [S[st[j]] for j in _SHIFT]                 # SubBytes then ShiftRows
        if rnd < 10:
            out = []
            for c in range(0, 16, 4):
                a0, a1, a2, a3 = st[c:c + 4]
                out += [_XT[a0] ^ _XT[a1] ^ a1 ^ a2 ^ a3, a0 ^ _XT[a1] ^ _XT[a2] ^ a2 ^ a3,
                        a0 ^ a1 ^ _XT[a2] ^ _XT[a3] ^ a3, _XT[a0] ^ a0 ^ a1 ^ a2 ^ _XT[a3]]
            st = out
        st = [b ^ k for b, k in zip(st, rk[rnd])]
    return bytes(st)


def xor(a, b):
    return bytes(x ^ y for x, y in zip(a, b))


# ------------------------------------------------------------------------- the fixed-key streams ----
KEY_PRG = bytes(16)                                     # prg.rs:283
KEY_HASH = bytes([1]) * 16                              # prg.rs:520


def mmo(key, block):                                    # AES_k(x) XOR x, prg.rs:326-330, 582-588
    return xor(aes(key, block), block)


def inc(ctr):                                           # prg.rs:450-452: _mm_add_epi64(v, set_epi64x(1, 0)) -- the HIGH 64-bit lane
    hi = (int.from_bytes(ctr[8:], "little") + 1) & 0xFFFFFFFFFFFFFFFF
    return ctr[:8] + hi.to_bytes(8, "little")


def stream(seed, nbytes):
    """set_key(seed) then fill_bytes(nbytes) (prg.rs:296-301, 487-509): refill and refill8 walk the same counters."""
    out, ctr = b"", bytes(seed)
    while len(out) < nbytes:
        out += mmo(KEY_PRG, ctr)
        ctr = inc(ctr)
    return out[:nbytes]


def expand(seed):                                       # prg.rs:133-174 with left = right = true
    bits = ((seed[0] & 1) == 0, (seed[0] & 2) == 0)
    short = bytes([seed[0] & 0xFC]) + bytes(seed[1:])
    return bits, (mmo(KEY_PRG, short), mmo(KEY_PRG, inc(short)))


def hash2to4(seed, x_bytes):                            # prg.rs:202-241
    xb = bytes(x_bytes) + bytes(16 - len(x_bytes))
    s0 = mmo(KEY_PRG, xb)
    s1 = mmo(KEY_PRG, bytes(seed))
    return [s0, s1, mmo(KEY_PRG, s0), mmo(KEY_PRG, s1)]


def hash4to4(h):                                        # prg.rs:599-631
    return [mmo(KEY_HASH, b) for b in h]


def xor4(a, b):
    return [xor(x, y) for x, y in zip(a, b)]


# ----------------------------------------------------------------------------------- output groups ----
class U32:                                              # group.rs:77-122; convert: next_u32 = the stream's first four bytes
    @staticmethod
    def convert(seed):
        return int.from_bytes(stream(seed, 4), "little")

    add = staticmethod(lambda a, b: (a + b) & M32)
    neg = staticmethod(lambda a: (-a) & M32)


class BucketU32:                                        # bucket.rs:27-110: 256 data words, sign_data, ctr, element-wise
    @staticmethod
    def convert(seed):                                  # get_random_batch(258): one fill_bytes of 1032 bytes, words little endian
        b = stream(seed, 4 * (STRING_SIZE + 2))
        return [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(STRING_SIZE + 2)]

    add = staticmethod(lambda a, b: [(x + y) & M32 for x, y in zip(a, b)])
    neg = staticmethod(lambda a: [(-x) & M32 for x in a])


# ------------------------------------------------------------------------------------------ decode ----
class Cur:
    def __init__(self, d):
        self.d, self.p = bytes(d), 0

    def take(self, n):
        assert self.p + n <= len(self.d), "key chain ends early"
        out = self.d[self.p:self.p + n]
        self.p += n
        return out

    def u64(self):
        return int.from_bytes(self.take(8), "little")

    def boolean(self):
        b = self.take(1)[0]
        assert b <= 1
        return b == 1


def decode_vdpf_key(cur, bucket):
    k = {"key_id": cur.boolean(), "root": cur.take(16)}
    k["cws"] = [(cur.take(16), (cur.boolean(), cur.boolean())) for _ in range(cur.u64())]
    k["cs"] = [cur.take(16) for _ in range(4)]
    if bucket:
        n = cur.u64()
        assert n == STRING_SIZE
        k["oc"] = [int.from_bytes(cur.take(4), "little") for _ in range(n + 2)]
    else:
        k["oc"] = int.from_bytes(cur.take(4), "little")
    return k


def decode_key_chain(data):
    cur = Cur(data)
    kc = {"id": int.from_bytes(cur.take(16), "little"), "main": decode_vdpf_key(cur, True)}
    kc["support"] = [decode_vdpf_key(cur, False) for _ in range(cur.u64())]
    return kc


# ------------------------------------------------------------------------------------------- VDPF ----
def eval_all(k, depth, G, x_is_bool):
    """VDPFKey::eval_all (vdpf.rs:123-214): (the 2^depth output words, sha256 hex of pi)."""
    assert len(k["cws"]) == depth
    nodes = [(0, k["root"], k["key_id"])]
    for i in range((1 << depth) - 1):
        level, seed, bit = nodes[i]
        bits, (left, right) = expand(seed)
        lb, rb = bits
        if bit:
            cw_seed, cw_bits = k["cws"][level]
            left, right = xor(left, cw_seed), xor(right, cw_seed)
            lb, rb = lb ^ cw_bits[0], rb ^ cw_bits[1]
        nodes += [(level + 1, left, lb), (level + 1, right, rb)]
    pi = list(k["cs"])
    x = False if x_is_bool else 0
    res = []
    for i in range((1 << depth) - 1, 2 * (1 << depth) - 1):
        _, seed, _ = nodes[i]
        tilde = hash2to4(seed, bytes([1 if x else 0]) if x_is_bool else bytes(2))     # IntModN::as_bytes is two bytes of val
        word = G.convert(seed)
        if seed[0] & 1:                                 # get_lsb
            word = G.add(word, k["oc"])
            tilde = xor4(tilde, k["cs"])
        if k["key_id"]:
            word = G.neg(word)
        res.append(word)
        pi = xor4(pi, hash4to4(xor4(pi, tilde)))
        if x_is_bool:
            x = not x                                   # bool::add is XOR with one() = true; IntModN::one() is 0: x stays
    return res, hashlib.sha256(b"".join(pi)).hexdigest()


def eval_and_check(data, depth):
    """BCSketch::eval_and_check::<IntModN<2^depth>> for Bucket<u32> (protocol.rs:78-176): (bucket vectors, 64-character tag)."""
    kc = decode_key_chain(data)
    main = kc["main"]
    dat, proof = eval_all(main, depth, BucketU32, False)
    buck = [0] * (STRING_SIZE + 2)
    for d in dat:
        buck = BucketU32.add(buck, d)
    proof_string = proof
    assert len(kc["support"]) == STRING_SIZE + 1
    support_vec = []
    for i, sk in enumerate(kc["support"]):
        small, proof_small = eval_all(sk, 1, U32, True)
        right = (small[1] - small[0]) & M32
        support_vec.append((small[0] + small[1]) & M32)
        buck[i] = (buck[i] - right) & M32               # i == 256 is sign_data, which sits right behind data
        proof_string += proof_small
    if main["key_id"]:
        buck[STRING_SIZE + 1] = (buck[STRING_SIZE + 1] - 1) & M32
        buck = BucketU32.neg(buck)
        support_vec = [(-(v - 1)) & M32 for v in support_vec]
    buck_bytes = b"".join(v.to_bytes(4, "little") for v in buck)
    ones_bytes = b"".join(v.to_bytes(4, "little") for v in support_vec)
    proof_string += hashlib.sha256(buck_bytes).hexdigest() + hashlib.sha256(ones_bytes).hexdigest()
    return kc["id"], dat, hashlib.sha256(proof_string.encode()).hexdigest()
