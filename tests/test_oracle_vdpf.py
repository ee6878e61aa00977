"""CPU tests of the heavy-hitter oracle (oracle/vdpf.c) and of the product-side CPU client.

The reference holds no known-answer vector for any VDPF value (its tests draw from thread_rng), so the tag and bucket
VALUES are "parity unpinned".  Pinned here: the primitives (SHA-256 against hashlib, AES-128 against FIPS-197 C.1, MMO
against AES), an independent pure-Python restatement of eval_and_check, and the reference's own properties
(hhcore/src/vdpf.rs:225-305, protocol.rs:191-231).
"""
import hashlib

import numpy as np
import pytest

import oracle_lib as o

SBOX = None


def _sbox():
    global SBOX
    if SBOX is None:
        # S[x] from AES itself: with the all-zero key the first round key is zero, so encrypting is not a shortcut;
        # take it from the oracle's single-block AES pinned by FIPS-197: S[x] = first key-schedule step
        def xt(a):
            return ((a << 1) ^ (0x1b if a & 0x80 else 0)) & 0xff

        def mul(a, b):
            r = 0
            while b:
                if b & 1:
                    r ^= a
                a = xt(a); b >>= 1
            return r
        inv = [0] * 256
        for x in range(1, 256):
            for y in range(1, 256):
                if mul(x, y) == 1:
                    inv[x] = y
                    break
        SBOX = []
        for x in range(256):
            s = r = inv[x]
            for _ in range(4):
                r = ((r << 1) | (r >> 7)) & 0xff
                s ^= r
            SBOX.append(s ^ 0x63)
    return SBOX


def py_aes(key, block):
    """Plain FIPS-197 AES-128, byte oriented."""
    S = _sbox()

    def xt(a):
        return ((a << 1) ^ (0x1b if a & 0x80 else 0)) & 0xff
    rk = list(key)
    rcon = 1
    for i in range(16, 176, 4):
        t = rk[i - 4:i]
        if i % 16 == 0:
            t = [S[t[1]] ^ rcon, S[t[2]], S[t[3]], S[t[0]]]
            rcon = xt(rcon)
        rk += [rk[i - 16 + k] ^ t[k] for k in range(4)]
    s = [b ^ k for b, k in zip(block, rk[:16])]
    for rnd in range(1, 11):
        t = [S[s[4 * ((c + r) & 3) + r]] for c in range(4) for r in range(4)]
        if rnd < 10:
            u = []
            for c in range(4):
                a = t[4 * c:4 * c + 4]
                u += [xt(a[0]) ^ xt(a[1]) ^ a[1] ^ a[2] ^ a[3], a[0] ^ xt(a[1]) ^ xt(a[2]) ^ a[2] ^ a[3],
                      a[0] ^ a[1] ^ xt(a[2]) ^ xt(a[3]) ^ a[3], xt(a[0]) ^ a[0] ^ a[1] ^ a[2] ^ xt(a[3])]
            t = u
        s = [a ^ b for a, b in zip(t, rk[16 * rnd:16 * rnd + 16])]
    return bytes(s)


def test_fips197_and_mmo():
    key = bytes(range(16)); pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    want = bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")  # FIPS-197 appendix C.1
    assert py_aes(key, pt) == want
    assert o.aes128_encrypt_block(key, pt) == want
    rng = np.random.default_rng(5)
    for kb in (0, 1):
        for _ in range(8):
            x = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
            e = py_aes(bytes([kb] * 16), x)
            assert o.mmo_fixed(kb, x) == bytes(a ^ b for a, b in zip(e, x))


def test_sha256_against_hashlib():
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1028, 1032, 16640):
        d = bytes((i * 7 + 3) & 255 for i in range(n))
        assert o.sha256(d) == hashlib.sha256(d).digest(), n


# ---- an independent restatement of eval_and_check on the bincode bytes (hhcore/src/protocol.rs:78-176) ----
def _mmo(kb, x):
    return bytes(a ^ b for a, b in zip(py_aes(bytes([kb] * 16), x), x))


def _bump(c, k):
    hi = (int.from_bytes(c[8:], "little") + k) & (2**64 - 1)
    return c[:8] + hi.to_bytes(8, "little")


def _xor(a, b):
    return bytes(x ^ y for x, y in zip(a, b))


def _py_eval_all(key_id, root, cws, cs, oc, depth, words, x_counts):
    nodes = [(root, key_id)]
    for lvl in range(depth):
        nxt = []
        for seed, bit in nodes:
            bl, br = int(seed[0] & 1 == 0), int(seed[0] & 2 == 0)
            c = bytes([seed[0] & 0xfc]) + seed[1:]
            left, right = _mmo(0, c), _mmo(0, _bump(c, 1))
            if bit:
                cw, b0, b1 = cws[lvl]
                left, right, bl, br = _xor(left, cw), _xor(right, cw), bl ^ b0, br ^ b1
            nxt += [(left, bl), (right, br)]
        nodes = nxt
    pi, out = cs, []
    for x, (seed, _) in enumerate(nodes):
        xb = (x if x_counts else 0).to_bytes(2, "little") + bytes(14)
        s0 = _mmo(0, xb); s1 = _mmo(0, seed)
        tp = s0 + s1 + _mmo(0, s0) + _mmo(0, s1)
        c, stream = seed, b""
        if words == 1:
            stream = s1
        else:
            for _ in range(65):
                stream += _mmo(0, c); c = _bump(c, 1)
        w = np.frombuffer(stream[:4 * words], "<u4").astype(np.uint32)
        if seed[0] & 1:
            w = w + oc; tp = _xor(tp, cs)
        if key_id:
            w = (0 - w).astype(np.uint32)
        out.append(w.astype(np.uint32))
        hin = _xor(pi, tp)
        pi = _xor(pi, b"".join(_mmo(1, hin[16 * k:16 * k + 16]) for k in range(4)))
    return np.array(out, np.uint32), hashlib.sha256(pi).hexdigest()


def _py_parse_key(b, at, depth, bucket):
    key_id = b[at]; root = b[at + 1:at + 17]
    assert int.from_bytes(b[at + 17:at + 25], "little") == depth
    at += 25
    cws = []
    for _ in range(depth):
        cws.append((b[at:at + 16], b[at + 16], b[at + 17])); at += 18
    cs = b[at:at + 64]; at += 64
    if bucket:
        assert int.from_bytes(b[at:at + 8], "little") == 256
        at += 8
    words = 258 if bucket else 1
    oc = np.frombuffer(b[at:at + 4 * words], "<u4").astype(np.uint32)
    return (key_id, root, cws, cs, oc), at + 4 * words


def py_eval_and_check(kc, depth):
    main, at = _py_parse_key(kc, 16, depth, True)
    dat, proof = _py_eval_all(*main, depth, 258, False)
    buck = dat.sum(axis=0, dtype=np.uint32)
    assert int.from_bytes(kc[at:at + 8], "little") == 257
    at += 8
    proofs, ones = [proof], []
    for i in range(257):
        sk, at = _py_parse_key(kc, at, 1, False)
        d, pf = _py_eval_all(*sk, 1, 1, True)
        ones.append(np.uint32(d[0][0] + d[1][0]))
        buck[i] -= np.uint32(d[1][0] - d[0][0])
        proofs.append(pf)
    ones = np.array(ones, np.uint32)
    if main[0]:
        buck[257] -= np.uint32(1)
        buck = (0 - buck).astype(np.uint32)
        ones = (0 - (ones - np.uint32(1))).astype(np.uint32)
    proofs.append(hashlib.sha256(buck.astype("<u4").tobytes()).hexdigest())
    proofs.append(hashlib.sha256(ones.astype("<u4").tobytes()).hexdigest())
    return dat, hashlib.sha256("".join(proofs).encode()).hexdigest()


def _bucket(rng):
    bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32)
    bk[257] = 1
    return bk


def test_python_restatement_agrees_with_oracle():
    rng = np.random.default_rng(11)
    depth, alpha = 2, 3
    with np.errstate(over="ignore"):
        a, b = o.hh_gen_key_chain(depth, 5, alpha, _bucket(rng), bytes(range(32)))
        for kc in (a, b):
            rc, dat, tag = o.hh_eval_and_check(depth, kc)
            pdat, ptag = py_eval_and_check(kc, depth)
            assert rc == 0 and tag.decode() == ptag
            assert (dat == pdat).all()


@pytest.mark.parametrize("depth,alpha", [(1, 1), (3, 0), (5, 19), (9, 300)])
def test_reference_properties(depth, alpha):
    """protocol.rs:191-231 / vdpf.rs:225-305: equal proofs, reconstruction to beta at alpha and zero elsewhere."""
    rng = np.random.default_rng(depth)
    bk = _bucket(rng)
    a, b = o.hh_gen_key_chain(depth, 1234, alpha, bk, bytes([depth] * 32))
    assert len(a) == len(b) == o.hh_key_chain_len(depth)
    rc0, d0, t0 = o.hh_eval_and_check(depth, a)
    rc1, d1, t1 = o.hh_eval_and_check(depth, b)
    assert rc0 == rc1 == 0 and t0 == t1
    s = d0 + d1
    assert (s[alpha] == bk).all()
    s[alpha] = 0
    assert not s.any()


def test_malformed_client_is_caught():
    """client-hh/src/main.rs:84-118: a bad client sends Alice one key chain and Bob another; the tags differ."""
    rng = np.random.default_rng(3)
    depth = 4
    a0, _ = o.hh_gen_key_chain(depth, 9, 2, _bucket(rng), bytes([1] * 32))
    _, b1 = o.hh_gen_key_chain(depth, 9, 7, _bucket(rng), bytes([2] * 32))
    assert o.hh_eval_and_check(depth, a0)[2] != o.hh_eval_and_check(depth, b1)[2]
    # a flipped byte in anything this server reads changes the tag or is refused (server 0 never applies the
    # correction words of the first level, so those are left out)
    good = o.hh_eval_and_check(depth, a0)[2]
    for at in (17, 41 + 18 * depth + 5, 200, len(a0) - 3, len(a0) - 40, len(a0) - 100):
        bad = bytearray(a0); bad[at] ^= 0x10
        rc, _, t = o.hh_eval_and_check(depth, bytes(bad))
        assert rc != 0 or t != good, at


def test_key_format_errors():
    rng = np.random.default_rng(4)
    depth = 3
    a, _ = o.hh_gen_key_chain(depth, 1, 1, _bucket(rng), bytes(32))
    for at, val in ((16, 2), (33, depth + 1), (41 + 16, 2), (41 + 18 * depth + 64, 255), (41 + 18 * depth + 64 + 8 + 1032, 0),
                    (o.hh_key_chain_len(depth) - 111, 3)):
        bad = bytearray(a); bad[at] = val
        assert o.hh_eval_and_check(depth, bytes(bad))[0] == o.ERR_KEY_FORMAT, at
    assert o.hh_eval_and_check(depth, a[:-1])[0] == o.ERR_SHORT


def test_product_client_matches_oracle_generator():
    """whisper_amd's CPU client (hh_client.cpp) and the oracle's generator are separate code: same bytes."""
    hh = pytest.importorskip("whisper_amd.hh")
    rng = np.random.default_rng(8)
    for depth, alpha in ((1, 0), (4, 9), (9, 511)):
        bk = _bucket(rng)
        seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        a, b = o.hh_gen_key_chain(depth, 2**100 + 7, alpha, bk, seed)
        pa, pb = hh.gen_key_chain(depth, 2**100 + 7, alpha, bk, seed)
        assert pa.tobytes() == a and pb.tobytes() == b
        assert hh.key_chain_len(depth) == len(a)
