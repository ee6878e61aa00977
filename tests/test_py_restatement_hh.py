"""The C oracle's eval_and_check against a second restatement written independently of it (tests/py_restatement_hh.py: pure
Python, from the Rust sources alone).  Every VDPF / eval_and_check output is "parity unpinned" (the reference holds no known-answer
vector for it); this makes a misreading of hhcore / common::prg something two independent readings would both have had to make.
CPU only; small depths, since the AES here is Python."""
import numpy as np
import pytest

import frozen
import oracle_lib as O
import py_restatement as R
import py_restatement_hh as H


def _bucket(rng):
    bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32)
    bk[257] = 1
    return bk


def test_stream_primitives():
    key = bytes(range(16))
    blk = bytes.fromhex("00112233445566778899aabbccddeeff")
    assert H.aes(key, blk).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a" and H.aes(key, blk) == R.aes128_encrypt_block(key, blk)
    assert H.mmo(H.KEY_PRG, blk) == O.mmo_fixed(0, blk) and H.mmo(H.KEY_HASH, blk) == O.mmo_fixed(1, blk)
    # the counter steps in its high 64-bit lane and does not carry into the low one
    assert H.inc(bytes(8) + b"\xff" * 8) == bytes(16) and H.inc(bytes(16)) == bytes(8) + b"\x01" + bytes(7)


@pytest.mark.parametrize("depth,n", [(1, 2), (2, 2), (3, 2), (4, 1)])
def test_eval_and_check_equals_oracle(depth, n):
    rng = np.random.default_rng(900 + depth)
    vk = bytes(range(40, 56))
    for i in range(n):
        alpha = int(rng.integers(0, 1 << depth))
        bk = _bucket(rng)
        cid = (1 << 100) + 7 * i + depth
        chains = O.hh_gen_key_chain(depth, cid, alpha, bk, rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
        outs = []
        for kc in chains:
            rc, o_dat, o_tag = O.hh_eval_and_check(depth, kc)
            assert rc == 0
            got_id, dat, tag = H.eval_and_check(kc, depth)
            assert got_id == cid
            assert tag.encode() == o_tag
            assert np.array_equal(np.array(dat, dtype=np.uint32), o_dat)
            tid = int.from_bytes(R.aes128_encrypt_block(vk, got_id.to_bytes(16, "little")), "little")
            _, _, o_ids, o_status = O.hh_eval_and_check_batch(depth, vk, np.frombuffer(kc, np.uint8).reshape(1, -1), want_dat=False)
            assert not o_status.any() and tid.to_bytes(16, "little") == o_ids[0].tobytes()
            outs.append((dat, tag))
        # the reference's own property (protocol.rs:219-230): equal tags, and the two servers' vectors add up to the bucket at alpha
        assert outs[0][1] == outs[1][1]
        s = (np.array(outs[0][0], dtype=np.uint32) + np.array(outs[1][0], dtype=np.uint32))
        assert np.array_equal(s[alpha], bk)
        s[alpha] = 0
        assert not s.any()


def test_tags_differ_for_a_client_that_sends_two_different_chains():
    depth = 2
    rng = np.random.default_rng(5)
    a, _ = O.hh_gen_key_chain(depth, 9, 1, _bucket(rng), bytes(range(32)))
    _, b = O.hh_gen_key_chain(depth, 9, 2, _bucket(rng), bytes(range(1, 33)))
    ta, tb = H.eval_and_check(a, depth)[2], H.eval_and_check(b, depth)[2]
    assert ta != tb
    assert ta.encode() == O.hh_eval_and_check(depth, a, want_dat=False)[2] and tb.encode() == O.hh_eval_and_check(depth, b, want_dat=False)[2]


@pytest.mark.parametrize("path", [p for p in frozen.cases("hh") if frozen.load(p)["depth"] <= 4], ids=lambda p: p.split("/")[-1][:-5])
def test_frozen_hh_fixture_reproduced_by_the_python_restatement(path):
    fx = frozen.load(path)
    chains = frozen.product_key_chains(fx)
    for who, idx in (("alice", 0), ("bob", 1)):
        want = fx["servers"][who]
        for j, pair in enumerate(chains[:3]):
            if want["status"][j]:
                continue
            _, dat, tag = H.eval_and_check(pair[idx].tobytes(), fx["depth"])
            assert tag == want["tags"][j]
            assert frozen.sha(np.array(dat, dtype=np.uint32).tobytes()) == want["buckets_sha256"][j]
