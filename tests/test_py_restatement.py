"""The C oracle against a second restatement of the reference written independently of it (tests/py_restatement.py: pure
Python, from the Rust sources alone).  Neither can be pinned against the Rust binary for the batched-only glue, but they were
read by different sessions, so a shared misreading of prio3.rs:2466-2879 would have to be made twice to pass here.  CPU only."""
import numpy as np
import pytest

import oracle_lib as O
import py_restatement as R
import whisper_amd as W
from common import VK, blob_of, make_reports, server_views

CASES = [
    ("sumvec_2_5_3", (W.SUMVEC, 2, 5, 3), 6),
    ("sumvec_ragged_16_3_7", (W.SUMVEC, 16, 3, 7), 4),          # the last chunk is partial
    ("sumvec_8_10_9", (W.SUMVEC, 8, 10, 9), 4),                 # the draft-07 vector's shape
    ("sumvec_cfg1", (W.SUMVEC, 16, 128, 64), 2),                # configs/prio3-server-alice.json
    ("histogram_4_2", (W.HISTOGRAM, 0, 4, 2), 6),
    ("histogram_11_3", (W.HISTOGRAM, 0, 11, 3), 4),
    ("average_64", (W.AVERAGE, 64, 0, 0), 4),
]


def test_field_constants_and_transform():
    assert R.P == O.P128 and R.root(0) == 1 and R.root(1) == R.P - 1
    for l in range(21):
        assert R.root(l) == O.f_root(l)
        assert pow(R.root(l), 1 << l, R.P) == 1 and (l == 0 or pow(R.root(l), 1 << (l - 1), R.P) == R.P - 1)
    vals = [int.from_bytes(O.shake128(b"dft%d" % i, 16), "little") % R.P for i in range(13)]
    for size in (1, 2, 16, 64):
        want = O.fft(vals[:min(13, size)], size)
        assert R.dft(vals[:min(13, size)], size) == want
        w = R.root(R.log2(size))
        assert want == [sum(v * pow(w, i * j, R.P) for j, v in enumerate(vals[:min(13, size)])) % R.P for i in range(size)]
        inv = R.dft(want, size)
        R.dft_inv_finish(inv, size, pow(size, -1, R.P))
        assert inv == (vals[:min(13, size)] + [0] * size)[:size]


def test_aes_block_fips197():
    key = bytes(range(16))
    assert R.aes128_encrypt_block(key, bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    assert R.aes128_encrypt_block(key, bytes(16)) == O.aes128_encrypt_block(key, bytes(16))


@pytest.mark.parametrize("name,cfg,n", CASES, ids=[c[0] for c in CASES])
def test_prepare_encoded_key_equals_oracle(name, cfg, n):
    p = W.Prio3Params(*cfg)
    op = O.params(*cfg)
    c = R.Circuit(*cfg)
    assert (c.input_len, c.proof_len, c.verifier_len, c.output_len) == (p.sizes.input_len, p.sizes.proof_len, p.sizes.verifier_len, p.sizes.output_len)
    lr, hr, _ = make_reports(p, n, seed=77, first_client_id=1 << 70, nthreads=2)
    alice, bob = server_views(lr, hr, bad={1})
    mine = {}
    for who, reports in (("alice", alice), ("bob", bob)):
        blob, off = blob_of(reports)
        o_tags, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob, off, nthreads=2)
        assert not o_status.any()
        got = [R.prepare_encoded_key(c, 2, VK, rep) for rep in reports]
        for i, (tid, tag, out) in enumerate(got):
            assert R.enc(tag) == o_tags[i].tobytes(), (who, i)
            assert tid.to_bytes(16, "little") == o_ids[i].tobytes(), (who, i)
            assert b"".join(R.enc(e) for e in out) == o_outs[i].tobytes(), (who, i)
        mine[who] = got
    # the protocol's own property (vdaf.rs:818-900): the two servers' tags agree exactly for the untampered clients
    for i in range(n):
        assert mine["alice"][i][0] == mine["bob"][i][0]
        assert (mine["alice"][i][1] == mine["bob"][i][1]) == (i != 1)


def test_undecodable_reports_are_rejected_by_both():
    cfg = (W.SUMVEC, 2, 5, 3)
    p, op, c = W.Prio3Params(*cfg), O.params(*cfg), R.Circuit(*cfg)
    lr, hr, _ = make_reports(p, 2, seed=78, nthreads=1)
    leader, helper = lr[0].tobytes(), hr[1].tobytes()
    over = bytearray(leader)
    over[49:65] = R.P.to_bytes(16, "little")                     # the first measurement element is p itself
    wrong_nq = bytearray(helper)
    nq_at = 49 + 64 + 32
    wrong_nq[nq_at:nq_at + 8] = (c.verifier_len - 1).to_bytes(8, "big")
    cases = [leader[:100], helper[:-1], b"", leader[:48] + b"\x02" + leader[49:], bytes(over), bytes(wrong_nq)]
    blob, off = blob_of(cases)
    _, _, status, _ = O.prio3_prepare_batch(op, VK, blob, off, nthreads=1)
    assert all(int(s) != 0 for s in status)
    for rep in cases:
        with pytest.raises(R.Malformed):
            R.prepare_encoded_key(c, 2, VK, rep)
    # and the untouched ones pass both
    blob, off = blob_of([leader, helper])
    _, _, status, _ = O.prio3_prepare_batch(op, VK, blob, off, nthreads=1)
    assert not status.any()
    R.prepare_encoded_key(c, 2, VK, leader), R.prepare_encoded_key(c, 2, VK, helper)


import frozen  # noqa: E402


@pytest.mark.parametrize("path", frozen.cases("batched"), ids=lambda p: p.split("/")[-1][:-5])
def test_frozen_fixtures_reproduced_by_the_python_restatement(path):
    """The frozen known-answer files (made from the C oracle) are reproduced by the independent restatement as well: tags,
    testing ids, out shares and the aggregate of both servers."""
    fx = frozen.load(path)
    cc = fx["circuit"]
    c = R.Circuit(cc["circuit"], cc["bits"], cc["len"], cc["chunk"])
    vk = bytes.fromhex(fx["verify_key"])
    _, prod = frozen.product_reports(fx)
    for who, reps in frozen.server_views(fx, prod).items():
        want = fx["servers"][who]
        assert not any(want["status"])
        got = [R.prepare_encoded_key(c, 2, vk, rep) for rep in reps]
        assert [R.enc(tag).hex() for _, tag, _ in got] == want["tags"]
        assert [tid.to_bytes(16, "little").hex() for tid, _, _ in got] == want["testing_ids"]
        assert [frozen.sha(b"".join(R.enc(e) for e in out)) for _, _, out in got] == want["out_share_sha256"]
        agg = [sum(out[e] for _, _, out in got) % R.P for e in range(c.output_len)]
        assert b"".join(R.enc(e) for e in agg).hex() == want["aggregate_all"]
