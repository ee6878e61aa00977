"""Property tests of the oracle's batched path, after the reference's own run_vdaf_batched
(vdaf.rs:818-900: the aggregators' tags are shares of zero) and the client's tamper rule
(client-batch-prio3/src/main.rs:146-149).  No known-answer tag exists in the reference
("parity unpinned" for the tag value, see oracle/wh_oracle.h)."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O

VK = bytes(range(16, 32))

CASES = [
    ("sumvec_small", O.params(O.SUMVEC, 2, 5, 3)),
    ("sumvec_cfg1", O.params(O.SUMVEC, 16, 128, 64)),          # configs/prio3-server-alice.json
    ("sumvec_ragged", O.params(O.SUMVEC, 16, 3, 7)),            # last chunk partial
    ("histogram_4_2", O.params(O.HISTOGRAM, 0, 4, 2)),          # prio3.rs:3228-3233
    ("histogram_128_64", O.params(O.HISTOGRAM, 0, 128, 64)),
    ("average_64", O.params(O.AVERAGE, 64)),                    # prio3.rs:3263-3276, AVG_BITS
]


def det(seed, n):
    return hashlib.shake_128(seed).digest(n)


def meas_for(p, i):
    s = O.sizes(p)
    if p.circuit == O.SUMVEC:
        r = np.frombuffer(det(b"m%d" % i, 8 * p.len), dtype=np.uint64)
        return [int(x) & ((1 << p.bits) - 1) for x in r]
    if p.circuit == O.HISTOGRAM:
        return [i % p.len]
    return [int.from_bytes(det(b"m%d" % i, 8), "little") & 0xFFFF]


@pytest.mark.parametrize("name,p", CASES, ids=[c[0] for c in CASES])
def test_tags_are_shares_of_zero_and_outputs_sum(name, p):
    s = O.sizes(p)
    total = None
    for i in range(2):
        meas = meas_for(p, i)
        nonce, rand = det(b"n%d" % i, 16), det(b"r%d" % i, 144)
        lr, hr = O.prio3_shard_batched(p, meas, nonce, rand, client_id=1000 + i)
        assert len(lr) == s.leader_report_len and len(hr) == s.helper_report_len
        rc0, tag0, id0, out0 = O.prio3_prepare_encoded_key(p, VK, lr)
        rc1, tag1, id1, out1 = O.prio3_prepare_encoded_key(p, VK, hr)
        assert rc0 == 0 and rc1 == 0
        # helper's tag is negated by prepare_encoded_key -> both servers hold the same value
        assert tag0 == tag1
        assert id0 == id1 == O.aes128_encrypt_block(VK, (1000 + i).to_bytes(16, "little"))
        # out shares add up to the truncated measurement
        got = [(O.f_dec(out0[16 * e:16 * e + 16]) + O.f_dec(out1[16 * e:16 * e + 16])) % O.P128 for e in range(s.output_len)]
        if p.circuit == O.SUMVEC:
            assert got == meas
        elif p.circuit == O.HISTOGRAM:
            assert got == [1 if e == meas[0] else 0 for e in range(p.len)]
        else:
            assert got == meas
        # tamper as the meta-client does: nonce[0] += 1 on Bob's copy
        bad = bytearray(hr); bad[-16] = (bad[-16] + 1) & 0xFF
        rcb, tagb, _, _ = O.prio3_prepare_encoded_key(p, VK, bytes(bad))
        assert rcb == 0 and tagb != tag0
        # a flipped measurement bit on the leader side must change the tag too
        bad = bytearray(lr); bad[49] ^= 1
        rcb, tagb, _, _ = O.prio3_prepare_encoded_key(p, VK, bytes(bad))
        assert rcb == 0 and tagb != tag0


def test_decode_errors():
    p = O.params(O.SUMVEC, 2, 5, 3)
    s = O.sizes(p)
    lr, hr = O.prio3_shard_batched(p, [1, 2, 3, 0, 1], det(b"n", 16), det(b"r", 144), client_id=7)
    assert O.prio3_prepare_encoded_key(p, VK, lr[:-1])[0] == O.ERR_SHORT
    assert O.prio3_prepare_encoded_key(p, VK, hr[:40])[0] == O.ERR_SHORT
    assert O.prio3_prepare_encoded_key(p, VK, lr + b"\x00" * 5)[0] == 0   # trailing bytes are not looked at
    bad = bytearray(lr); bad[48] = 2
    assert O.prio3_prepare_encoded_key(p, VK, bytes(bad))[0] == O.ERR_AGG_ID
    bad = bytearray(lr); bad[49:65] = (O.P128).to_bytes(16, "little")
    assert O.prio3_prepare_encoded_key(p, VK, bytes(bad))[0] == O.ERR_MODULUS
    bad = bytearray(lr); bad[49:65] = (O.P128 - 1).to_bytes(16, "little")
    assert O.prio3_prepare_encoded_key(p, VK, bytes(bad))[0] == 0
    nq_off = 49 + 64 + 32
    bad = bytearray(hr); bad[nq_off:nq_off + 8] = (s.verifier_len - 1).to_bytes(8, "big")
    assert O.prio3_prepare_encoded_key(p, VK, bytes(bad))[0] == O.ERR_NUM_QUERIES
    bad = bytearray(hr); bad[nq_off:nq_off + 8] = (2**63).to_bytes(8, "big")
    assert O.prio3_prepare_encoded_key(p, VK, bytes(bad))[0] == O.ERR_SHORT


def test_batch_threads_equal_single():
    p = O.params(O.SUMVEC, 16, 8, 4)
    reports = []
    for i in range(12):
        lr, hr = O.prio3_shard_batched(p, meas_for(p, i), det(b"n%d" % i, 16), det(b"r%d" % i, 144), client_id=i)
        reports.append(lr if i % 2 == 0 else hr)
    offsets = np.cumsum([0] + [len(r) for r in reports]).astype(np.uint64)
    blob = np.frombuffer(b"".join(reports), dtype=np.uint8)
    a = O.prio3_prepare_batch(p, VK, blob, offsets, nthreads=1)
    b = O.prio3_prepare_batch(p, VK, blob, offsets, nthreads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for i, r in enumerate(reports):
        rc, tag, tid, out = O.prio3_prepare_encoded_key(p, VK, r)
        assert rc == a[2][i] and tag == a[0][i].tobytes() and tid == a[1][i].tobytes() and out == a[3][i].tobytes()
    keep = np.ones(12, np.uint8); keep[3] = 0
    agg = O.prio3_aggregate(a[3], O.sizes(p).output_len, keep)
    want = [0] * 8
    for i in range(12):
        if keep[i]:
            for e in range(8):
                want[e] = (want[e] + O.f_dec(a[3][i, 16 * e:16 * e + 16].tobytes())) % O.P128
    assert agg.tobytes() == b"".join(O.f_enc(x) for x in want)
