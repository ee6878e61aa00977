"""Pin the CPU oracle against every golden vector the reference holds for this path.

Vectors: tests/golden/draft07/*.json are byte copies of the DATA files
batched-flpcp-rs/src/vdaf/test_vec/07/{XofShake128,Prio3SumVec_*,Prio3Histogram_*,Prio3Sum_*}.json
(the reference's own tests for them: xof.rs:550-571, prio3_test.rs:179-254).
"""
import hashlib
import json
import os
import random

import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden", "draft07")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_field_constants_match_reference_table():
    # fp.rs:391-397
    p, r2, g_mont, mu = O.f_constants()
    assert p == 340282366920938462946865773367900766209
    assert r2 == 403909908237944342183153
    assert g_mont == 107630958476043550189608038630704257141
    assert mu == 18446744073709551615
    # roots[] of fp.rs:398-420 are Montgomery forms of root(l); spot-check via residues
    R = 2**128
    table = [516508834063867445247, 340282366920938462430356939304033320962,
             129526470195413442198896969089616959958, 169031622068548287099117778531474117974]
    for l, mont in enumerate(table):
        assert O.f_root(l) == mont * pow(R, -1, p) % p
    assert O.f_root(20) == 148221243758794364405224645520862378432 * pow(R, -1, p) % p


def test_field_ops_vs_bigint():
    # the recipe of fp.rs:448-532
    rnd = random.Random(1)
    p = O.P128
    edge = [0, 1, 2, p - 1, p - 2, 2**64, 2**64 - 1, 2**127, 7 * 2**66]
    vals = edge + [rnd.randrange(p) for _ in range(100)]
    for a in vals:
        for b in vals[:12]:
            assert O.f_mul(a, b) == a * b % p
            assert O.f_add(a, b) == (a + b) % p
            assert O.f_sub(a, b) == (a - b) % p
    for a in vals[1:30]:
        assert O.f_mul(O.f_inv(a), a) == 1


def test_shake128_vs_hashlib():
    rnd = random.Random(2)
    for n in [0, 1, 167, 168, 169, 335, 336, 337, 1000]:
        data = bytes(rnd.randrange(256) for _ in range(n))
        for out in [1, 16, 168, 169, 400]:
            assert O.shake128(data, out) == hashlib.shake_128(data).digest(out)


def test_xof_shake128_kat():
    v = load("XofShake128.json")
    seed, dst, binder = bytes.fromhex(v["seed"]), bytes.fromhex(v["dst"]), bytes.fromhex(v["binder"])
    assert O.xof_derive_seed(seed, dst, binder) == bytes.fromhex(v["derived_seed"])
    assert O.xof_expand(seed, dst, binder, v["length"]) == bytes.fromhex(v["expanded_vec_field128"])


def test_fft_roundtrip_and_definition():
    # fft.rs:126-226
    rnd = random.Random(3)
    p = O.P128
    for size in [1, 2, 4, 8, 64, 256]:
        vals = [rnd.randrange(p) for _ in range(size)]
        f = O.fft(vals, size)
        assert O.fft(f, size, inverse=True) == vals
        if 1 < size <= 8:
            w = O.f_root(size.bit_length() - 1)
            assert f == [sum(vals[j] * pow(w, i * j, p) for j in range(size)) % p for i in range(size)]
    # zero padding of short inputs (fft.rs:51-54)
    vals = [rnd.randrange(p) for _ in range(5)]
    assert O.fft(vals, 8) == O.fft(vals + [0, 0, 0], 8)


def _params_for(name, v):
    na = v["shares"]
    if name.startswith("Prio3SumVec"):
        return O.params(O.SUMVEC, v["bits"], v["length"], v["chunk_length"], na)
    if name.startswith("Prio3Histogram"):
        return O.params(O.HISTOGRAM, 0, v["length"], v["chunk_length"], na)
    # Prio3Sum: Average's circuit with algorithm id 1 (types.rs:142-208 vs 241-313)
    return O.params(O.AVERAGE, v["bits"], 0, 0, na, alg_id=1)


VECTORS = ["Prio3SumVec_0.json", "Prio3SumVec_1.json", "Prio3Histogram_0.json", "Prio3Histogram_1.json",
           "Prio3Sum_0.json", "Prio3Sum_1.json"]


@pytest.mark.parametrize("name", VECTORS)
def test_prio3_draft07_vectors(name):
    v = load(name)
    p = _params_for(name, v)
    s = O.sizes(p)
    vk = bytes.fromhex(v["verify_key"])
    all_out = [[] for _ in range(v["shares"])]
    for prep in v["prep"]:
        nonce, rand = bytes.fromhex(prep["nonce"]), bytes.fromhex(prep["rand"])
        # sharding (pins encode, prove, helper-share expansion, joint-rand parts)
        public_share, input_shares = O.prio3_shard(p, prep["measurement"], nonce, rand)
        assert public_share == bytes.fromhex(prep["public_share"])
        assert [x.hex() for x in input_shares] == prep["input_shares"]
        # preparation (pins query, joint-rand derivation, truncate)
        verifiers = []
        for agg_id in range(v["shares"]):
            rc, prep_share, out_share = O.prio3_prepare_init(p, vk, agg_id, nonce, public_share, input_shares[agg_id])
            assert rc == 0
            assert prep_share.hex() == prep["prep_shares"][0][agg_id]
            assert [out_share[16 * i:16 * i + 16].hex() for i in range(s.output_len)] == prep["out_shares"][agg_id]
            all_out[agg_id].append(out_share)
            verifiers.append(prep_share[:16 * s.verifier_len])
        # decide on the summed verifier accepts (prepare_shares_to_prepare_message, prio3.rs:2386)
        total = [sum(O.f_dec(vv[16 * i:16 * i + 16]) for vv in verifiers) % O.P128 for i in range(s.verifier_len)]
        assert O.flp_decide(p, b"".join(O.f_enc(x) for x in total))
    import numpy as np
    for agg_id in range(v["shares"]):
        outs = np.frombuffer(b"".join(all_out[agg_id]), dtype=np.uint8).reshape(len(v["prep"]), -1)
        assert O.prio3_aggregate(outs, s.output_len).tobytes().hex() == v["agg_shares"][agg_id]


def test_aes128_fips197():
    # FIPS-197 appendix C.1 (Prf::compute_prf has no vector in the reference, prg.rs:694-701)
    key = bytes(range(16))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    assert O.aes128_encrypt_block(key, pt).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    # appendix B
    assert O.aes128_encrypt_block(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c"),
                                  bytes.fromhex("3243f6a8885a308d313198a2e0370734")).hex() == "3925841d02dc09fbdc118597196a0b32"
