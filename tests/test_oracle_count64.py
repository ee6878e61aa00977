"""The oracle's Prio3Count (Field64) against the reference's own draft-07 vectors (test_vec/07/Prio3Count_{0,1}.json, copied
as data): shard, prepare_init, prepare_shares_to_prepare_message, out shares, aggregate -- every byte."""
import json
import os

import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "draft07")


def vectors():
    return [json.load(open(os.path.join(GOLD, f"Prio3Count_{i}.json"))) for i in (0, 1)]


@pytest.mark.parametrize("v", vectors(), ids=["Prio3Count_0", "Prio3Count_1"])
def test_count64_reference_vectors(v):
    na = v["shares"]
    vk = bytes.fromhex(v["verify_key"])
    outs = [[] for _ in range(na)]
    for p in v["prep"]:
        nonce, rand = bytes.fromhex(p["nonce"]), bytes.fromhex(p["rand"])
        assert p["public_share"] == ""
        shares = O.count_shard(na, p["measurement"], nonce, rand)
        assert [s.hex() for s in shares] == p["input_shares"]
        preps = []
        for a in range(na):
            rc, prep, out = O.count_prepare_init(na, vk, a, nonce, shares[a])
            assert rc == 0
            preps.append(prep)
            assert out.hex() == p["out_shares"][a][0]
            outs[a].append(out)
            # the same through the encoded VdafKey
            rc2, prep2, out2 = O.count_prepare_init_encoded(na, vk, bytes([a]) + shares[a] + nonce)
            assert (rc2, prep2, out2) == (0, prep, out)
        assert [x.hex() for x in preps] == p["prep_shares"][0]
        assert p["prep_messages"] == [""]
        assert O.count_prepare_finish(na, preps) == 0
    assert [O.count_aggregate(outs[a]).hex() for a in range(na)] == v["agg_shares"]
    total = sum(int.from_bytes(O.count_aggregate(outs[a]), "little") for a in range(na)) % O.P64
    assert total == v["agg_result"]


def test_count64_rejects_and_errors():
    vk = bytes(range(16)); nonce = bytes(range(16, 32)); rand = bytes(range(48))
    for meas in (0, 1):
        sh = O.count_shard(2, meas, nonce, rand)
        preps = [O.count_prepare_init(2, vk, a, nonce, sh[a])[1] for a in range(2)]
        assert O.count_prepare_finish(2, preps) == 0
    # a Leader share of 2 (not a bit): rejected by decide, not by decoding
    sh = O.count_shard(2, 1, nonce, rand)
    x = (int.from_bytes(sh[0][:8], "little") + 1) % O.P64
    bad = x.to_bytes(8, "little") + sh[0][8:]
    preps = [O.count_prepare_init(2, vk, 0, nonce, bad)[1], O.count_prepare_init(2, vk, 1, nonce, sh[1])[1]]
    assert O.count_prepare_finish(2, preps) == 1
    # non-canonical element, short share, bad aggregator id
    assert O.count_prepare_init(2, vk, 0, nonce, (O.P64).to_bytes(8, "little") + sh[0][8:])[0] == O.ERR_MODULUS
    assert O.count_prepare_init(2, vk, 0, nonce, sh[0][:47])[0] == O.ERR_SHORT
    assert O.count_prepare_init(2, vk, 2, nonce, sh[0])[0] == O.ERR_AGG_ID
    assert O.count_prepare_init_encoded(2, vk, bytes([1]) + sh[1] + nonce[:15])[0] == O.ERR_SHORT
    # three aggregators
    sh3 = O.count_shard(3, 1, nonce, bytes(range(80)))
    preps = [O.count_prepare_init(3, vk, a, nonce, sh3[a])[1] for a in range(3)]
    assert O.count_prepare_finish(3, preps) == 0
