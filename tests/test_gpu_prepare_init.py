"""GPU parity for the un-batched (draft-07) Prio3 path, SURVEY section 8(f) row 3: prepare_init /
prepare_next through the C ABI against the reference's own golden vectors (tests/golden/draft07, copied as
data from batched-flpcp-rs/src/vdaf/test_vec/07/) and against the oracle on seeded inputs.  Bit-exact."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import det

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "draft07")

VECTORS = ["Prio3SumVec_0.json", "Prio3SumVec_1.json", "Prio3Histogram_0.json", "Prio3Histogram_1.json",
           "Prio3Sum_0.json", "Prio3Sum_1.json"]


def params_for(name, v):
    na = v["shares"]
    if name.startswith("Prio3SumVec"):
        return (W.Prio3Params(W.SUMVEC, v["bits"], v["length"], v["chunk_length"], na),
                O.params(O.SUMVEC, v["bits"], v["length"], v["chunk_length"], na))
    if name.startswith("Prio3Histogram"):
        return (W.Prio3Params(W.HISTOGRAM, 0, v["length"], v["chunk_length"], na),
                O.params(O.HISTOGRAM, 0, v["length"], v["chunk_length"], na))
    # Prio3Sum: Average's circuit bound to algorithm id 1 (types.rs:142-208 vs 241-313)
    return (W.Prio3Params(W.AVERAGE, v["bits"], 0, 0, na, alg_id=1), O.params(O.AVERAGE, v["bits"], 0, 0, na, alg_id=1))


@pytest.mark.parametrize("name", VECTORS)
def test_prepare_init_reproduces_draft07_vectors(name):
    v = json.load(open(os.path.join(GOLD, name)))
    p, _ = params_for(name, v)
    vk = bytes.fromhex(v["verify_key"])
    na = v["shares"]
    vdaf = W.Prio3(p)
    try:
        preps = v["prep"]
        per_agg_prep, per_agg_out = [], []
        for agg_id in range(na):
            keys = [W.Prio3.encode_vdaf_key(agg_id, bytes.fromhex(t["public_share"]), bytes.fromhex(t["input_shares"][agg_id]),
                                            bytes.fromhex(t["nonce"])) for t in preps]
            prep, status = vdaf.prepare_init_encoded_keys(vk, keys)
            assert not status.any()
            outs = vdaf.out_shares()
            for j, t in enumerate(preps):
                assert prep[j].tobytes().hex() == t["prep_shares"][0][agg_id]
                assert [outs[j][16 * e:16 * e + 16].tobytes().hex() for e in range(p.output_len)] == t["out_shares"][agg_id]
            # nothing is aggregated before prepare_next has accepted the prepare message (prio3.rs:2396-2442)
            with pytest.raises(W.WhError) as e:
                vdaf.aggregate(None)
            assert e.value.code == -102
            per_agg_prep.append(prep)
            per_agg_out.append(outs)
        for agg_id in range(na):
            keys = [W.Prio3.encode_vdaf_key(agg_id, bytes.fromhex(t["public_share"]), bytes.fromhex(t["input_shares"][agg_id]),
                                            bytes.fromhex(t["nonce"])) for t in preps]
            vdaf.prepare_init_encoded_keys(vk, keys)
            _, status = vdaf.prepare_finish(per_agg_prep)
            assert not status.any()
            assert vdaf.aggregate(None).tobytes().hex() == v["agg_shares"][agg_id]
        # prepare_shares_to_prepare_message + prepare_next, with the last aggregator's state in the context
        msgs, status = vdaf.prepare_finish(per_agg_prep)
        assert not status.any()
        for j, t in enumerate(preps):
            assert msgs[j].tobytes().hex() == t["prep_messages"][0]
        # a verifier share that does not belong makes decide fail; a foreign joint-rand part the seed check
        bad = [a.copy() for a in per_agg_prep]
        bad[0][0, 20] ^= 1
        _, status = vdaf.prepare_finish(bad)
        assert status[0] == -7 and not status[1:].any()
    finally:
        vdaf.close()


CASES = [
    ("sumvec_16_16_8", (W.SUMVEC, 16, 16, 8), 2, 9),
    ("sumvec_cfg1", (W.SUMVEC, 16, 128, 64), 2, 66),
    ("sumvec_chunk128_three", (W.SUMVEC, 16, 64, 128), 3, 5),
    ("sumvec_ragged", (W.SUMVEC, 8, 40, 5), 2, 5),
    ("histogram_128_64", (W.HISTOGRAM, 0, 128, 64), 2, 10),
    ("average_64", (W.AVERAGE, 64, 0, 0), 2, 10),
]


def measurement_for(circ, bits, ln, i):
    if circ == W.SUMVEC:
        r = np.frombuffer(det(b"pm%d" % i, 8 * ln), dtype=np.uint64)
        return [int(x) & ((1 << min(bits, 16)) - 1) for x in r]
    if circ == W.HISTOGRAM:
        return [i % ln]
    return [int.from_bytes(det(b"pm%d" % i, 8), "little") & 0xFFFF]


@pytest.mark.parametrize("name,cfg,na,n", CASES, ids=[c[0] for c in CASES])
def test_prepare_init_matches_oracle(name, cfg, na, n):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch, na)
    op = O.params(circ, bits, ln, ch, na)
    vk = bytes(det(b"vk" + name.encode(), 16))
    shards = []
    for i in range(n):
        nonce = bytes(det(b"n%d" % i, 16))
        rand = bytes(det(b"r%d" % i, O.prio3_rand_size(op)))
        public_share, input_shares = O.prio3_shard(op, measurement_for(circ, bits, ln, i), nonce, rand)
        shards.append((nonce, public_share, input_shares))
    vdaf = W.Prio3(p)
    try:
        all_prep = []
        for agg_id in range(na):
            keys = [W.Prio3.encode_vdaf_key(agg_id, ps, ish[agg_id], nonce) for nonce, ps, ish in shards]
            # malformed keys: short, bad aggregator id, element >= p (Leader only)
            extra = [keys[0][:-1], bytes([na]) + keys[0][1:]]
            if agg_id == 0:
                x = bytearray(keys[1]); off = 1 + 16 * na; x[off:off + 16] = O.P128.to_bytes(16, "little"); extra.append(bytes(x))
            prep, status = vdaf.prepare_init_encoded_keys(vk, keys + extra)
            outs = vdaf.out_shares()
            assert list(status[n:]) == ([-2, -4, -3] if agg_id == 0 else [-2, -4])
            assert not prep[n:].any() and not outs[n:].any()
            for j, (nonce, ps, ish) in enumerate(shards):
                rc, want_prep, want_out = O.prio3_prepare_init(op, vk, agg_id, nonce, ps, ish[agg_id])
                assert rc == 0 and status[j] == 0
                assert prep[j].tobytes() == want_prep
                assert outs[j].tobytes() == want_out
            prep, status = vdaf.prepare_init_encoded_keys(vk, keys)      # the state the finish step uses
            all_prep.append(prep)
        msgs, status = vdaf.prepare_finish(all_prep)
        assert not status.any()
        # tampered nonce on the last aggregator's copy: its joint randomness differs, decide fails
        nonce, ps, ish = shards[0]
        bad_key = W.Prio3.encode_vdaf_key(na - 1, ps, ish[na - 1], bytes([nonce[0] ^ 1]) + nonce[1:])
        keys = [bad_key] + [W.Prio3.encode_vdaf_key(na - 1, ps2, ish2[na - 1], n2) for n2, ps2, ish2 in shards[1:]]
        prep, status = vdaf.prepare_init_encoded_keys(vk, keys)
        assert not status.any()
        _, status = vdaf.prepare_finish(all_prep[:-1] + [prep])
        assert status[0] in (-7, -8) and not status[1:].any()
        keep = np.ones(n, np.uint8)
        assert vdaf.aggregate(keep).tobytes() == O.prio3_aggregate(vdaf.out_shares()[1:], p.output_len).tobytes()
    finally:
        vdaf.close()


@pytest.mark.parametrize("cfg,na", [((W.SUMVEC, 16, 16, 8), 2), ((W.SUMVEC, 8, 40, 5), 3), ((W.HISTOGRAM, 0, 128, 64), 2)],
                         ids=["sumvec_2", "sumvec_ragged_3", "histogram_2"])
def test_prepare_init_sampler_rejection_branch(cfg, na):
    """The un-batched path shares k_meas with the batched one: a Helper whose share stream has a rejected chunk leaves the lane-pair
    hand-over and is redone by the general path.  With the test bound at ~p/3 (two of three chunks skipped, prng.rs:73-99) every
    Helper key takes that route; prepare shares and out shares still equal the oracle's, which samples with the same bound."""
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch, na)
    op = O.params(circ, bits, ln, ch, na)
    vk = bytes(det(b"vk-rej", 16))
    bound = O.P128 // 3 + 999
    n = 7
    vdaf = W.Prio3(p)
    try:
        O.set_test_sampler_bound(bound)
        vdaf.set_test_sampler_bound(bound)
        shards = []
        for i in range(n):
            nonce = bytes(det(b"rn%d" % i, 16))
            rand = bytes(det(b"rr%d" % i, O.prio3_rand_size(op)))
            shards.append((nonce,) + tuple(O.prio3_shard(op, measurement_for(circ, bits, ln, i), nonce, rand)))
        all_prep = []
        for agg_id in range(na):
            keys = [W.Prio3.encode_vdaf_key(agg_id, ps, ish[agg_id], nonce) for nonce, ps, ish in shards]
            prep, status = vdaf.prepare_init_encoded_keys(vk, keys)
            outs = vdaf.out_shares()
            assert not status.any()
            for j, (nonce, ps, ish) in enumerate(shards):
                rc, want_prep, want_out = O.prio3_prepare_init(op, vk, agg_id, nonce, ps, ish[agg_id])
                assert rc == 0 and prep[j].tobytes() == want_prep and outs[j].tobytes() == want_out
            all_prep.append(prep)
        _, status = vdaf.prepare_finish(all_prep)
        assert not status.any()
    finally:
        O.set_test_sampler_bound(None)
        vdaf.close()


@pytest.mark.parametrize("cfg,na", [((W.SUMVEC, 8, 10, 9), 2), ((W.HISTOGRAM, 0, 11, 3), 3), ((W.AVERAGE, 16, 0, 0), 2)],
                         ids=["sumvec_2", "histogram_3", "sum_2"])
def test_prepare_init_mutated_keys_differential_fuzz(cfg, na):
    """Seeded in-place mutations of valid VdafKeys (byte flips, 16-byte windows set to p - 1, p, p + 1, all ones) through
    the GPU and the oracle's prepare_init: the same status, and the same prepare share and out share where it is zero."""
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch, na)
    op = O.params(circ, bits, ln, ch, na)
    vk = bytes(det(b"fuzzvk", 16))
    rng = np.random.default_rng(5 + na + ln)
    special = [O.P128, O.P128 - 1, O.P128 + 1, (1 << 128) - 1, 0]
    n = 40
    vdaf = W.Prio3(p)
    try:
        for agg_id in range(na):
            keys = []
            for i in range(n):
                nonce = bytes(det(b"fn%d" % i, 16))
                rand = bytes(det(b"fr%d" % i, O.prio3_rand_size(op)))
                ps, ish = O.prio3_shard(op, measurement_for(circ, bits, ln, i), nonce, rand)
                x = bytearray(W.Prio3.encode_vdaf_key(agg_id, ps, ish[agg_id], nonce))
                kind = int(rng.integers(0, 4))
                if kind == 0:
                    x[int(rng.integers(1, len(x)))] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:
                    off = 1 + 16 * int(rng.integers(0, (len(x) - 1) // 16))
                    x[off:off + 16] = special[int(rng.integers(0, len(special)))].to_bytes(16, "little")
                elif kind == 2:
                    for _ in range(4):
                        x[int(rng.integers(1, len(x)))] = int(rng.integers(0, 256))
                keys.append(bytes(x))
            prep, status = vdaf.prepare_init_encoded_keys(vk, keys)
            outs = vdaf.out_shares()
            n_ok = 0
            for j, k in enumerate(keys):
                ps, rest = k[1:1 + 16 * na], k[1 + 16 * na:]
                rc, want_prep, want_out = O.prio3_prepare_init(op, vk, agg_id, rest[-16:], ps, rest[:-16])
                assert status[j] == rc, (agg_id, j, status[j], rc)
                if rc == 0:
                    n_ok += 1
                    assert prep[j].tobytes() == want_prep and outs[j].tobytes() == want_out, (agg_id, j)
                else:
                    assert not prep[j].any() and not outs[j].any()
            assert n_ok
    finally:
        vdaf.close()


def test_prepare_init_takes_any_batch_length_in_sub_batches():
    """server-base-prio3's loop takes a Vec of any length (main.rs:41-61): a batch larger than the workspace runs in sub-batches,
    the prepare state (out shares, joint-rand seeds) lives outside the workspace, and prepare_finish sees the whole batch.
    Same prepare shares, out shares, prepare messages, status and aggregate as the same batch in one piece."""
    circ, bits, ln, ch, na, n = W.SUMVEC, 8, 10, 9, 2, 200
    p = W.Prio3Params(circ, bits, ln, ch, na)
    op = O.params(circ, bits, ln, ch, na)
    vk = bytes(det(b"vk-sub", 16))
    shards = []
    for i in range(n):
        nonce = bytes(det(b"sn%d" % i, 16))
        rand = bytes(det(b"sr%d" % i, O.prio3_rand_size(op)))
        shards.append((nonce,) + tuple(O.prio3_shard(op, measurement_for(circ, bits, ln, i), nonce, rand)))
    keys = [[W.Prio3.encode_vdaf_key(a, ps, ish[a], nonce) for nonce, ps, ish in shards] for a in range(na)]
    keys[0][17] = keys[0][17][:-3]                                           # one malformed key, one tampered nonce
    keys[1][150] = keys[1][150][:-1] + bytes([keys[1][150][-1] ^ 1])
    results = []
    for sub in (0, 64):
        vdaf = W.Prio3(p)
        if sub:
            vdaf.set_option("max_subbatch", sub)                              # 200 keys: sub-batches of 64 + 64 + 64 + 8 ... (equalised: 4 x 50)
        try:
            preps = [vdaf.prepare_init_encoded_keys(vk, keys[a]) for a in range(na)]      # the context ends with aggregator na - 1's state
            outs = vdaf.out_shares()
            msgs, status = vdaf.prepare_finish([pr for pr, _ in preps])
            results.append(([pr.copy() for pr, _ in preps], [st.copy() for _, st in preps], outs, msgs, status, vdaf.aggregate(None)))
        finally:
            vdaf.close()
    one, split = results
    for x, y in zip(one, split):
        if isinstance(x, list):
            assert all(np.array_equal(a, b) for a, b in zip(x, y))
        else:
            assert np.array_equal(x, y)
    # key 17 did not decode at aggregator 0 (its prepare share is zeros, so aggregator 1's decide fails); key 150's nonce differs
    assert one[1][0][17] == -2 and one[4][17] == -7 and one[4][150] in (-7, -8) and int((one[4] != 0).sum()) == 2
    j = 33
    rc, want_prep, want_out = O.prio3_prepare_init(op, vk, na - 1, shards[j][0], shards[j][1], shards[j][2][na - 1])
    assert rc == 0 and split[0][na - 1][j].tobytes() == want_prep and split[2][j].tobytes() == want_out


def test_prepare_init_batch_that_mixes_aggregator_ids():
    """One batch with keys of all aggregator ids interleaved (a wavefront then holds lanes whose joint-rand seed sponges absorb
    segments of different lengths, keccak.hpp: xor_word): every prepare share and out share equals the oracle's."""
    circ, bits, ln, ch, na, n = W.SUMVEC, 3, 20, 6, 3, 200
    p = W.Prio3Params(circ, bits, ln, ch, na)
    op = O.params(circ, bits, ln, ch, na)
    vk = bytes(det(b"vk-mixed", 16))
    keys, want = [], []
    for i in range(n):
        nonce = bytes(det(b"mn%d" % i, 16))
        rand = bytes(det(b"mr%d" % i, O.prio3_rand_size(op)))
        ps, ish = O.prio3_shard(op, measurement_for(circ, bits, ln, i), nonce, rand)
        a = (i * 7 + i // 5) % na
        keys.append(W.Prio3.encode_vdaf_key(a, ps, ish[a], nonce))
        rc, wp, wo = O.prio3_prepare_init(op, vk, a, nonce, ps, ish[a])
        assert rc == 0
        want.append((wp, wo))
    vdaf = W.Prio3(p)
    try:
        prep, status = vdaf.prepare_init_encoded_keys(vk, keys)
        outs = vdaf.out_shares()
        assert not status.any()
        for j in range(n):
            assert prep[j].tobytes() == want[j][0] and outs[j].tobytes() == want[j][1], j
    finally:
        vdaf.close()
