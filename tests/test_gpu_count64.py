"""Prio3Count over Field64 on the GPU, through the C ABI: the reference's own draft-07 vectors (test_vec/07/Prio3Count_{0,1}.json)
byte for byte -- prep shares, out shares, aggregate shares, the accept decision -- then seeded batches with rejected, malformed
and non-canonical keys against the oracle, two and three aggregators, and the Field64 sampler's rejection branch against
hashlib's SHAKE128."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "draft07")
P64 = 2**64 - 2**32 + 1


@pytest.mark.parametrize("name", ["Prio3Count_0", "Prio3Count_1"])
def test_count64_reference_vectors_on_gpu(name):
    import whisper_amd as W
    v = json.load(open(os.path.join(GOLD, name + ".json")))
    na, vk = v["shares"], bytes.fromhex(v["verify_key"])
    servers = [W.new_count(na) for _ in range(na)]
    try:
        keys = [[W.Prio3Count.encode_vdaf_key(a, bytes.fromhex(p["input_shares"][a]), bytes.fromhex(p["nonce"])) for p in v["prep"]] for a in range(na)]
        preps = []
        for a in range(na):
            prep, status = servers[a].prepare_init_encoded_keys(vk, keys[a])
            assert not status.any()
            assert [r.tobytes().hex() for r in prep] == [p["prep_shares"][0][a] for p in v["prep"]]
            preps.append(prep)
        for a in range(na):
            assert not servers[a].prepare_finish(preps).any()
            assert [r.tobytes().hex() for r in servers[a].out_shares()] == [p["out_shares"][a][0] for p in v["prep"]]
            assert servers[a].aggregate().tobytes().hex() == v["agg_shares"][a]
        assert sum(int.from_bytes(s.aggregate().tobytes(), "little") for s in servers) % P64 == v["agg_result"]
    finally:
        for s in servers:
            s.close()


@pytest.mark.parametrize("na", [2, 3])
def test_count64_batches_vs_oracle(na):
    import whisper_amd as W
    rng = np.random.default_rng(64 + na)
    n = 3000
    vk = rng.bytes(16)
    meas = rng.integers(0, 2, n)
    keys = [[] for _ in range(na)]
    for i in range(n):
        nonce, rand = rng.bytes(16), rng.bytes(32 * (na - 1) + 16)
        sh = O.count_shard(na, int(meas[i]), nonce, rand)
        for a in range(na):
            keys[a].append(bytearray(W.Prio3Count.encode_vdaf_key(a, sh[a], nonce)))
    # spoil some keys: not a bit (decide rejects), non-canonical proof element, short, bad aggregator id, tampered nonce, empty
    def bump(k):
        x = (int.from_bytes(k[1:9], "little") + 1) % P64
        k[1:9] = x.to_bytes(8, "little")
    bump(keys[0][5]); keys[0][6][9:17] = P64.to_bytes(8, "little"); keys[0][7] = keys[0][7][:40]; keys[0][8][0] = na
    keys[1][9][-1] ^= 1; keys[0][10] = bytearray(); keys[0][11][1:9] = (2**64 - 1).to_bytes(8, "little")
    servers = [W.new_count(na) for _ in range(na)]
    try:
        preps, stats = [], []
        for a in range(na):
            prep, status = servers[a].prepare_init_encoded_keys(vk, [bytes(k) for k in keys[a]])
            for i in list(range(0, 40)) + list(rng.integers(0, n, 200)):
                rc, oprep, oout = O.count_prepare_init_encoded(na, vk, bytes(keys[a][i]))
                assert status[i] == rc, (a, i)
                if rc == 0:
                    assert prep[i].tobytes() == oprep
            preps.append(prep); stats.append(status.copy())
        assert [int(stats[0][i]) for i in (6, 7, 8, 10, 11)] == [-3, -2, -4, -2, -3]
        want_reject = {5, 9}
        for a in range(na):
            status = servers[a].prepare_finish(preps)
            for i in range(n):
                if stats[a][i] != 0:
                    assert status[i] == stats[a][i]
                elif all(stats[b][i] == 0 for b in range(na)):
                    assert (status[i] == -7) == (i in want_reject), (a, i, status[i])
                    assert (O.count_prepare_finish(na, [preps[b][i].tobytes() for b in range(na)]) == 1) == (i in want_reject)
            outs = servers[a].out_shares()
            ok = status == 0
            keep = np.ones(n, np.uint8); keep[::7] = 0
            want = O.count_aggregate([outs[i].tobytes() for i in range(n) if ok[i] and keep[i]])
            assert servers[a].aggregate(keep).tobytes() == want
            if a == 0:
                agg0, ok0 = servers[a].aggregate(None), ok
        # the accepted reports reconstruct the count (every aggregator accepted the same set here, except where only one failed)
        good = np.ones(n, bool)
        for a in range(na):
            good &= servers[a].prepare_finish(preps) == 0
        keep = good.astype(np.uint8)
        total = sum(int.from_bytes(s.aggregate(keep).tobytes(), "little") for s in servers) % P64
        assert total == int(meas[good].sum())
        # offsets that descend or point past the blob: empty keys, not reads outside the blob
        import ctypes as C
        ks = [bytes(k) for k in keys[0][:4]]
        blob = np.frombuffer(b"".join(ks) + b"\0", np.uint8)
        off = np.zeros(5, np.uint64); off[1:] = np.cumsum([len(k) for k in ks])
        off[2] = 1 << 40                                   # key 1 ends and key 2 starts far outside
        prep = np.zeros((4, 32), np.uint8); st = np.zeros(4, np.int32)
        vkb = np.frombuffer(vk, np.uint8)
        assert W.lib.wh_prio3_count_prepare_init_batch(servers[0]._ctx, na, vkb.ctypes.data, blob.ctypes.data, off.ctypes.data, 4,
                                                       prep.ctypes.data, st.ctypes.data) == 0
        assert list(st) == [0, -2, -2, 0] and prep[0].tobytes() == O.count_prepare_init_encoded(na, vk, ks[0])[1]
        # empty batch, then state errors
        p0, s0 = servers[0].prepare_init_encoded_keys(vk, [])
        assert p0.shape == (0, 32) and servers[0].aggregate().tobytes() == bytes(8)
    finally:
        for s in servers:
            s.close()


def test_field64_sampler_rejection_branch_vs_shake128():
    import whisper_amd as W
    rng = np.random.default_rng(7)
    c = W.new_count(2)
    try:
        seeds = rng.integers(0, 256, (300, 16), dtype=np.uint8)
        for usage, binder, ne, bound in ((1, b"\x01", 1, P64), (2, b"\x02", 5, P64), (5, bytes(range(16)), 1, P64),
                                         (2, b"\x01", 5, P64 // 3), (2, b"", 5, P64 // 40), (1, b"\x07", 1, 2**58)):
            got = c.test_f64_xof_expand(seeds, usage, binder, ne, bound)
            for i in (0, 1, 17, 299):
                dst = bytes([7, 0, 0, 0, 0, 0, usage >> 8, usage & 0xff])
                stream = hashlib.shake_128(bytes([8]) + dst + seeds[i].tobytes() + binder).digest(8 * 4000)
                want = []
                for k in range(0, len(stream), 8):
                    v = int.from_bytes(stream[k:k + 8], "little")
                    if v < bound:
                        want.append(v)
                    if len(want) == ne:
                        break
                assert [int(x) for x in got[i]] == want, (usage, bound, i)
    finally:
        c.close()
