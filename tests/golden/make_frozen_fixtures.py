#!/usr/bin/env python3
"""Freezes known-answer fixtures for the results the reference holds no vector for: the batched Prio3 tag / testing id / out
share (tests/golden/batched/*.json) and the heavy-hitter tag / testing id / bucket digests (tests/golden/hh/*.json).

THESE DO NOT PIN PARITY WITH THE REFERENCE.  The values come from oracle/ (the C restatement of the reference, itself pinned by
the reference's draft-07 vectors for everything those cover); nothing here was produced by the Rust binary, which cannot be built
in this image.  What the files do: stop the oracle and the GPU product from drifting TOGETHER unnoticed -- both are checked
against these frozen bytes (tests/test_frozen_fixtures.py on the CPU, tests/test_gpu_frozen_fixtures.py on the GPU).

Inputs are stored as seeds (measurement, nonce, the 144 random bytes, client id; key-chain seed) with SHA-256 digests of the
encoded reports / key chains they must produce, so the generators are frozen too and the files stay small.

    python tests/golden/make_frozen_fixtures.py          # rewrites the JSON files (only after a deliberate change)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

VK = bytes.fromhex("000102030405060708090a0b0c0d0e0f")


def det(tag, n):
    return hashlib.shake_128(tag).digest(n)


def prio3_case(name, circuit, bits, length, chunk, measurements):
    p = O.params(circuit, bits, length, chunk)
    s = O.sizes(p)
    clients = []
    reports = {"alice": [], "bob": []}
    for i, m in enumerate(measurements):
        nonce, rand, cid = det(b"nonce-%s-%d" % (name.encode(), i), 16), det(b"rand-%s-%d" % (name.encode(), i), 144), (7 << 64) + 1000 + i
        lr, hr = O.prio3_shard_batched(p, m, nonce, rand, cid)
        clients.append({"measurement": m, "nonce": nonce.hex(), "rand": rand.hex(), "client_id": str(cid),
                        "leader_report_sha256": hashlib.sha256(lr).hexdigest(), "helper_report_sha256": hashlib.sha256(hr).hexdigest()})
        a, b = (lr, hr) if i % 2 == 0 else (hr, lr)            # client-batch-prio3/src/main.rs:114-115
        if i == len(measurements) - 1:                         # the last client is tampered on Bob's copy (main.rs:146-149)
            b = bytearray(b); b[-16] = (b[-16] + 1) & 0xff; b = bytes(b)
        reports["alice"].append(a); reports["bob"].append(b)
    out = {"comment": "frozen from oracle/ (NOT from the reference binary): guards against silent co-drift of oracle and product",
           "circuit": {"circuit": circuit, "bits": bits, "len": length, "chunk": chunk}, "verify_key": VK.hex(),
           "tampered_on_bob": len(measurements) - 1, "clients": clients, "servers": {}}
    for who in ("alice", "bob"):
        blob = b"".join(reports[who])
        off = np.zeros(len(reports[who]) + 1, np.uint64)
        off[1:] = np.cumsum([len(r) for r in reports[who]])
        tags, ids, status, outs = O.prio3_prepare_batch(p, VK, blob, off)
        agg = O.prio3_aggregate(outs, s.output_len)
        out["servers"][who] = {"tags": [t.tobytes().hex() for t in tags], "testing_ids": [t.tobytes().hex() for t in ids],
                               "status": [int(x) for x in status],
                               "out_share_sha256": [hashlib.sha256(o.tobytes()).hexdigest() for o in outs],
                               "out_share_first_element": [o[:16].tobytes().hex() for o in outs],
                               "aggregate_all": agg.tobytes().hex()}
    # the property the reference itself tests (vdaf.rs:889-897): the two aggregators' tags of an honest client cancel, i.e. are
    # equal once the Helper-role one is negated (server-batch-prio3/src/main.rs:53-59) -- what the group test compares
    for i in range(len(measurements)):
        same = out["servers"]["alice"]["tags"][i] == out["servers"]["bob"]["tags"][i]
        assert same == (i != len(measurements) - 1), (name, i)
    with open(os.path.join(HERE, "batched", name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


def hh_case(name, depth, n):
    rng = np.random.default_rng(depth * 1000 + n)
    chains = []
    keys = {"alice": [], "bob": []}
    for i in range(n):
        bucket = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32)
        bucket[257] = 1
        alpha, seed, cid = int(rng.integers(0, 1 << depth)), det(b"hh-%s-%d" % (name.encode(), i), 32), (3 << 64) + 500 + i
        k0, k1 = O.hh_gen_key_chain(depth, cid, alpha, bucket, seed)
        chains.append({"client_id": str(cid), "alpha": alpha, "bucket": [int(x) for x in bucket], "rng_seed": seed.hex(),
                       "key_chain_0_sha256": hashlib.sha256(k0).hexdigest(), "key_chain_1_sha256": hashlib.sha256(k1).hexdigest()})
        keys["alice"].append(np.frombuffer(k0, np.uint8)); keys["bob"].append(np.frombuffer(k1, np.uint8))
    out = {"comment": "frozen from oracle/ (NOT from the reference binary): guards against silent co-drift of oracle and product",
           "depth": depth, "verify_key": VK.hex(), "key_chains": chains, "servers": {}}
    for who in ("alice", "bob"):
        dat, tags, ids, status = O.hh_eval_and_check_batch(depth, VK, np.stack(keys[who]))
        out["servers"][who] = {"tags": [bytes(t).decode() for t in tags], "testing_ids": [bytes(t).hex() for t in ids],
                               "status": [int(x) for x in status],
                               "buckets_sha256": [hashlib.sha256(d.tobytes()).hexdigest() for d in dat],
                               "sketch_sha256": hashlib.sha256(O.hh_bucket_sum_u32(dat).tobytes()).hexdigest()}
    assert out["servers"]["alice"]["tags"] == out["servers"]["bob"]["tags"]         # protocol.rs:219-230
    with open(os.path.join(HERE, "hh", name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    prio3_case("sumvec_bits2_len5_chunk3", O.SUMVEC, 2, 5, 3, [[1, 0, 3, 2, 1], [0, 0, 0, 0, 0], [3, 3, 3, 3, 3], [2, 1, 0, 1, 2]])
    prio3_case("sumvec_bits16_len16_chunk8", O.SUMVEC, 16, 16, 8, [[(i * 7919 + j * 104729) & 0xffff for j in range(16)] for i in range(4)])
    prio3_case("histogram_len11_chunk3", O.HISTOGRAM, 0, 11, 3, [[0], [10], [4], [7]])
    prio3_case("average_bits64", O.AVERAGE, 64, 0, 0, [[0], [2**64 - 1], [12345678901234], [65535]])
    hh_case("depth3", 3, 3)
    hh_case("depth9", 9, 2)
    print("fixtures written under", HERE)
