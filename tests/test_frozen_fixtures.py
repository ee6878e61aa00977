"""The frozen fixtures (tests/golden/batched, tests/golden/hh) on the CPU: the oracle still gives the frozen answers, and the
product's CPU generators still make byte-identical reports and key chains.  These files do NOT pin the reference binary
("parity unpinned" for the batched tag and every VDPF output: DESIGN.md); they freeze the agreed answers against co-drift."""
import numpy as np
import pytest

import frozen
import oracle_lib as O


@pytest.mark.parametrize("path", frozen.cases("batched"), ids=lambda p: p.split("/")[-1][:-5])
def test_batched_prio3_fixture_oracle_and_cpu_client(path):
    fx = frozen.load(path)
    c = fx["circuit"]
    op = O.params(c["circuit"], c["bits"], c["len"], c["chunk"])
    vk = bytes.fromhex(fx["verify_key"])
    p, prod = frozen.product_reports(fx)
    reports = []
    for cl, (plr, phr) in zip(fx["clients"], prod):
        lr, hr = O.prio3_shard_batched(op, cl["measurement"], bytes.fromhex(cl["nonce"]), bytes.fromhex(cl["rand"]), int(cl["client_id"]))
        assert frozen.sha(lr) == cl["leader_report_sha256"] and frozen.sha(hr) == cl["helper_report_sha256"]
        assert plr == lr and phr == hr                    # the product's CPU client, byte for byte
        reports.append((lr, hr))
    for who, reps in frozen.server_views(fx, reports).items():
        want = fx["servers"][who]
        off = np.zeros(len(reps) + 1, np.uint64)
        off[1:] = np.cumsum([len(r) for r in reps])
        tags, ids, status, outs = O.prio3_prepare_batch(op, vk, b"".join(reps), off)
        assert [t.tobytes().hex() for t in tags] == want["tags"]
        assert [t.tobytes().hex() for t in ids] == want["testing_ids"]
        assert [int(x) for x in status] == want["status"]
        assert [frozen.sha(o.tobytes()) for o in outs] == want["out_share_sha256"]
        assert O.prio3_aggregate(outs, O.sizes(op).output_len).tobytes().hex() == want["aggregate_all"]


@pytest.mark.parametrize("path", frozen.cases("hh"), ids=lambda p: p.split("/")[-1][:-5])
def test_heavy_hitter_fixture_oracle_and_cpu_client(path):
    fx = frozen.load(path)
    vk = bytes.fromhex(fx["verify_key"])
    prod = frozen.product_key_chains(fx)
    keys = {"alice": [], "bob": []}
    for kc, (pa, pb) in zip(fx["key_chains"], prod):
        k0, k1 = O.hh_gen_key_chain(fx["depth"], int(kc["client_id"]), kc["alpha"], np.array(kc["bucket"], np.uint32), bytes.fromhex(kc["rng_seed"]))
        assert frozen.sha(k0) == kc["key_chain_0_sha256"] and frozen.sha(k1) == kc["key_chain_1_sha256"]
        assert pa.tobytes() == k0 and pb.tobytes() == k1
        keys["alice"].append(np.frombuffer(k0, np.uint8)); keys["bob"].append(np.frombuffer(k1, np.uint8))
    for who in ("alice", "bob"):
        want = fx["servers"][who]
        dat, tags, ids, status = O.hh_eval_and_check_batch(fx["depth"], vk, np.stack(keys[who]))
        assert [bytes(t).decode() for t in tags] == want["tags"]
        assert [bytes(t).hex() for t in ids] == want["testing_ids"]
        assert [int(x) for x in status] == want["status"]
        assert [frozen.sha(d.tobytes()) for d in dat] == want["buckets_sha256"]
        assert frozen.sha(O.hh_bucket_sum_u32(dat).tobytes()) == want["sketch_sha256"]
