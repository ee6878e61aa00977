"""The frozen fixtures on the GPU product, through the C ABI: batched Prio3 tags / testing ids / status / out shares / aggregate
and heavy-hitter tags / testing ids / bucket vectors equal the frozen bytes (made from oracle/, NOT from the reference binary:
they guard against oracle and product drifting together, they do not pin the reference)."""
import numpy as np
import pytest

import frozen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", frozen.cases("batched"), ids=lambda p: p.split("/")[-1][:-5])
def test_batched_prio3_fixture_on_gpu(path):
    import whisper_amd as W
    fx = frozen.load(path)
    vk = bytes.fromhex(fx["verify_key"])
    p, reports = frozen.product_reports(fx)
    for cl, (lr, hr) in zip(fx["clients"], reports):
        assert frozen.sha(lr) == cl["leader_report_sha256"] and frozen.sha(hr) == cl["helper_report_sha256"]
    vdaf = W.Prio3(p)
    try:
        for who, reps in frozen.server_views(fx, reports).items():
            want = fx["servers"][who]
            tags, ids, status = vdaf.prepare_encoded_keys(vk, reps)
            outs = vdaf.out_shares()
            assert [t.tobytes().hex() for t in tags] == want["tags"]
            assert [t.tobytes().hex() for t in ids] == want["testing_ids"]
            assert [int(x) for x in status] == want["status"]
            assert [frozen.sha(o.tobytes()) for o in outs] == want["out_share_sha256"]
            assert vdaf.aggregate(None).tobytes().hex() == want["aggregate_all"]
    finally:
        vdaf.close()


@pytest.mark.parametrize("path", frozen.cases("hh"), ids=lambda p: p.split("/")[-1][:-5])
def test_heavy_hitter_fixture_on_gpu(path):
    from whisper_amd import hh
    fx = frozen.load(path)
    vk = bytes.fromhex(fx["verify_key"])
    chains = frozen.product_key_chains(fx)
    h = hh.HeavyHitters(1 << fx["depth"])
    try:
        for who, side in (("alice", 0), ("bob", 1)):
            want = fx["servers"][who]
            keys = np.stack([c[side] for c in chains])
            tags, ids, status = h.eval_and_check(vk, keys)
            dat = h.buckets()
            assert list(tags) == want["tags"]
            assert [int(i).to_bytes(16, "little").hex() for i in ids] == want["testing_ids"]
            assert [int(x) for x in status] == want["status"]
            assert [frozen.sha(d.tobytes()) for d in dat] == want["buckets_sha256"]
            assert frozen.sha(h.aggregate_hhs(None).tobytes()) == want["sketch_sha256"]
    finally:
        h.close()
