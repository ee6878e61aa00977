"""The GPU product against the SECOND restatement of the reference directly (tests/py_restatement.py, py_restatement_hh.py: pure
Python, written from the Rust alone), not through the C oracle: tags, testing ids, out shares and the aggregate of a Prio3 batch;
bucket vectors, tags and ids of heavy-hitter key chains.  Bit-exact."""
import numpy as np
import pytest

import py_restatement as R
import py_restatement_hh as H
import whisper_amd as W
from common import VK, make_reports, server_views

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(W.SUMVEC, 2, 5, 3), (W.SUMVEC, 16, 3, 7), (W.HISTOGRAM, 0, 11, 3), (W.AVERAGE, 64, 0, 0)],
                         ids=["sumvec_2_5_3", "sumvec_ragged", "histogram_11_3", "average_64"])
def test_prio3_batch_equals_python_restatement(cfg):
    p = W.Prio3Params(*cfg)
    c = R.Circuit(*cfg)
    n = 6
    lr, hr, _ = make_reports(p, n, seed=91, first_client_id=3 << 64, nthreads=2)
    alice, bob = server_views(lr, hr, bad={4})
    vdaf = W.Prio3(p)
    try:
        for reports in (alice, bob):
            tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            assert not status.any()
            outs = vdaf.out_shares().reshape(n, -1)
            want = [R.prepare_encoded_key(c, 2, VK, rep) for rep in reports]
            for i, (tid, tag, out) in enumerate(want):
                assert tags[i].tobytes() == R.enc(tag) and ids[i].tobytes() == tid.to_bytes(16, "little")
                assert outs[i].tobytes() == b"".join(R.enc(e) for e in out)
            keep = np.ones(n, np.uint8)
            keep[4] = 0
            agg = [sum(want[i][2][e] for i in range(n) if keep[i]) % R.P for e in range(c.output_len)]
            assert vdaf.aggregate(keep).tobytes() == b"".join(R.enc(e) for e in agg)
    finally:
        vdaf.close()


@pytest.mark.parametrize("depth", [1, 3])
def test_heavy_hitter_expansion_equals_python_restatement(depth):
    from whisper_amd import hh
    rng = np.random.default_rng(40 + depth)
    vk = bytes(range(60, 76))
    n = 2
    keys = []
    for i in range(n):
        bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32)
        bk[257] = 1
        a, b = hh.gen_key_chain(depth, 500 + i, int(rng.integers(0, 1 << depth)), bk, rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
        keys += [np.asarray(a, np.uint8), np.asarray(b, np.uint8)]
    K = np.stack(keys)
    h = hh.HeavyHitters(1 << depth)
    try:
        tags, ids, status = h.eval_and_check(vk, K)
        dat = h.buckets()
        assert not status.any()
        for j in range(len(keys)):
            cid, want_dat, want_tag = H.eval_and_check(K[j].tobytes(), depth)
            assert tags[j] == want_tag
            assert ids[j] == int.from_bytes(R.aes128_encrypt_block(vk, cid.to_bytes(16, "little")), "little")
            assert np.array_equal(dat[j], np.array(want_dat, dtype=np.uint32))
    finally:
        h.close()
