"""Differential test over random circuit shapes: the batched path and the draft-07 prepare_init path on the GPU
against the oracle, bit-exact, for shapes that hit every launch variant of the query kernel (1, 2 or 4 waves per
report, LDS or global truncate, partial last call, lazy-sum flush, several slot blocks per wave)."""
import random

import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import VK, blob_of, det, make_reports, server_views

pytestmark = pytest.mark.gpu


def shapes():
    rnd = random.Random(20241110)
    out = []
    for _ in range(14):
        bits = rnd.choice([1, 2, 3, 5, 8, 8, 16, 16, 16, 32])
        ln = rnd.randint(1, 90)
        chunk = rnd.choice([1, 2, 7, 16, 31, 64, 64, 100, 128, 128, 192, 256, 300])
        out.append((W.SUMVEC, bits, ln, chunk))
    for _ in range(6):
        ln = rnd.randint(2, 700)
        chunk = rnd.choice([1, 3, 16, 64, 128, 130, 256, 400])
        out.append((W.HISTOGRAM, 0, ln, chunk))
    out += [(W.SUMVEC, 16, 64, 64), (W.SUMVEC, 8, 128, 128), (W.SUMVEC, 32, 16, 64), (W.SUMVEC, 16, 100, 256),
            (W.AVERAGE, 1, 0, 0), (W.AVERAGE, 17, 0, 0), (W.AVERAGE, 100, 0, 0)]
    return out


SHAPES = shapes()


@pytest.mark.parametrize("cfg", SHAPES, ids=["%d_b%d_l%d_c%d" % c for c in SHAPES])
def test_random_shape_matches_oracle(cfg):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    n = 5
    lr, hr, _ = make_reports(p, n, seed=sum(cfg))
    alice, bob = server_views(lr, hr, {2})
    vdaf = W.Prio3(p)
    try:
        got = []
        for reports in (alice, bob):
            tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            outs = vdaf.out_shares()
            blob, offsets = blob_of(reports)
            o_tags, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=4)
            assert np.array_equal(status, o_status) and not status.any()
            assert np.array_equal(tags, o_tags) and np.array_equal(ids, o_ids) and np.array_equal(outs, o_outs)
            assert np.array_equal(vdaf.aggregate(None), O.prio3_aggregate(o_outs, p.output_len))
            got.append(tags)
        for i in range(n):
            assert (got[0][i].tobytes() != got[1][i].tobytes()) == (i == 2)
        # the un-batched path on the same shape
        vk = bytes(det(b"fz", 16))
        for j in range(2):
            nonce = bytes(det(b"fn%d" % j, 16))
            rand = bytes(det(b"fr%d" % j, O.prio3_rand_size(op)))
            if circ == W.SUMVEC:
                meas = [int(x) & ((1 << min(bits, 16)) - 1) for x in np.frombuffer(det(b"fm%d" % j, 8 * ln), np.uint64)]
            elif circ == W.HISTOGRAM:
                meas = [(7 * j + 1) % ln]
            else:
                meas = [(12345 + j) & ((1 << min(bits, 16)) - 1)]
            ps, ish = O.prio3_shard(op, meas, nonce, rand)
            for agg_id in range(2):
                prep, status = vdaf.prepare_init_encoded_keys(vk, [W.Prio3.encode_vdaf_key(agg_id, ps, ish[agg_id], nonce)])
                rc, want_prep, want_out = O.prio3_prepare_init(op, vk, agg_id, nonce, ps, ish[agg_id])
                assert rc == 0 and not status.any()
                assert prep[0].tobytes() == want_prep and vdaf.out_shares()[0].tobytes() == want_out
    finally:
        vdaf.close()
