"""The batched path with more than two aggregators.  The servers run Alice and Bob, but Prio3::prepare_batched is generic in
the number of aggregators (vdaf/prio3.rs:2843-2855) and the reference's own batched test runs three
(test_prio3_sum_batched, :3077-3085, through run_vdaf_batched, vdaf.rs:818-898): aggregator 0 holds explicit shares, 1.. hold
seeds, a public share carries one part per aggregator, Helper a's share of zero from the hash of the public parts sits at
RLC position a - 1 and the Leader's at all of them.

Checked here: every aggregator's view through the HIP path against the oracle, bit for bit (both pipelines); the property the
reference's test asserts -- the aggregators' proof tags sum to zero (here: Leader tag == sum of the Helpers' returned tags,
because prepare_encoded_key negates a Helper's, main.rs:53-59); a tampered report breaks it; the out shares unshard to the
true aggregate.  parity unpinned beyond that: the reference holds no fixture for the batched glue."""
import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import VK, blob_of, det, measurements

pytestmark = pytest.mark.gpu

P128 = O.P128


def reports_for(p, n, seed):
    na = p.c.num_aggregators
    meas = measurements(p, n, seed)
    nonces = det(b"nonce%d" % seed, 16 * n).reshape(n, 16)
    rands = det(b"rand%d" % seed, 16 * (5 * na - 1) * n).reshape(n, -1)
    lr, hr = W.shard_batched_many(p, meas, nonces, rands, 7, 8)
    hr = hr.reshape(n, na - 1, -1)
    return lr, hr, meas, nonces, rands


def fsum(rows):
    return sum(int.from_bytes(r.tobytes(), "little") for r in rows) % P128


@pytest.mark.parametrize("path", ("default", "long_share"))
@pytest.mark.parametrize("na,cfg", [(3, (W.SUMVEC, 16, 3, 7)), (4, (W.HISTOGRAM, 0, 11, 3)), (3, (W.AVERAGE, 8, 0, 0)),
                                    (5, (W.SUMVEC, 2, 40, 16)), (3, (W.HISTOGRAM, 0, 128, 64))])
def test_every_aggregators_view_matches_the_oracle_and_tags_cancel(na, cfg, path):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch, num_aggregators=na)
    op = O.params(circ, bits, ln, ch, num_aggregators=na)
    n = 9
    lr, hr, meas, nonces, rands = reports_for(p, n, seed=na * 10 + circ)
    # the CPU client is the oracle's sharder, byte for byte
    for i in range(3):
        o_l, o_h = O.prio3_shard_batched(op, meas[i], nonces[i].tobytes(), rands[i].tobytes(), 7 + i)
        assert o_l == lr[i].tobytes() and o_h == hr[i].tobytes()
    views = [[lr[i].tobytes() for i in range(n)]] + [[hr[i, a - 1].tobytes() for i in range(n)] for a in range(1, na)]
    tampered = 4
    x = bytearray(views[na - 1][tampered]); x[-40] ^= 1; views[na - 1][tampered] = bytes(x)      # a byte of the last public proof element
    vdaf = W.Prio3(p)
    if path == "long_share":
        vdaf.set_option("chain_max_input_len", 0); vdaf.set_option("query_tiny", 0)
    try:
        tags, aggs = [], []
        for a, reports in enumerate(views):
            # a malformed report in every view: an aggregator id out of range (vdaf.rs:168-169, prio3.rs:1496-1501)
            bad = bytearray(reports[0]); bad[16 + 16 * na] = na
            rs = reports + [bytes(bad)]
            t, ids, status = vdaf.prepare_encoded_keys(VK, rs)
            outs = vdaf.out_shares()
            blob, offsets = blob_of(rs)
            o_t, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=8)
            assert np.array_equal(status, o_status) and status[:n].tolist() == [0] * n and status[n] == -4
            assert np.array_equal(t, o_t) and np.array_equal(ids, o_ids) and np.array_equal(outs, o_outs)
            keep = np.ones(n + 1, np.uint8); keep[tampered] = 0
            aggs.append(vdaf.aggregate(keep))
            tags.append(t[:n])
        for i in range(n):
            leader = int.from_bytes(tags[0][i].tobytes(), "little")
            helpers = fsum([tags[a][i] for a in range(1, na)])
            assert (leader == helpers) == (i != tampered), i
        total = W.Prio3.unshard(aggs)
        good = [i for i in range(n) if i != tampered]
        if circ == W.SUMVEC:
            want = [int(sum(int(meas[i][e]) for i in good)) for e in range(ln)]
        elif circ == W.HISTOGRAM:
            want = [sum(1 for i in good if int(meas[i][0]) == e) for e in range(ln)]
        else:
            want = [int(sum(int(meas[i][0]) for i in good))]
        assert total == want
    finally:
        vdaf.close()


@pytest.mark.parametrize("path", ("default", "long_share"))
@pytest.mark.parametrize("na", (2, 3))
def test_a_batch_that_mixes_aggregator_ids_in_one_wavefront(na, path):
    """The servers' batches mix roles (the meta-client alternates them, main.rs:114-115), so a wavefront of the per-report
    kernels can hold lanes of different aggregator ids: their joint-rand seed sponges absorb the parts before / after the own
    one in segments of different lengths and go out of phase (keccak.hpp: xor_word).  With the Leader count not a multiple
    of 64 the boundary wavefront mixes Leaders and Helpers; every report must still match the oracle, twice in a row."""
    circ, bits, ln, ch = W.SUMVEC, 4, 30, 8
    p = W.Prio3Params(circ, bits, ln, ch, num_aggregators=na)
    op = O.params(circ, bits, ln, ch, num_aggregators=na)
    n = 300
    lr, hr, meas, nonces, rands = reports_for(p, n, seed=77 + na)
    # client i is seen in role i % na, except that Leaders are a little rarer: 100 + 7 of 300 -> boundary wave is mixed
    roles = [(i % na) if (i % 17) else (1 + i % (na - 1)) for i in range(n)]
    reports = [lr[i].tobytes() if roles[i] == 0 else hr[i, roles[i] - 1].tobytes() for i in range(n)]
    assert sum(1 for r in roles if r == 0) % 64 != 0
    blob, offsets = blob_of(reports)
    o_t, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=8)
    assert not o_status.any()
    vdaf = W.Prio3(p)
    if path == "long_share":
        vdaf.set_option("chain_max_input_len", 0); vdaf.set_option("query_tiny", 0)
    try:
        for _ in range(2):
            t, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            assert not status.any()
            assert np.array_equal(t, o_t) and np.array_equal(ids, o_ids) and np.array_equal(vdaf.out_shares(), o_outs)
    finally:
        vdaf.close()
