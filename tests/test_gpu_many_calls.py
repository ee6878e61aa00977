"""Circuits with very many gadget calls (a chunk length far below sqrt(input_len)).  The reference takes any chunk_length
(types.rs:508-560, 353-380); the query kernel keeps 48 B of weight limbs per gadget call in LDS and asks for up to the
160 KiB a gfx950 workgroup can have, so about 3 200 calls verify (slowly: one workgroup per CU) and more are refused loudly
(WH_ERR_UNSUPPORTED), never computed some other way.  Found by tools/stress_prio3.py, which drew SumVec(33, 39, chunk 1).

Checked: both servers' views against the oracle bit for bit (status, tags, testing ids, out shares) in both pipelines, with
a truncated and a tampered report in the batch; the largest call count that fits; the refusal one call above it."""
import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import VK, blob_of, make_reports

pytestmark = pytest.mark.gpu

LDS_MAX = 160 * 1024
STAGE = 8 * 64 * 17          # truncate staging per wavefront (kernels.hip: QT_STAGE_U4 * 16)
SHARED = 10 * 16 + 4 * 4     # QueryShared


def max_calls(waves):
    return (LDS_MAX - SHARED - waves * STAGE) // 48


CASES = [
    ("sumvec_33_39_chunk1", (W.SUMVEC, 33, 39, 1), 5),                # 1 287 calls, m = 2 048: the stress tool's draw
    ("histogram_chunk65_calls1001", (W.HISTOGRAM, 0, 65 * 1001, 65), 4),   # two wavefronts per report, 82 KiB of LDS
    ("sumvec_most_calls_that_fit", (W.SUMVEC, 1, max_calls(1), 1), 4),     # 3 228 calls, m = 4 096, 160 KiB less 32 B
]


@pytest.mark.parametrize("path", ("default", "long_share"))
@pytest.mark.parametrize("name,cfg,n", CASES, ids=[c[0] for c in CASES])
def test_many_gadget_calls_match_oracle(name, cfg, n, path):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    assert p.calls > 1000
    lr, hr, _ = make_reports(p, n, seed=len(name))
    alice = [(lr[i] if i % 2 == 0 else hr[i]).tobytes() for i in range(n)]
    bob = [(hr[i] if i % 2 == 0 else lr[i]).tobytes() for i in range(n)]
    x = bytearray(alice[1]); x[-40] ^= 1; alice[1] = bytes(x)        # a public proof element: wrong tag, still decodes
    alice.append(alice[0][:-3])                                       # truncated
    vdaf = W.Prio3(p)
    if path == "long_share":
        vdaf.set_option("chain_max_input_len", 0); vdaf.set_option("query_tiny", 0)
    try:
        for reports in (alice, bob):
            tags, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            outs = vdaf.out_shares()
            blob, offsets = blob_of(reports)
            o_tags, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=8)
            assert np.array_equal(status, o_status), (status, o_status)
            ok = status == 0
            assert ok.sum() >= n - 1
            assert np.array_equal(tags[ok], o_tags[ok])
            assert np.array_equal(ids[ok], o_ids[ok])
            assert np.array_equal(outs[ok], o_outs[ok])
            assert not tags[~ok].any() and not outs[~ok].any()
    finally:
        vdaf.close()


def test_one_call_more_than_fits_is_refused():
    p = W.Prio3Params(W.SUMVEC, 1, max_calls(1) + 1, 1)
    lr, hr, _ = make_reports(p, 2, seed=3)
    vdaf = W.Prio3(p)
    try:
        with pytest.raises(W.WhError) as e:
            vdaf.prepare_encoded_keys(VK, [lr[0].tobytes(), hr[1].tobytes()])
        assert e.value.code == -103 and "gadget calls" in str(e.value)      # WH_ERR_UNSUPPORTED
    finally:
        vdaf.close()
