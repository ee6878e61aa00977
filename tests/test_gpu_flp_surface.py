"""The flp::Type surface (batched-flpcp-rs/src/flp.rs:186-196) through the C ABI: encode and prove on the CPU,
query / truncate / decide on the GPU, against the oracle's restatement (which the draft-07 vectors pin).  After the
reference's own flp_validity_test (flp/types.rs:1247-1410): a valid input's proof verifies, mutated inputs and
proofs do not, and every intermediate value agrees bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import det

pytestmark = pytest.mark.gpu

CASES = [
    ("sumvec_16_16_8", (W.SUMVEC, 16, 16, 8)),
    ("sumvec_cfg1", (W.SUMVEC, 16, 128, 64)),
    ("sumvec_chunk128", (W.SUMVEC, 8, 100, 128)),
    ("sumvec_ragged", (W.SUMVEC, 3, 11, 7)),
    ("histogram_128_64", (W.HISTOGRAM, 0, 128, 64)),
    ("histogram_11_3", (W.HISTOGRAM, 0, 11, 3)),
    ("average_64", (W.AVERAGE, 64, 0, 0)),
]


def rand_elems(tag, n):
    raw = bytes(det(tag, 16 * n + 16))
    return b"".join(O.f_enc(int.from_bytes(raw[16 * i:16 * i + 16], "little") % O.P128) for i in range(n))


@pytest.mark.parametrize("name,cfg", CASES, ids=[c[0] for c in CASES])
def test_prove_query_decide_match_oracle(name, cfg):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    s = p.sizes
    vdaf = W.Prio3(p)
    try:
        inputs, proofs, qrs, jrs, expect_ok = [], [], [], [], []
        for i in range(6):
            if circ == W.SUMVEC:
                meas = [int(x) & ((1 << bits) - 1) for x in np.frombuffer(det(b"fs%d" % i, 8 * ln), np.uint64)]
            elif circ == W.HISTOGRAM:
                meas = [(5 * i + 2) % ln]
            else:
                meas = [int.from_bytes(bytes(det(b"fs%d" % i, 8)), "little") & 0xffffffff]
            inp = W.flp_encode(p, meas).tobytes()
            assert inp == O.flp_encode(op, meas)
            prove_rand = rand_elems(b"pr%d" % i, s.arity)
            jr = rand_elems(b"jr%d" % i, s.joint_rand_len)
            proof = W.flp_prove(p, np.frombuffer(inp, np.uint8), np.frombuffer(prove_rand, np.uint8), np.frombuffer(jr, np.uint8)).tobytes()
            assert proof == O.flp_prove(op, inp, prove_rand, jr)
            ok = True
            if i == 3:      # an input that is not a valid encoding (a "bit" equal to 2)
                inp = O.f_enc(2) + inp[16:]
                ok = False
            if i == 4:      # a corrupted proof
                proof = proof[:32] + O.f_enc(12345) + proof[48:]
                ok = False
            inputs.append(inp); proofs.append(proof); qrs.append(rand_elems(b"qr%d" % i, 1)); jrs.append(jr); expect_ok.append(ok)
        # one query randomness that is an m-th root of unity: FlpError::Query
        qrs[5] = O.f_enc(O.f_root(max(1, s.m.bit_length() - 1)))
        arr = lambda rows: np.frombuffer(b"".join(rows), np.uint8).reshape(len(rows), -1)
        ver, status = vdaf.flp_query(arr(inputs), arr(proofs), arr(qrs), arr(jrs), 1)
        outs = vdaf.out_shares()
        for i in range(6):
            rc, want = O.flp_query(op, inputs[i], proofs[i], qrs[i], jrs[i], 1)
            assert status[i] == rc, (i, status[i], rc)
            if rc == 0:
                assert ver[i].tobytes() == want
                assert outs[i].tobytes() == O.flp_truncate(op, inputs[i])
        assert status[5] == -6 and not status[:5].any()
        dec = vdaf.flp_decide(ver[:5])
        for i in range(5):
            assert bool(dec[i]) == O.flp_decide(op, ver[i].tobytes()) == expect_ok[i], i
        # two shares of the same input and proof: the verifier shares add up to the verifier (linearity), num_shares = 2
        x = [O.f_dec(inputs[0][16 * e:16 * e + 16]) for e in range(s.input_len)]
        pf = [O.f_dec(proofs[0][16 * e:16 * e + 16]) for e in range(s.proof_len)]
        x0 = [O.f_dec(rand_elems(b"sx", s.input_len)[16 * e:16 * e + 16]) for e in range(s.input_len)]
        p0 = [O.f_dec(rand_elems(b"sp", s.proof_len)[16 * e:16 * e + 16]) for e in range(s.proof_len)]
        enc = lambda v: b"".join(O.f_enc(e % O.P128) for e in v)
        sh_in = [enc(x0), enc([(a - b) for a, b in zip(x, x0)])]
        sh_pf = [enc(p0), enc([(a - b) for a, b in zip(pf, p0)])]
        ver2, st2 = vdaf.flp_query(arr(sh_in), arr(sh_pf), arr([qrs[0]] * 2), arr([jrs[0]] * 2), 2)
        assert not st2.any()
        for k in range(2):
            assert ver2[k].tobytes() == O.flp_query(op, sh_in[k], sh_pf[k], qrs[0], jrs[0], 2)[1]
        total = b"".join(O.f_enc((O.f_dec(ver2[0][16 * e:16 * e + 16].tobytes()) + O.f_dec(ver2[1][16 * e:16 * e + 16].tobytes())) % O.P128)
                         for e in range(s.verifier_len))
        assert total == ver[0].tobytes()
    finally:
        vdaf.close()
