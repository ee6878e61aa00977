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


PROVE_CASES = CASES + [
    ("sumvec_large", (W.SUMVEC, 16, 1024, 128)),     # m = 256, 256 wires: BASELINE.json's headline shape
    ("sumvec_m512", (W.SUMVEC, 4, 700, 8)),          # 350 calls -> m = 512, the largest the kernel's LDS takes
    ("sumvec_tiny", (W.SUMVEC, 2, 5, 3)),            # m = 8: fewer points than lanes
    ("histogram_2_1", (W.HISTOGRAM, 0, 2, 1)),       # one wire pair
]


@pytest.mark.parametrize("name,cfg", PROVE_CASES, ids=[c[0] for c in PROVE_CASES])
def test_gpu_prove_matches_oracle(name, cfg):
    """Type::prove on the GPU (wh_flp_prove_batch) against the oracle's prove (pinned by the draft-07 shard vectors) and
    the CPU client's, bit for bit: valid encodings, arbitrary field elements as input (prove does not need a valid
    measurement), and a non-canonical element."""
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    s = p.sizes
    vdaf = W.Prio3(p)
    try:
        n = 5
        inputs, prs, jrs = [], [], []
        for i in range(n):
            if i < 2:
                if circ == W.SUMVEC:
                    meas = [int(x) & ((1 << bits) - 1) for x in np.frombuffer(det(b"gp%d" % i, 8 * ln), np.uint64)]
                elif circ == W.HISTOGRAM:
                    meas = [(3 * i + 1) % ln]
                else:
                    meas = [int.from_bytes(bytes(det(b"gp%d" % i, 8)), "little") & 0xffffffff]
                inp = W.flp_encode(p, meas).tobytes()
            else:
                inp = rand_elems(b"gx%d" % i, s.input_len)
            inputs.append(inp); prs.append(rand_elems(b"gr%d" % i, s.arity)); jrs.append(rand_elems(b"gj%d" % i, s.joint_rand_len))
        inputs[4] = inputs[4][:16 * (s.input_len - 1)] + (O.P128 + 1).to_bytes(16, "little")     # not a canonical residue
        arr = lambda rows: np.frombuffer(b"".join(rows), np.uint8).reshape(len(rows), -1)
        proofs, status = vdaf.flp_prove(arr(inputs), arr(prs), arr(jrs))
        assert [int(x) for x in status] == [0, 0, 0, 0, -3]
        assert not proofs[4].any()
        for i in range(4):
            want = O.flp_prove(op, inputs[i], prs[i], jrs[i])
            assert proofs[i].tobytes() == want, i
        cpu = W.flp_prove(p, np.frombuffer(inputs[0], np.uint8), np.frombuffer(prs[0], np.uint8), np.frombuffer(jrs[0], np.uint8))
        assert proofs[0].tobytes() == cpu.tobytes()
        # and the proof of a valid input verifies on the GPU
        qr = rand_elems(b"gq", 1)
        ver, st = vdaf.flp_query(arr(inputs[:2]), proofs[:2], arr([qr, qr]), arr(jrs[:2]), 1)
        assert not st.any() and all(vdaf.flp_decide(ver))
    finally:
        vdaf.close()


# ---- discrete_fourier_transform / inv_finish (fft.rs:31-115) as a stand-alone batch entry point ----
def _elems(tag, n):
    raw = bytes(det(tag.encode(), 16 * n + 16))
    return [int.from_bytes(raw[16 * i:16 * i + 16], "little") % O.P128 for i in range(n)]


@pytest.mark.parametrize("size,n_in", [(1, 1), (2, 2), (4, 3), (64, 64), (64, 17), (256, 129), (512, 512), (1024, 511), (4096, 4096), (8, 0)])
def test_ntt_batch_matches_oracle_fft(size, n_in):
    """wh_ntt_f128_batch against the oracle's restatement of fft.rs, forward and inverse, zero-padded inputs, and the algebra
    the reference's own test checks (fft.rs:131-160): inverse(forward(x)) = x."""
    vdaf = W.Prio3(W.Prio3Params(W.SUMVEC, 16, 16, 8))
    try:
        n = 3
        rows = [_elems(f"ntt-{size}-{n_in}-{r}", n_in) for r in range(n)]
        rows[1] = [O.P128 - 1] * n_in if n_in else []              # the largest residue everywhere
        v = np.frombuffer(b"".join(O.f_enc(x) for row in rows for x in row), np.uint8).reshape(n, n_in, 16)
        fwd = vdaf.ntt(v, size)
        inv = vdaf.ntt(v, size, inverse=True)
        for r in range(n):
            assert [O.f_dec(fwd[r, i].tobytes()) for i in range(size)] == O.fft(rows[r], size)
            assert [O.f_dec(inv[r, i].tobytes()) for i in range(size)] == O.fft(rows[r], size, inverse=True)
        back = vdaf.ntt(fwd, size, inverse=True)
        for r in range(n):
            assert [O.f_dec(back[r, i].tobytes()) for i in range(size)] == rows[r] + [0] * (size - n_in)
    finally:
        vdaf.close()


def test_ntt_batch_definition_and_errors():
    """Against the definition out[i] = sum_j in[j] w^(ij) with Python integers (w = the reference's root for that size), and
    the error classes: a non-canonical element, a size that is no power of two, more inputs than points, sizes past the limits."""
    vdaf = W.Prio3(W.Prio3Params(W.HISTOGRAM, 0, 4, 2))
    try:
        size, n_in = 16, 11
        x = _elems("ntt-def", n_in)
        w = O.f_root(4)
        assert pow(w, 16, O.P128) == 1 and pow(w, 8, O.P128) != 1
        v = np.frombuffer(b"".join(O.f_enc(e) for e in x), np.uint8).reshape(1, n_in, 16)
        got = vdaf.ntt(v, size)
        assert [O.f_dec(got[0, i].tobytes()) for i in range(size)] == [sum(x[j] * pow(w, i * j, O.P128) for j in range(n_in)) % O.P128 for i in range(size)]
        bad = v.copy()
        bad[0, 5] = np.frombuffer(int(O.P128).to_bytes(16, "little"), np.uint8)          # p itself
        for args, code in (((bad, 16), -3), ((v, 24), -1), ((v, 8), -1), ((v, 1 << 21), -1), ((v, 8192), -103)):
            with pytest.raises(W.WhError) as e:
                vdaf.ntt(*args)
            assert e.value.code == code
        assert vdaf.ntt(np.zeros((0, 4, 16), np.uint8), 8).shape == (0, 8, 16)
    finally:
        vdaf.close()
