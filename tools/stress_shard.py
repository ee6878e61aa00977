"""Randomised differential run of the GPU sharder / prover and the flp::Type surface (a tool, not a test: minutes of GPU time).
Random circuits and shapes (SumVec bits to 64, vectors to 300, histograms to 3 000, chunk lengths 1 ... 1 000 or around
sqrt(input_len), Average to 64 bits), batches of 1 / 2 / 65 / 130 clients:
  * wh_client_shard_many_gpu against the CPU client byte for byte, and a sample of clients against the oracle's shard_batched;
  * wh_flp_prove_batch, wh_flp_query_batch, wh_flp_decide_batch on random inputs (valid and not) against the oracle's
    prove / query / decide.
The only refusal allowed is the documented one (more than 511 gadget calls for the GPU prover, include/whisper_gfx950.h).
Usage on the GPU box: python tools/stress_shard.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O          # noqa: E402  (the checker)
import whisper_amd as W         # noqa: E402
from common import det          # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
rounds = clients = tuples = refused = 0


def chunk_for(input_len):
    k = int(rng.integers(0, 4))
    if k == 0:
        return int(rng.integers(1, 11))
    if k == 1:
        return int(rng.choice([63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 512, 1000]))
    root = max(1, int(input_len ** 0.5))
    return int(rng.integers(max(1, root // 2), 2 * root + 2))


def fail(what, **kw):
    print("MISMATCH", what, kw, flush=True)
    sys.exit(1)


while time.time() - t0 < budget:
    circ = int(rng.integers(0, 3))
    if circ == W.SUMVEC:
        bits = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64])); ln = int(rng.integers(1, 301)); ch = chunk_for(bits * ln)
    elif circ == W.HISTOGRAM:
        bits = 0; ln = int(rng.integers(2, 3001)); ch = chunk_for(ln)
    else:
        bits = int(rng.integers(1, 65)); ln = ch = 0
    try:
        p = W.Prio3Params(circ, bits, ln, ch)
    except W.WhError:
        continue
    if p.calls > 1200:
        continue
    op = O.params(circ, bits, ln, ch)
    shape = dict(circ=circ, bits=bits, ln=ln, ch=ch, calls=p.calls)
    n = int(rng.choice([1, 2, 65, 130]))
    if p.input_len > 4000:
        n = min(n, 2)
    seed = int(rng.integers(0, 1 << 30))
    tag = b"ss%d" % seed
    if circ == W.SUMVEC:
        meas = np.frombuffer(det(tag, 8 * ln * n), np.uint64).reshape(n, ln)
        meas = meas if bits == 64 else meas & np.uint64((1 << bits) - 1)
    else:
        raw = np.frombuffer(det(tag, 8 * n), np.uint64).reshape(n, 1)
        meas = raw % np.uint64(ln) if circ == W.HISTOGRAM else (raw if bits == 64 else raw & np.uint64((1 << bits) - 1))
    nonces = np.frombuffer(det(tag + b"n", 16 * n), np.uint8).reshape(n, 16)
    rands = np.frombuffer(det(tag + b"r", 144 * n), np.uint8).reshape(n, 144)
    first = int(rng.integers(0, 1 << 60))
    vdaf = W.Prio3(p)
    try:
        # ---- the sharder
        try:
            lr, hr = vdaf.shard_batched_many(meas, nonces, rands, first)
        except W.WhError as e:
            if e.code == -103 and p.calls > 511:
                refused += 1
                lr = None
            else:
                print("REFUSED", shape, e, flush=True)
                sys.exit(1)
        if lr is not None:
            lr_cpu, hr_cpu = W.shard_batched_many(p, meas, nonces, rands, first, nthreads=8)
            if not (np.array_equal(lr, lr_cpu) and np.array_equal(hr, hr_cpu)):
                fail("GPU sharder against the CPU client", **shape, n=n, seed=seed)
            i = int(rng.integers(0, n))
            want = O.prio3_shard_batched(op, [int(x) for x in meas[i]], nonces[i].tobytes(), rands[i].tobytes(), first + i)
            if lr[i].tobytes() != want[0] or hr[i].tobytes() != want[1]:
                fail("sharder against the oracle", **shape, n=n, seed=seed, i=i)
            clients += n
        # ---- prove / query / decide on random tuples: a valid encoding, and one with a flipped element
        k = min(n, 3)
        s = p.sizes
        inputs, prs, jrs, qrs = [], [], [], []
        for j in range(k):
            enc = O.flp_encode(op, [int(x) for x in meas[j]])
            if j == 1:
                e2 = bytearray(enc); e2[0] ^= 2; enc = bytes(e2)          # not a valid encoding any more: decide must say so
            inputs.append(enc)
            prs.append(b"".join(O.f_enc(int.from_bytes(bytes(det(tag + b"p%d-%d" % (j, q), 16)), "little") % O.P128) for q in range(s.arity)))
            jrs.append(b"".join(O.f_enc(int.from_bytes(bytes(det(tag + b"j%d-%d" % (j, q), 16)), "little") % O.P128) for q in range(s.joint_rand_len)))
            qrs.append(O.f_enc(int.from_bytes(bytes(det(tag + b"q%d" % j, 16)), "little") % O.P128))
        a_in = np.frombuffer(b"".join(inputs), np.uint8).reshape(k, -1)
        a_pr = np.frombuffer(b"".join(prs), np.uint8).reshape(k, -1)
        a_jr = np.frombuffer(b"".join(jrs), np.uint8).reshape(k, -1) if s.joint_rand_len else np.zeros((k, 0), np.uint8)
        a_qr = np.frombuffer(b"".join(qrs), np.uint8).reshape(k, 16)
        want_proofs = [O.flp_prove(op, inputs[j], prs[j], jrs[j]) for j in range(k)]
        if p.calls <= 511:
            proofs, st = vdaf.flp_prove(a_in, a_pr, a_jr)
            if st.any() or any(proofs[j].tobytes() != want_proofs[j] for j in range(k)):
                fail("flp_prove", **shape, seed=seed)
        a_proofs = np.frombuffer(b"".join(want_proofs), np.uint8).reshape(k, -1)
        ver, st = vdaf.flp_query(a_in, a_proofs, a_qr, a_jr, 1)
        want_ver = [O.flp_query(op, inputs[j], want_proofs[j], qrs[j], jrs[j], 1) for j in range(k)]
        for j in range(k):
            rc, wv = want_ver[j] if isinstance(want_ver[j], tuple) else (0, want_ver[j])
            if int(st[j]) != rc or (rc == 0 and ver[j].tobytes() != wv):
                fail("flp_query", **shape, seed=seed, j=j, status=int(st[j]), want=rc)
        ok = vdaf.flp_decide(ver)
        for j in range(k):
            rc, wv = want_ver[j] if isinstance(want_ver[j], tuple) else (0, want_ver[j])
            if rc == 0 and bool(ok[j]) != bool(O.flp_decide(op, wv)):
                fail("flp_decide", **shape, seed=seed, j=j)
            if rc == 0 and bool(ok[j]) != (j != 1):
                fail("flp_decide against the encoding's validity", **shape, seed=seed, j=j)
        tuples += k
    finally:
        vdaf.close()
    rounds += 1
    if rounds % 20 == 0:
        print(f"{rounds} shapes, {clients} clients, {tuples} tuples, {time.time() - t0:.0f} s", flush=True)
print(f"ok: {rounds} shapes, {clients} clients sharded on the GPU identical to the CPU client (one per shape also to the oracle), "
      f"{tuples} prove/query/decide tuples identical to the oracle in {time.time() - t0:.0f} s "
      f"({refused} shapes of more than 511 gadget calls refused by the GPU sharder)")
