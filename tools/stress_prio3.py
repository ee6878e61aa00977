"""Randomised differential run of the batched Prio3 path against the oracle (a tool, not a test: minutes of GPU time).
Random circuits and shapes, 2..4 aggregators, batches that mix aggregator ids in random order and random sizes around the
wavefront boundaries, truncated / mutated reports, both pipelines; with `base` as third argument the un-batched (draft-07)
prepare_init path with keys of mixed aggregator ids; with `wide` (`widebase` for the un-batched path) shapes from far wider ranges (bits to 127, vectors to 400,
histograms to 5 000, chunk lengths from 1 to 1 000) in small batches, where the only refusal allowed is the documented one
(more gadget calls than the query kernel's 160 KiB of LDS rows hold).  Half of the batches run with random settings of the
context's launch-shape knobs (sub-batch size, side streams, prefetch, H2D chunk), on which results must not depend.
Usage on the GPU box: python tools/stress_prio3.py [seconds] [seed] [base|wide|widebase]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O          # noqa: E402  (the checker)
import whisper_amd as W         # noqa: E402
from common import VK, blob_of, det, measurements   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
rounds = reports_checked = 0
base_mode = len(sys.argv) > 3 and sys.argv[3] in ("base", "widebase")
wide_mode = len(sys.argv) > 3 and sys.argv[3] in ("wide", "widebase")
refused = 0


def wide_chunk(input_len):
    k = int(rng.integers(0, 4))
    if k == 0:
        return int(rng.integers(1, 11))
    if k == 1:
        return int(rng.choice([63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 512, 1000]))
    root = max(1, int(input_len ** 0.5))
    return int(rng.integers(max(1, root // 2), 2 * root + 2))


def random_knobs(vdaf):
    """Results never depend on a knob (include/whisper_gfx950.h): draw some."""
    if rng.integers(0, 2):
        return
    vdaf.set_option("max_subbatch", int(rng.choice([0, 64, 65, 100, 128])))
    for name in ("side_stream", "hash_b_side", "third_stream", "base_paired", "chain_prefetch"):
        vdaf.set_option(name, int(rng.integers(0, 2)))
    vdaf.set_option("host_chunk_mb", int(rng.choice([1, 2, 2048])))


def calls_limit(chunk):           # include/whisper_gfx950.h, wh_prio3_prepare_batch_dev
    return 3228 if chunk <= 64 else 3047 if chunk <= 128 else 2684


while base_mode and time.time() - t0 < budget:
    circ = int(rng.integers(0, 3))
    na = int(rng.choice([2, 3, 5, 17, 254])) if wide_mode else int(rng.integers(2, 6))
    if wide_mode and circ == W.SUMVEC:
        bits = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 100, 127])); ln = int(rng.integers(1, 201)); ch = wide_chunk(bits * ln)
    elif wide_mode and circ == W.HISTOGRAM:
        bits = 0; ln = int(rng.integers(2, 3001)); ch = wide_chunk(ln)
    elif wide_mode:
        bits = int(rng.integers(1, 128)); ln = ch = 0
    elif circ == W.SUMVEC:
        bits = int(rng.choice([1, 3, 8, 16])); ln = int(rng.integers(1, 24)); ch = int(rng.integers(1, 40))
    elif circ == W.HISTOGRAM:
        bits = 0; ln = int(rng.integers(2, 100)); ch = int(rng.integers(1, 50))
    else:
        bits = int(rng.integers(1, 100)); ln = ch = 0
    try:
        p = W.Prio3Params(circ, bits, ln, ch, na)
    except W.WhError:
        continue
    if wide_mode and p.calls > 4000:
        continue
    op = O.params(circ, bits, ln, ch, na)
    n = int(rng.choice([1, 3, 65])) if wide_mode else int(rng.choice([1, 63, 65, 130, 200]))
    if wide_mode and p.input_len > 4000:
        n = min(n, 3)
    seed = int(rng.integers(0, 1 << 30))
    meas = measurements(p, n, seed)
    vk = bytes(det(b"bvk%d" % seed, 16))
    keys, want = [], []
    for i in range(n):
        nonce = bytes(det(b"bn%d-%d" % (seed, i), 16))
        rand = bytes(det(b"br%d-%d" % (seed, i), O.prio3_rand_size(op)))
        ps, ish = O.prio3_shard(op, [int(x) for x in meas[i]], nonce, rand)
        a = int(rng.integers(0, na))
        key = W.Prio3.encode_vdaf_key(a, ps, ish[a], nonce)
        if rng.integers(0, 10) == 0:
            key = key[: int(rng.integers(0, len(key)))]
            want.append(None)
        else:
            rc, wp, wo = O.prio3_prepare_init(op, vk, a, nonce, ps, ish[a])
            want.append((rc, wp, wo))
        keys.append(key)
    vdaf = W.Prio3(p)
    random_knobs(vdaf)
    if rng.integers(0, 3) == 0:
        vdaf.set_option("chain_max_input_len", 0)
    try:
        try:
            prep, status = vdaf.prepare_init_encoded_keys(vk, keys)
        except W.WhError as e:
            if e.code == -103 and circ != W.AVERAGE and p.calls > calls_limit(ch):     # the documented refusal
                refused += 1
                continue
            print("REFUSED base", dict(circ=circ, bits=bits, ln=ln, ch=ch, na=na, n=n, seed=seed, calls=p.calls), e, flush=True)
            sys.exit(1)
        outs = vdaf.out_shares()
        for j in range(n):
            if want[j] is None:
                bad = status[j] == 0
            else:
                bad = status[j] != want[j][0] or (status[j] == 0 and (prep[j].tobytes() != want[j][1] or outs[j].tobytes() != want[j][2]))
            if bad:
                print("MISMATCH base", dict(circ=circ, bits=bits, ln=ln, ch=ch, na=na, n=n, seed=seed, j=j, status=int(status[j])), flush=True)
                sys.exit(1)
    finally:
        vdaf.close()
    rounds += 1; reports_checked += n
    if rounds % 20 == 0:
        print(f"{rounds} batches, {reports_checked} keys, {time.time() - t0:.0f} s", flush=True)
if base_mode:
    print(f"ok (un-batched): {rounds} batches, {reports_checked} keys checked against the oracle in {time.time() - t0:.0f} s"
          + (f" ({refused} shapes beyond the documented gadget-call limit refused)" if wide_mode else ""))
    sys.exit(0)
while time.time() - t0 < budget:
    circ = int(rng.integers(0, 3))
    na = int(rng.choice([2, 2, 3, 4, 9, 64, 254])) if wide_mode else int(rng.choice([2, 2, 3, 4]))
    if wide_mode and circ == W.SUMVEC:
        bits = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 100, 127])); ln = int(rng.integers(1, 401)); ch = wide_chunk(bits * ln)
    elif wide_mode and circ == W.HISTOGRAM:
        bits = 0; ln = int(rng.integers(2, 5001)); ch = wide_chunk(ln)
    elif wide_mode:
        bits = int(rng.integers(1, 128)); ln = ch = 0
    elif circ == W.SUMVEC:
        bits = int(rng.choice([1, 2, 5, 8, 16, 33])); ln = int(rng.integers(1, 40)); ch = int(rng.integers(1, 70))
    elif circ == W.HISTOGRAM:
        bits = 0; ln = int(rng.integers(2, 200)); ch = int(rng.integers(1, 80))
    else:
        bits = int(rng.integers(1, 120)); ln = ch = 0
    try:
        p = W.Prio3Params(circ, bits, ln, ch, num_aggregators=na)
    except W.WhError:
        continue
    if wide_mode and p.calls > 4000:          # far beyond the limit (and the context's tables grow with calls^2)
        continue
    op = O.params(circ, bits, ln, ch, num_aggregators=na)
    n = int(rng.choice([1, 3, 65])) if wide_mode else int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 200, 257, 300]))
    if wide_mode and p.input_len > 8000:
        n = min(n, 3)
    seed = int(rng.integers(0, 1 << 30))
    meas = measurements(p, n, seed)
    nonces = det(b"sn%d" % seed, 16 * n).reshape(n, 16)
    rands = det(b"sr%d" % seed, 16 * (5 * na - 1) * n).reshape(n, -1)
    lr, hr = W.shard_batched_many(p, meas, nonces, rands, seed, 8)
    hr = hr.reshape(n, na - 1, -1)
    roles = rng.integers(0, na, n)
    reports = []
    for i in range(n):
        r = lr[i].tobytes() if roles[i] == 0 else hr[i, roles[i] - 1].tobytes()
        k = rng.integers(0, 12)
        if k == 0:
            r = r[: int(rng.integers(0, len(r)))]                       # truncated
        elif k == 1:
            x = bytearray(r); x[int(rng.integers(0, len(x)))] ^= 1 << int(rng.integers(0, 8)); r = bytes(x)   # one bit anywhere
        elif k == 2:
            x = bytearray(r); pos = int(rng.integers(0, max(1, len(x) - 16))); x[pos:pos + 16] = b"\xff" * 16; r = bytes(x)
        reports.append(r)
    blob, offsets = blob_of(reports)
    o_t, o_ids, o_status, o_outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=16)
    vdaf = W.Prio3(p)
    random_knobs(vdaf)
    try:
        for path in ("default", "long_share"):
            if path == "long_share":
                vdaf.set_option("chain_max_input_len", 0); vdaf.set_option("query_tiny", 0)
            try:
                t, ids, status = vdaf.prepare_encoded_keys(VK, reports)
            except W.WhError as e:
                if e.code == -103 and circ != W.AVERAGE and p.calls > calls_limit(ch):     # the documented refusal
                    refused += 1
                    break
                print("REFUSED", dict(circ=circ, bits=bits, ln=ln, ch=ch, na=na, n=n, seed=seed, path=path, calls=p.calls), e, flush=True)
                sys.exit(1)
            outs = vdaf.out_shares()
            ok = status == 0
            good = (np.array_equal(status, o_status) and np.array_equal(t[ok], o_t[ok]) and np.array_equal(ids[ok], o_ids[ok])
                    and np.array_equal(outs[ok], o_outs[ok]) and not t[~ok].any() and not outs[~ok].any())
            if not good:
                print("MISMATCH", dict(circ=circ, bits=bits, ln=ln, ch=ch, na=na, n=n, seed=seed, path=path), flush=True)
                print(" status", status.tolist(), o_status.tolist(), flush=True)
                sys.exit(1)
    finally:
        vdaf.close()
    rounds += 1; reports_checked += 2 * n
    if rounds % 20 == 0:
        print(f"{rounds} batches, {reports_checked} reports, {time.time() - t0:.0f} s", flush=True)
print(f"ok: {rounds} batches, {reports_checked} reports checked against the oracle in {time.time() - t0:.0f} s"
      + (f" ({refused} shapes beyond the documented gadget-call limit refused)" if wide_mode else ""))
