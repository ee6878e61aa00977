"""Randomised differential run of the heavy-hitter expansion (k_hh_main / support / chain / final) against the oracle (a tool,
not a test: minutes of GPU time): random depths 1..11, batch sizes around the persistent grid's boundaries (fewer key chains
than CUs, one more, several per CU), key chains with mutated headers, the bucket vectors, tags, testing ids and status of both
servers' views.  Usage on the GPU box: python tools/stress_hh.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as o          # noqa: E402  (the checker)
from whisper_amd import hh      # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
rounds = checked = 0
while time.time() - t0 < budget:
    depth = int(rng.integers(1, 12))
    n = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 300, 513, 700])) if depth <= 7 else int(rng.choice([1, 2, 5, 33, 257, 300]))
    ln = hh.key_chain_len(depth)
    keys = np.zeros((n, ln), np.uint8)
    for i in range(n):
        bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32); bk[257] = 1
        a, b = hh.gen_key_chain(depth, int(rng.integers(0, 1 << 40)), int(rng.integers(0, 1 << depth)), bk,
                                rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
        keys[i] = a if rng.integers(0, 2) else b
        k = rng.integers(0, 10)
        if k == 0:
            keys[i, int(rng.integers(0, ln))] ^= np.uint8(1 << int(rng.integers(0, 8)))     # one bit anywhere (a header byte or a payload byte)
        elif k == 1:
            keys[i, int(rng.integers(0, min(ln, 200)))] = np.uint8(rng.integers(0, 256))     # a byte in the main key's header
    vk = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
    h = hh.HeavyHitters(1 << depth)
    try:
        tags, ids, status = h.eval_and_check(vk, keys)
        dat = h.buckets()
        odat, otags, oids, ostatus = o.hh_eval_and_check_batch(depth, vk, keys, nthreads=16)
        good = ((status == ostatus).all()
                and [t.encode() if s == 0 else bytes(64) for t, s in zip(tags, status)] == [bytes(t) for t in otags]
                and ids == [int.from_bytes(bytes(i), "little") for i in oids] and (dat == odat).all())
        if not good:
            print("MISMATCH", dict(depth=depth, n=n), status.tolist()[:20], ostatus.tolist()[:20], flush=True)
            sys.exit(1)
    finally:
        h.close()
    rounds += 1; checked += n
    if rounds % 10 == 0:
        print(f"{rounds} batches, {checked} key chains, {time.time() - t0:.0f} s", flush=True)
print(f"ok: {rounds} batches, {checked} key chains checked against the oracle in {time.time() - t0:.0f} s")
