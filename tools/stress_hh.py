"""Randomised differential run of the heavy-hitter expansion (k_hh_main / support / chain / final) against the oracle (a tool,
not a test: minutes of GPU time): random depths 1..11, batch sizes around the persistent grid's boundaries (fewer key chains
than CUs, one more, several per CU), key chains with mutated headers, the bucket vectors, tags, testing ids and status of both
servers' views.  With `sketch` as third argument: the fused path (wh_hh_eval_and_sketch_batch_dev, sub-batches summed into
per-repetition sketches on a side stream; wh_hh_sketch_subtract_dev) with random repetition counts, group sizes, sub-batch
sizes, rejected sets and accumulation into a running sketch, against sums of the ORACLE's bucket vectors.
Usage on the GPU box: python tools/stress_hh.py [seconds] [seed] [sketch]"""
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
sketch_mode = len(sys.argv) > 3 and sys.argv[3] == "sketch"


def random_keys(depth, n):
    ln = hh.key_chain_len(depth)
    keys = np.zeros((n, ln), np.uint8)
    for i in range(n):
        bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32); bk[257] = 1
        a, b = hh.gen_key_chain(depth, int(rng.integers(0, 1 << 40)), int(rng.integers(0, 1 << depth)), bk,
                                rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
        keys[i] = a if rng.integers(0, 2) else b
        if rng.integers(0, 8) == 0:
            keys[i, int(rng.integers(0, min(ln, 200)))] = np.uint8(rng.integers(0, 256))     # a byte in the main key's header
    return keys


while sketch_mode and time.time() - t0 < budget:
    import torch
    depth = int(rng.choice([1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 9, 10, 11]))
    reps = int(rng.integers(1, 5))
    clients = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 65])) if depth <= 7 else int(rng.choice([1, 2, 5, 9]))
    n = reps * clients
    sub = rng.choice(["", "1", "2", "3", "5", "8", "13", "64"])
    if sub:
        os.environ["WH_HH_SUBBATCH"] = str(sub)
    else:
        os.environ.pop("WH_HH_SUBBATCH", None)
    keys = random_keys(depth, n)
    vk = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
    odat, otags, oids, ostatus = o.hh_eval_and_check_batch(depth, vk, keys, nthreads=16)
    odat = np.asarray(odat, dtype=np.uint32).reshape(n, 1 << depth, 258)
    odat[np.asarray(ostatus) != 0] = 0                      # a rejected key chain contributes nothing
    rejected = sorted(set(int(x) for x in rng.integers(0, n, int(rng.integers(0, 4)))))
    keep = np.ones(n, np.uint32); keep[rejected] = 0
    want_all = np.stack([odat[r * clients:(r + 1) * clients].sum(axis=0, dtype=np.uint32) for r in range(reps)])
    want_kept = np.stack([(odat[r * clients:(r + 1) * clients] * keep[r * clients:(r + 1) * clients, None, None]).sum(axis=0, dtype=np.uint32)
                          for r in range(reps)])
    dev = torch.device("cuda:0")
    N = 1 << depth
    h = hh.HeavyHitters(N)
    try:
        d_keys = torch.from_numpy(keys).to(dev)
        d_sk = torch.zeros((reps, N, 258), dtype=torch.int32, device=dev)
        d_tags = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
        d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
        d_status = torch.full((n,), 99, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        acc = bool(rng.integers(0, 2))
        if acc:                                               # a running sketch from an earlier slice
            prior = rng.integers(0, 1 << 32, (reps, N, 258), dtype=np.uint64).astype(np.uint32)
            d_sk.copy_(torch.from_numpy(prior.view(np.int32)).to(dev))
        else:
            prior = np.zeros((reps, N, 258), np.uint32)
            d_sk.fill_(12345)                                 # must be overwritten
        h.eval_and_sketch_dev(vk, d_keys.data_ptr(), keys.shape[1], n, clients, d_sk.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(),
                              d_status.data_ptr(), accumulate=acc)
        h.synchronize()
        got = d_sk.cpu().numpy().view(np.uint32)
        status = d_status.cpu().numpy()
        good = (np.array_equal(got, prior + want_all) and np.array_equal(status, np.asarray(ostatus))
                and [bytes(t) for t in d_tags.cpu().numpy()] == [bytes(t) for t in otags]
                and [bytes(i) for i in d_ids.cpu().numpy()] == [bytes(i) for i in oids])
        if good and rejected:
            h.sketch_subtract_dev(vk, d_keys.data_ptr(), keys.shape[1], rejected, clients, d_sk.data_ptr())
            h.synchronize()
            good = np.array_equal(d_sk.cpu().numpy().view(np.uint32), prior + want_kept)
        if not good:
            print("MISMATCH sketch", dict(depth=depth, reps=reps, clients=clients, sub=str(sub), acc=acc, rejected=rejected), flush=True)
            sys.exit(1)
    finally:
        h.close()
    rounds += 1; checked += n
    if rounds % 10 == 0:
        print(f"{rounds} batches, {checked} key chains, {time.time() - t0:.0f} s", flush=True)
if sketch_mode:
    print(f"ok (sketch path): {rounds} batches, {checked} key chains: sketches, tags, ids, status against the oracle in {time.time() - t0:.0f} s")
    sys.exit(0)
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
