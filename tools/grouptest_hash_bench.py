"""Batch proofs of the group test's first split (common/src/grouptest.rs:182-208) over one million sorted tags: the d slices on
the GPU (wh_grouptest_split_proofs_dev, one lane per slice) against hashlib on one host core.  Prints one line per d."""
import hashlib
import sys
import time

import numpy as np
import torch

import whisper_amd as W
from whisper_amd.grouptest import DeviceHasher, U128_MAX

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
vk = bytes(range(16))
rng = np.random.default_rng(1)
for tag_bytes, tag_size in ((16, 16), (64, 32)):
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    v = ids.view(np.uint64).reshape(-1, 2)
    ids = ids[np.lexsort((v[:, 0], v[:, 1]))]
    tags = rng.integers(0, 256, (n, tag_bytes), dtype=np.uint8)
    ctx = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
    d_ids, d_tags = torch.from_numpy(ids).cuda(), torch.from_numpy(tags).cuda()
    torch.cuda.synchronize()
    h = DeviceHasher(ctx, d_ids.data_ptr(), d_tags.data_ptr(), n, tag_bytes, vk, tag_size)
    blob = tags.tobytes()
    t0 = time.perf_counter()
    for _ in range(3):
        (hashlib.shake_128(vk + blob).digest(16) if tag_size == 16 else hashlib.sha256(vk + blob).digest())
    host_ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"n={n} tag_bytes={tag_bytes} proof={tag_size}: host, one stream over all tags: {host_ms:.1f} ms "
          f"({n * tag_bytes / host_ms / 1e6:.2f} GB/s)", flush=True)
    for d in (1, 16, 64, 256, 512, 768, 1024, 1536, 2048, 4096, 16384, 65536):
        if d == 1 and n > 200_000:
            uppers = [int.from_bytes(ids[n // 50].tobytes(), "little")]       # a 2 % slice: one lane's speed
        else:
            q, r = divmod(U128_MAX, d)
            uppers = [(q + 1) * (i + 1) if i < r else r * (q + 1) + (i + 1 - r) * q for i in range(d)]
        h.split(uppers)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = h.split(uppers)
            times.append((time.perf_counter() - t0) * 1e3)
        ms = min(times)
        covered = sum(c for _, _, c in out)
        print(f"  d={d:6d}: GPU {ms:8.2f} ms (best of 5; all: {' '.join(f'{t:.2f}' for t in times)}) for {covered} tags "
              f"({covered * tag_bytes / ms / 1e6:.3f} GB/s)", flush=True)
    ctx.close()
