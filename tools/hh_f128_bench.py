"""BASELINE.json's streaming heavy-hitter configuration over the Field128 ring (an extension: the reference's AggRing is u32):
masked mod-p sums of per-client sketch-shaped vectors of 16-byte counters, in batch_size slices added into a running sketch
(wh_hh_bucket_{sum,accumulate}_f128_dev), timed with HIP events on the context stream against the 8 TB/s HBM peak.
A sample of the counters is checked against Python integers."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import hh  # noqa: E402

P = 2**128 - 7 * 2**66 + 1
dev = torch.device("cuda:0")
out = {}
for name, (reps, buckets, n, batch) in {"hh_t0.01_f128": (14, 512, 240, 60), "hh_t0.001_f128": (17, 2048, 60, 20)}.items():
    words = reps * buckets * 258
    h = hh.HeavyHitters(buckets)
    ext = torch.cuda.ExternalStream(h.stream, device=dev)
    # residues below p: the top 32-bit word of every counter is kept under 2^31
    d_b = torch.randint(0, 2**31 - 1, (n, words, 4), dtype=torch.int32, device=dev)
    d_keep = torch.ones(n, dtype=torch.uint8, device=dev)
    d_keep[::50] = 0
    d_acc = torch.zeros((words, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ts = []
    for it in range(5):
        with torch.cuda.stream(ext):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for k, lo in enumerate(range(0, n, batch)):            # the streaming server's slices
                m = min(batch, n - lo)
                h.bucket_sum_f128_dev(d_b.data_ptr() + lo * words * 16, d_keep.data_ptr() + lo, m, words, d_acc.data_ptr(), accumulate=k > 0)
            e1.record(ext)
        h.synchronize()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    keep = d_keep.cpu().numpy().astype(bool)
    for w in (0, 1, words // 2, words - 1):
        col = d_b[:, w, :].cpu().numpy().view(np.uint32).astype(object)
        vals = [int(r[0]) | int(r[1]) << 32 | int(r[2]) << 64 | int(r[3]) << 96 for r in col]
        want = sum(v for v, k in zip(vals, keep) if k) % P
        a = d_acc[w].cpu().numpy().view(np.uint32).astype(object)
        assert int(a[0]) | int(a[1]) << 32 | int(a[2]) << 64 | int(a[3]) << 96 == want
    ms = min(ts[1:])
    nbytes = n * words * 16
    out[name] = {"repetitions": reps, "buckets": buckets, "clients": n, "batch_size": batch, "counters": words, "ms": ms,
                 "GB/s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000, "clients_per_s": n / ms * 1e3}
    h.close()
    del d_b, d_acc
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
