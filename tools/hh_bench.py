"""HBM-bound side of the path: heavy-hitter bucket sums (server-hh/src/main.rs:72-104) and the masked Prio3
out-share aggregation, timed with torch events against the 8 TB/s HBM peak."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whisper_amd as W

dev = torch.device("cuda:0")
vdaf = W.new_sum_vec_256(2, 16, 1024, 128)
ext = torch.cuda.ExternalStream(vdaf.stream, device=dev)
out = {}
for name, (reps, buckets, n) in {"hh_t0.01": (14, 512, 1000), "hh_t0.001": (17, 2048, 250)}.items():
    words = reps * buckets * 258
    d_b = torch.randint(0, 2**31 - 1, (n, words), dtype=torch.int32, device=dev)
    d_keep = torch.ones(n, dtype=torch.uint8, device=dev); d_keep[::100] = 0
    d_acc = torch.zeros(words, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ts = []
    for it in range(6):
        d_acc.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(ext):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            vdaf.hh_bucket_sum_u32_dev(d_b.data_ptr(), d_keep.data_ptr(), n, words, d_acc.data_ptr())
            e1.record(ext)
        vdaf.synchronize(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    want = (d_b[d_keep.bool()].to(torch.int64).sum(0) & 0xffffffff)
    assert torch.equal(want, d_acc.to(torch.int64) & 0xffffffff)
    ms = min(ts[1:])
    out[name] = {"clients": n, "words": words, "ms": ms, "GB/s": n * words * 4 / ms / 1e6, "frac_of_8TBs": n * words * 4 / ms / 1e6 / 8000}
    del d_b
print(json.dumps(out, indent=1))
