"""Does the verify step gain from running sub-batches concurrently?  L contexts (one stream each) on one GPU, each with
n/L reports of the sumvec_large workload; all launched back to back per step.  python tools/overlap_probe.py [n]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import whisper_amd as W
from common import VK, make_reports

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
params = W.Prio3Params(W.SUMVEC, 16, 1024, 128)
s = params.sizes
dev = torch.device("cuda", 0)
K = 64
lr, hr, _ = make_reports(params, 2 * K, seed=5, nthreads=16)
tile_reports = [(lr[i] if i % 2 == 0 else hr[i]) for i in range(2 * K)]
tile = np.concatenate(tile_reports)
tile_off = np.zeros(2 * K + 1, np.int64); tile_off[1:] = np.cumsum([len(r) for r in tile_reports])
for lanes in (1, 2, 3, 4):
    m = n // lanes
    reps = (m + 2 * K - 1) // (2 * K)
    d_blob = torch.from_numpy(tile).to(dev).repeat(reps)
    offs = (torch.from_numpy(tile_off[:-1]).to(dev)[None, :] + (torch.arange(reps, device=dev, dtype=torch.int64) * int(tile_off[-1]))[:, None]).reshape(-1)
    d_off = torch.cat([offs, torch.tensor([reps * int(tile_off[-1])], device=dev, dtype=torch.int64)])[: m + 1].contiguous()
    ctxs = [W.Prio3(params, device=0) for _ in range(lanes)]
    bufs = [(torch.zeros(16 * m, dtype=torch.uint8, device=dev), torch.zeros(16 * m, dtype=torch.uint8, device=dev),
             torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(16 * s.output_len, dtype=torch.uint8, device=dev)) for _ in range(lanes)]
    def step():
        for c, (t, i, st, a) in zip(ctxs, bufs):
            c.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), m, t.data_ptr(), i.data_ptr(), st.data_ptr())
        for c, (t, i, st, a) in zip(ctxs, bufs):
            c.aggregate_dev(None, a.data_ptr())
    def sync():
        torch.cuda.synchronize()
        for c in ctxs: c.synchronize()
    step(); sync()
    t0 = time.perf_counter()
    for _ in range(4): step()
    sync()
    dt = (time.perf_counter() - t0) / 4
    bad = sum(int((b[2] != 0).sum().item()) for b in bufs)
    print(f"lanes={lanes} reports={m * lanes} ms/step={1e3 * dt:.2f} reports/s={m * lanes / dt / 1e6:.3f}M bad={bad}", flush=True)
    for c in ctxs: c.close()
    del bufs, d_blob, d_off
    torch.cuda.empty_cache()
