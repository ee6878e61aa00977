"""wh_sort_by_testing_id_dev: time per batch (HIP events on the context stream) beside a stable host sort of the same ids.
python tools/sort_bench.py"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W

dev = torch.device("cuda:0")
vdaf = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3))
out = []
for n in (1000, 125_000, 1_000_000, 8_000_000):
    rng = np.random.default_rng(n)
    ids = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    tags = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    d_ids, d_tags = torch.from_numpy(ids).to(dev), torch.from_numpy(tags).to(dev)
    d_order = torch.zeros(n, dtype=torch.int32, device=dev)
    d_sids, d_stags = torch.zeros_like(d_ids), torch.zeros_like(d_tags)
    torch.cuda.synchronize()
    call = lambda: vdaf.sort_by_testing_id_dev(d_ids.data_ptr(), n, d_order.data_ptr(), d_tags.data_ptr(), 16, d_sids.data_ptr(), d_stags.data_ptr())
    call(); vdaf.synchronize()
    vdaf.set_profiling(True); vdaf.kernel_times(reset=True)
    reps = 10
    for _ in range(reps): call()
    vdaf.synchronize()
    ms = vdaf.kernel_times(reset=True)["sort"][0] / reps
    vdaf.set_profiling(False)
    v = ids.view(np.uint64).reshape(-1, 2)
    t0 = time.perf_counter(); order = np.lexsort((v[:, 0], v[:, 1])); t_host = time.perf_counter() - t0
    assert np.array_equal(d_order.cpu().numpy().view(np.uint32), order.astype(np.uint32))
    out.append({"n": n, "gpu_ms": round(ms, 3), "ids_per_s": round(n / ms * 1e3), "numpy_lexsort_ms": round(1e3 * t_host, 2)})
    print(out[-1], flush=True)
print(json.dumps({"sort_by_testing_id": out}))
vdaf.close()
