"""The un-batched draft-07 path on the GPU (server-base-prio3/src/main.rs:26-103: decode + Prio3::prepare_init per VdafKey), keys
resident in HBM, alternating Leader / Helper keys tiled from 64 distinct clients made by the oracle's sharder; HIP-event-free
wall clock around synchronised calls.  For comparison with the batched path's bench.py lines (same circuits, same report counts)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402  (input generator only)
import whisper_amd as W  # noqa: E402

dev = torch.device("cuda:0")
VK = bytes(range(16))
out = {}
for name, cfg, n in (("sumvec_cfg1", (W.SUMVEC, 16, 128, 64), 500000), ("histogram", (W.HISTOGRAM, 0, 128, 64), 100000),
                     ("sumvec_large", (W.SUMVEC, 16, 1024, 128), 125000)):
    p, op = W.Prio3Params(*cfg), O.params(*cfg)
    rng = np.random.default_rng(3)
    K = 64 if name != "sumvec_large" else 8
    keys = []
    for i in range(K):
        meas = [int(x) for x in rng.integers(0, 1 << 16, cfg[2])] if cfg[0] == W.SUMVEC else [int(rng.integers(0, cfg[2]))]
        nonce, rand = rng.integers(0, 256, 16, dtype=np.uint8).tobytes(), rng.integers(0, 256, O.prio3_rand_size(op), dtype=np.uint8).tobytes()
        ps, ish = O.prio3_shard(op, meas, nonce, rand)
        keys.append(W.Prio3.encode_vdaf_key(i % 2, ps, ish[i % 2], nonce))
    tile = np.frombuffer(b"".join(keys), np.uint8)
    lens = np.array([len(k) for k in keys], np.int64)
    reps = -(-n // K)
    d_blob = torch.from_numpy(tile.copy()).to(dev).repeat(reps)
    d_blob = torch.cat([d_blob, torch.zeros(16, dtype=torch.uint8, device=dev)])
    off1 = np.concatenate([[0], np.cumsum(lens)])
    offs = (off1[:-1][None, :] + (np.arange(reps, dtype=np.int64) * int(off1[-1]))[:, None]).reshape(-1)
    offs = np.concatenate([offs, [reps * int(off1[-1])]])[: n + 1]
    d_off = torch.from_numpy(offs).to(dev)
    vdaf = W.Prio3(p)
    d_prep = torch.zeros(n * 16 * (p.sizes.verifier_len + 1), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    vk = np.frombuffer(VK, np.uint8)
    torch.cuda.synchronize()

    def step():
        rc = W.lib.wh_prio3_prepare_init_batch_dev(vdaf._ctx, vk.ctypes.data, d_blob.data_ptr(), d_off.data_ptr(), n, d_prep.data_ptr(), d_status.data_ptr())
        assert rc == 0, rc
    step(); vdaf.synchronize()
    assert not d_status.any().item()
    t0 = time.perf_counter()
    steps = 3
    for _ in range(steps):
        step()
    vdaf.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out[name] = {"keys_per_step": n, "ms_per_step": 1e3 * dt, "keys_per_s": n / dt, "key_bytes_leader_helper": [int(lens[0]), int(lens[1])]}
    vdaf.close()
    del d_blob, d_prep
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
