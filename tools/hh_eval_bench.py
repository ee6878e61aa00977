"""Heavy-hitter expansion (BCSketch::eval_and_check, hhcore/src/protocol.rs:78-176) on the GPU: key chains per second
with the key chains resident in HBM, the AES block rate against the LDS-lookup bound of the table AES, and the
oracle (AES-NI when the host has it) on the host cores beside it.

  python tools/hh_eval_bench.py [--buckets 512] [--clients 2000] [--steps 5] [--cpu-threads 16] [--no-cpu]
"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_amd import hh

ap = argparse.ArgumentParser()
ap.add_argument("--buckets", type=int, default=512)
ap.add_argument("--clients", type=int, default=14000, help="key chains per call: clients x repetitions (14 x 1000 at threshold 0.01)")
ap.add_argument("--unique", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--cpu-threads", type=int, default=16)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()

depth = a.buckets.bit_length() - 1
n, N = a.clients, a.buckets
rng = np.random.default_rng(0x5748)
ln = hh.key_chain_len(depth)
t0 = time.time()
uniq = np.zeros((a.unique, ln), np.uint8)
for i in range(a.unique):
    bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32); bk[257] = 1
    uniq[i], _ = hh.gen_key_chain(depth, i, int(rng.integers(0, N)), bk, rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
gen_s = time.time() - t0
keys = uniq[np.arange(n) % a.unique].copy()
keys[:, :8] = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)  # fresh client ids

dev = torch.device("cuda:0")
h = hh.HeavyHitters(N)
ext = torch.cuda.ExternalStream(h.stream, device=dev)
d_keys = torch.from_numpy(keys).to(dev)
d_dat = torch.empty((n, N, 258), dtype=torch.int32, device=dev)
d_tags = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
d_status = torch.zeros(n, dtype=torch.int32, device=dev)
d_acc = torch.zeros((N, 258), dtype=torch.int32, device=dev)
vk = bytes(range(16))
torch.cuda.synchronize()
ts = []
for it in range(a.steps + 1):
    with torch.cuda.stream(ext):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(ext)
        h.eval_and_check_dev(vk, d_keys.data_ptr(), ln, n, d_dat.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
        e1.record(ext)
        h.aggregate_hhs_dev(None, d_acc.data_ptr())
        e2.record(ext)
    h.synchronize(); torch.cuda.synchronize()
    ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
assert int(d_status.abs().sum()) == 0
ev, ag = min(t[0] for t in ts[1:]), min(t[1] for t in ts[1:])
# AES blocks the algorithm needs per key chain: tree 2(N-1), leaf hashes 2N, conversion 65N, chain 4N,
# support keys 257 * (2 + 2*2 + 2*4), testing id 1
aes = 2 * (N - 1) + 2 * N + 65 * N + 4 * N + 257 * 14 + 1
# table AES on gfx950 is bound by instruction issue (tools/aes_microbench.hip: VALU and LDS reads do not overlap):
# per block and wavefront 160 x (v_perm 4.2 + ds_read 2.6) + 16 x 2.6 + 80 x 2.3 cycles on one of 4 SIMDs
issue_cycles = 160 * (4.2 + 2.6) + 16 * 2.6 + 80 * 2.3
lds_peak = min(256 * 2.4e9 * 4 * 64 / issue_cycles, 256 * 2.4e9 * 32 / 176)   # ... and the LDS: 176 reads per block at 32 lanes/clk/CU
out = {"buckets": N, "clients": n, "key_chain_bytes": ln, "eval_ms": ev, "aggregate_ms": ag,
       "key_chains_per_s": n / ev * 1e3, "key_chains_per_s_with_sum": n / (ev + ag) * 1e3,
       "aes_blocks_per_key_chain": aes, "aes_blocks_per_s": n * aes / ev * 1e3, "aes_bound_blocks_per_s": lds_peak,
       "frac_of_bound": n * aes / ev * 1e3 / lds_peak, "dat_write_GBps": n * N * 258 * 4 / ev / 1e6, "client_gen_s": gen_s}
if not a.no_cpu:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    m = min(n, a.cpu_threads * 1500)   # about ten seconds of host work at 512 buckets
    t0 = time.time()
    odat, otags, oids, ost = o.hh_eval_and_check_batch(depth, vk, keys[:m], nthreads=a.cpu_threads)
    cpu_s = time.time() - t0
    assert (d_tags[:m].cpu().numpy() == otags).all() and (d_dat[:m].cpu().numpy().view(np.uint32) == odat).all()
    out["cpu_baseline"] = {"value": m / cpu_s, "unit": "key chains/s", "cores": a.cpu_threads, "kind": "port",
                           "aes": "AES-NI" if o.aes_backend() else "portable C",
                           "sample": f"{m} key chains in {cpu_s:.2f} s; oracle/vdpf.c = C restatement of eval_and_check"}
print(json.dumps(out, indent=1))
