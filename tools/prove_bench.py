"""Type::prove on the GPU (wh_flp_prove_batch, prove.hip): proofs per second of the kernel alone (HIP events on the
context stream) for the headline SumVec shape, with the CPU client's prove and the oracle's beside it.
  python tools/prove_bench.py [--n 4096] [--shape 16,1024,128]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_amd as W

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--shape", default="16,1024,128")
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
bits, ln, ch = (int(v) for v in a.shape.split(","))
p = W.Prio3Params(W.SUMVEC, bits, ln, ch)
s = p.sizes
rng = np.random.default_rng(7)
def elems(n, k):
    v = rng.integers(0, 2**62, (n, k, 2), dtype=np.uint64)      # < 2^126 < p: canonical
    return v.view(np.uint8).reshape(n, 16 * k)
K = 64
meas = rng.integers(0, 1 << bits, (K, ln), dtype=np.uint64)
inp = np.stack([W.flp_encode(p, m) for m in meas])[np.arange(a.n) % K]
pr, jr = elems(a.n, s.arity), elems(a.n, s.joint_rand_len)
vdaf = W.Prio3(p)
vdaf.flp_prove(inp[:8], pr[:8], jr[:8])
vdaf.set_profiling(True); vdaf.kernel_times(reset=True)
t0 = time.perf_counter()
proofs, status = vdaf.flp_prove(inp, pr, jr)
wall = time.perf_counter() - t0
ms, cnt = vdaf.kernel_times(reset=True)["prove"]
assert not status.any()
# field multiplications per proof: per wire two size-m transforms (m/2 log2 m each) + twist m + build 2 per call;
# products 2m per pair; final size-2m transform
m, log2m = s.m, s.m.bit_length() - 1
mults = s.arity * (m * log2m + m) + (s.arity // 2) * (2 * m + 2 * s.calls) + m * (log2m + 1) + 2 * m
out = {"shape": {"bits": bits, "len": ln, "chunk": ch, "m": m, "wires": s.arity, "proof_len": s.proof_len}, "n": a.n,
       "kernel_ms": ms, "proofs_per_s_kernel": a.n / ms * 1e3, "proofs_per_s_with_pcie": a.n / wall,
       "field_mults_per_proof": mults, "field_mults_per_s": mults * a.n / ms * 1e3}
if not a.no_cpu:
    t0 = time.perf_counter()
    k = 8
    for i in range(k):
        cpu = W.flp_prove(p, inp[i], pr[i], jr[i])
        assert cpu.tobytes() == proofs[i].tobytes()
    out["cpu_client_proofs_per_s_one_thread"] = k / (time.perf_counter() - t0)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    op = O.params(O.SUMVEC, bits, ln, ch)
    t0 = time.perf_counter()
    for i in range(k):
        assert O.flp_prove(op, inp[i].tobytes(), pr[i].tobytes(), jr[i].tobytes()) == proofs[i].tobytes()
    out["oracle_proofs_per_s_one_thread"] = k / (time.perf_counter() - t0)
print(json.dumps(out, indent=1))
