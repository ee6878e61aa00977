"""HBM bytes per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
FETCH_SIZE is doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (128-B requests tallied as 64 B).
  pmc_traffic.py <fetch dir> <write dir> <tag> [<workload> <size key> <size> [<json to merge into>]]"""
import csv, glob, json, re, sys
from collections import defaultdict


def per_kernel(d):
    """average per launch; instantiations of one template that both ran are kept apart (k_query_psum<false, false> is the
    verifier's launch, <false, true> the sharder's public proofs), a lone instantiation goes by the bare name"""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"\b(k_[a-z0-9_]+)(<[^>]*>)?", row["Kernel_Name"])
            if m:
                acc[m.group(1)][m.group(2) or ""].append(float(row["Counter_Value"]))
    out = {}
    for base, inst in acc.items():
        for targs, v in inst.items():
            out[base + (targs if len(inst) > 1 else "")] = sum(v) / len(v)
    return out


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
workload, size_key, size = (sys.argv[4], sys.argv[5], int(sys.argv[6])) if len(sys.argv) > 6 else ("sumvec_large", "reports_per_gpu", 125000)
out = {}
if len(sys.argv) > 7:
    try:
        out = json.load(open(sys.argv[7]))
    except (OSError, ValueError):
        out = {}
out["source"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (%s, tools/profile_round.sh, tools/hh_pmc.sh); "
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; bytes per launch" % sys.argv[3])
out[workload] = {size_key: size,
                 "kernels": {k: {"fetch_bytes": 2 * 1024 * fetch[k], "write_bytes": 1024 * write.get(k, 0.0)} for k in sorted(fetch)}}
print(json.dumps(out, indent=1))
