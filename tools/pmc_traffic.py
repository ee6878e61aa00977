"""HBM bytes per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
FETCH_SIZE is doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (128-B requests tallied as 64 B)."""
import csv, glob, json, sys
from collections import defaultdict


def per_kernel(d):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("wh::", "").split("<")[0].strip()
            acc[k].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if k.startswith("k_")}


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (%s, tools/profile_round.sh); FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md HBM section; bytes per launch" % sys.argv[3],
       "sumvec_large": {"reports_per_gpu": 125000,
                        "kernels": {k: {"fetch_bytes": 2 * 1024 * fetch[k], "write_bytes": 1024 * write.get(k, 0.0)} for k in sorted(fetch)}}}
print(json.dumps(out, indent=1))
