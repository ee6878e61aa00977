"""Per-kernel averages of the counters in a rocprofv3 --pmc output directory."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not (k.startswith("wh::") or k.startswith("k_") or k.startswith("void wh::")):
        continue
    print(k, " ".join("%s=%.4g(x%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
