#!/bin/bash
# instruction mix and stall counters of the verify kernels (one --pmc pass, 8 SQ counters): gpurun_out/meas_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/meas_pmc
( cd $R && timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/meas_pmc --output-format csv -- python3 bench.py --tiled --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/meas_pmc.err ) || { echo pmc failed; tail -5 $O/meas_pmc.err; exit 1; }
python3 - <<PY > $O/meas_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/meas_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if not k.startswith("k_"): continue
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
cat $O/meas_pmc.txt
