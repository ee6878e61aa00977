#!/bin/bash
# LDS side of the heavy-hitter kernels: bank conflicts and LDS-array cycles against wave cycles, one --pmc pass.
# Usage on the GPU box: tools/hh_lds_pmc.sh <tag> [workload]
tag=${1:-rX}; wl=${2:-hh_t0.01}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${tag}_hhlds
( cd $R && timeout -s KILL 200 rocprofv3 --pmc ${WH_PMC:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} -d $O/${tag}_hhlds --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/${tag}_hhlds.err ) || { echo pmc failed; tail -3 $O/${tag}_hhlds.err; exit 1; }
python3 - "$O/${tag}_hhlds" > $O/${tag}_hh_lds_pmc.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); cnt[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(acc, key=lambda k: -dur[k])[:8]:
    print(k, "launches", cnt[k], "total ms %.3f" % (dur[k] / 1e6), {c: "%.4g" % v for c, v in sorted(acc[k].items())})
PY
cat $O/${tag}_hh_lds_pmc.txt
