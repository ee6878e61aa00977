#!/bin/bash
# Effective shader clock from the PMC side: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 / kernel duration, for
# k_meas inside bench.py and for the bare permutation (tools/keccak_bench).  One counter, its own pass.  tools/clock_pmc.sh <tag>
tag=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${tag}_clk_bench $O/${tag}_clk_keccak
( cd $R && timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/${tag}_clk_bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/${tag}_clk_bench.err ) || { echo bench pass failed; tail -3 $O/${tag}_clk_bench.err; exit 1; }
( cd $R && timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/${tag}_clk_keccak --output-format csv -- ./tools/keccak_bench > /dev/null 2> $O/${tag}_clk_keccak.err ) || { echo keccak pass failed; tail -3 $O/${tag}_clk_keccak.err; exit 1; }
python3 - "$O/${tag}_clk_bench" "$O/${tag}_clk_keccak" > $O/${tag}_clock_pmc.txt <<'PY'
import csv, glob, sys, collections
for d in sys.argv[1:]:
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != "GRBM_GUI_ACTIVE": continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            rows[r["Kernel_Name"].split("(")[0][-60:]].append((float(r["Counter_Value"]), dur))
    print("==", d.split("/")[-1])
    for k, v in sorted(rows.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:8]:
        big = [x for x in v if x[1] > 0.5 * max(y[1] for y in v)]
        val = sum(x[0] for x in big) / len(big); dur = sum(x[1] for x in big) / len(big)
        print(f"{k:60s} launches {len(v):3d}  avg duration {dur/1e6:8.3f} ms  GRBM_GUI_ACTIVE {val:.4g}  -> {val/8/dur:.3f} GHz (value / 8 XCDs / duration)")
PY
cat $O/${tag}_clock_pmc.txt
