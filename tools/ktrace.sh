#!/bin/bash
# kernel timeline of one bench step: start/end of every kernel relative to the first, in ms
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/ktrace; rm -rf $out; mkdir -p $out
( cd $R && WHISPER_GFX950_LIB=${WHISPER_GFX950_LIB:-} timeout -s KILL 240 rocprofv3 --kernel-trace -d $out --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/run.log 2>&1 ) || { tail -5 $out/run.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "wh::" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step only: find the last k_parse
idx = max(i for i, r in enumerate(rows) if "k_parse" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-40s q=%s start %8.3f end %8.3f  dur %7.3f ms" % (r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
