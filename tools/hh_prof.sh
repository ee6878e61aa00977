#!/bin/bash
# per-kernel times of the heavy-hitter expansion: rocprofv3 --kernel-trace --stats over tools/hh_eval_bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/hhprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 tools/hh_eval_bench.py --no-cpu "$@" > $O/bench.json 2> $O/err.log ) || { tail -5 $O/err.log; exit 1; }
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/hh_kernel_stats.csv
head -10 $f | cut -c1-200
grep -E "eval_ms|key_chains_per_s\"|frac_of" $O/bench.json
