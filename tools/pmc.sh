#!/bin/bash
# one rocprofv3 counter pass per argument ("NAME:CTR1 CTR2 ..."), each in its own run, summarised per kernel
# usage: tools/pmc.sh "sq:SQ_INSTS_VALU SQ_WAVE_CYCLES" "tcc:FETCH_SIZE TCC_HIT_sum"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  out=$R/gpurun_out/pmc_$name
  rm -rf $out; mkdir -p $out
  ( cd $R && timeout -s KILL 240 rocprofv3 --pmc $ctrs -d $out --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $out/run.log 2>&1 ) || { echo "pass $name failed"; tail -5 $out/run.log; exit 1; }
  python3 $R/tools/pmc_summary.py $out | tee $out/summary.txt
done
