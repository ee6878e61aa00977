#!/bin/bash
# Evidence for profiles/: kernel-trace stats, HBM traffic counters (separate passes), the Keccak ceiling, and a
# bench line with the CPU baseline.  Usage on the GPU box: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>_*)
tag=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${tag}_stats && ( cd $R && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/${tag}_stats_bench.json 2> $O/${tag}_stats.err ) || { echo stats failed; exit 1; }
cp $(find $O/${tag}_stats -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${tag}_pmc_$ctr
  ( cd $R && timeout -s KILL 300 rocprofv3 --pmc $ctr -d $O/${tag}_pmc_$ctr --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/${tag}_pmc_$ctr.err ) || { echo pmc $ctr failed; exit 1; }
done
python3 $R/tools/pmc_traffic.py $O/${tag}_pmc_FETCH_SIZE $O/${tag}_pmc_WRITE_SIZE "$tag" > $O/${tag}_pmc_traffic.json || exit 1
timeout -k 5 120 $R/tools/keccak_bench > $O/${tag}_keccak_bench.txt || exit 1
cd $R && timeout -k 10 400 python3 bench.py --steps 5 --warmup 1 > $O/${tag}_bench.json 2> $O/${tag}_bench.err || exit 1
cat $O/${tag}_bench.json
