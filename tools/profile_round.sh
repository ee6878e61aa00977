#!/bin/bash
# Evidence for profiles/: kernel-trace stats, HBM traffic counters (separate passes), the Keccak ceiling, and the default bench
# line (CPU baseline + the `extra` lines of the other configurations).  The small workloads get their own kernel stats and
# traffic passes.  Usage on the GPU box: tools/profile_round.sh <tag> [hh]  (writes gpurun_out/<tag>_*; `hh`: only the kernel
# stats of the two heavy-hitter workloads)
tag=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <suffix> <bench args...>
  local sfx=$1; shift
  rm -rf $O/${tag}_stats$sfx && ( cd $R && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_stats$sfx --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $O/${tag}_stats_bench$sfx.json 2> $O/${tag}_stats$sfx.err ) || { echo stats$sfx failed; tail -3 $O/${tag}_stats$sfx.err; return 1; }
  cp $(find $O/${tag}_stats$sfx -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats$sfx.csv
}
pmc() {     # pmc <suffix> <workload> <reports> <bench args...>: FETCH_SIZE and WRITE_SIZE, a pass each, merged into <tag>_pmc_traffic.json
  local sfx=$1 wl=$2 n=$3; shift 3
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${tag}_pmc_$ctr$sfx
    ( cd $R && timeout -s KILL 300 rocprofv3 --pmc $ctr -d $O/${tag}_pmc_$ctr$sfx --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $O/${tag}_pmc_$ctr$sfx.err ) || { echo pmc $ctr$sfx failed; return 1; }
  done
  python3 $R/tools/pmc_traffic.py $O/${tag}_pmc_FETCH_SIZE$sfx $O/${tag}_pmc_WRITE_SIZE$sfx "$tag" $wl reports_per_gpu $n $O/${tag}_pmc_traffic.json > $O/${tag}_pmc_traffic.tmp && mv $O/${tag}_pmc_traffic.tmp $O/${tag}_pmc_traffic.json
}
if [ "$2" = hh ]; then
  for w in hh_t0.01 hh_t0.001; do stats _$w --workload $w || exit 1; done
  exit 0
fi
rm -f $O/${tag}_pmc_traffic.json
stats "" || exit 1
pmc "" sumvec_large 125000 || exit 1
for w in sumvec_cfg1:500000 histogram:100000 average:500000; do
  wl=${w%%:*}; n=${w##*:}
  stats _$wl --workload $wl || exit 1
  pmc _$wl $wl $n --workload $wl || exit 1
done
timeout -k 5 120 $R/tools/keccak_bench > $O/${tag}_keccak_bench.txt || exit 1
cd $R && timeout -k 10 600 python3 bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || exit 1
python3 -c "
import json; d=json.loads(open('$O/${tag}_bench.json').read().strip().split('\n')[-1])
print(round(d['value']), round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})
for k,v in d.get('extra',{}).items(): print(k, round(v['value']), round(v['ms_per_step'],3), v['kernels'])"
