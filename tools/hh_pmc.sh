#!/bin/bash
# HBM traffic of the heavy-hitter kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over one bench step,
# merged into profiles/pmc_traffic.json under the workload's key.  Usage on the GPU box: tools/hh_pmc.sh <tag> [workload]
tag=${1:-rX}; wl=${2:-hh_t0.01}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${tag}_hhpmc_$ctr
  ( cd $R && timeout -s KILL 200 rocprofv3 --pmc $ctr -d $O/${tag}_hhpmc_$ctr --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline > $O/${tag}_hhpmc_$ctr.json 2> $O/${tag}_hhpmc_$ctr.err ) || { echo pmc $ctr failed; tail -3 $O/${tag}_hhpmc_$ctr.err; exit 1; }
done
n=$(python3 -c "import json;print(json.loads(open('$O/${tag}_hhpmc_FETCH_SIZE.json').read().strip().splitlines()[-1])['config']['key_chains_per_step_per_gpu'])")
python3 $R/tools/pmc_traffic.py $O/${tag}_hhpmc_FETCH_SIZE $O/${tag}_hhpmc_WRITE_SIZE "$tag" $wl key_chains_per_gpu $n $R/profiles/pmc_traffic.json > $O/${tag}_pmc_traffic_merged.json || exit 1
python3 -c "import json;d=json.load(open('$O/${tag}_pmc_traffic_merged.json'));print({k:(round(v['fetch_bytes']/1e9,2),round(v['write_bytes']/1e9,2)) for k,v in d['$wl']['kernels'].items()})"
