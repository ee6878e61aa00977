#!/bin/bash
# The three small-vector workloads (BASELINE configs 1-2 and the servers' Average) with their per-kernel times, one line each.
# Usage on the GPU box: tools/small_bench.sh <tag> [extra bench.py flags]   (writes gpurun_out/<tag>_bench_<workload>.json)
tag=${1:-rX}; shift
O=gpurun_out; mkdir -p $O
for wl in histogram average sumvec_cfg1; do
  timeout -k 10 300 python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/${tag}_bench_$wl.json 2> $O/${tag}_bench_$wl.err || { tail -5 $O/${tag}_bench_$wl.err; exit 1; }
  python3 -c "import json; d=json.loads(open('$O/${tag}_bench_$wl.json').read().strip().split('\n')[-1]); print('$wl', round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"
done
