#!/bin/bash
# quick loop on the GPU box: parity tests, then one short bench line with the per-kernel times
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -${TAIL:-3} gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2> gpurun_out/bench_now.err > gpurun_out/bench_now.json || exit 1
python3 -c "import sys,json; d=json.loads(open('gpurun_out/bench_now.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
