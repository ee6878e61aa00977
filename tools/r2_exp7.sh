#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_hh_eval.py tests/test_gpu_hh_e2e.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/exp7_tests.log 2>&1 || { tail -40 $O/exp7_tests.log; exit 1; }
tail -3 $O/exp7_tests.log
for wl in hh_t0.01 hh_t0.001; do
for mode in "WH_BENCH_HH_MATERIALIZE=1" "WH_X=1"; do
env $mode timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2> $O/exp7_hh.err > $O/exp7_hh.json || { tail $O/exp7_hh.err; exit 1; }
python3 -c "import sys,json; d=json.loads(open('$O/exp7_hh.json').read()); print('$wl $mode', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"
done
done
