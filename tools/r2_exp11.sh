#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_hh_eval.py tests/test_gpu_frozen_fixtures.py tests/test_gpu_hh_full_size.py tests/test_gpu_hh_e2e.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/exp11_tests.log 2>&1 || { tail -40 $O/exp11_tests.log; exit 1; }
tail -2 $O/exp11_tests.log
for wl in hh_t0.01 hh_t0.001; do
  timeout -k 10 300 python -u bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/exp11_$wl.json || exit 1
  python3 -c "import sys,json; d=json.loads(open('$O/exp11_$wl.json').read()); print('$wl', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items()})"
done
