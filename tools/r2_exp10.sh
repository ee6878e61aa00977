#!/bin/bash
# parity subset, then bench lines (short and long) with per-kernel times
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_frozen_fixtures.py tests/test_gpu_hh_eval.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/exp10_tests.log 2>&1 || { tail -40 $O/exp10_tests.log; exit 1; }
tail -2 $O/exp10_tests.log
for st in 5 20; do
timeout -k 10 300 python bench.py --steps $st --warmup 2 --no-cpu-baseline 2> $O/bench_now.err > $O/bench_now_$st.json || exit 1
python3 -c "import sys,json; d=json.loads(open('$O/bench_now_$st.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
done
