#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/exp5_tests.log 2>&1 || { tail -40 $O/exp5_tests.log; exit 1; }
tail -3 $O/exp5_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2> $O/exp5_bench.err > $O/exp5_bench.json || { tail $O/exp5_bench.err; exit 1; }
python3 -c "import sys,json; d=json.loads(open('$O/exp5_bench.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
