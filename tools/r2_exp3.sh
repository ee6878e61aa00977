#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/exp3_tests.log 2>&1 || { tail -30 $O/exp3_tests.log; exit 1; }
tail -3 $O/exp3_tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2> $O/exp3_bench.err > $O/exp3_bench.json || { tail $O/exp3_bench.err; exit 1; }
python3 -c "import sys,json; d=json.loads(open('$O/exp3_bench.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
done
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --in-flight 2 2> $O/exp3_bench2.err > $O/exp3_bench2.json || { tail $O/exp3_bench2.err; exit 1; }
python3 -c "import sys,json; d=json.loads(open('$O/exp3_bench2.json').read()); print('in-flight 2', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
