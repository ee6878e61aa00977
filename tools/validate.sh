#!/bin/bash
# round-end check on the GPU box: full GPU suite, smoke(), the default bench line (tools/validate.sh)
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python3 -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().split('\n')[-1]); print(round(d['value']), d['unit'], round(d['ms_per_step'],2), 'roofline', d['roofline']['bound'], round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value']), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
