#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_api_state.py tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/exp6_tests.log 2>&1 || { tail -40 $O/exp6_tests.log; exit 1; }
tail -3 $O/exp6_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2> $O/exp6_bench.err > $O/exp6_bench.json || { tail $O/exp6_bench.err; exit 1; }
python3 -c "import sys,json; d=json.loads(open('$O/exp6_bench.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()}); print(json.dumps(d['roofline']))"
for m in pageable pinned; do
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --host-buffers $m 2> $O/exp6_host_$m.err > $O/exp6_host_$m.json || { tail $O/exp6_host_$m.err; exit 1; }
cat $O/exp6_host_$m.json
done
