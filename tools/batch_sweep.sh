#!/bin/bash
# reports/s against the batch size (one context, inputs in HBM): gpurun_out/batch_sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; : > $O/batch_sweep.txt
cd $R
for n in 1000 4000 16000 32000 62500 125000 250000; do
  timeout -k 10 200 python3 -u bench.py --reports-per-gpu $n --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'reports': d['config']['reports_per_gpu'], 'reports_per_s': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernels_ms': {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}}))" >> $O/batch_sweep.txt || exit 1
done
cat $O/batch_sweep.txt
