#!/bin/bash
# A/B/n of library builds on the heavy-hitter workloads: tools/abn_hh.sh libA.so libB.so ...
O=gpurun_out; mkdir -p $O
WHISPER_GFX950_LIB=$1 timeout -k 10 300 python -u -m pytest tests/test_gpu_hh_eval.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/abn_hh_tests_first.log 2>&1 || { tail -20 $O/abn_hh_tests_first.log; exit 1; }
for round in 1 2; do
  for wl in hh_t0.01 hh_t0.001; do
    for L in "$@"; do
      WHISPER_GFX950_LIB=$L timeout -k 10 300 python -u bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', '$wl', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items()})" || exit 1
    done
  done
done
