#!/bin/bash
# A/B/n several builds of the library on the same GPU box: tools/abn.sh libA.so libB.so ... 
for round in 1 2; do
  for L in "$@"; do
    WHISPER_GFX950_LIB=$L timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})" || exit 1
  done
done
