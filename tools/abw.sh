#!/bin/bash
# A/B/n several builds of the library on one GPU box, two rounds: tools/abw.sh "<bench.py args>" libA.so libB.so ...
args=$1; shift
for round in 1 2; do
  for L in "$@"; do
    WHISPER_GFX950_LIB=$PWD/$L timeout -k 10 300 python3 -u bench.py --steps 10 --warmup 2 --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" || exit 1
  done
done
