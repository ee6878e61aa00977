#!/bin/bash
# round 2, first GPU call: the allocation-free sort, sponge parity, k_meas variants alone, the Keccak ceiling with clocks
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/exp1_tests.log 2>&1 || { tail -30 $O/exp1_tests.log; exit 1; }
tail -3 $O/exp1_tests.log
timeout -k 5 120 tools/keccak_bench > $O/exp1_keccak.txt 2>&1 || { tail $O/exp1_keccak.txt; exit 1; }
cat $O/exp1_keccak.txt
for b in base v2 probe v2_probe; do
  for cfg in "62500 62500" "187500 0" "0 93750"; do
    echo "== $b $cfg" >> $O/exp1_meas.txt
    timeout -k 5 120 build_exp/meas_bench_$b $cfg >> $O/exp1_meas.txt 2>&1 || { tail $O/exp1_meas.txt; exit 1; }
  done
done
cat $O/exp1_meas.txt
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2> $O/exp1_bench.err > $O/exp1_bench.json || { tail $O/exp1_bench.err; exit 1; }
python3 -c "import sys,json; d=json.loads(open('$O/exp1_bench.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
