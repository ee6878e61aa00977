#!/bin/bash
O=gpurun_out; mkdir -p $O; rm -f $O/exp2_meas.txt
for b in v5e20_probe v5e21_probe v5e22_probe v5e20 v5e21 v5e22; do
for cfg in "62500 62500"; do
  echo "== $b $cfg" >> $O/exp2_meas.txt
  timeout -k 5 120 build_exp/meas_bench_$b $cfg >> $O/exp2_meas.txt 2>&1 || { tail $O/exp2_meas.txt; exit 1; }
done
done
cat $O/exp2_meas.txt
