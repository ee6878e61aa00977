#!/bin/bash
# full GPU suite, then the heavy-hitter bench lines and kernel stats for profiles/
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
for wl in hh_t0.01 hh_t0.001; do
  timeout -k 10 400 python -u bench.py --workload $wl --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/r2b_bench_$wl.json || exit 1
done
./tools/hh_prof.sh > $O/r2b_hh_prof.log 2>&1 || { tail -5 $O/r2b_hh_prof.log; exit 1; }
cp $O/hh_kernel_stats.csv $O/r2b_hh_kernel_stats.csv
timeout -k 10 120 ./build_exp/inst_rate > $O/r2_inst_rate_full.txt 2>&1 || exit 1
echo done
