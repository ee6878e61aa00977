#!/bin/bash
O=gpurun_out; mkdir -p $O
for b in "$@"; do
  echo "== $b" >> $O/exp9.txt
  timeout -k 10 120 ./build_exp/$b >> $O/exp9.txt 2>&1 || exit 1
done
