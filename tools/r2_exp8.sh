#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_hh_full_size.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/exp8_tests.log 2>&1 || { tail -40 $O/exp8_tests.log; exit 1; }
tail -3 $O/exp8_tests.log
