#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_prepare_init.py tests/test_gpu_flp_surface.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/exp4_tests.log 2>&1 || { tail -30 $O/exp4_tests.log; exit 1; }
tail -2 $O/exp4_tests.log
bash tools/abn.sh ab_libs/libwh_prev.so whisper_amd/libwhisper_gfx950.so
