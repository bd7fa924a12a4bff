#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_round3.py tests/test_zz_no_alloc.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q --timeout 300 --timeout-method thread > gpurun_out/xcd_test.txt 2>&1 || { tail -30 gpurun_out/xcd_test.txt; exit 1; }
tail -n 2 gpurun_out/xcd_test.txt
O=gpurun_out/xcd_ab.txt
: > $O
timeout -k 10 300 python benchmarks/xcd_slices_ab.py 10000000 >> $O
timeout -k 10 300 python benchmarks/xcd_slices_ab.py 1250000 >> $O
cat $O
