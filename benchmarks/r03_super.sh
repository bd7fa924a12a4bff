#!/bin/bash
# super-round partition: parity first, then the A/B at four store sizes
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_round3.py tests/test_zz_no_alloc.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/super_test.txt 2>&1 || { tail -30 gpurun_out/super_test.txt; exit 1; }
tail -n 2 gpurun_out/super_test.txt
O=gpurun_out/super_ab.txt
: > $O
timeout -k 10 300 python benchmarks/super_rounds_ab.py 12500000 4 >> $O
timeout -k 10 300 python benchmarks/super_rounds_ab.py 10000000 3 >> $O
timeout -k 10 300 python benchmarks/super_rounds_ab.py 20000000 3 >> $O
timeout -k 10 300 python benchmarks/super_rounds_ab.py 1250000 3 >> $O
cat $O
