#!/bin/bash
# the 2-D group launch: its parity test, then the A/B at the reference's scale, a shard's and the full store
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_round3.py -x -q -k "group_launch" --timeout 200 --timeout-method thread > gpurun_out/group_test.txt 2>&1 || { tail -30 gpurun_out/group_test.txt; exit 1; }
tail -3 gpurun_out/group_test.txt
timeout -k 10 120 python benchmarks/group_launch_ab.py 14777 f32 0 > gpurun_out/group_ab.txt &&
timeout -k 10 120 python benchmarks/group_launch_ab.py 14777 bf16 0 >> gpurun_out/group_ab.txt &&
timeout -k 10 120 python benchmarks/group_launch_ab.py 200000 f32 0 >> gpurun_out/group_ab.txt &&
timeout -k 10 180 python benchmarks/group_launch_ab.py 1250000 bf16 0 >> gpurun_out/group_ab.txt &&
timeout -k 10 240 python benchmarks/group_launch_ab.py 10000000 bf16 0 >> gpurun_out/group_ab.txt
cat gpurun_out/group_ab.txt
