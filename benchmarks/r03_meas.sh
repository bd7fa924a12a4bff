#!/bin/bash
# Round-3 measurement set (one MI355X box): everything lands under gpurun_out/r03f/.  Each GPU step has its own time limit
# and the steps are chained: a failure ends the script.
set -e
O=gpurun_out/r03u
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-pass > $O/bench_under_rocprof.json 2> $O/rocprof.err
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
timeout -k 10 120 python bench.py --rows 1250000 --steps 400 --warmup 40 --no-cpu-baseline --no-traffic-pass --inproc > $O/bench_shard_size.json 2> $O/bench_shard.err
timeout -k 10 200 python benchmarks/batch_sweep.py --batches 1,2,3,4,8,16,32,48,64,65,128,256,512,1024 > $O/batch_sweep_10m.txt 2> $O/sweep.err
timeout -k 10 120 python benchmarks/batch_sweep.py --rows 1250000 --batches 1,2,3,4,8,16,32,64,65,128,256,1024 > $O/batch_sweep_1250k.txt 2>> $O/sweep.err
timeout -k 10 120 python benchmarks/sharded_issue_cost.py --burst 4 > $O/issue_cost_tiny_shards.json 2> $O/issue.err
timeout -k 10 120 python benchmarks/sharded_issue_cost.py --burst 4 --rows-per-shard 156250 > $O/issue_cost_10m_over_8_shards_on_one_gpu.json 2>> $O/issue.err
timeout -k 10 400 python tests/fuzz_gpu.py --family state --cases 400 --budget-s 300 --seed 31 > $O/fuzz_state.txt 2>&1
tail -3 $O/fuzz_state.txt
