#!/bin/bash
# end-of-round check of the final build: the whole GPU suite, the driver's bench command, its rocprofv3 twin, the shard-size step
set -e
O=gpurun_out/r03final
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-pass > $O/bench_under_rocprof.json 2> $O/rocprof.err
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
timeout -k 10 200 python bench.py --rows 1250000 --steps 400 --warmup 40 --no-cpu-baseline --no-traffic-pass --inproc > $O/bench_shard_size.json 2> $O/bench_shard.err
timeout -k 10 200 python benchmarks/stream_ab.py 1250000 > $O/stream_ab.txt 2>/dev/null
timeout -k 10 200 python benchmarks/stream_ab.py 10000000 >> $O/stream_ab.txt 2>/dev/null
cat $O/stream_ab.txt
