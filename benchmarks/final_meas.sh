set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
timeout -k 10 400 python3 benchmarks/batch_sweep.py --batches 1,2,3,4,16,32,64,65,128,256,512,1024,2048,4096 > gpurun_out/r02_batch_sweep_new.txt 2>&1
timeout -k 10 120 python3 bench.py --rows 1250000 --steps 400 --warmup 40 --no-cpu-baseline --no-traffic-pass > gpurun_out/r02_shard_size.json 2>/dev/null
timeout -k 10 300 python3 benchmarks/selfjoin_profile.py > gpurun_out/r02_selfjoin_8blocks.txt 2>&1
timeout -k 10 600 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "c5" --timeout 500 --timeout-method thread > gpurun_out/r02_c5.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
