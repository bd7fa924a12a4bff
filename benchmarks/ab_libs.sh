#!/bin/bash
# A/B of two builds of the library on one box, runs alternated: vectordb_amd/lib/libvdb_hip_prev.so (copy of an earlier
# build, selected through VECTORDB_AMD_LIB) against the in-tree build.   BATCHES="4 32" STEPS=200 bash benchmarks/ab_libs.sh
set -e
BATCHES=${BATCHES:-1024}
STEPS=${STEPS:-20}
for r in 1 2 3 4; do
  for lib in prev new; do
    if [ $lib = prev ]; then export VECTORDB_AMD_LIB=$PWD/vectordb_amd/lib/libvdb_hip_prev.so; else unset VECTORDB_AMD_LIB; fi
    for b in $BATCHES; do
      timeout -k 10 120 python3 bench.py --batch $b --batch2 0 --steps $STEPS --warmup 3 --no-cpu-baseline --no-traffic-pass > gpurun_out/abl_tmp.json 2>gpurun_out/abl_tmp.err
      python3 -c "import json;d=json.load(open('gpurun_out/abl_tmp.json'));print('$lib', $b, round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4))"
    done
  done
done
