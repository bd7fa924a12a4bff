#!/bin/bash
# A/B of the dense sample's size (option mfma_dense_tiles) by batch size and store size; parity tests of the touched kernels first.
set -e
O=gpurun_out/r03h
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q --timeout 200 --timeout-method thread -k "mfma or stream or batch or shard" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for cfg in "1024 10000000 6 6" "32 10000000 6 20" "256 10000000 6 10" "1024 1250000 8 20" "32 1250000 8 50" "4096 2000000 4 4"; do
  set -- $cfg
  timeout -k 10 200 python benchmarks/ab_option.py --option mfma_dense_tiles --values 16,32,64 --batch $1 --rows $2 --rounds $3 --iters $4 >> $O/dense_ab.txt 2>> $O/err.txt
done
cat $O/dense_ab.txt
