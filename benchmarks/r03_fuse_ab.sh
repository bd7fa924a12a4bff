#!/bin/bash
set -e
O=gpurun_out/r03q
mkdir -p $O
for n in 14777 200000 1250000 10000000; do for b in 1 2 3 4; do
  timeout -k 10 100 python benchmarks/ab_option.py --option fuse_small_batch --values 0,1 --set batch_mfma_min_b=1048576 --batch $b --rows $n --rounds 5 --iters 30 >> $O/fuse_ab.txt 2>> $O/err.txt
done; done
python - <<'PY'
import json
for ln in open('gpurun_out/r03q/fuse_ab.txt'):
    d=json.loads(ln); print(d["batch"], d["rows"], {k: round(v["median"],4) for k,v in d["ms_per_step"].items()})
PY
