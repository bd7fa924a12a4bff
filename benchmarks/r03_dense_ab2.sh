#!/bin/bash
set -e
O=gpurun_out/r03i
mkdir -p $O
ARMS="mfma_dense_tiles=16,mfma_one_pass_span=204;mfma_dense_tiles=16,mfma_one_pass_span=24;mfma_dense_tiles=32,mfma_one_pass_span=24;mfma_dense_tiles=64,mfma_one_pass_span=24;mfma_dense_tiles=64,mfma_one_pass_span=8"
for cfg in "1024 1250000 6 20" "32 1250000 6 50" "4096 2000000 4 4" "1024 200000 6 30" "32 200000 6 50" "128 1250000 6 30" "1024 40000 6 30"; do
  set -- $cfg
  timeout -k 10 200 python benchmarks/ab_option.py --arms "$ARMS" --batch $1 --rows $2 --rounds $3 --iters $4 >> $O/dense_ab2.txt 2>> $O/err.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/r03i/dense_ab2.txt'):
    d=json.loads(ln); print(d["batch"], d["rows"], {k.replace("mfma_dense_tiles=","D").replace("mfma_one_pass_span=","S"): round(v["median"],4) for k,v in d["ms_per_step"].items()})
PY
