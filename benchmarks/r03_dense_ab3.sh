#!/bin/bash
set -e
O=gpurun_out/r03j
mkdir -p $O
for cfg in "1024 10000000 5 6" "32 10000000 5 20" "1024 1250000 6 20" "32 1250000 6 50" "4096 2000000 4 4"; do
  set -- $cfg
  timeout -k 10 200 python benchmarks/ab_option.py --option mfma_dense_tiles --values 64,96,128,16 --batch $1 --rows $2 --rounds $3 --iters $4 >> $O/dense_ab3.txt 2>> $O/err.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/r03j/dense_ab3.txt'):
    d=json.loads(ln); print(d["batch"], d["rows"], {k: round(v["median"],4) for k,v in d["ms_per_step"].items()})
PY
