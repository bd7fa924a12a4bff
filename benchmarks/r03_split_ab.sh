#!/bin/bash
set -e
O=gpurun_out/r03l
mkdir -p $O
for cfg in "1024 10000000 5 6" "1024 1250000 6 20" "4096 2000000 4 4"; do
  set -- $cfg
  timeout -k 10 200 python benchmarks/ab_option.py --option mfma_split_x1000 --values 400,450,500,550,600,650 --batch $1 --rows $2 --rounds $3 --iters $4 >> $O/split_ab.txt 2>> $O/err.txt
done
for cfg in "32 10000000 5 20" "32 1250000 6 50"; do
  set -- $cfg
  timeout -k 10 200 python benchmarks/ab_option.py --option mfma_split_x1000 --values 450,500,550,600,650,700 --batch $1 --rows $2 --rounds $3 --iters $4 >> $O/split_ab.txt 2>> $O/err.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/r03l/split_ab.txt'):
    d=json.loads(ln); print(d["batch"], d["rows"], {k: round(v["median"],4) for k,v in d["ms_per_step"].items()})
PY
