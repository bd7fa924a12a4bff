#!/bin/bash
# batch-1 step by scan blocks per CU at store sizes where the rate dips
for n in 12500000 20000000; do for bpc in 2 3 4; do
  timeout -k 10 150 python bench.py --rows $n --blocks-per-cu $bpc --no-cpu-baseline --no-traffic-pass --batch2 0 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print(b['config']['rows'], 'blocks/CU', $bpc, round(b['ms_per_step'],4), 'ms', round(r['achieved']), 'GB/s')"
done; done
