#!/bin/bash
# allocation order A/B: one fresh process per store, orders alternating, three sizes
set -e
O=gpurun_out/alloc_order.txt
: > $O
for rows in 12500000 10000000 20000000; do
  for ord in 0 1 0 1; do
    timeout -k 10 200 python benchmarks/alloc_order_ab.py $rows $ord >> $O
  done
done
cat $O
