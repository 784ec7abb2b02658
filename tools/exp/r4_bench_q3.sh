#!/bin/bash
for q in 8 16 24 32; do
  for c in "2 2 6" "2 3 9"; do
    set -- $c
    echo "== GPU_MAX_HW_QUEUES=$q batch $1 copy-streams $2 host-threads $3"
    GPU_MAX_HW_QUEUES=$q tools/exp/r4_bench_s.sh $1 $2 $3 | head -2
  done
done
