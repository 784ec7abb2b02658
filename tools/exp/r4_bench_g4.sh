#!/bin/bash
for g in 128 256 512 1024; do
  for c in "2 2" "2 1" "4 1"; do
    set -- $c
    echo "== run-grid $g batch $1 copy-streams $2"
    DE265HIP_SCAN_RUN_GRID=$g tools/exp/r4_bench_s.sh $1 $2 9 | head -1
  done
done
