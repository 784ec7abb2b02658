#!/bin/bash
# host-stage timing on the GPU box, best of several runs: tools/exp/time_build_ab.sh [lib.so ...]
for so in "${@:-libde265_amd/libde265_hip.so}"; do
  best_i=999; best_b=999
  for rep in 1 2 3 4; do
    out=$(DE265HIP_SO=$so python tools/time_build.py 2>/dev/null)
    i=$(echo "$out" | sed -n 1p | sed -e 's/.*build \([0-9.]*\) ms.*/\1/'); b=$(echo "$out" | sed -n 2p | sed -e 's/.*build \([0-9.]*\) ms.*/\1/')
    best_i=$(python3 -c "print(min($best_i,$i))"); best_b=$(python3 -c "print(min($best_b,$b))")
  done
  echo "$so: I $best_i ms, B $best_b ms"
done
