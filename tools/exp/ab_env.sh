#!/bin/bash
# A/B of environment settings on one box: tools/exp/ab_env.sh "VAR=a" "VAR=b" ...   (each run twice, 3 GOP streams)
for rep in 1 2; do
  for e in "$@"; do
    v=$(env $e timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive 2>/dev/null \
        | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernels_isolated']; print(d['value'], 'intra_iso_us', k['intra']['us_per_picture'])")
    echo "$e : $v"
  done
done
