#!/bin/bash
# A/B of environment settings on the bench itself (device replay, 3 GOP streams and 1): bench_env_ab.sh "A=1" "B=2" ... (two rounds)
for round in 1 2; do for e in "$@"; do
  for s in 3 1; do
    echo -n "== $e streams=$s: "
    env $e timeout -k 10 300 python bench.py --streams $s --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('device_replay',{}).get('value', d['value']), 'fps; iso', {k:v['us_per_picture'] for k,v in d['kernels_isolated'].items()})" || exit 1
  done
done; done
