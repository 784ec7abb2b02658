#!/bin/bash
# product-path rate of the full-length bench for environment settings, interleaved: tools/exp/product_ab_long.sh REPS "A=1" "A=2" ...
reps=$1; shift
for r in $(seq $reps); do
  for e in "$@"; do
    env $e timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'product fps %.0f replay %.0f 1thr %.0f' % (d['value'], d['device_replay']['value'], d['product_path']['value_1_host_thread']))" || exit 1
  done
done
