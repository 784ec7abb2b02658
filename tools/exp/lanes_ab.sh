#!/bin/bash
# device replay rate against streams x lanes: tools/exp/lanes_ab.sh "1 1" "1 2" "1 3" "3 1" "3 2"
for sl in "$@"; do
  set -- $sl
  timeout -k 10 300 python bench.py --streams $1 --lanes $2 --steps 20 --warmup 3 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 lanes $2: device replay %.0f frames/s, aggregate roofline %.4f' % (d['device_replay']['value'], d['roofline_aggregate']['frac']))" || exit 1
done
