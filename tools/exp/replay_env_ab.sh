#!/bin/bash
# device replay (3 streams and 1 stream) for environment settings: tools/exp/replay_env_ab.sh "A=1" "A=2" ...
for e in "$@"; do
  for s in 3 1; do
    env $e timeout -k 10 300 python bench.py --streams $s --steps 20 --warmup 3 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e streams $s: device replay %.0f' % d['device_replay']['value'])" || exit 1
  done
done
