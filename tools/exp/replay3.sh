#!/bin/bash
# 3-stream device replay rate of the bench for environment settings: tools/exp/replay3.sh "A=1" "A=2" ...
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-inclusive --no-e2e 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'replay fps %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items() if v.get('ms_per_step')})" || exit 1
done
