#!/bin/bash
# A/B of environment settings on the bench (3 GOP streams, no CPU baseline): bench_ab.sh "A=1" "A=2" ...
for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items() if v.get('ms_per_step')})" || exit 1
done
