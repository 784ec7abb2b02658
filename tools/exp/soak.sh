#!/bin/bash
# long product-path runs (pool / event / lane bookkeeping under sustained load): tools/exp/soak.sh [steps]
steps=${1:-400}
for args in "--streams 3" "--streams 3 --lanes 2" "--streams 1 --lanes 2" "--streams 2 --lanes 3 --gop 8"; do
  /usr/bin/env python bench.py $args --steps $steps --warmup 3 2>/tmp/soak.err | python -c "
import sys,json,resource; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args: product %.0f replay %.0f parity %s' % (d['value'], d['device_replay']['value'], d['parity_vs_reference']))" || { tail -n 5 /tmp/soak.err; exit 1; }
  grep -i "error\|fault\|hang" /tmp/soak.err | head -n 3
done
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -n 2
