#!/bin/bash
# the bench's own bit-exactness check (last picture of stream 0 against the compiled reference) with lanes: tools/exp/bench_parity_lanes.sh
for args in "--streams 1 --lanes 2" "--streams 3 --lanes 2" "--streams 2 --lanes 3"; do
  python bench.py $args --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args:', 'product', d['value'], 'replay', d['device_replay']['value'], 'parity', d['parity_vs_reference'])"
done
