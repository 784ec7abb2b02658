#!/bin/bash
# product path against the scan chains a decoder keeps in flight and the pictures per chain
for t in 6 9; do for k in 1 2 3; do for b in 2 4; do
  DE265HIP_PIPE_CHAINS=$k DE265HIP_PIPE_BATCH=$b DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads $t --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('host-threads $t chains $k batch $b: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: 2" /tmp/err.txt | head -1
done; done; done
