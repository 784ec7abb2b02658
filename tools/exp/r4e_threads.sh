#!/bin/bash
for t in 6 9 12 15; do
  DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads $t --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('host-threads $t: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -3
  grep "enqueue sections" /tmp/err.txt | head -1
done
