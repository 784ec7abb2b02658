#!/bin/bash
for p in 20 50 100 200; do for t in 9 12; do
  DE265HIP_PIPE_POLL_US=$p DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads $t --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('poll $p us host-threads $t: value', d['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done; done
