#!/bin/bash
for n in 0 8 4 2; do
  DE265HIP_SCAN_CUS=$n DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads 9 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan on every $n-th CU: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done
