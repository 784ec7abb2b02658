#!/bin/bash
# round 4: where the pipeline workers' time goes, against the number of host threads
for t in 3 6 9 15; do
  DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads $t --no-cpu-baseline --no-copy-out 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $t value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline" /tmp/err.txt | head -3
done
