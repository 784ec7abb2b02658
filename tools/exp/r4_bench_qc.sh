#!/bin/bash
# round 4: product-path rate against hardware queues x copy streams per decoder
for q in 4 8 16; do
  for c in 1 2 3; do
    GPU_MAX_HW_QUEUES=$q DE265HIP_COPY_STREAMS=$c DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --no-cpu-baseline --no-copy-out 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('queues $q copy-streams $c value', d['value'], 'replay', d['device_replay']['value'])"
    grep "de265hip pipeline" /tmp/err.txt | head -1
  done
done
