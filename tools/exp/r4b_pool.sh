#!/bin/bash
for c in 2 3 4; do
  DE265HIP_COPY_STREAMS=$c DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads 6 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pooled streams, copy-streams $c: value', d['value'], 'replay', d['device_replay']['value'])"
done
