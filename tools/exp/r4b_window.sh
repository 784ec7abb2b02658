#!/bin/bash
# product path against the pipeline window (pictures between parser and device) and the scan batch, pooled kernel streams
for w in 12 24 48; do for b in 2 4; do for c in 2 3; do
  DE265HIP_PIPE_WINDOW=$w DE265HIP_PIPE_BATCH=$b DE265HIP_COPY_STREAMS=$c DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads 6 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('window $w batch $b copy-streams $c: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: 2" /tmp/err.txt | head -1
done; done; done
