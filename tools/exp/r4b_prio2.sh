#!/bin/bash
# pooled kernel streams: device replay and product path against copy streams / hardware queues per priority pool
for q in 4 8; do for c in 1 2 3; do for o in 0 1; do
  GPU_MAX_HW_QUEUES=$q DE265HIP_OWN_STREAMS=$o DE265HIP_COPY_STREAMS=$c python bench.py --streams 3 --steps 10 --host-threads 6 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hwq $q copy-streams $c own-streams $o: value', d['value'], 'replay', d['device_replay']['value'])"
done; done; done
