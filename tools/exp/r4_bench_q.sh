#!/bin/bash
# round 4: product-path rate against the number of hardware queues and GOP streams (the scan's kernels live on the copy streams)
for q in 4 8 16; do
  for s in 1 3; do
    GPU_MAX_HW_QUEUES=$q python bench.py --streams $s --steps 10 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('queues $q streams $s value', d['value'], 'replay', d['device_replay']['value'], '1thr', d['product_path']['value_1_host_thread'])"
  done
done
